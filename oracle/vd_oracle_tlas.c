/*
 * vd_oracle_tlas.c — CPU restatement of voidin's agglomerative TLAS builder (+ refit).
 * TEST INFRASTRUCTURE ONLY (see vd_oracle.h; parity unpinned).
 *
 * Literal port of crates/bvh/src/tlas.rs:31-105, quirks included:
 * leaf boxes are seeded with the OBJECT-space mesh box (tlas.rs:39), the merge loop runs
 * while instance_count > 0 (tlas.rs:61; one extra self-merge of the root), nodes[0] is a
 * copy of the last node (tlas.rs:84).
 * glam 0.24.1 Mat4::transform_point3 (spec decision, source not on disk):
 *   ((X*p.x + Y*p.y) + Z*p.z) + W, no FMA.
 * Refit (T3) is NEW: same topology, boxes recomputed bottom-up.
 */
#include "vd_oracle.h"
#include "vd_oracle_math.h"

#include <stdlib.h>

typedef struct { v3 mn, mx; } aabb;

/* tlas.rs:35-44 */
static aabb leaf_bounds(const VdInstance* inst, const VdMeshInfo* meshes, uint32_t n_mesh) {
    uint32_t mid = inst->mesh < n_mesh ? inst->mesh : n_mesh - 1;
    const VdMeshInfo* mesh = &meshes[mid];
    const float* T = inst->transform;
    v3 bound[2] = {v3_load(mesh->min), v3_load(mesh->max)};
    aabb r = {bound[0], bound[1]}; /* fold seed: [mesh.min, mesh.max] */
    for (int i = 0; i < 8; ++i) {
        /* [i & 1, i & 2, i & 4].map(|i| i == 0).map(usize::from) */
        int ix = (i & 1) == 0, iy = (i & 2) == 0, iz = (i & 4) == 0;
        float px = bound[ix].x, py = bound[iy].y, pz = bound[iz].z;
        v3 p;
        p.x = ((T[0] * px + T[4] * py) + T[8] * pz) + T[12];
        p.y = ((T[1] * px + T[5] * py) + T[9] * pz) + T[13];
        p.z = ((T[2] * px + T[6] * py) + T[10] * pz) + T[14];
        r.mn = v3_min_to(r.mn, p);
        r.mx = v3_max_to(r.mx, p);
    }
    return r;
}

/* tlas.rs:87-105 */
static size_t find_best_match(const aabb* boxes, const uint32_t* indices, size_t num_unused,
                              size_t target) {
    float smallest = 1e30f;
    size_t best = target;
    aabb t = boxes[indices[target]];
    for (size_t i = 0; i < num_unused; ++i) {
        if (target == i) continue;
        aabb o = boxes[indices[i]];
        v3 bmin = v3_min_to(t.mn, o.mn);
        v3 bmax = v3_max_to(t.mx, o.mx);
        float area = aabb_area(bmin, bmax);
        if (area < smallest) {
            smallest = area;
            best = i;
        }
    }
    return best;
}

/* Shared build core: fills boxes[0..2n], left[], right[], inst[] for 2n+1 nodes. */
static int tlas_build_core(const VdInstance* instances, uint32_t n, const VdMeshInfo* meshes,
                           uint32_t n_mesh, aabb* boxes, uint32_t* left, uint32_t* right,
                           uint32_t* inst_idx) {
    size_t total = 2u * (size_t)n + 1u;
    for (size_t k = 0; k < total; ++k) {
        boxes[k].mn = v3_make(0, 0, 0); boxes[k].mx = v3_make(0, 0, 0);
        left[k] = right[k] = 0; inst_idx[k] = 0; /* TlasNode::default() */
    }
    for (uint32_t i = 0; i < n; ++i) { /* tlas.rs:34-54 */
        boxes[i + 1] = leaf_bounds(&instances[i], meshes, n_mesh);
        inst_idx[i + 1] = i;
    }
    uint32_t* node_indices = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    if (!node_indices) return VD_ERR_OOM;
    for (uint32_t i = 0; i < n; ++i) node_indices[i] = i + 1;

    size_t instance_count = n;         /* tlas.rs:56-60 */
    size_t nodes_used = 1 + instance_count;
    size_t a = 0;
    size_t b = find_best_match(boxes, node_indices, instance_count, a);
    while (instance_count > 0) {       /* tlas.rs:61 */
        size_t c = find_best_match(boxes, node_indices, instance_count, b);
        if (a == c) {
            uint32_t idx_a = node_indices[a], idx_b = node_indices[b];
            boxes[nodes_used].mn = v3_min_to(boxes[idx_a].mn, boxes[idx_b].mn);
            boxes[nodes_used].mx = v3_max_to(boxes[idx_a].mx, boxes[idx_b].mx);
            left[nodes_used] = idx_a;
            right[nodes_used] = idx_b;
            inst_idx[nodes_used] = 0xffffffffu;
            node_indices[a] = (uint32_t)nodes_used;
            nodes_used += 1;
            node_indices[b] = node_indices[instance_count - 1];
            instance_count -= 1;
            b = find_best_match(boxes, node_indices, instance_count, a);
        } else {
            a = b;
            b = c;
        }
    }
    /* tlas.rs:84 nodes[0] = nodes[node_indices[a]] */
    uint32_t root = node_indices[a];
    boxes[0] = boxes[root]; left[0] = left[root]; right[0] = right[root];
    inst_idx[0] = inst_idx[root];
    free(node_indices);
    return VD_OK;
}

static int tlas_alloc(uint32_t n, aabb** boxes, uint32_t** l, uint32_t** r, uint32_t** ii) {
    size_t total = 2u * (size_t)n + 1u;
    *boxes = (aabb*)malloc(sizeof(aabb) * total);
    *l = (uint32_t*)malloc(sizeof(uint32_t) * total);
    *r = (uint32_t*)malloc(sizeof(uint32_t) * total);
    *ii = (uint32_t*)malloc(sizeof(uint32_t) * total);
    if (!*boxes || !*l || !*r || !*ii) { free(*boxes); free(*l); free(*r); free(*ii); return VD_ERR_OOM; }
    return VD_OK;
}

int vd_ref_tlas_build(const VdInstance* instances, uint32_t n, const VdMeshInfo* meshes,
                      uint32_t n_mesh, VdTlasNode* out_nodes) {
    if (!instances || !meshes || !out_nodes || n == 0 || n_mesh == 0) return VD_ERR_INVALID_ARG;
    if (n > VD_TLAS_MAX_INSTANCES) return VD_ERR_TLAS_OVERFLOW; /* tlas.rs:71 packs 16-bit ids */
    aabb* boxes; uint32_t *l, *r, *ii;
    int rc = tlas_alloc(n, &boxes, &l, &r, &ii);
    if (rc) return rc;
    rc = tlas_build_core(instances, n, meshes, n_mesh, boxes, l, r, ii);
    if (rc == VD_OK) {
        for (size_t k = 0; k < 2u * (size_t)n + 1u; ++k) {
            v3_store(out_nodes[k].min, boxes[k].mn);
            v3_store(out_nodes[k].max, boxes[k].mx);
            out_nodes[k].left_right = l[k] + (r[k] << 16); /* tlas.rs:71 */
            out_nodes[k].instance_idx = ii[k];
        }
    }
    free(boxes); free(l); free(r); free(ii);
    return rc;
}

int vd_ref_tlas_build_wide(const VdInstance* instances, uint32_t n, const VdMeshInfo* meshes,
                           uint32_t n_mesh, VdTlasNodeWide* out_nodes) {
    if (!instances || !meshes || !out_nodes || n == 0 || n_mesh == 0) return VD_ERR_INVALID_ARG;
    if (n > 0x3fffffffu) return VD_ERR_INVALID_ARG;
    aabb* boxes; uint32_t *l, *r, *ii;
    int rc = tlas_alloc(n, &boxes, &l, &r, &ii);
    if (rc) return rc;
    rc = tlas_build_core(instances, n, meshes, n_mesh, boxes, l, r, ii);
    if (rc == VD_OK) {
        for (size_t k = 0; k < 2u * (size_t)n + 1u; ++k) {
            memset(&out_nodes[k], 0, sizeof(out_nodes[k]));
            v3_store(out_nodes[k].min, boxes[k].mn);
            v3_store(out_nodes[k].max, boxes[k].mx);
            out_nodes[k].left = l[k];
            out_nodes[k].right = r[k];
            out_nodes[k].instance_idx = ii[k];
        }
    }
    free(boxes); free(l); free(r); free(ii);
    return rc;
}

/* SURVEY.md §8a T3 */
int vd_ref_tlas_refit(const VdInstance* instances, uint32_t n, const VdMeshInfo* meshes,
                      uint32_t n_mesh, VdTlasNode* nodes) {
    if (!instances || !meshes || !nodes || n == 0 || n_mesh == 0) return VD_ERR_INVALID_ARG;
    if (n > VD_TLAS_MAX_INSTANCES) return VD_ERR_TLAS_OVERFLOW;
    for (uint32_t i = 0; i < n; ++i) {
        aabb b = leaf_bounds(&instances[nodes[i + 1].instance_idx], meshes, n_mesh);
        v3_store(nodes[i + 1].min, b.mn);
        v3_store(nodes[i + 1].max, b.mx);
    }
    for (size_t k = (size_t)n + 1; k <= 2u * (size_t)n; ++k) {
        uint32_t l = nodes[k].left_right & 0xffffu, r = nodes[k].left_right >> 16;
        v3_store(nodes[k].min, v3_min_to(v3_load(nodes[l].min), v3_load(nodes[r].min)));
        v3_store(nodes[k].max, v3_max_to(v3_load(nodes[l].max), v3_load(nodes[r].max)));
    }
    /* node 0 is a COPY of the node the chain ended on (tlas.rs:84) - node 2n in every ordinary build, but any node when
     * the chain ended on a stale slot (a cluster whose unions all reach 1e30 is dropped: tlas.rs:62-79, 88-104; the root
     * can then be a single leaf): its box follows from its OWN payload, which a refit never touches */
    {
        uint32_t l = nodes[0].left_right & 0xffffu, r = nodes[0].left_right >> 16;
        if (nodes[0].left_right == 0u) {
            aabb b = leaf_bounds(&instances[nodes[0].instance_idx], meshes, n_mesh);
            v3_store(nodes[0].min, b.mn); v3_store(nodes[0].max, b.mx);
        } else {
            v3 mn = v3_min_to(v3_load(nodes[l].min), v3_load(nodes[r].min)), mx = v3_max_to(v3_load(nodes[l].max), v3_load(nodes[r].max));
            v3_store(nodes[0].min, mn); v3_store(nodes[0].max, mx);
        }
    }
    return VD_OK;
}

int vd_ref_tlas_refit_wide(const VdInstance* instances, uint32_t n, const VdMeshInfo* meshes,
                           uint32_t n_mesh, VdTlasNodeWide* nodes) {
    if (!instances || !meshes || !nodes || n == 0 || n_mesh == 0) return VD_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        aabb b = leaf_bounds(&instances[nodes[i + 1].instance_idx], meshes, n_mesh);
        v3_store(nodes[i + 1].min, b.mn);
        v3_store(nodes[i + 1].max, b.mx);
    }
    for (size_t k = (size_t)n + 1; k <= 2u * (size_t)n; ++k) {
        uint32_t l = nodes[k].left, r = nodes[k].right;
        v3_store(nodes[k].min, v3_min_to(v3_load(nodes[l].min), v3_load(nodes[r].min)));
        v3_store(nodes[k].max, v3_max_to(v3_load(nodes[l].max), v3_load(nodes[r].max)));
    }
    {   /* node 0 from its own payload: see vd_ref_tlas_refit */
        uint32_t l = nodes[0].left, r = nodes[0].right;
        if (l == 0u && r == 0u) {
            aabb b = leaf_bounds(&instances[nodes[0].instance_idx], meshes, n_mesh);
            v3_store(nodes[0].min, b.mn); v3_store(nodes[0].max, b.mx);
        } else {
            v3 mn = v3_min_to(v3_load(nodes[l].min), v3_load(nodes[r].min)), mx = v3_max_to(v3_load(nodes[l].max), v3_load(nodes[r].max));
            v3_store(nodes[0].min, mn); v3_store(nodes[0].max, mx);
        }
    }
    return VD_OK;
}

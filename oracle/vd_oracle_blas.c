/*
 * vd_oracle_blas.c — CPU restatement of voidin's SAH BLAS builder.
 * TEST INFRASTRUCTURE ONLY (see vd_oracle.h; parity unpinned).
 *
 * Literal sequential port of crates/bvh/src/blas.rs:51-204, quirks included
 * (SURVEY.md §8a B1-B8): hard-coded 8 bins (blas.rs:136), order-dependent
 * partition_shuffle with its never-examined element (blas.rs:168-182), NaN-rejected empty
 * splits (blas.rs:155-156), stale optimal_pivot after the final re-shuffle (blas.rs:164-165),
 * node 1 never used (blas.rs:90), count reset to 0 for interior nodes (blas.rs:127).
 * This is deliberately the NAIVE loop: the HIP builder's closed-form emulation is tested
 * against it.
 *
 * glam 0.24.1 (Cargo.lock:844) is not on disk; the two glam-dependent choices are spec
 * decisions from its published source: Vec3::lerp(rhs,s) = self + (rhs - self)*s and
 * Vec3 / f32 = component-wise true division.  min/max are Rust's f32::min/max (a NaN operand
 * is ignored; -0 < +0 as the tie rule Rust leaves open): vd_oracle_math.h.
 */
#include "vd_oracle.h"
#include "vd_oracle_math.h"

#include <stdlib.h>

typedef struct {
    const float* verts;
    const uint32_t* indices; /* original (unpermuted) triples */
    v3* centroids;
    uint32_t* tri_ids;
    VdBvhNode* nodes;
} builder;

typedef struct { v3 mn, mx; } aabb;

/* blas.rs:184-204 */
static aabb calculate_bounds(const builder* b, uint32_t first, uint32_t amount, int centroids) {
    aabb r;
    r.mx = v3_make(-VD_REF_MAX_DIST, -VD_REF_MAX_DIST, -VD_REF_MAX_DIST);
    r.mn = v3_make(VD_REF_MAX_DIST, VD_REF_MAX_DIST, VD_REF_MAX_DIST);
    for (uint32_t k = 0; k < amount; ++k) {
        uint32_t idx = b->tri_ids[first + k];
        if (centroids) {
            v3 v = b->centroids[idx];
            r.mx = v3_max_to(r.mx, v);
            r.mn = v3_min_to(r.mn, v);
        } else {
            for (int c = 0; c < 3; ++c) {
                v3 v = v3_load(b->verts + 3u * (size_t)b->indices[3u * (size_t)idx + c]);
                r.mx = v3_max_to(r.mx, v);
                r.mn = v3_min_to(r.mn, v);
            }
        }
    }
    return r;
}

static inline float v3_axis(v3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

/* blas.rs:168-182 */
static uint32_t partition_shuffle(builder* b, int axis, float pos, uint32_t start, uint32_t count) {
    size_t end = (size_t)start + count - 1;
    size_t i = start;
    while (i < end) {
        if (v3_axis(b->centroids[b->tri_ids[i]], axis) < pos) {
            i += 1;
        } else {
            uint32_t t = b->tri_ids[i];
            b->tri_ids[i] = b->tri_ids[end];
            b->tri_ids[end] = t;
            end -= 1;
        }
    }
    return (uint32_t)i;
}

uint32_t vd_ref_partition_shuffle(const float* keys_by_id, uint32_t* ids, uint32_t start,
                                  uint32_t count, float pos) {
    size_t end = (size_t)start + count - 1;
    size_t i = start;
    while (i < end) {
        if (keys_by_id[ids[i]] < pos) {
            i += 1;
        } else {
            uint32_t t = ids[i]; ids[i] = ids[end]; ids[end] = t;
            end -= 1;
        }
    }
    return (uint32_t)i;
}

/* blas.rs:135-166. Returns 0 and sets *degenerate when every candidate was rejected
 * (the reference then underflows / recurses forever: SURVEY.md §8a B7). */
static uint32_t partition(builder* b, uint32_t start, uint32_t count, int* degenerate) {
    const int bins = 8; /* blas.rs:136 — num_bins is ignored */
    int optimal_axis = 0;
    float optimal_pos = 0.0f;
    uint32_t optimal_pivot = 0;
    float optimal_cost = 3.40282347e+38f; /* f32::MAX */
    int accepted = 0;

    aabb cb = calculate_bounds(b, start, count, 1);
    for (int axis = 0; axis < 3; ++axis) {
        for (int k = 1; k < bins; ++k) {
            float scale = (float)k / (float)bins;
            /* glam Vec3::lerp: self + (rhs - self) * s */
            v3 p = v3_add(cb.mn, v3_scale(v3_sub(cb.mx, cb.mn), scale));
            float pos = v3_axis(p, axis);
            uint32_t pivot = partition_shuffle(b, axis, pos, start, count);

            uint32_t bb1_count = pivot - start;
            uint32_t bb2_count = count - bb1_count;
            aabb bb1 = calculate_bounds(b, start, bb1_count, 0);
            aabb bb2 = calculate_bounds(b, pivot, bb2_count, 0);
            float cost = aabb_area(bb1.mn, bb1.mx) * (float)bb1_count +
                         aabb_area(bb2.mn, bb2.mx) * (float)bb2_count;
            if (cost < optimal_cost) {
                optimal_axis = axis;
                optimal_pos = pos;
                optimal_pivot = pivot;
                optimal_cost = cost;
                accepted = 1;
            }
        }
    }
    partition_shuffle(b, optimal_axis, optimal_pos, start, count); /* result discarded */
    if (!accepted) *degenerate = 1;
    return optimal_pivot;
}

static void set_bound(VdBvhNode* n, const aabb* a) {
    v3_store(n->max, a->mx);
    v3_store(n->min, a->mn);
}

int vd_ref_bvh_build(const float* verts_xyz, uint32_t n_vert, uint32_t* indices_inout,
                     uint32_t n_tri, VdBvhNode* out_nodes, uint32_t node_cap,
                     uint32_t* out_n_nodes) {
    if (!verts_xyz || !indices_inout || !out_nodes || !out_n_nodes || n_tri == 0 || n_vert == 0)
        return VD_ERR_INVALID_ARG;
    if (node_cap < 2u * n_tri || n_tri > 0x7fffffffu / 2u) return VD_ERR_INVALID_ARG;
    for (size_t k = 0; k < 3u * (size_t)n_tri; ++k)
        if (indices_inout[k] >= n_vert) return VD_ERR_INVALID_ARG;

    builder b;
    b.verts = verts_xyz;
    uint32_t* orig = (uint32_t*)malloc(sizeof(uint32_t) * 3u * (size_t)n_tri);
    b.centroids = (v3*)malloc(sizeof(v3) * (size_t)n_tri);
    b.tri_ids = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n_tri);
    /* explicit DFS stack: (node, start) pairs; the reference recurses (blas.rs:125-126) */
    size_t stack_cap = 2u * (size_t)n_tri + 2u;
    uint32_t* stack = (uint32_t*)malloc(sizeof(uint32_t) * 2u * stack_cap);
    if (!orig || !b.centroids || !b.tri_ids || !stack) {
        free(orig); free(b.centroids); free(b.tri_ids); free(stack);
        return VD_ERR_OOM;
    }
    memcpy(orig, indices_inout, sizeof(uint32_t) * 3u * (size_t)n_tri);
    b.indices = orig;
    b.nodes = out_nodes;
    memset(out_nodes, 0, sizeof(VdBvhNode) * 2u * (size_t)n_tri); /* blas.rs:52 */

    /* blas.rs:70-81 centroids = ((v0 + v1) + v2) / 3.0 */
    for (uint32_t t = 0; t < n_tri; ++t) {
        v3 v0 = v3_load(verts_xyz + 3u * (size_t)orig[3u * (size_t)t + 0]);
        v3 v1 = v3_load(verts_xyz + 3u * (size_t)orig[3u * (size_t)t + 1]);
        v3 v2 = v3_load(verts_xyz + 3u * (size_t)orig[3u * (size_t)t + 2]);
        v3 s = v3_add(v3_add(v0, v1), v2);
        b.centroids[t] = v3_make(s.x / 3.0f, s.y / 3.0f, s.z / 3.0f);
        b.tri_ids[t] = t; /* blas.rs:83 */
    }
    out_nodes[0].left_first = 0;
    out_nodes[0].count = n_tri; /* blas.rs:84-85 */
    aabb root = calculate_bounds(&b, 0, n_tri, 0);
    set_bound(&out_nodes[0], &root); /* blas.rs:87-88 */

    uint32_t pool = 2; /* blas.rs:90 */
    int degenerate = 0;
    size_t sp = 0;
    stack[2 * sp] = 0; stack[2 * sp + 1] = 0; ++sp;
    while (sp > 0 && !degenerate) {
        --sp;
        uint32_t cur = stack[2 * sp], start = stack[2 * sp + 1];
        VdBvhNode* node = &out_nodes[cur];
        if (node->count <= 3) { /* blas.rs:106-109 */
            node->left_first = start;
            continue;
        }
        uint32_t index = pool; /* blas.rs:110-112 */
        pool += 2;
        uint32_t count = node->count;
        node->left_first = index;

        uint32_t pivot = partition(&b, start, count, &degenerate); /* blas.rs:114 */
        if (degenerate) break;
        uint32_t left_count = pivot - start;
        out_nodes[index].count = left_count;
        aabb bl = calculate_bounds(&b, start, left_count, 0);
        set_bound(&out_nodes[index], &bl);
        uint32_t right_count = count - left_count;
        out_nodes[index + 1].count = right_count;
        aabb br = calculate_bounds(&b, pivot, right_count, 0);
        set_bound(&out_nodes[index + 1], &br);

        node->count = 0; /* blas.rs:127 (after the recursion there; nothing reads it between) */
        /* left first, then right (blas.rs:125-126): push right below left */
        stack[2 * sp] = index + 1; stack[2 * sp + 1] = pivot; ++sp;
        stack[2 * sp] = index; stack[2 * sp + 1] = start; ++sp;
    }

    int rc = VD_OK;
    if (degenerate) {
        rc = VD_ERR_DEGENERATE;
    } else {
        *out_n_nodes = pool; /* blas.rs:93 truncate */
        /* blas.rs:95-100: indices[i] = old_indices[tri_ids[i]] */
        for (uint32_t i = 0; i < n_tri; ++i) {
            uint32_t t = b.tri_ids[i];
            indices_inout[3u * (size_t)i + 0] = orig[3u * (size_t)t + 0];
            indices_inout[3u * (size_t)i + 1] = orig[3u * (size_t)t + 1];
            indices_inout[3u * (size_t)i + 2] = orig[3u * (size_t)t + 2];
        }
    }
    free(orig); free(b.centroids); free(b.tri_ids); free(stack);
    return rc;
}

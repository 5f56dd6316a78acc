/*
 * vd_oracle_trace.c — CPU restatement of voidin's TLAS/BLAS ray traversal.
 * TEST INFRASTRUCTURE ONLY (see vd_oracle.h; parity unpinned).
 *
 * R1: shaders/utils/bvh.wgsl:30-123 + shaders/utils/intersections.wgsl:13-45 +
 *     shaders/utils/stack.wgsl:1-20 (the GPU path; multiply by inv_dir, backface cull).
 * R2: crates/bvh/src/blas.rs:247-295 + crates/bvh/src/intersection.rs:47-92 (the Rust CPU
 *     harness; divide by dir, two-sided, EPS = 1e-4).
 * WGSL min/max on NaN operands is implementation-defined; spec decision: fminf/fmaxf
 * (IEEE minNum/maxNum), which is also what v_min_f32/v_max_f32 do on gfx950.
 */
#include "vd_oracle.h"
#include "vd_oracle_math.h"

#include <pthread.h>
#include <stdlib.h>

#define REF_STACK 256

typedef struct { v3 eye, dir, inv_dir; } ray_t;

static inline float min_element(v3 v) { return fminf(v.x, fminf(v.y, v.z)); }   /* math.wgsl:19-21 */
static inline float max_element(v3 v) { return fmaxf(v.x, fmaxf(v.y, v.z)); }   /* math.wgsl:23-25 */
static inline v3 v3_fmin(v3 a, v3 b) { return v3_make(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static inline v3 v3_fmax(v3 a, v3 b) { return v3_make(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }

/* intersections.wgsl:13-23 */
static inline float intersect_aabb_wgsl(const ray_t* r, v3 bmin, v3 bmax, float t) {
    v3 tx1 = v3_mul(v3_sub(bmin, r->eye), r->inv_dir);
    v3 tx2 = v3_mul(v3_sub(bmax, r->eye), r->inv_dir);
    float tmax = min_element(v3_fmax(tx1, tx2));
    float tmin = max_element(v3_fmin(tx1, tx2));
    if (tmax >= tmin && tmin < t && tmax > 0.0f) return tmin;
    return VD_REF_MAX_DIST;
}

/* intersections.wgsl:25-45 */
static inline int intersect_trig_wgsl(const ray_t* r, v3 v0, v3 v1, v3 v2, float* hit) {
    v3 edge1 = v3_sub(v1, v0);
    v3 edge2 = v3_sub(v2, v0);
    v3 uvec = v3_cross(r->dir, edge2);
    float det = v3_dot(edge1, uvec);
    if (det < 1e-10f) return 0; /* cull backface */
    float inv_det = 1.0f / det;
    v3 orig = v3_sub(r->eye, v0);
    float u = inv_det * v3_dot(orig, uvec);
    if (u < 0.0f || 1.0f < u) return 0;
    v3 vvec = v3_cross(orig, edge1);
    float v = inv_det * v3_dot(r->dir, vvec);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = inv_det * v3_dot(edge2, vvec);
    if (t > 0.0f && t < *hit) {
        *hit = t;
        return 1;
    }
    return 0;
}

/* bvh.wgsl:30-33 */
static inline v3 fetch_vertex(const VdTraceScene* s, uint32_t idx, const VdMeshInfo* mesh) {
    uint32_t i = (uint32_t)mesh->vertex_offset + s->indices[mesh->base_index + idx];
    return v3_load(s->vertices + 3u * (size_t)i);
}

/* bvh.wgsl:35-76 */
static int traverse_bvh(const VdTraceScene* s, const ray_t* ray, const VdMeshInfo* mesh,
                        VdHit* res, uint32_t inst_id, uint32_t* max_stack) {
    uint32_t stack[REF_STACK];
    uint32_t head = 0;
    stack[head++] = mesh->bvh_index;
    float hit = res->dist;
    while (head > 0) {
        VdBvhNode node = s->bvh_nodes[stack[--head]];
        if (node.count > 0) {
            for (uint32_t i = 0; i < node.count; ++i) {
                uint32_t idx = node.left_first + i;
                v3 v0 = fetch_vertex(s, 3u * idx + 0u, mesh);
                v3 v1 = fetch_vertex(s, 3u * idx + 1u, mesh);
                v3 v2 = fetch_vertex(s, 3u * idx + 2u, mesh);
                if (intersect_trig_wgsl(ray, v0, v1, v2, &hit)) {
                    res->dist = hit; res->hit = 1; res->instance = inst_id; res->triangle = idx;
                }
            }
        } else {
            uint32_t min_index = mesh->bvh_index + node.left_first;
            uint32_t max_index = mesh->bvh_index + node.left_first + 1u;
            VdBvhNode min_child = s->bvh_nodes[min_index];
            VdBvhNode max_child = s->bvh_nodes[max_index];
            float min_dist = intersect_aabb_wgsl(ray, v3_load(min_child.min), v3_load(min_child.max), hit);
            float max_dist = intersect_aabb_wgsl(ray, v3_load(max_child.min), v3_load(max_child.max), hit);
            if (min_dist > max_dist) {
                uint32_t ti = min_index; min_index = max_index; max_index = ti;
                float tf = min_dist; min_dist = max_dist; max_dist = tf;
            }
            if (min_dist >= hit) continue;
            if (head + 2 > REF_STACK) return VD_ERR_STACK_OVERFLOW;
            if (max_dist <= hit) stack[head++] = max_index;
            stack[head++] = min_index;
            if (head > *max_stack) *max_stack = head;
        }
    }
    return VD_OK;
}

/* bvh.wgsl:78-87 */
static int instance_intersect(const VdTraceScene* s, const ray_t* ray, uint32_t inst_id,
                              VdHit* res, uint32_t* max_stack) {
    const VdInstance* instance = &s->instances[inst_id];
    uint32_t mid = instance->mesh < s->n_meshes ? instance->mesh : s->n_meshes - 1;
    const VdMeshInfo* mesh = &s->meshes[mid];
    const float* M = instance->inv_transform;
    ray_t nr;
    /* (inv_transform * vec4(eye, 1.)).xyz and (inv_transform * vec4(dir, 0.)).xyz */
    float e[3], d[3];
    for (int r = 0; r < 3; ++r) {
        e[r] = ((M[0 + r] * ray->eye.x + M[4 + r] * ray->eye.y) + M[8 + r] * ray->eye.z) + M[12 + r] * 1.0f;
        d[r] = ((M[0 + r] * ray->dir.x + M[4 + r] * ray->dir.y) + M[8 + r] * ray->dir.z) + M[12 + r] * 0.0f;
    }
    nr.eye = v3_make(e[0], e[1], e[2]);
    nr.dir = v3_make(d[0], d[1], d[2]);
    nr.inv_dir = v3_make(1.0f / nr.dir.x, 1.0f / nr.dir.y, 1.0f / nr.dir.z);
    return traverse_bvh(s, &nr, mesh, res, inst_id, max_stack);
}

/* bvh.wgsl:89-123 */
static int traverse_tlas(const VdTraceScene* s, const ray_t* ray, VdHit* res, uint32_t* max_stack) {
    uint32_t stack[REF_STACK];
    uint32_t head = 0;
    stack[head++] = 0u;
    res->dist = VD_REF_MAX_DIST; res->hit = 0; res->instance = 0xffffffffu; res->triangle = 0xffffffffu;
    while (head > 0) {
        VdTlasNode node = s->tlas_nodes[stack[--head]];
        if (node.left_right == 0u) {
            int rc = instance_intersect(s, ray, node.instance_idx, res, max_stack);
            if (rc) return rc;
        } else {
            uint32_t min_index = node.left_right & 0xffffu;
            uint32_t max_index = node.left_right >> 16u;
            VdTlasNode min_child = s->tlas_nodes[min_index];
            VdTlasNode max_child = s->tlas_nodes[max_index];
            float min_dist = intersect_aabb_wgsl(ray, v3_load(min_child.min), v3_load(min_child.max), res->dist);
            float max_dist = intersect_aabb_wgsl(ray, v3_load(max_child.min), v3_load(max_child.max), res->dist);
            if (min_dist > max_dist) {
                uint32_t ti = min_index; min_index = max_index; max_index = ti;
                float tf = min_dist; min_dist = max_dist; max_dist = tf;
            }
            if (min_dist >= res->dist) continue;
            if (head + 2 > REF_STACK) return VD_ERR_STACK_OVERFLOW;
            if (max_dist < res->dist) stack[head++] = max_index;
            stack[head++] = min_index;
            if (head > *max_stack) *max_stack = head;
        }
    }
    return VD_OK;
}

typedef struct {
    const VdTraceScene* s; const VdRay* rays; uint32_t begin, end; VdHit* out;
    uint32_t max_stack; int rc;
} trace_job;

static void* trace_thread(void* p) {
    trace_job* j = (trace_job*)p;
    j->rc = VD_OK; j->max_stack = 0;
    for (uint32_t i = j->begin; i < j->end; ++i) {
        ray_t r;
        r.eye = v3_load(j->rays[i].eye);
        r.dir = v3_load(j->rays[i].dir);
        /* intersections.wgsl:9-11 ray_new: inv_dir = 1. / dir */
        r.inv_dir = v3_make(1.0f / r.dir.x, 1.0f / r.dir.y, 1.0f / r.dir.z);
        int rc = traverse_tlas(j->s, &r, &j->out[i], &j->max_stack);
        if (rc && !j->rc) j->rc = rc;
    }
    return NULL;
}

int vd_ref_trace(const VdTraceScene* scene, const VdRay* rays, uint32_t n_rays, VdHit* out,
                 uint32_t* out_max_stack, int threads) {
    if (!scene || !scene->tlas_nodes || !scene->instances || !scene->meshes || !scene->bvh_nodes ||
        !scene->vertices || !scene->indices || (n_rays && (!rays || !out)))
        return VD_ERR_INVALID_ARG;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (n_rays < 1024) threads = 1;
    pthread_t tid[256];
    trace_job job[256];
    for (int t = 0; t < threads; ++t) {
        job[t].s = scene; job[t].rays = rays; job[t].out = out;
        job[t].begin = (uint32_t)((uint64_t)n_rays * t / threads);
        job[t].end = (uint32_t)((uint64_t)n_rays * (t + 1) / threads);
        if (threads == 1 || pthread_create(&tid[t], NULL, trace_thread, &job[t]) != 0) {
            trace_thread(&job[t]);
            tid[t] = 0;
        }
    }
    int rc = VD_OK;
    uint32_t ms = 0;
    for (int t = 0; t < threads; ++t) {
        if (threads > 1 && tid[t]) pthread_join(tid[t], NULL);
        if (job[t].rc && !rc) rc = job[t].rc;
        if (job[t].max_stack > ms) ms = job[t].max_stack;
    }
    if (out_max_stack) *out_max_stack = ms;
    return rc;
}

/* ---------------- R2: Rust CPU harness variant (BLAS only) ---------------- */

typedef struct { int hit; float t; } dist_t; /* intersection.rs:22-26 enum Dist { Hit(f32), Miss } */

/* intersection.rs:47-55 */
static inline dist_t intersect_aabb_rs(v3 orig, v3 dir, v3 bmin, v3 bmax, float t) {
    v3 a = v3_sub(bmin, orig), b = v3_sub(bmax, orig);
    v3 tx1 = v3_make(a.x / dir.x, a.y / dir.y, a.z / dir.z);
    v3 tx2 = v3_make(b.x / dir.x, b.y / dir.y, b.z / dir.z);
    float tmax = min_element(v3_fmax(tx1, tx2));
    float tmin = max_element(v3_fmin(tx1, tx2));
    dist_t d;
    d.hit = (tmax >= tmin && tmin < t && tmax > 0.0f);
    d.t = tmin;
    return d;
}

/* intersection.rs:68-92; returns t or -1 (Miss) */
static inline float ray_intersect_rs(v3 orig, v3 dir, v3 v0, v3 v1, v3 v2) {
    const float EPS = 0.0001f;
    v3 edge1 = v3_sub(v1, v0), edge2 = v3_sub(v2, v0);
    v3 h = v3_cross(dir, edge2);
    float a = v3_dot(edge1, h);
    if (-EPS < a && a < EPS) return -1.0f;
    float f = 1.0f / a;
    v3 s = v3_sub(orig, v0);
    float u = f * v3_dot(s, h);
    if (!(0.0f <= u && u <= 1.0f)) return -1.0f;
    v3 q = v3_cross(s, edge1);
    float v = f * v3_dot(dir, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    float t = f * v3_dot(edge2, q);
    return t > EPS ? t : -1.0f;
}

/* Dist ordering (intersection.rs:22-26, derive(PartialOrd)): Hit(x) < Miss; Hit by value. */
static inline int dist_gt(dist_t a, dist_t b) {
    if (a.hit != b.hit) return !a.hit; /* Miss > Hit */
    if (!a.hit) return 0;
    return a.t > b.t;
}

int vd_ref_traverse_iter(const VdBvhNode* nodes, uint32_t n_nodes, const float* verts_xyz,
                         const uint32_t* indices, const VdRay* rays, uint32_t n_rays,
                         float* out_dist) {
    if (!nodes || !verts_xyz || !indices || (n_rays && (!rays || !out_dist)) || n_nodes == 0)
        return VD_ERR_INVALID_ARG;
    for (uint32_t r = 0; r < n_rays; ++r) {
        v3 orig = v3_load(rays[r].eye), dir = v3_load(rays[r].dir);
        uint32_t stack[REF_STACK];
        uint32_t head = 0;
        stack[head++] = 0;
        float hit = -1.0f; /* Miss */
        while (head > 0) {
            VdBvhNode node = nodes[stack[--head]];
            if (node.count > 0) {
                for (uint32_t i = 0; i < node.count; ++i) {
                    const uint32_t* idx = indices + 3u * (size_t)(node.left_first + i);
                    float d = ray_intersect_rs(orig, dir, v3_load(verts_xyz + 3u * (size_t)idx[0]),
                                               v3_load(verts_xyz + 3u * (size_t)idx[1]),
                                               v3_load(verts_xyz + 3u * (size_t)idx[2]));
                    if (d >= 0.0f) hit = (hit >= 0.0f) ? fminf(hit, d) : d;
                }
            } else {
                uint32_t min_index = node.left_first, max_index = node.left_first + 1;
                VdBvhNode mc = nodes[min_index], xc = nodes[max_index];
                float lim = hit >= 0.0f ? hit : VD_REF_MAX_DIST;
                dist_t min_dist = intersect_aabb_rs(orig, dir, v3_load(mc.min), v3_load(mc.max), lim);
                dist_t max_dist = intersect_aabb_rs(orig, dir, v3_load(xc.min), v3_load(xc.max), lim);
                if (dist_gt(min_dist, max_dist)) {
                    uint32_t ti = min_index; min_index = max_index; max_index = ti;
                    dist_t tf = min_dist; min_dist = max_dist; max_dist = tf;
                }
                if (!min_dist.hit) continue; /* blas.rs:285-288 */
                if (head + 2 > REF_STACK) return VD_ERR_STACK_OVERFLOW;
                stack[head++] = min_index;
                if (max_dist.hit) stack[head++] = max_index; /* blas.rs:289-291 */
            }
        }
        out_dist[r] = hit;
    }
    return VD_OK;
}

/* ---------------- R3: Bvh::traverse, the recursive walk (crates/bvh/src/blas.rs:211-245) ----------------
 * Dead code in the reference (its only call is commented out: src/bin/bvh_cpu.rs:86 `self.bvh.traverse(.., ray, 0, 1e30)`);
 * restated for completeness of SURVEY.md 8a.  It takes Vec4 / UVec4 arrays there (xyz used); the arrays here are the
 * Vec3 / UVec3 ones of traverse_iter.  Quirk kept: once a node's box is hit the call returns Hit(t) with the t it was
 * GIVEN when nothing closer was found, so the top-level call returns Hit(1e30) for a ray that enters the root box and
 * hits no triangle, and Miss only when the root box itself is missed.                                               */
static dist_t traverse_rec(const VdBvhNode* nodes, const float* verts_xyz, const uint32_t* indices, v3 orig, v3 dir,
                           uint32_t node_idx, float t, uint32_t depth, int* overflow) {
    dist_t miss = {0, 0.0f};
    if (depth > 4096u) { *overflow = 1; return miss; }
    const VdBvhNode* node = &nodes[node_idx];
    dist_t box = intersect_aabb_rs(orig, dir, v3_load(node->min), v3_load(node->max), t);
    if (!box.hit) return miss;                                   /* blas.rs:220-222 */
    if (node->count > 0) {                                       /* blas.rs:223-235 */
        for (uint32_t i = 0; i < node->count; ++i) {
            const uint32_t* idx = indices + 3u * (size_t)(node->left_first + i);
            float d = ray_intersect_rs(orig, dir, v3_load(verts_xyz + 3u * (size_t)idx[0]), v3_load(verts_xyz + 3u * (size_t)idx[1]),
                                       v3_load(verts_xyz + 3u * (size_t)idx[2]));
            if (d >= 0.0f) t = fminf(t, d);
        }
    } else {                                                     /* blas.rs:236-243 */
        dist_t l = traverse_rec(nodes, verts_xyz, indices, orig, dir, node->left_first, t, depth + 1u, overflow);
        if (l.hit) t = fminf(t, l.t);
        dist_t r = traverse_rec(nodes, verts_xyz, indices, orig, dir, node->left_first + 1u, t, depth + 1u, overflow);
        if (r.hit) t = fminf(t, r.t);
    }
    dist_t h = {1, t};
    return h;                                                    /* blas.rs:244 */
}

int vd_ref_traverse(const VdBvhNode* nodes, uint32_t n_nodes, const float* verts_xyz, const uint32_t* indices,
                    const VdRay* rays, uint32_t n_rays, float t0, float* out_dist) {
    if (!nodes || !verts_xyz || !indices || (n_rays && (!rays || !out_dist)) || n_nodes == 0) return VD_ERR_INVALID_ARG;
    for (uint32_t r = 0; r < n_rays; ++r) {
        int overflow = 0;
        dist_t d = traverse_rec(nodes, verts_xyz, indices, v3_load(rays[r].eye), v3_load(rays[r].dir), 0u, t0, 0u, &overflow);
        if (overflow) return VD_ERR_STACK_OVERFLOW;
        out_dist[r] = d.hit ? d.t : -1.0f;
    }
    return VD_OK;
}

/* Per-pixel primary rays of the CPU harness (src/bin/bvh_cpu.rs:71-83):
 *   x = (i % WIDTH) / WIDTH; y = (i / HEIGHT) / HEIGHT; (x, y) = ((x, y) - 0.5) * (2, -2);
 *   view_pos = clip_to_world * (x, y, 1, 1); view_tang = clip_to_world * (x, y, 0, 1);
 *   eye = view_pos.xyz / view_pos.w; dir = view_tang.xyz.normalize()
 * The row of pixel i is i / HEIGHT as the source has it (WIDTH == HEIGHT == 640 there).  glam 0.24.1 (not on disk,
 * from memory): Mat4 * Vec4 = ((c0*x + c1*y) + c2*z) + c3*w; Vec3::normalize = v * (1 / sqrt((x*x + y*y) + z*z)).
 * The fragment-shader harness (src/bin/bvh_trace.wgsl:225-234) builds the same rays from rasteriser-interpolated
 * uv and WGSL normalize, neither of which is specified to the bit; this follows the Rust source. */
int vd_ref_primary_rays(const VdCameraUniform* camera, uint32_t width, uint32_t height, VdRay* out) {
    const size_t n = (size_t)width * height;
    if (!camera || (n != 0 && !out)) return VD_ERR_INVALID_ARG;
    const float* M = camera->clip_to_world;
    for (size_t i = 0; i < n; ++i) {
        float x = (float)(i % width) / (float)width;
        float y = (float)(i / height) / (float)height;
        x = (x - 0.5f) * 2.0f;
        y = (y - 0.5f) * -2.0f;
        float p[4], t[4];
        for (int r = 0; r < 4; ++r) {
            p[r] = ((M[r] * x + M[4 + r] * y) + M[8 + r] * 1.0f) + M[12 + r] * 1.0f;
            t[r] = ((M[r] * x + M[4 + r] * y) + M[8 + r] * 0.0f) + M[12 + r] * 1.0f;
        }
        const float rl = 1.0f / sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
        memset(&out[i], 0, sizeof(out[i]));
        for (int k = 0; k < 3; ++k) {
            out[i].eye[k] = p[k] / p[3];
            out[i].dir[k] = t[k] * rl;
        }
    }
    return VD_OK;
}

/* src/bin/raytraced_shadows.wgsl:90 `let light_vec = light.position - pos;` and :97
 * `ray_new(pos + nor * 0.0001, light_vec)`; the pass reads only `.hit` of the traversal (:98-101). */
int vd_ref_shadow_rays(const float* positions, const float* normals, uint32_t n_points, const float* light_position,
                       VdRay* out) {
    if (n_points && (!positions || !normals || !light_position || !out)) return VD_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n_points; ++i) {
        memset(&out[i], 0, sizeof(out[i]));
        for (int k = 0; k < 3; ++k) {
            const float p = positions[3 * (size_t)i + k];
            const float off = normals[3 * (size_t)i + k] * 0.0001f;
            out[i].eye[k] = p + off;
            out[i].dir[k] = light_position[k] - p;
        }
    }
    return VD_OK;
}

// trace.hip — batch TLAS -> instance -> BLAS ray traversal on gfx950.
//
// Replaces `traverse_tlas(ray)` (reference: shaders/utils/bvh.wgsl:89-123) and its callees
// `instance_intersect` (bvh.wgsl:78-87), `traverse_bvh` (bvh.wgsl:35-76), `fetch_vertex`
// (bvh.wgsl:30-33), `intersect_aabb` / `intersect_trig` (shaders/utils/intersections.wgsl:13-45).
// One lane per ray; the arithmetic order is the WGSL source order with no FMA, so hit distances
// are reproduced to the bit on the oracle's evaluation model (tolerance in tests: 1e-5).
// The reference's 24-entry stack is unchecked (shaders/utils/stack.wgsl:1-20); here the stack is
// 64 deep per traversal level and overflow is reported, not ignored.
#include "vd_common.hpp"

namespace {

#ifndef VD_TRACE_STACK
#define VD_TRACE_STACK 64
#endif
constexpr int kStack = VD_TRACE_STACK;
constexpr float kMaxDist = 1e30f;

struct Ray { float ex, ey, ez, dx, dy, dz, ix, iy, iz; };

__device__ __forceinline__ float min3(float a, float b, float c) { return fminf(a, fminf(b, c)); }   // math.wgsl:19-21
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }   // math.wgsl:23-25

// intersections.wgsl:13-23
__device__ __forceinline__ float intersect_aabb(const Ray& r, const float* mn, const float* mx, float t) {
    const float ax = (mn[0] - r.ex) * r.ix, ay = (mn[1] - r.ey) * r.iy, az = (mn[2] - r.ez) * r.iz;
    const float bx = (mx[0] - r.ex) * r.ix, by = (mx[1] - r.ey) * r.iy, bz = (mx[2] - r.ez) * r.iz;
    const float tmax = min3(fmaxf(ax, bx), fmaxf(ay, by), fmaxf(az, bz));
    const float tmin = max3(fminf(ax, bx), fminf(ay, by), fminf(az, bz));
    return (tmax >= tmin && tmin < t && tmax > 0.0f) ? tmin : kMaxDist;
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;
}

// intersections.wgsl:25-45 (Moeller-Trumbore, backface cull det < 1e-10)
__device__ __forceinline__ bool intersect_trig(const Ray& r, const float* v0, const float* v1, const float* v2, float& hit) {
    const float e1x = v1[0] - v0[0], e1y = v1[1] - v0[1], e1z = v1[2] - v0[2];
    const float e2x = v2[0] - v0[0], e2y = v2[1] - v0[1], e2z = v2[2] - v0[2];
    // cross(dir, edge2)
    const float ux = r.dy * e2z - e2y * r.dz, uy = r.dz * e2x - e2z * r.dx, uz = r.dx * e2y - e2x * r.dy;
    const float det = dot3(e1x, e1y, e1z, ux, uy, uz);
    if (det < 1e-10f) return false;
    const float inv_det = 1.0f / det;
    const float ox = r.ex - v0[0], oy = r.ey - v0[1], oz = r.ez - v0[2];
    const float u = inv_det * dot3(ox, oy, oz, ux, uy, uz);
    if (u < 0.0f || 1.0f < u) return false;
    // cross(orig, edge1)
    const float vx = oy * e1z - e1y * oz, vy = oz * e1x - e1z * ox, vz = ox * e1y - e1x * oy;
    const float v = inv_det * dot3(r.dx, r.dy, r.dz, vx, vy, vz);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = inv_det * dot3(e2x, e2y, e2z, vx, vy, vz);
    if (t > 0.0f && t < hit) {
        hit = t;
        return true;
    }
    return false;
}

struct Scene {
    const VdTlasNode* tlas; const VdInstance* inst; const VdMeshInfo* meshes; const VdBvhNode* bvh;
    const float* verts; const unsigned* indices; unsigned n_meshes;
};

// ANY: occlusion query - the lane stops at the first accepted triangle and only `hit` is reported.  That flag is
// the same as the closest-hit traversal's: until something is accepted nothing is pruned by distance, so both walks
// visit the same nodes up to that point (the reference's shadow pass uses only `.hit`: raytraced_shadows.wgsl:97-102).
template <bool ANY>
__global__ __launch_bounds__(64) void trace_kernel(Scene s, const VdRay* __restrict__ rays, unsigned n_rays,
                                                   VdHit* __restrict__ out, unsigned* __restrict__ out_any,
                                                   unsigned* __restrict__ overflow) {
    const unsigned i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n_rays) return;
    // a stack entry is the popped node's payload, packed into one word: its box is never looked at again
    // (bvh.wgsl:45-47, 96-98), so a pop costs no node fetch.  The stacks live in scratch memory, and the scratch a
    // wave needs bounds how many waves the runtime keeps resident: 512 B per lane instead of 1 KB.
    //   BLAS: count (<= 3... any u2 would do, kept as 2 bits) << 30 | left_first (n_tri <= 2^30 - 1)
    //   TLAS: interior = left_right (its low half, the left child, is never 0: node 0 is only the root copy);
    //         leaf = instance_idx << 16 (low half 0)
    unsigned tstack[kStack], bstack[kStack];
    Ray ray;
    {
        const float4 a = reinterpret_cast<const float4*>(rays + i)[0], b = reinterpret_cast<const float4*>(rays + i)[1];
        ray.ex = a.x; ray.ey = a.y; ray.ez = a.z; ray.dx = b.x; ray.dy = b.y; ray.dz = b.z;
        ray.ix = 1.0f / ray.dx; ray.iy = 1.0f / ray.dy; ray.iz = 1.0f / ray.dz;   // ray_new: inv_dir = 1. / dir
    }
    VdHit res; res.dist = kMaxDist; res.hit = 0u; res.instance = 0xffffffffu; res.triangle = 0xffffffffu;
    // The reference pushes the near child last and pops it straight away (bvh.wgsl:66-74, 113-121).  Here the near
    // child stays in registers - including the payload that was fetched with its box - and only the far child
    // touches the (scratch-memory) stack: same visiting order, one dependent fetch per step instead of two.
    bool ovf = false;
    unsigned thead = 0;
    uint2 tn;                                                      // {left_right, instance_idx} of the current node
    { const VdTlasNode root = s.tlas[0]; tn.x = root.left_right; tn.y = root.instance_idx; }
    for (;;) {                                                     // bvh.wgsl:94
        bool pop = true;
        if (tn.x == 0u) {                                          // leaf: instance_intersect (bvh.wgsl:78-87)
            const unsigned instance_idx = tn.y;
            const VdInstance* I = s.inst + instance_idx;
            const unsigned mesh_id = min(I->mesh, s.n_meshes - 1u);
            const VdMeshInfo mesh = s.meshes[mesh_id];
            const float* M = I->inv_transform;
            Ray nr;
            nr.ex = ((M[0] * ray.ex + M[4] * ray.ey) + M[8] * ray.ez) + M[12] * 1.0f;
            nr.ey = ((M[1] * ray.ex + M[5] * ray.ey) + M[9] * ray.ez) + M[13] * 1.0f;
            nr.ez = ((M[2] * ray.ex + M[6] * ray.ey) + M[10] * ray.ez) + M[14] * 1.0f;
            nr.dx = ((M[0] * ray.dx + M[4] * ray.dy) + M[8] * ray.dz) + M[12] * 0.0f;
            nr.dy = ((M[1] * ray.dx + M[5] * ray.dy) + M[9] * ray.dz) + M[13] * 0.0f;
            nr.dz = ((M[2] * ray.dx + M[6] * ray.dy) + M[10] * ray.dz) + M[14] * 0.0f;
            nr.ix = 1.0f / nr.dx; nr.iy = 1.0f / nr.dy; nr.iz = 1.0f / nr.dz;
            // traverse_bvh (bvh.wgsl:35-76)
            unsigned bhead = 0;
            uint2 bn;                                              // {left_first, count} of the current node
            { const VdBvhNode root = s.bvh[mesh.bvh_index]; bn.x = root.left_first; bn.y = root.count; }
            float hit = res.dist;
            for (;;) {
                bool bpop = true;
                if (bn.y > 0u) {
                    for (unsigned k = 0; k < bn.y; ++k) {
                        const unsigned idx = bn.x + k;
                        const unsigned i0 = (unsigned)mesh.vertex_offset + s.indices[mesh.base_index + 3u * idx + 0u];
                        const unsigned i1 = (unsigned)mesh.vertex_offset + s.indices[mesh.base_index + 3u * idx + 1u];
                        const unsigned i2 = (unsigned)mesh.vertex_offset + s.indices[mesh.base_index + 3u * idx + 2u];
                        const float* v0 = s.verts + 3u * (size_t)i0;
                        const float* v1 = s.verts + 3u * (size_t)i1;
                        const float* v2 = s.verts + 3u * (size_t)i2;
                        const float a0[3] = {v0[0], v0[1], v0[2]}, a1[3] = {v1[0], v1[1], v1[2]}, a2[3] = {v2[0], v2[1], v2[2]};
                        if (intersect_trig(nr, a0, a1, a2, hit)) {
                            res.dist = hit; res.hit = 1u; res.instance = instance_idx; res.triangle = idx;
                            if (ANY) break;
                        }
                    }
                } else {
                    const unsigned pair = mesh.bvh_index + bn.x;
                    const VdBvhNode c0 = s.bvh[pair], c1 = s.bvh[pair + 1u];
                    float min_dist = intersect_aabb(nr, c0.min, c0.max, hit);
                    float max_dist = intersect_aabb(nr, c1.min, c1.max, hit);
                    uint2 near = make_uint2(c0.left_first, c0.count), far = make_uint2(c1.left_first, c1.count);
                    if (min_dist > max_dist) {
                        const uint2 tu = near; near = far; far = tu;
                        const float tf = min_dist; min_dist = max_dist; max_dist = tf;
                    }
                    if (!(min_dist >= hit)) {
                        if (max_dist <= hit) {
                            if (bhead + 1u > (unsigned)kStack) { ovf = true; break; }
                            bstack[bhead++] = far.x | (far.y << 30);
                        }
                        bn = near;
                        bpop = false;
                    }
                }
                if (ANY && res.hit) break;
                if (bpop) {
                    if (bhead == 0u) break;
                    { const unsigned w = bstack[--bhead]; bn = make_uint2(w & 0x3fffffffu, w >> 30); }
                }
            }
            if (ovf || (ANY && res.hit)) break;
        } else {
            const VdTlasNode c0 = s.tlas[tn.x & 0xffffu], c1 = s.tlas[tn.x >> 16u];
            float min_dist = intersect_aabb(ray, c0.min, c0.max, res.dist);
            float max_dist = intersect_aabb(ray, c1.min, c1.max, res.dist);
            uint2 near = make_uint2(c0.left_right, c0.instance_idx), far = make_uint2(c1.left_right, c1.instance_idx);
            if (min_dist > max_dist) {
                const uint2 tu = near; near = far; far = tu;
                const float tf = min_dist; min_dist = max_dist; max_dist = tf;
            }
            if (!(min_dist >= res.dist)) {
                if (max_dist < res.dist) {
                    if (thead + 1u > (unsigned)kStack) { ovf = true; break; }
                    tstack[thead++] = far.x != 0u ? far.x : (far.y << 16);
                }
                tn = near;
                pop = false;
            }
        }
        if (pop) {
            if (thead == 0u) break;
            { const unsigned w = tstack[--thead]; tn = (w & 0xffffu) ? make_uint2(w, 0xffffffffu) : make_uint2(0u, w >> 16); }
        }
    }
    if (ovf) atomicOr(overflow, 1u);
    if (ANY) out_any[i] = res.hit; else out[i] = res;
}

// Shadow rays of the reference's deferred pass (src/bin/raytraced_shadows.wgsl:97): origin = pos + nor * 0.0001,
// dir = light.position - pos (not normalised: t is in units of the light vector, and the pass ignores it).
__global__ __launch_bounds__(256) void shadow_rays_kernel(const float* __restrict__ pos, const float* __restrict__ nor, unsigned n,
                                                          float lx, float ly, float lz, VdRay* __restrict__ rays) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float px = pos[3u * i], py = pos[3u * i + 1u], pz = pos[3u * i + 2u];
    VdRay r;
    r.eye[0] = px + nor[3u * i] * 0.0001f; r.eye[1] = py + nor[3u * i + 1u] * 0.0001f; r.eye[2] = pz + nor[3u * i + 2u] * 0.0001f;
    r._pad0 = 0.0f;
    r.dir[0] = lx - px; r.dir[1] = ly - py; r.dir[2] = lz - pz;
    r._pad1 = 0.0f;
    rays[i] = r;
}

int launch_trace(VdCtx* ctx, const VdTraceScene* sc, const VdRay* d_rays, uint32_t n_rays, VdHit* d_out, uint32_t* d_any = nullptr) {
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, 256);
    if (rc) return rc;
    unsigned* d_flag = reinterpret_cast<unsigned*>(ctx->scratch);
    Scene s{sc->tlas_nodes, sc->instances, sc->meshes, sc->bvh_nodes, sc->vertices, sc->indices, sc->n_meshes};
    vd_time_begin(ctx);
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_flag, 0, 16, ctx->stream));
    if (d_any) hipLaunchKernelGGL(trace_kernel<true>, dim3((n_rays + 63) / 64), dim3(64), 0, ctx->stream, s, d_rays, n_rays, d_out, d_any, d_flag);
    else hipLaunchKernelGGL(trace_kernel<false>, dim3((n_rays + 63) / 64), dim3(64), 0, ctx->stream, s, d_rays, n_rays, d_out, d_any, d_flag);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->host_pinned, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->host_pinned[0]) VD_FAIL(ctx, VD_ERR_STACK_OVERFLOW, "vd_trace: traversal stack (64) exceeded");
    return VD_OK;
}

bool scene_ok(const VdTraceScene* s) {
    return s && s->tlas_nodes && s->instances && s->meshes && s->bvh_nodes && s->vertices && s->indices && s->n_meshes &&
           s->n_tlas_nodes && s->n_instances;
}

}  // namespace

extern "C" {

int vd_trace_dev(VdCtx* ctx, const VdTraceScene* d_scene, const VdRay* d_rays, uint32_t n_rays, VdHit* d_out) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!scene_ok(d_scene)) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace: incomplete scene");
    if (n_rays == 0) return VD_OK;
    if (!d_rays || !d_out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace: null rays/out");
    return launch_trace(ctx, d_scene, d_rays, n_rays, d_out);
}

int vd_trace_any_dev(VdCtx* ctx, const VdTraceScene* d_scene, const VdRay* d_rays, uint32_t n_rays, uint32_t* d_out_hit) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!scene_ok(d_scene)) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace_any: incomplete scene");
    if (n_rays == 0) return VD_OK;
    if (!d_rays || !d_out_hit) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace_any: null rays/out");
    return launch_trace(ctx, d_scene, d_rays, n_rays, nullptr, d_out_hit);
}

int vd_shadow_rays_dev(VdCtx* ctx, const float* d_positions, const float* d_normals, uint32_t n_points, const float* light_position,
                       VdRay* d_rays) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (n_points == 0) return VD_OK;
    if (!d_positions || !d_normals || !light_position || !d_rays) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_shadow_rays: null pointer");
    hipLaunchKernelGGL(shadow_rays_kernel, dim3((n_points + 255) / 256), dim3(256), 0, ctx->stream, d_positions, d_normals, n_points,
                       light_position[0], light_position[1], light_position[2], d_rays);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_trace(VdCtx* ctx, const VdTraceScene* scene, const VdRay* rays, uint32_t n_rays, VdHit* out) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!scene_ok(scene)) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace: incomplete scene");
    if (n_rays == 0) return VD_OK;
    if (!rays || !out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace: null rays/out");
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // one staging arena, sub-allocated at 256-B boundaries
    const size_t sz[8] = {(size_t)scene->n_tlas_nodes * sizeof(VdTlasNode), (size_t)scene->n_instances * sizeof(VdInstance),
                          (size_t)scene->n_meshes * sizeof(VdMeshInfo), (size_t)scene->n_bvh_nodes * sizeof(VdBvhNode),
                          (size_t)scene->n_vertices * 12, (size_t)scene->n_indices * 4, (size_t)n_rays * sizeof(VdRay),
                          (size_t)n_rays * sizeof(VdHit)};
    const void* src[7] = {scene->tlas_nodes, scene->instances, scene->meshes, scene->bvh_nodes, scene->vertices, scene->indices, rays};
    size_t off[8], total = 0;
    for (int k = 0; k < 8; ++k) { off[k] = total; total += (sz[k] + 255) & ~(size_t)255; }
    int rc = vd_ensure(ctx, &ctx->stage_in, &ctx->stage_in_bytes, total);
    if (rc) return rc;
    char* base = reinterpret_cast<char*>(ctx->stage_in);
    for (int k = 0; k < 7; ++k)
        if (sz[k]) VD_HIP_CHECK(ctx, hipMemcpyAsync(base + off[k], src[k], sz[k], hipMemcpyHostToDevice, ctx->stream));
    VdTraceScene d = *scene;
    d.tlas_nodes = reinterpret_cast<const VdTlasNode*>(base + off[0]);
    d.instances = reinterpret_cast<const VdInstance*>(base + off[1]);
    d.meshes = reinterpret_cast<const VdMeshInfo*>(base + off[2]);
    d.bvh_nodes = reinterpret_cast<const VdBvhNode*>(base + off[3]);
    d.vertices = reinterpret_cast<const float*>(base + off[4]);
    d.indices = reinterpret_cast<const uint32_t*>(base + off[5]);
    VdHit* d_out = reinterpret_cast<VdHit*>(base + off[7]);
    rc = launch_trace(ctx, &d, reinterpret_cast<const VdRay*>(base + off[6]), n_rays, d_out);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(out, d_out, sz[7], hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VD_OK;
}

}  // extern "C"

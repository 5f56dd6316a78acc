// trace.hip — batch TLAS -> instance -> BLAS ray traversal on gfx950.
//
// Replaces `traverse_tlas(ray)` (reference: shaders/utils/bvh.wgsl:89-123) and its callees
// `instance_intersect` (bvh.wgsl:78-87), `traverse_bvh` (bvh.wgsl:35-76), `fetch_vertex`
// (bvh.wgsl:30-33), `intersect_aabb` / `intersect_trig` (shaders/utils/intersections.wgsl:13-45).
// One lane per ray at a time, persistent waves that refill (see trace_body); the arithmetic order is the WGSL
// source order with no FMA, so hit distances are reproduced to the bit on the oracle's evaluation model (tolerance
// in tests: 1e-5).  The reference's 24-entry stack is unchecked (shaders/utils/stack.wgsl:1-20); here one 128-entry
// stack serves the TLAS and the BLAS walk of a ray, and a ray that needs more is walked AGAIN, from its start, by a second
// pass whose stack goes on in a slab of global memory (launch_trace: the call is total - VD_ERR_STACK_OVERFLOW is left for
// scenes whose trees are deeper than the slab the context may allocate).  Leaves of more than 3
// triangles do not occur (BvhBuilder stops at <= 3: blas.rs:108) and are not representable in a stack entry.
#include "vd_common.hpp"

#include <algorithm>
#include <new>

// vd_trace_prepare_dev: per-scene data derived once from the six trace buffers
struct VdTraceAccel {
    VdTraceScene scene; float* tris = nullptr;
    // private top level (VD_OPT_TRACE_TIGHT_TLAS): its nodes, which builder made it (1 agglomerative, 2 LBVH), the scene's own top
    // level (what the walk goes back to when an update finds an instance that does not qualify), work memory of the builders
    VdTlasNode* tight = nullptr; unsigned tight_mode = 0, tight_fallbacks = 0;
    const VdTlasNode* user_tlas = nullptr; unsigned user_n_nodes = 0;
    float* boxes = nullptr; unsigned* lbvh = nullptr;
};

namespace {

constexpr int kStack = 64;
constexpr int kLdsStack = 24;            // stack entries per ray kept in LDS: 6 KB per wave, 24 waves per CU = 144 of the 160 KB
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
    const float* tris;          // vd_trace_prepare_dev: 9 floats per triangle in index-buffer order (nullptr: not prepared)
    const float4* irec;         // entry records, 4 x float4 per TLAS node (records_kernel)
    const float4* tpair;        // child pairs of the TLAS, 4 x float4 per node, at the LEFT child's index
    const float4* mrec;         // mesh records, 8 x float4 per mesh
    unsigned yield;             // waiting lanes at which a wave leaves the stepping loop (VD_OPT_TRACE_YIELD)
    unsigned* ovf_bits = nullptr;   // first pass: bit r is set when ray r ran out of its 128 entries (the second pass walks those again)
    unsigned* deep = nullptr;       // second pass (DEEP): entries 128.. of a lane's stack, [wave][entry][lane]
    unsigned deep_cap = 0;          // ... and how many there are per lane
};

// Where a wave's rays come from.  `order` (nullptr = identity) lists the ray ids in the order they are handed out (the
// optional binning pre-pass sorts them by origin cell and direction).  chunk == 1 (default): idle lanes draw single
// rays from one global counter.  chunk >= 64: rays are handed out in CHUNKS of consecutive positions, a workgroup (several
// waves = one CU's L1) working one chunk off before it takes the next, so that the lanes of a wave and the waves of a CU
// sit in a small window of the order.  Both were built to test whether locality pays on this part; it does not
// (launch_trace), and the defaults are the round-2 supply.
// Fan-out (single-ray supply, see kFan*): one job = one TLAS subtree of one ray.  48 bytes: as a pending job {ray, the distance
// found so far, ray id, order key, the subtree as a stack-entry word}; when finished the same slot holds the job's result.
struct FanJob { float ex, ey, ez, dx, dy, dz; float lim; unsigned ray_id, key, node, state, tri; };
static_assert(sizeof(FanJob) == 48, "FanJob is three uint4");
enum : unsigned { FAN_PENDING = 0u, FAN_MISS = 1u, FAN_HIT = 2u, FAN_NULL = 3u };
struct Fan {                       // one launch's place in the fan-out (all nullptr / 0: a plain call)
    FanJob* jobs; const unsigned* begin; const unsigned* end;      // this launch's supply = jobs[*begin, *end) instead of fresh rays (end == nullptr: fresh rays)
    unsigned* append; unsigned cap;                                // where new jobs go: jobs[atomicAdd(append, n)], n <= cap
    unsigned long long* best;                                      // per ray: min over finished jobs of {distance bits, key}
    unsigned below, level;                                         // fan out when fewer than `below` rays are alive in a draining wave (0: never); key digit position
};
struct RaySource { const unsigned* order; unsigned n_rays, chunk, n_chunks; unsigned* next_chunk; Fan fan = {}; };

// Fan-out.  The stress scene's call is as long as its longest rays - 8 500 dependent steps of ~1.4 us against a mean of 517 -
// and for most of it a few waves each nurse a handful of such rays while the rest of the machine is idle (the "relay" that
// only repacked whole rays into full waves made it slower: profiles/r04_trace_relay_experiment.log).  A ray's remaining work
// IS its stack: every entry is a TLAS subtree still to be visited, and visits of different subtrees do not depend on each
// other except through the distance found so far.  So a call runs as a few launches: once the rays are handed out, a wave
// that is down to fewer than kFanBelow live rays turns each of them (when it is between instances: the stack then holds TLAS
// entries only) into jobs - one for the node it is at, one per stack entry - that start from the distance the ray has
// found so far, and ends; the next launch's waves draw jobs like rays, 64 per wave, and may fan out again.
// Same result, bit for bit: the walk's answer is the minimum of (t, visit order) over the triangles the ray truly
// intersects - pruning by the current distance only ever skips candidates that cannot beat it, and a later candidate at an
// equal t loses (intersections.wgsl:40 `t < hit`).  Visit order over the tree is fixed by the ray alone (near child first,
// by slab distance), so it is carried as a key: the ray's own result so far is digit 0, the node it is at digit 1, its stack
// entries top to bottom digits 2, 3, ... - one digit per launch that fanned the job out - and a job accepts only t strictly
// below the distance it started from, so an equal t never beats anything found before the fan-out.  Each finished job
// records {distance bits, key} with an atomic min per ray; vd_fan_resolve_kernel lets the job that holds the minimum write the
// ray's record.  (The reference's root is visited twice - tlas.rs:59 merges the true root with itself - and the second
// visit can only tie: its job is not created.)
#ifndef VD_FAN_PHASES
#define VD_FAN_PHASES 3
#endif
#ifndef VD_FAN_BELOW
#define VD_FAN_BELOW 32
#endif
#ifndef VD_FAN_GRACE
#define VD_FAN_GRACE 128
#endif
#ifndef VD_FAN_AGE
#define VD_FAN_AGE 512
#endif
constexpr unsigned kFanPhases = VD_FAN_PHASES;   // launches per call (VD_OPT_TRACE_FAN; 1 = one launch, no fan-out)
constexpr unsigned kFanBelow = VD_FAN_BELOW;     // live rays below which a draining wave fans its rays out at once
constexpr unsigned kFanGrace = VD_FAN_GRACE;     // stepping iterations after the last draw before a wave fans out whatever is still alive
constexpr unsigned kFanAgain = 8;             // ... and between two looks at the rays that were inside an instance at the time
constexpr unsigned kFanAge = VD_FAN_AGE;      // only rays that have been stepping for at least this many of the wave's iterations fan out (0: all)

// A fixed grid of waves; a lane whose ray is finished draws the next ray from a counter, so a wave stays full while
// rays of very different cost (a few node visits to thousands) pass through it.  To let a lane restart at any point the
// TLAS walk and the per-instance BLAS walk are ONE state machine over ONE stack (TLAS and BLAS nodes share the 32-byte
// {min, u32, max, u32} layout).  Every busy lane makes ONE fetch per iteration of the stepping loop, whatever it is
// doing - an interior node's child pair, the entry record of the instance it enters, a de-indexed triangle - into one set
// of registers (see the loop).  Each ray still sees exactly the reference's sequence of node visits and triangle tests
// (bvh.wgsl:35-123); only the interleaving across lanes changes.
// ANY: occlusion query - the lane stops at the first accepted triangle and only `hit` is reported.  That flag is
// the same as the closest-hit traversal's: until something is accepted nothing is pruned by distance, so both walks
// visit the same nodes up to that point (the reference's shadow pass uses only `.hit`: raytraced_shadows.wgsl:97-102).
constexpr long long kYieldDefault = 1;   // see the stepping loop (sweep in profiles/r03_ab_trace.log: 1 is best)
constexpr unsigned kRefillBelow = 56;   // draw new rays when fewer than this many lanes are busy
constexpr unsigned kWavesPerCu = 24;    // persistent grid = what is resident (6 waves per SIMD at <= 84 VGPRs: the walk holds ~76, and with 72 it spilled): no wave starts late
constexpr int kWgWaves = 6;             // waves per workgroup of the chunked form: 4 workgroups per CU
// PREP: leaf triangles come de-indexed from Scene::tris (one contiguous fetch instead of indices[] -> verts[]).
// CHUNKS = false (default): idle lanes draw single rays from one global counter, one wave per workgroup - the round-2
// form, kept apart so that it carries none of the chunk machinery (inside multi-wave workgroups with the chunk state live
// the closest-hit walk spilled registers and lost 10 %: 38.5 -> 34.5 Mrays/s, same-session A/B against the round-2 library).
// DEEP: the second pass over the rays whose stack overflowed (launch_trace) - same walk, entries 128.. in s.deep.
template <bool ANY, bool PREP, bool CHUNKS, bool FAN = false, bool DEEP = false>
__device__ __forceinline__ void trace_body(const Scene& s, const VdRay* __restrict__ rays, const RaySource& src, VdHit* __restrict__ out,
                                           unsigned* __restrict__ out_any, unsigned* __restrict__ overflow) {
    const unsigned lane = threadIdx.x & 63u;
    const bool from_jobs = FAN && src.fan.end != nullptr;                       // this launch hands out jobs, not fresh rays
    const unsigned job_begin = from_jobs ? *src.fan.begin : 0u;
    const unsigned n_rays = from_jobs ? *src.fan.end - job_begin : src.n_rays;     // what this launch hands out
    bool fan_off = false;                  // wave-uniform: the job list was full when this wave wanted to fan out
    unsigned drain_iters = 0;              // wave-uniform: stepping iterations since this wave found the supply empty
    unsigned wave_iters = 0;               // wave-uniform: stepping iterations of this wave (FAN: a ray's age = now - s_born[lane])
    __shared__ unsigned s_born[(FAN && kFanAge) ? 64 : 1];
    __shared__ vd_u64 s_word;              // chunked supply: {next, end} positions of the workgroup's current chunk
    unsigned p_next = 0, p_end = 0;        // a chunk this wave could not publish (another wave's was installed first)
    if (CHUNKS) {
        if (threadIdx.x == 0) s_word = 0ull;
        __syncthreads();
    }
    // The first kLdsStack entries of a ray's stack live in LDS ([slot][lane]: a lane always hits its own bank), the rest in
    // the private array.  A private array indexed by a per-lane depth is scratch memory, and 64 lanes at 64 depths are 64
    // lines per push and per pop - as many as the node fetch itself, out of the same L1 miss bandwidth that bounds the walk.
    __shared__ unsigned s_stack[CHUNKS ? kWgWaves : 1][kLdsStack][64];
    const unsigned wv = CHUNKS ? (threadIdx.x >> 6) : 0u;     // (indexed, not through a pointer: a 64-bit pointer was spilled and reloaded at every push)
    unsigned stack[2 * kStack - kLdsStack];  // BLAS entries sit above the TLAS entries of the same ray
    const size_t deep_base = DEEP ? (size_t)blockIdx.x * s.deep_cap * 64u + lane : 0u;       // one wave per workgroup in the DEEP kernel
    Ray world, ray;                        // `ray` is the active one (object space inside an instance)
    VdHit res;
    unsigned ray_id = 0;
    // ((st & kBusy) != 0u) / ((st & kDone) != 0u) / ((st & kInBlas) != 0u) live as bits of one per-lane word: as `bool`s they are lane masks in scalar registers, and every
    // assignment under divergent control flow is mask algebra (550 of the loop's 1060 instructions were s_and / s_or / s_andn2)
    unsigned st = 0;
    constexpr unsigned kBusy = 1u, kDone = 2u, kInBlas = 4u, kOvf = 8u, kBadLeaf = 16u, kBadEntry = 32u;
    bool exhausted = false;               // wave-uniform
    unsigned head = 0, blas_base = 0;
    unsigned tl_leaf = 0, bvh_index = 0, base_index = 0, vertex_offset = 0;   // tl_leaf: the TLAS leaf node the ray is inside
    uint2 cn = make_uint2(0u, 0u);         // payload of the current node
#ifdef VD_TUNING
    unsigned dbg_outer = 0, dbg_iter = 0, dbg_lanes = 0, dbg_kind[3] = {0, 0, 0}, dbg_ray_steps = 0, dbg_ray_max = 0, dbg_drain = 0, dbg_lone = 0, dbg_few = 0;
    // timeline (words 16..63 of the flag block): [16,17] = earliest wave start (100 MHz ticks), [18..40] = waves that ended in
    // each 0.5 ms slot, [41..63] = stepping iterations done in each slot
    if (lane == 0) atomicMin(reinterpret_cast<unsigned long long*>(overflow + 16), (unsigned long long)wall_clock64());
    unsigned dbg_slot_iter = 0, dbg_slot = 0;
#endif

    auto pop = [&]() {                     // leave the current node
        if (((st & kInBlas) != 0u) && head == blas_base) { st &= ~kInBlas; ray = world; }
        if (head == 0u) { st |= kDone; return; }
        --head;
        unsigned w;
        if (DEEP && head >= 2u * (unsigned)kStack) w = s.deep[deep_base + (size_t)(head - 2u * (unsigned)kStack) * 64u];
        else w = head < (unsigned)kLdsStack ? s_stack[wv][head][lane] : stack[head - (unsigned)kLdsStack];
        if (((st & kInBlas) != 0u)) cn = make_uint2(w & 0x3fffffffu, w >> 30);
        else cn = (w & 0xffffu) ? make_uint2(w, 0xffffffffu) : make_uint2(0u, w >> 16);
    };
    for (;;) {
        // ---- retire finished rays, refill idle lanes ----
        if (((st & kBusy) != 0u) && ((st & kDone) != 0u)) {
#ifdef VD_TUNING
            dbg_ray_max = max(dbg_ray_max, dbg_ray_steps); dbg_ray_steps = 0;
#endif
            if (FAN && (ray_id & 0x80000000u)) {          // a job: its slot takes the result, the ray's minimum is updated
                FanJob* J = src.fan.jobs + (ray_id & 0x7fffffffu);
                if (res.hit) {
                    const unsigned rid = J->ray_id;
                    if (ANY) out_any[rid] = 1u;
                    else {
                        J->lim = res.dist; J->node = s.tlas[res.instance].instance_idx; J->tri = res.triangle;
                        atomicMin(src.fan.best + rid, ((unsigned long long)__float_as_uint(res.dist) << 32) | J->key);
                    }
                }
                J->state = (res.hit && !ANY) ? FAN_HIT : FAN_MISS;
            } else
            if (ANY) out_any[ray_id] = res.hit;
            else { if (res.hit) res.instance = s.tlas[res.instance].instance_idx; out[ray_id] = res; }   // hits carry the leaf node until here
            st &= ~kBusy;
        }
        const unsigned long long busy_mask = __ballot(((st & kBusy) != 0u));
        if (!exhausted && (unsigned)__popcll(busy_mask) < kRefillBelow) {
            const unsigned long long idle = ~busy_mask;
            const unsigned want = (unsigned)__popcll(idle);
            unsigned base = 0, got = 0, done = 0;
            if (!CHUNKS) {                                          // single rays from one global counter: the finest balance
                if (lane == 0) base = atomicAdd(src.next_chunk, want);
                base = __shfl(base, 0);
                got = want;                                         // ids are checked against n_rays below (limit)
                done = base + want >= n_rays ? 1u : 0u;
            } else {
                if (lane == 0) {
                    if (p_next < p_end) { got = min(want, p_end - p_next); base = p_next; p_next += got; }
                    else for (;;) {
                        const vd_u64 old = __hip_atomic_load(&s_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const unsigned nx = (unsigned)old, en = (unsigned)(old >> 32);
                        if (nx < en) {                                  // the workgroup's chunk still has rays
                            const unsigned take = min(want, en - nx);
                            vd_u64 expect = old;
                            if (__hip_atomic_compare_exchange_strong(&s_word, &expect, ((vd_u64)en << 32) | (nx + take), __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_WORKGROUP)) { base = nx; got = take; break; }
                            continue;
                        }
                        const unsigned c = atomicAdd(src.next_chunk, 1u);
                        if (c >= src.n_chunks) { done = 1u; break; }
                        const unsigned b = c * src.chunk, e = min(b + src.chunk, n_rays);
                        base = b; got = min(want, e - b);
                        vd_u64 expect = old;                            // the rest is for the whole workgroup - unless another wave installed a chunk meanwhile
                        if (!__hip_atomic_compare_exchange_strong(&s_word, &expect, ((vd_u64)e << 32) | (b + got), __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_WORKGROUP)) { p_next = b + got; p_end = e; }
                        break;
                    }
                }
                base = __shfl(base, 0); got = __shfl(got, 0); done = __shfl(done, 0);
            }
            // the position drawn is checked against an explicit upper limit (`pos < limit`), never through a difference: written as
            // `got = base < n_rays ? min(want, n_rays - base) : 0; if (k < got)` the one-wave kernel was compiled WITHOUT the
            // `base < n_rays` guard (v_sub_u32 + v_min_u32, no clamp, no select: profiles/r03_trace_guard_isa.txt) and every
            // draw past the end took `want` rays from beyond the array
            const unsigned limit = CHUNKS ? base + got : n_rays;
            if (!((st & kBusy) != 0u)) {
                const unsigned pos = base + vd_mbcnt(idle);
                if (from_jobs && pos < limit && pos >= base) {
                    const uint4* Jp = reinterpret_cast<const uint4*>(src.fan.jobs + (job_begin + pos));
                    const uint4 j0 = Jp[0], j1 = Jp[1], j2 = Jp[2];
                    const bool skip = j2.z != FAN_PENDING || (ANY && out_any[j1.w] != 0u);      // a filler slot, or the ray is known to be occluded
                    if (skip) { if (j2.z == FAN_PENDING) src.fan.jobs[job_begin + pos].state = FAN_MISS; }
                    else {
                        world.ex = __uint_as_float(j0.x); world.ey = __uint_as_float(j0.y); world.ez = __uint_as_float(j0.z);
                        world.dx = __uint_as_float(j0.w); world.dy = __uint_as_float(j1.x); world.dz = __uint_as_float(j1.y);
                        world.ix = 1.0f / world.dx; world.iy = 1.0f / world.dy; world.iz = 1.0f / world.dz;   // ray_new: inv_dir = 1. / dir
                        ray = world;
                        // only t below the distance found before the fan-out counts - or, when another job of this ray has finished
                        // with a hit meanwhile, t up to and including that distance (an equal t is decided by the keys)
                        float lim = __uint_as_float(j1.z);
                        if (!ANY) {
                            const unsigned bd = (unsigned)(src.fan.best[j1.w] >> 32);
                            if (bd < 0x7f800000u && __uint_as_float(bd + 1u) < lim) lim = __uint_as_float(bd + 1u);      // next float above a positive distance
                        }
                        res.dist = lim; res.hit = 0u; res.instance = 0xffffffffu; res.triangle = 0xffffffffu;
                        const unsigned w = j2.y;                       // the subtree, as the stack held it (pop)
                        cn = (w & 0xffffu) ? make_uint2(w, 0xffffffffu) : make_uint2(0u, w >> 16);
                        ray_id = 0x80000000u | (job_begin + pos); st |= kBusy; st &= ~kDone; st &= ~kInBlas; head = 0; blas_base = 0;
                        if (FAN && kFanAge) s_born[lane] = wave_iters;
                    }
                } else
                if (pos < limit && pos >= base) {
                    const unsigned id = src.order ? src.order[pos] : pos;
                    const float4 a = reinterpret_cast<const float4*>(rays + id)[0], b = reinterpret_cast<const float4*>(rays + id)[1];
                    world.ex = a.x; world.ey = a.y; world.ez = a.z; world.dx = b.x; world.dy = b.y; world.dz = b.z;
                    world.ix = 1.0f / world.dx; world.iy = 1.0f / world.dy; world.iz = 1.0f / world.dz;   // ray_new: inv_dir = 1. / dir
                    ray = world;
                    res.dist = kMaxDist; res.hit = 0u; res.instance = 0xffffffffu; res.triangle = 0xffffffffu;
                    const VdTlasNode root = s.tlas[0];
                    cn = make_uint2(root.left_right, 0u);          // .y of a TLAS leaf = its node index
                    ray_id = id; st |= kBusy; st &= ~kDone; st &= ~kInBlas; head = 0; blas_base = 0;
                    if (FAN && kFanAge) s_born[lane] = wave_iters;
                }
            }
            if (done) exhausted = true;     // wave-uniform
        }
        if (!__ballot(((st & kBusy) != 0u))) break;
        if (FAN && exhausted && src.fan.below != 0u && !fan_off) {
            // nothing left to draw and few rays alive here: every ray that is between instances (its stack holds TLAS entries
            // only) becomes jobs and leaves; rays inside an instance go on and are looked at again when the wave next gets here
            const bool live = (st & kBusy) != 0u;
            if ((unsigned)__popcll(__ballot(live)) < src.fan.below || drain_iters >= kFanGrace) {
                if (drain_iters >= kFanGrace) drain_iters = kFanGrace - kFanAgain;
                const bool can = live && (st & kInBlas) == 0u && (!kFanAge || wave_iters - s_born[kFanAge ? lane : 0u] >= kFanAge);
                // the bottom entry of a ray that came through the reference's root: the true root's second visit (see above)
                unsigned dup_word = 0u;
                {
                    const unsigned lr = s.tlas[0].left_right, l = lr & 0xffffu;
                    if (lr != 0u && l == (lr >> 16)) { const unsigned w2 = s.tlas[l].left_right; dup_word = w2 != 0u ? w2 : (l << 16); }
                }
                const bool dup = can && head != 0u && dup_word != 0u && (ray_id & 0x80000000u) == 0u && s_stack[wv][0][lane] == dup_word;
                const unsigned n_ent = can ? head - (dup ? 1u : 0u) : 0u;
                const unsigned m = can ? n_ent + 2u : 0u;           // the result so far, the node it is at, the stack entries
                unsigned incl = m;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const unsigned v = __shfl_up(incl, off); if (lane >= (unsigned)off) incl += v; }
                const unsigned total = __shfl(incl, 63);
                if (total != 0u) {
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(src.fan.append, total);
                    base = __shfl(base, 0);
                    if (base + total > src.fan.cap) {
                        // list full: nobody of this wave fans out (now or later); the part of the reservation that lies inside
                        // the list is filled with slots that say "nothing here"
                        for (unsigned k = lane; k < total && base + k < src.fan.cap; k += 64u) src.fan.jobs[base + k].state = FAN_NULL;
                        fan_off = true;
                    } else if (can) {
                        unsigned rid = ray_id, pkey = 0u;
                        if (ray_id & 0x80000000u) {                 // a job fans out again: one more digit behind its own key
                            FanJob* P = src.fan.jobs + (ray_id & 0x7fffffffu);
                            rid = P->ray_id; pkey = P->key;
                            P->state = FAN_MISS;                    // its result so far moves to the new slot below
                        }
                        const unsigned shift = 8u * (3u - src.fan.level);
                        uint4* O = reinterpret_cast<uint4*>(src.fan.jobs + (base + incl - m));
                        const uint4 r0 = make_uint4(__float_as_uint(world.ex), __float_as_uint(world.ey), __float_as_uint(world.ez), __float_as_uint(world.dx));
                        const unsigned dyb = __float_as_uint(world.dy), dzb = __float_as_uint(world.dz), limb = __float_as_uint(res.dist);
                        // digit 0: what the ray has found so far, as a finished job
                        const unsigned inst0 = res.hit ? s.tlas[res.instance].instance_idx : 0xffffffffu;
                        O[0] = r0; O[1] = make_uint4(dyb, dzb, limb, rid);
                        O[2] = make_uint4(pkey, inst0, (res.hit && !ANY) ? FAN_HIT : FAN_MISS, res.triangle);
                        if (res.hit && !ANY) atomicMin(src.fan.best + rid, ((unsigned long long)limb << 32) | pkey);
                        if ((ray_id & 0x80000000u) == 0u) {         // a ray's first fan-out: its record so far (a miss unless a job says otherwise)
                            if (ANY) out_any[rid] = res.hit;
                            else { VdHit h = res; h.instance = inst0; out[rid] = h; }
                        }
                        // digit 1: the node the ray is at; digits 2..: the stack, top first
                        O[3] = r0; O[4] = make_uint4(dyb, dzb, limb, rid);
                        O[5] = make_uint4(pkey | (1u << shift), cn.x != 0u ? cn.x : (cn.y << 16), FAN_PENDING, 0u);
                        for (unsigned e = 0; e < n_ent; ++e) {
                            const unsigned at = head - 1u - e;
                            const unsigned w = at < (unsigned)kLdsStack ? s_stack[wv][at][lane] : stack[at - (unsigned)kLdsStack];
                            O[6u + 3u * e] = r0; O[7u + 3u * e] = make_uint4(dyb, dzb, limb, rid);
                            O[8u + 3u * e] = make_uint4(pkey | ((2u + e) << shift), w, FAN_PENDING, 0u);
                        }
                        st &= ~kBusy;
                    }
                    if (!__ballot(((st & kBusy) != 0u))) break;
                }
            }
        }
        // ---- interior steps (bvh.wgsl:56-74 and 104-121) and instance entries (bvh.wgsl:78-87) ----
        // A TLAS leaf is an instance to enter: instance -> inv_transform + mesh -> MeshInfo -> the mesh's root node -> its two
        // children is a chain of four dependent fetches, taken 140 times per ray on the stress scene (2000 overlapping
        // instances; 1.9 BLAS steps per entry: most entries end at the root's children) against 475 interior steps.  The
        // entry record of the leaf (entry_records_kernel: one 128-byte line per TLAS node, rewritten at every call) holds
        // all of it, so an entry is ONE trip to L2 (and a second read of the same line out of L1) and runs the root's step in
        // the same iteration - same values, same arithmetic.
        // The wave leaves the loop to serve the lanes that wait - at a BLAS leaf, or finished - once `yield` of them do
        // (or nobody steps any more): leaves are rare next to steps (13 against 615 per ray on the stress scene), so waiting
        // for every lane to reach one would idle most of the wave, and serving each at once would run the triangle code
        // for one lane at a time.
        const unsigned n_busy = (unsigned)__popcll(__ballot(((st & kBusy) != 0u)));
#ifdef VD_TUNING
        ++dbg_outer;
#endif
        for (;;) {
            // PREP: a lane at a BLAS leaf steps too - its fetch is one de-indexed triangle (36 contiguous bytes), in flight
            // together with the other lanes' node pairs and entry records: one trip to memory per iteration whatever the
            // lanes are doing.  (Indexed leaves are two dependent fetches of another shape: served outside the loop.)
            const bool leaf = ((st & kInBlas) != 0u) && cn.y != 0u;
            const bool stepping = ((st & kBusy) != 0u) && !((st & kDone) != 0u) && (PREP || !leaf);
            const unsigned n_step = (unsigned)__popcll(__ballot(stepping));
            if (n_step == 0u || n_busy - n_step >= s.yield) break;
            if (FAN && kFanAge) wave_iters = (unsigned)__builtin_amdgcn_readfirstlane((int)(wave_iters + 1u));
            if (FAN && exhausted && src.fan.below != 0u && !fan_off) {      // time to fan out what is still alive?
                drain_iters = (unsigned)__builtin_amdgcn_readfirstlane((int)(drain_iters + 1u));      // (kept in a scalar register: it was a spilled VGPR otherwise)
                if (drain_iters >= kFanGrace) break;
            }
#ifdef VD_TUNING
            ++dbg_iter; dbg_lanes += n_step; if (stepping) ++dbg_ray_steps; if (exhausted) ++dbg_drain;
            if ((dbg_iter & 15u) == 0u) {       // every 16 iterations: which 0.5 ms slot are we in
                const unsigned long long t0 = *reinterpret_cast<volatile unsigned long long*>(overflow + 16);
                const unsigned slot = min(22u, (unsigned)((wall_clock64() - t0) / 200000ull));
                if (slot != dbg_slot) { if (lane == 0 && dbg_slot_iter) atomicAdd(overflow + 41 + dbg_slot, dbg_slot_iter); dbg_slot = slot; dbg_slot_iter = 0; }
                dbg_slot_iter += 16u;
            }
            if (n_busy == 1u) ++dbg_lone; else if (n_busy <= 4u) ++dbg_few;
            dbg_kind[0] += (unsigned)__popcll(__ballot(stepping && leaf)); dbg_kind[1] += (unsigned)__popcll(__ballot(stepping && !leaf && !((st & kInBlas) != 0u) && cn.x == 0u));
            dbg_kind[2] += (unsigned)__popcll(__ballot(stepping && !leaf && !((st & kInBlas) != 0u) && cn.x != 0u));
#endif
            if (!stepping) continue;
            const bool enter = !((st & kInBlas) != 0u) && cn.x == 0u;
            const char* p0; const char* p1;
            unsigned idx0 = 0u, idx1 = 0u;
            if (PREP && leaf) {               // triangle cn.x of the mesh: the vertices fetch_vertex (bvh.wgsl:30-33) returns
                p0 = reinterpret_cast<const char*>(s.tris + 9u * ((size_t)(base_index / 3u) + cn.x));
                p1 = p0 + sizeof(VdBvhNode);
            } else if (enter) {               // the leaf's entry record: matrix rows + mesh id
                p0 = reinterpret_cast<const char*>(s.irec + 4u * (size_t)cn.y);
                p1 = p0 + sizeof(VdBvhNode);
            } else if (((st & kInBlas) != 0u)) {
                p0 = reinterpret_cast<const char*>(s.bvh + (bvh_index + cn.x));
                p1 = p0 + sizeof(VdBvhNode);
            } else {
                idx0 = cn.x & 0xffffu; idx1 = cn.x >> 16u;       // both children as ONE line: the pair record at the left child's index
                p0 = reinterpret_cast<const char*>(s.tpair + 4u * (size_t)idx0);
                p1 = p0 + sizeof(VdTlasNode);
            }
            // one set of registers for either kind of lane: an entering lane's matrix rows travel with the other lanes' child pairs
            float4 a0 = reinterpret_cast<const float4*>(p0)[0], a1 = reinterpret_cast<const float4*>(p0)[1];
            float4 b0 = reinterpret_cast<const float4*>(p1)[0], b1 = reinterpret_cast<const float4*>(p1)[1];
            bool do_pop = false;              // one pop site for both kinds of lane
            if (PREP && leaf) {               // bvh.wgsl:48-55, one triangle per iteration, in leaf order
                const float v0[3] = {a0.x, a0.y, a0.z}, v1[3] = {a0.w, a1.x, a1.y}, v2[3] = {a1.z, a1.w, b0.x};
                float hit = res.dist;
                if (intersect_trig(ray, v0, v1, v2, hit)) {
                    res.dist = hit; res.hit = 1u; res.instance = tl_leaf; res.triangle = cn.x;
                    if (ANY) { st |= kDone; continue; }
                }
                if (--cn.y == 0u) do_pop = true; else ++cn.x;
            } else {
            if (enter) {
                // (inv_transform * vec4(eye, 1.)).xyz and (inv_transform * vec4(dir, 0.)).xyz; a0, a1, b0 = rows 0..2 of the matrix
                tl_leaf = cn.y;
                ray.ex = ((a0.x * world.ex + a0.y * world.ey) + a0.z * world.ez) + a0.w * 1.0f;
                ray.ey = ((a1.x * world.ex + a1.y * world.ey) + a1.z * world.ez) + a1.w * 1.0f;
                ray.ez = ((b0.x * world.ex + b0.y * world.ey) + b0.z * world.ez) + b0.w * 1.0f;
                ray.dx = ((a0.x * world.dx + a0.y * world.dy) + a0.z * world.dz) + a0.w * 0.0f;
                ray.dy = ((a1.x * world.dx + a1.y * world.dy) + a1.z * world.dz) + a1.w * 0.0f;
                ray.dz = ((b0.x * world.dx + b0.y * world.dy) + b0.z * world.dz) + b0.w * 0.0f;
                ray.ix = 1.0f / ray.dx; ray.iy = 1.0f / ray.dy; ray.iz = 1.0f / ray.dz;
                if (__float_as_uint(b1.y) != 0u) { st |= kBadEntry; st |= kDone; continue; }   // the leaf's instance lies outside the buffer
                // the mesh's words and its root's children: the mesh record (a handful of lines that stay in L1)
                const float4* Mr = s.mrec + 8u * (size_t)__float_as_uint(b1.x);
                const float4 idv = Mr[0];
                a0 = Mr[4]; a1 = Mr[5]; b0 = Mr[6]; b1 = Mr[7];
                bvh_index = __float_as_uint(idv.x); base_index = __float_as_uint(idv.y); vertex_offset = __float_as_uint(idv.z);
                const unsigned rw = __float_as_uint(idv.w);            // the mesh's root: left_first | count << 30 (traverse_bvh starts there)
                if (rw == 0xffffffffu) { st |= kBadEntry; st |= kDone; continue; }   // mesh root / its children outside the buffer
                st |= kInBlas;
                blas_base = head;
                cn = make_uint2(rw & 0x3fffffffu, rw >> 30);
                if (cn.y != 0u) continue;                                // a mesh of <= 3 triangles: its root is a leaf
            } else if (!((st & kInBlas) != 0u) && __float_as_uint(a1.w) != idx1) {
                // the pair record at idx0 was written for another right child: some unreachable slot of the TLAS array names
                // the same left child (records_kernel) - read the two nodes themselves
                const float4* n0 = reinterpret_cast<const float4*>(s.tlas + idx0);
                const float4* n1 = reinterpret_cast<const float4*>(s.tlas + idx1);
                a0 = n0[0]; a1 = n0[1]; b0 = n1[0]; b1 = n1[1];
            }
            const float mn0[3] = {a0.x, a0.y, a0.z}, mx0[3] = {a1.x, a1.y, a1.z};
            const float mn1[3] = {b0.x, b0.y, b0.z}, mx1[3] = {b1.x, b1.y, b1.z};
            float min_dist = intersect_aabb(ray, mn0, mx0, res.dist);
            float max_dist = intersect_aabb(ray, mn1, mx1, res.dist);
            // payload of a child: BLAS {left_first, count}; TLAS {left_right, its own node index}
            uint2 near = make_uint2(__float_as_uint(a0.w), ((st & kInBlas) != 0u) ? __float_as_uint(a1.w) : idx0);
            uint2 far = make_uint2(__float_as_uint(b0.w), ((st & kInBlas) != 0u) ? __float_as_uint(b1.w) : idx1);
            if (min_dist > max_dist) {
                const uint2 tu = near; near = far; far = tu;
                const float tf = min_dist; min_dist = max_dist; max_dist = tf;
            }
            if (min_dist >= res.dist) do_pop = true;
            else {
            // far child: the BLAS loop keeps it on `<=` (a missed child, 1e30, is pushed while nothing is hit yet),
            // the TLAS loop on `<`
            const bool blas_now = (st & kInBlas) != 0u;
            if (max_dist < res.dist || (blas_now && max_dist == res.dist)) {
                if (head + 1u > 2u * (unsigned)kStack) {
                    if (!DEEP) {          // out of entries: this ray is walked again by the second pass (its record is rewritten there)
                        if (s.ovf_bits) {
                            const unsigned rid = (FAN && (ray_id & 0x80000000u)) ? src.fan.jobs[ray_id & 0x7fffffffu].ray_id : ray_id;
                            atomicOr(s.ovf_bits + (rid >> 5), 1u << (rid & 31u));
                        }
                        st |= kOvf; st |= kDone; continue;
                    }
                    if (head - 2u * (unsigned)kStack >= s.deep_cap) { st |= kOvf; st |= kDone; continue; }
                }
                if (blas_now && far.y > 3u) st |= kBadLeaf;   // not representable in a stack entry
                const unsigned w_blas = far.x | (far.y << 30), w_tlas = far.x != 0u ? far.x : (far.y << 16);
                const unsigned w = blas_now ? w_blas : w_tlas;
                if (DEEP && head >= 2u * (unsigned)kStack) s.deep[deep_base + (size_t)(head - 2u * (unsigned)kStack) * 64u] = w;
                else if (head < (unsigned)kLdsStack) s_stack[wv][head][lane] = w; else stack[head - (unsigned)kLdsStack] = w;
                ++head;
            }
            cn = near;
            }
            }
            if (do_pop) pop();
        }
        if (PREP || !((st & kBusy) != 0u) || ((st & kDone) != 0u) || !(((st & kInBlas) != 0u) && cn.y != 0u)) continue;     // only lanes at an indexed BLAS leaf go on
        {
            // ---- BLAS leaf (bvh.wgsl:48-55) ----
            for (unsigned k = 0; k < cn.y; ++k) {
                const unsigned idx = cn.x + k;
                float a0[3], a1[3], a2[3];
                if (PREP) {          // the same three vertices fetch_vertex (bvh.wgsl:30-33) returns, stored side by side
                    const float* T = s.tris + 9u * ((size_t)(base_index / 3u) + idx);
                    a0[0] = T[0]; a0[1] = T[1]; a0[2] = T[2]; a1[0] = T[3]; a1[1] = T[4]; a1[2] = T[5]; a2[0] = T[6]; a2[1] = T[7]; a2[2] = T[8];
                } else {
                    const unsigned i0 = vertex_offset + s.indices[base_index + 3u * idx + 0u];
                    const unsigned i1 = vertex_offset + s.indices[base_index + 3u * idx + 1u];
                    const unsigned i2 = vertex_offset + s.indices[base_index + 3u * idx + 2u];
                    const float* v0 = s.verts + 3u * (size_t)i0;
                    const float* v1 = s.verts + 3u * (size_t)i1;
                    const float* v2 = s.verts + 3u * (size_t)i2;
                    a0[0] = v0[0]; a0[1] = v0[1]; a0[2] = v0[2]; a1[0] = v1[0]; a1[1] = v1[1]; a1[2] = v1[2]; a2[0] = v2[0]; a2[1] = v2[1]; a2[2] = v2[2];
                }
                float hit = res.dist;
                if (intersect_trig(ray, a0, a1, a2, hit)) {
                    res.dist = hit; res.hit = 1u; res.instance = tl_leaf; res.triangle = idx;
                    if (ANY) break;
                }
            }
            if (ANY && res.hit) st |= kDone; else pop();
        }
    }
#ifdef VD_TUNING
    if (lane == 0) {       // wave totals -> the words after the flag and the ray counter (vd_debug_trace_counters)
        atomicAdd(overflow + 4, dbg_outer); atomicAdd(overflow + 5, dbg_iter);
        atomicAdd(reinterpret_cast<unsigned long long*>(overflow + 6), (unsigned long long)dbg_lanes);
        atomicAdd(overflow + 8, dbg_kind[0]); atomicAdd(overflow + 9, dbg_kind[1]); atomicAdd(overflow + 10, dbg_kind[2]);
        atomicAdd(overflow + 12, dbg_drain);
        atomicMax(overflow + 13, dbg_iter);                 // the wave that iterates longest ...
        atomicMax(overflow + 14, dbg_lone);                 // ... and the longest stretches with one / two to four ((st & kBusy) != 0u) lanes
        atomicMax(overflow + 15, dbg_few);
        if (dbg_slot_iter) atomicAdd(overflow + 41 + dbg_slot, dbg_slot_iter);
        const unsigned long long t0 = *reinterpret_cast<volatile unsigned long long*>(overflow + 16);
        atomicAdd(overflow + 18 + min(22u, (unsigned)((wall_clock64() - t0) / 200000ull)), 1u);
    }
    atomicMax(overflow + 11, dbg_ray_max);
#endif
    if (st & kOvf) atomicOr(overflow, 1u);
    if (st & kBadLeaf) atomicOr(overflow, 2u);
    if (st & kBadEntry) atomicOr(overflow, 4u);
}

// Entry points.  The single-ray form takes the scene's buffers as plain kernel arguments (one wave per workgroup, the
// persistent grid = 24 waves per CU); the chunked form takes the scene / supply structs.
struct SceneArgs { const VdTlasNode* tlas; const VdInstance* inst; const VdMeshInfo* meshes; const VdBvhNode* bvh;
                   const float* verts; const unsigned* indices; unsigned n_meshes, yield; const float4* irec; const float4* tpair; const float4* mrec;
                   unsigned* ovf_bits; };
#ifndef VD_FAN_WPS
#define VD_FAN_WPS 5
#endif
template <bool ANY, bool FAN>      // FAN: the fan-out's code is compiled in (calls that run as one launch use the kernel without it)
__global__ __launch_bounds__(64, FAN ? VD_FAN_WPS : 6)   // second argument (HIP): waves per SIMD = 24 per CU; the fan-out's kernels take 96 registers at 5 per SIMD
                                                         // instead of spilling 9-12 at 6 (124 -> 129 / 192 -> 202 Mrays/s, profiles/r04_ab_trace_fan_wps.log)
void trace_single_kernel(SceneArgs a, const VdRay* __restrict__ rays, unsigned n_rays, VdHit* __restrict__ out, unsigned* __restrict__ out_any,
                         unsigned* __restrict__ overflow, unsigned* next_ray, const unsigned* __restrict__ gate, Fan fan) {
    if (gate && *gate == 0u) return;          // the call de-indexed the leaves itself and that went well: the other kernel runs
    const Scene s{a.tlas, a.inst, a.meshes, a.bvh, a.verts, a.indices, a.n_meshes, nullptr, a.irec, a.tpair, a.mrec, a.yield, a.ovf_bits};
    const RaySource src{nullptr, n_rays, 1u, n_rays, next_ray, fan};
    trace_body<ANY, false, false, FAN>(s, rays, src, out, out_any, overflow);
}
template <bool ANY, bool FAN>
__global__ __launch_bounds__(64, FAN ? VD_FAN_WPS : 6)
void trace_single_prep_kernel(SceneArgs a, const VdRay* __restrict__ rays, unsigned n_rays, VdHit* __restrict__ out, unsigned* __restrict__ out_any,
                              unsigned* __restrict__ overflow, unsigned* next_ray, const float* __restrict__ tris,
                              const unsigned* __restrict__ gate, Fan fan) {
    if (gate && *gate != 0u) return;          // the call's own de-indexing met an index range it cannot use: the indexed kernel runs
    const Scene s{a.tlas, a.inst, a.meshes, a.bvh, a.verts, a.indices, a.n_meshes, tris, a.irec, a.tpair, a.mrec, a.yield, a.ovf_bits};
    const RaySource src{nullptr, n_rays, 1u, n_rays, next_ray, fan};
    trace_body<ANY, true, false, FAN>(s, rays, src, out, out_any, overflow);
}
// The second pass: the rays listed in `list` (their ids, *n_list of them), one wave per workgroup, the stack beyond 128
// entries in s.deep.  Rare by construction - a handful of rays of a deep scene - so it is one plain launch.
template <bool ANY, bool PREP>
__global__ __launch_bounds__(64, 4)
void trace_deep_kernel(Scene s, const VdRay* __restrict__ rays, const unsigned* __restrict__ list, const unsigned* __restrict__ n_list,
                       VdHit* __restrict__ out, unsigned* __restrict__ out_any, unsigned* __restrict__ overflow, unsigned* next_ray) {
    const unsigned n = *n_list;
    const RaySource src{list, n, 1u, n, next_ray};
    trace_body<ANY, PREP, false, false, true>(s, rays, src, out, out_any, overflow);
}
// bitmap of the first pass -> list of ray ids (any order: a ray's record depends on the ray alone)
__global__ __launch_bounds__(256) void ovf_list_kernel(const unsigned* __restrict__ bits, unsigned n_words, unsigned* __restrict__ list, unsigned* __restrict__ count) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_words) return;
    unsigned w = bits[i];
    if (w == 0u) return;
    unsigned at = atomicAdd(count, (unsigned)__popc(w));
    while (w) { const unsigned b = (unsigned)__builtin_ctz(w); list[at++] = i * 32u + b; w &= w - 1u; }
}
template <bool ANY, bool PREP>
__global__ __launch_bounds__(64 * kWgWaves, 6)
void trace_chunk_kernel(Scene s, const VdRay* __restrict__ rays, RaySource src, VdHit* __restrict__ out, unsigned* __restrict__ out_any,
                        unsigned* __restrict__ overflow) {
    trace_body<ANY, PREP, true>(s, rays, src, out, out_any, overflow);
}

// fan-out, between two launches: the jobs appended so far are the next launch's supply
__global__ void fan_snapshot_kernel(const unsigned* append, unsigned cap, unsigned* end) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *end = min(*append, cap);
}
// fan-out, after the last launch: of a ray's finished jobs the one holding the minimum {distance, key} writes the ray's record
__global__ __launch_bounds__(256) void fan_resolve_kernel(const FanJob* __restrict__ jobs, const unsigned* __restrict__ append, unsigned cap,
                                                          const unsigned long long* __restrict__ best, VdHit* __restrict__ out) {
    const unsigned n = min(*append, cap);
    for (unsigned j = blockIdx.x * 256u + threadIdx.x; j < n; j += gridDim.x * 256u) {
        const FanJob J = jobs[j];
        if (J.state != FAN_HIT) continue;
        if (best[J.ray_id] != (((unsigned long long)__float_as_uint(J.lim) << 32) | J.key)) continue;
        VdHit h; h.dist = J.lim; h.hit = 1u; h.instance = J.node; h.triangle = J.tri;
        out[J.ray_id] = h;
    }
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

// One ray per pixel as the CPU harness makes them (src/bin/bvh_cpu.rs:71-83): pixel i -> x = (i % W) / W,
// y = (i / H) / H (the source divides by HEIGHT for the row; W == H == 640 there), ((x, y) - 0.5) * (2, -2),
// eye = (clip_to_world * (x, y, 1, 1)).xyz / .w, dir = normalize((clip_to_world * (x, y, 0, 1)).xyz).
struct Mat4 { float m[16]; };
__global__ __launch_bounds__(256) void primary_rays_kernel(Mat4 c2w, unsigned width, unsigned height, unsigned n, VdRay* __restrict__ rays) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float* M = c2w.m;
    float x = (float)(i % width) / (float)width;
    float y = (float)(i / height) / (float)height;
    x = (x - 0.5f) * 2.0f;
    y = (y - 0.5f) * -2.0f;
    float p[4], t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        p[r] = ((M[r] * x + M[4 + r] * y) + M[8 + r] * 1.0f) + M[12 + r] * 1.0f;
        t[r] = ((M[r] * x + M[4 + r] * y) + M[8 + r] * 0.0f) + M[12 + r] * 1.0f;
    }
    const float rl = 1.0f / sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
    VdRay r;
    r.eye[0] = p[0] / p[3]; r.eye[1] = p[1] / p[3]; r.eye[2] = p[2] / p[3]; r._pad0 = 0.0f;
    r.dir[0] = t[0] * rl; r.dir[1] = t[1] * rl; r.dir[2] = t[2] * rl; r._pad1 = 0.0f;
    rays[i] = r;
}

// `Bvh::traverse_iter` of the CPU harness (crates/bvh/src/blas.rs:247-295): one mesh, no TLAS.  It is NOT the WGSL
// walk above: the slab test divides by dir (intersection.rs:47-55), the triangle test is two-sided with EPS = 1e-4
// on the determinant and on t (intersection.rs:68-92), a child whose box is missed is never pushed, the near child
// is pushed first (so the far one is popped first), and pruning uses the closest hit so far or 1e30.  One lane per
// ray walks exactly that sequence.  The reference's stack holds 32 entries and panics past them (blas.rs:298-324);
// here 128, and overflow is an error code.  out_dist[r] = closest t, or -1 for Dist::Miss.
constexpr int kIterStack = 128;
struct DistRs { bool hit; float t; };
__device__ __forceinline__ DistRs intersect_aabb_rs(float ox, float oy, float oz, float dx, float dy, float dz,
                                                    const float* mn, const float* mx, float t) {
    const float ax = (mn[0] - ox) / dx, ay = (mn[1] - oy) / dy, az = (mn[2] - oz) / dz;
    const float bx = (mx[0] - ox) / dx, by = (mx[1] - oy) / dy, bz = (mx[2] - oz) / dz;
    const float tmax = min3(fmaxf(ax, bx), fmaxf(ay, by), fmaxf(az, bz));
    const float tmin = max3(fminf(ax, bx), fminf(ay, by), fminf(az, bz));
    return DistRs{tmax >= tmin && tmin < t && tmax > 0.0f, tmin};
}
// intersection.rs:68-92; t, or -1 for Miss
__device__ __forceinline__ float ray_intersect_rs(float ox, float oy, float oz, float dx, float dy, float dz,
                                                  const float* v0, const float* v1, const float* v2) {
    constexpr float EPS = 0.0001f;
    const float e1x = v1[0] - v0[0], e1y = v1[1] - v0[1], e1z = v1[2] - v0[2];
    const float e2x = v2[0] - v0[0], e2y = v2[1] - v0[1], e2z = v2[2] - v0[2];
    const float hx = dy * e2z - e2y * dz, hy = dz * e2x - e2z * dx, hz = dx * e2y - e2x * dy;   // dir x edge2
    const float a = dot3(e1x, e1y, e1z, hx, hy, hz);
    if (-EPS < a && a < EPS) return -1.0f;
    const float f = 1.0f / a;
    const float sx = ox - v0[0], sy = oy - v0[1], sz = oz - v0[2];
    const float u = f * dot3(sx, sy, sz, hx, hy, hz);
    if (!(0.0f <= u && u <= 1.0f)) return -1.0f;
    const float qx = sy * e1z - e1y * sz, qy = sz * e1x - e1z * sx, qz = sx * e1y - e1x * sy;   // s x edge1
    const float v = f * dot3(dx, dy, dz, qx, qy, qz);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    const float t = f * dot3(e2x, e2y, e2z, qx, qy, qz);
    return t > EPS ? t : -1.0f;
}

__global__ __launch_bounds__(64) void traverse_iter_kernel(const VdBvhNode* __restrict__ nodes, const float* __restrict__ verts,
                                                           const unsigned* __restrict__ indices, const VdRay* __restrict__ rays,
                                                           unsigned n_rays, float* __restrict__ out_dist, unsigned* __restrict__ overflow) {
    const unsigned r = blockIdx.x * 64u + threadIdx.x;
    if (r >= n_rays) return;
    const float4 e4 = reinterpret_cast<const float4*>(rays + r)[0], d4 = reinterpret_cast<const float4*>(rays + r)[1];
    const float ox = e4.x, oy = e4.y, oz = e4.z, dx = d4.x, dy = d4.y, dz = d4.z;
    unsigned stack[kIterStack];
    int head = 0;
    stack[head++] = 0u;
    float hit = -1.0f;   // Dist::Miss
    bool ovf = false;
    while (head > 0) {
        const VdBvhNode node = nodes[stack[--head]];
        if (node.count > 0u) {
            for (unsigned i = 0; i < node.count; ++i) {
                const unsigned* idx = indices + 3u * (size_t)(node.left_first + i);
                const float d = ray_intersect_rs(ox, oy, oz, dx, dy, dz, verts + 3u * (size_t)idx[0], verts + 3u * (size_t)idx[1],
                                                 verts + 3u * (size_t)idx[2]);
                if (d >= 0.0f) hit = hit >= 0.0f ? fminf(hit, d) : d;
            }
        } else {
            unsigned min_index = node.left_first, max_index = node.left_first + 1u;
            const VdBvhNode mc = nodes[min_index], xc = nodes[max_index];
            const float lim = hit >= 0.0f ? hit : kMaxDist;
            DistRs min_dist = intersect_aabb_rs(ox, oy, oz, dx, dy, dz, mc.min, mc.max, lim);
            DistRs max_dist = intersect_aabb_rs(ox, oy, oz, dx, dy, dz, xc.min, xc.max, lim);
            // derive(PartialOrd) on enum Dist { Hit(f32), Miss } (intersection.rs:22-26): Hit(x) < Miss
            const bool gt = min_dist.hit != max_dist.hit ? !min_dist.hit : (min_dist.hit && min_dist.t > max_dist.t);
            if (gt) {
                const unsigned ti = min_index; min_index = max_index; max_index = ti;
                const DistRs td = min_dist; min_dist = max_dist; max_dist = td;
            }
            if (!min_dist.hit) continue;
            if (head + 2 > kIterStack) { ovf = true; break; }
            stack[head++] = min_index;
            if (max_dist.hit) stack[head++] = max_index;
        }
    }
    out_dist[r] = hit;
    if (ovf) atomicOr(overflow, 1u);
}

// `Bvh::traverse` (crates/bvh/src/blas.rs:211-245), the RECURSIVE walk of the reference (its only call
// site is commented out at src/bin/bvh_cpu.rs:86).  traverse(node, t): Miss when the node's box is missed under t
// (same slab test as traverse_iter, intersection.rs:47-55); a leaf folds its triangles into t (`t = t.min(dist)`); an
// interior node calls left THEN right - no near / far ordering - each with the t found so far; every entered node
// returns Hit(t).  The returned t is only ever the running minimum handed on, so the recursion equals a pre-order walk
// with one running t and a stack of pending right children; one lane per ray runs that.  Quirk kept: a ray that enters
// the root box and hits nothing returns Hit(t0) - the t it was GIVEN - and Miss (-1) only when the root box is missed.
__global__ __launch_bounds__(64) void traverse_rec_kernel(const VdBvhNode* __restrict__ nodes, const float* __restrict__ verts,
                                                          const unsigned* __restrict__ indices, const VdRay* __restrict__ rays,
                                                          unsigned n_rays, float t0, float* __restrict__ out_dist,
                                                          unsigned* __restrict__ overflow) {
    const unsigned r = blockIdx.x * 64u + threadIdx.x;
    if (r >= n_rays) return;
    const float4 e4 = reinterpret_cast<const float4*>(rays + r)[0], d4 = reinterpret_cast<const float4*>(rays + r)[1];
    const float ox = e4.x, oy = e4.y, oz = e4.z, dx = d4.x, dy = d4.y, dz = d4.z;
    unsigned stack[kIterStack];
    int head = 0;
    float t = t0;
    bool ovf = false, entered_root = false;
    unsigned cur = 0u;
    bool have = true;
    while (have) {
        const VdBvhNode node = nodes[cur];
        have = false;
        const DistRs box = intersect_aabb_rs(ox, oy, oz, dx, dy, dz, node.min, node.max, t);   // blas.rs:220-222
        if (box.hit) {
            if (cur == 0u) entered_root = true;
            if (node.count > 0u) {                                                             // blas.rs:223-235
                for (unsigned i = 0; i < node.count; ++i) {
                    const unsigned* idx = indices + 3u * (size_t)(node.left_first + i);
                    const float d = ray_intersect_rs(ox, oy, oz, dx, dy, dz, verts + 3u * (size_t)idx[0], verts + 3u * (size_t)idx[1],
                                                     verts + 3u * (size_t)idx[2]);
                    if (d >= 0.0f) t = fminf(t, d);
                }
            } else {                                                                           // blas.rs:236-243: left, then right
                if (head + 1 > kIterStack) { ovf = true; break; }
                stack[head++] = node.left_first + 1u;
                cur = node.left_first; have = true;
                continue;
            }
        }
        if (head > 0) { cur = stack[--head]; have = true; }
    }
    out_dist[r] = entered_root ? t : -1.0f;                                                    // blas.rs:244 / :221
    if (ovf) atomicOr(overflow, 1u);
}

// ---- ray binning (order only: results are per ray, so any permutation gives the same output) ----------------------------
// key = origin cell (4 bits per axis over a box three times the scene's, Morton) in the high 12 bits, direction cell
// (octahedral map of the normalised direction, 10 + 10 bits, Morton) in the low 20: rays that start close together and
// point the same way - the pixels of a screen tile, the shadow rays of a surface patch - become neighbours.
__device__ __forceinline__ unsigned spread3(unsigned v) {    // 4 bits -> every third bit
    v &= 0xfu; v = (v | (v << 4)) & 0xc3u; v = (v | (v << 2)) & 0x249u; return v;
}
__device__ __forceinline__ unsigned spread2(unsigned v) {    // 10 bits -> every other bit
    v &= 0x3ffu; v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u; return v;
}
__global__ __launch_bounds__(256) void ray_keys_kernel(const VdRay* __restrict__ rays, unsigned n, const VdTlasNode* __restrict__ tlas,
                                                       unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = reinterpret_cast<const float4*>(rays + i)[0], b = reinterpret_cast<const float4*>(rays + i)[1];
    const VdTlasNode root = tlas[0];
    float q[3];
    const float o[3] = {a.x, a.y, a.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float lo = root.min[k], hi = root.max[k], w = hi - lo;
        const float t = (o[k] - (lo - w)) / (3.0f * w);                  // [lo - w, hi + w] -> [0, 1]
        q[k] = t > 0.0f ? (t < 1.0f ? t : 1.0f) : 0.0f;                  // NaN -> 0
    }
    const unsigned oc = spread3((unsigned)(q[0] * 15.99f)) | (spread3((unsigned)(q[1] * 15.99f)) << 1) | (spread3((unsigned)(q[2] * 15.99f)) << 2);
    const float l1 = (fabsf(b.x) + fabsf(b.y)) + fabsf(b.z);
    float u = b.x / l1, v = b.y / l1;
    if (b.z < 0.0f) { const float uu = (1.0f - fabsf(v)) * (u >= 0.0f ? 1.0f : -1.0f), vv = (1.0f - fabsf(u)) * (v >= 0.0f ? 1.0f : -1.0f); u = uu; v = vv; }
    u = u * 0.5f + 0.5f; v = v * 0.5f + 0.5f;
    u = u > 0.0f ? (u < 1.0f ? u : 1.0f) : 0.0f; v = v > 0.0f ? (v < 1.0f ? v : 1.0f) : 0.0f;
    const unsigned dc = spread2((unsigned)(u * 1023.99f)) | (spread2((unsigned)(v * 1023.99f)) << 1);
    keys[i] = (oc << 20) | dc;
    vals[i] = i;
}

// Stable LSD radix sort of (key, value) pairs, 8 bits per pass.  A UNIT is one wave's 1024 consecutive pairs; units are
// independent: pass 1 counts a unit's digits, a single-workgroup scan turns the [digit][unit] table into start offsets,
// pass 2 re-reads the unit in order - 16 groups of 64 - and ranks every pair among the equal digits before it (8 ballots
// give the lanes with the same digit; the unit's running offsets sit in LDS).
constexpr unsigned kSortUnit = 1024;
constexpr int kSortWaves = 4;
__global__ __launch_bounds__(64 * kSortWaves) void rs_count_kernel(const unsigned* __restrict__ keys, unsigned n, unsigned shift, unsigned n_units,
                                                                   unsigned* __restrict__ table) {
    __shared__ unsigned s_h[kSortWaves][256];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, unit = blockIdx.x * kSortWaves + wave;
    for (unsigned d = lane; d < 256u; d += 64u) s_h[wave][d] = 0u;
    vd_wave_lds_sync();
    if (unit < n_units) {
        const unsigned b0 = unit * kSortUnit;
        for (unsigned g = 0; g < kSortUnit; g += 64u) {
            const unsigned i = b0 + g + lane;
            if (i < n) atomicAdd(&s_h[wave][(keys[i] >> shift) & 255u], 1u);
        }
        vd_wave_lds_sync();
        for (unsigned d = lane; d < 256u; d += 64u) table[(size_t)d * n_units + unit] = s_h[wave][d];
    }
}
// exclusive scan of `m` counters in place (single workgroup; thread t owns a contiguous range)
__global__ __launch_bounds__(1024) void rs_scan_kernel(unsigned* __restrict__ table, unsigned m) {
    __shared__ unsigned s_wave[16];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned per = (m + 1023u) / 1024u;
    const unsigned lo = min(m, tid * per), hi = min(m, lo + per);
    unsigned sum = 0;
    for (unsigned i = lo; i < hi; ++i) sum += table[i];
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(incl, off); if (lane >= (unsigned)off) incl += t; }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    unsigned run = incl - sum;
    for (unsigned w = 0; w < wave; ++w) run += s_wave[w];
    for (unsigned i = lo; i < hi; ++i) { const unsigned c = table[i]; table[i] = run; run += c; }
}
__global__ __launch_bounds__(64 * kSortWaves) void rs_scatter_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ vals, unsigned n,
                                                                     unsigned shift, unsigned n_units, const unsigned* __restrict__ table,
                                                                     unsigned* __restrict__ keys_out, unsigned* __restrict__ vals_out) {
    __shared__ unsigned s_off[kSortWaves][256];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, unit = blockIdx.x * kSortWaves + wave;
    if (unit >= n_units) return;
    for (unsigned d = lane; d < 256u; d += 64u) s_off[wave][d] = table[(size_t)d * n_units + unit];
    vd_wave_lds_sync();
    const unsigned b0 = unit * kSortUnit;
    for (unsigned g = 0; g < kSortUnit; g += 64u) {
        const unsigned i = b0 + g + lane;
        const bool valid = i < n;
        const unsigned key = valid ? keys[i] : 0u, val = valid ? vals[i] : 0u;
        const unsigned d = (key >> shift) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long bm = __ballot((d >> bit) & 1u);
            same &= ((d >> bit) & 1u) ? bm : ~bm;
        }
        const unsigned rank = vd_mbcnt(same), cnt = (unsigned)__popcll(same);
        unsigned dst = 0;
        if (valid) dst = s_off[wave][d] + rank;
        vd_wave_lds_sync();
        if (valid && rank == 0u) s_off[wave][d] += cnt;      // the first lane of each digit advances the unit's offset
        vd_wave_lds_sync();
        if (valid) { keys_out[dst] = key; vals_out[dst] = val; }
    }
}

// Per-call records: what the walk reads at a TLAS step and when it enters an instance, re-laid so that each is ONE line.
// The bound of this kernel family is the number of distinct lines a CU can fetch (tools/probe_gather.hip:
// profiles/r03_probe_gather.log), so the lines per ray are what counts:
//   tpair[idx0]  (64 B) the two children of a TLAS node - tlas[idx0], tlas[idx1] - side by side at the LEFT child's index
//                (a TLAS step used to fetch two nodes at unrelated indices = two lines).  The left copy's unused
//                instance_idx word holds idx1 as a tag: a slot written for another right child (an unreachable slot of
//                the array naming the same left child) is detected and the step reads the two nodes themselves.  When
//                several slots name one left child, exactly ONE writes the record - the lowest node index
//                (pair_owner_kernel: atomicMin into an owner word, then records_kernel writes only as the owner) - so a
//                record is never a mixture of two writers' stores;
//   irec[k]      (64 B) for a TLAS leaf k: rows 0..2 of its instance's inv_transform (row r = {M[r], M[4 + r], M[8 + r],
//                M[12 + r]}: the operands of (inv_transform * vec4(p, w)).r in source order), {mesh id, bad-instance flag};
//   mrec[m]      (128 B) per mesh: {bvh_index, base_index, vertex_offset, root.left_first | root.count << 30} and the 64
//                bytes of the root's two children: `instance_intersect` (bvh.wgsl:78-87) + the first step of `traverse_bvh`
//                (bvh.wgsl:35-76) used to be instance -> MeshInfo -> root -> children, four dependent fetches, 140 times
//                per ray on the stress scene; few meshes, so these lines stay in L1.
// Written at the start of EVERY trace call from the scene's own buffers (<= 65 536 nodes: a few microseconds), so
// instances, TLAS nodes and meshes may change between calls as before.  A mesh whose root or root children lie outside
// the buffers gets 0xffffffff as its root word: a ray that ENTERS it reports VD_ERR_INVALID_ARG.
constexpr unsigned kTlasSlots = 65536;              // 16-bit child indices (bvh.wgsl:105-106)
__global__ __launch_bounds__(256) void pair_owner_kernel(const VdTlasNode* __restrict__ tlas, unsigned n_nodes, unsigned* __restrict__ owner) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n_nodes) return;
    const unsigned lr = tlas[k].left_right;
    if (lr != 0u && (lr & 0xffffu) < n_nodes && (lr >> 16u) < n_nodes) atomicMin(owner + (lr & 0xffffu), k);
}

__global__ __launch_bounds__(256) void records_kernel(const VdTlasNode* __restrict__ tlas, unsigned n_nodes, const VdInstance* __restrict__ inst,
                                                      unsigned n_inst, const VdMeshInfo* __restrict__ meshes, unsigned n_meshes,
                                                      const VdBvhNode* __restrict__ bvh, unsigned n_bvh, float4* __restrict__ irec,
                                                      float4* __restrict__ tpair, float4* __restrict__ mrec, const unsigned* __restrict__ owner) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k < n_meshes) {
        const VdMeshInfo mesh = meshes[k];
        float4* R = mrec + 8u * (size_t)k;
        unsigned rw = 0xffffffffu;
        if (mesh.bvh_index < n_bvh) {
            const VdBvhNode root = bvh[mesh.bvh_index];
            if (root.count <= 3u && root.left_first < (1u << 30)) {        // else: not a BvhBuilder tree (blas.rs:108)
                if (root.count != 0u) rw = root.left_first | (root.count << 30);
                else {
                    const size_t c = (size_t)mesh.bvh_index + root.left_first;
                    if (c + 1u < n_bvh) {
                        const float4* src = reinterpret_cast<const float4*>(bvh + c);
                        R[4] = src[0]; R[5] = src[1]; R[6] = src[2]; R[7] = src[3];
                        rw = root.left_first;
                    }
                }
            }
        }
        R[0] = make_float4(__uint_as_float(mesh.bvh_index), __uint_as_float(mesh.base_index), __uint_as_float((unsigned)mesh.vertex_offset),
                           __uint_as_float(rw));
    }
    if (k >= n_nodes) return;
    const VdTlasNode node = tlas[k];
    if (node.left_right != 0u) {
        const unsigned idx0 = node.left_right & 0xffffu, idx1 = node.left_right >> 16u;
        if (idx0 < n_nodes && idx1 < n_nodes && owner[idx0] == k) {
            const float4* c0 = reinterpret_cast<const float4*>(tlas + idx0);
            const float4* c1 = reinterpret_cast<const float4*>(tlas + idx1);
            float4 lo = c0[1]; lo.w = __uint_as_float(idx1);
            float4* P = tpair + 4u * (size_t)idx0;
            P[0] = c0[0]; P[1] = lo; P[2] = c1[0]; P[3] = c1[1];
        }
        return;
    }
    float4* R = irec + 4u * (size_t)k;
    if (node.instance_idx >= n_inst) { R[3] = make_float4(0.0f, __uint_as_float(1u), 0.0f, 0.0f); return; }
    const VdInstance* I = inst + node.instance_idx;
    const float* M = I->inv_transform;
    R[0] = make_float4(M[0], M[4], M[8], M[12]);
    R[1] = make_float4(M[1], M[5], M[9], M[13]);
    R[2] = make_float4(M[2], M[6], M[10], M[14]);
    R[3] = make_float4(__uint_as_float(min(I->mesh, n_meshes - 1u)), __uint_as_float(0u), 0.0f, 0.0f);
}

// De-indexed leaf triangles: tris[9 * (base_index / 3 + t)] = the three vertices fetch_vertex (bvh.wgsl:30-33) returns for
// triangle t of the mesh, i.e. vertices[vertex_offset + indices[base_index + 3 t + c]].
__global__ __launch_bounds__(256) void prepare_tris_kernel(const VdMeshInfo* __restrict__ meshes, const float* __restrict__ verts,
                                                           const unsigned* __restrict__ indices, unsigned n_indices, unsigned n_vertices,
                                                           float* __restrict__ tris, unsigned* __restrict__ err) {
    const VdMeshInfo m = meshes[blockIdx.y];
    const unsigned n_tri = m.index_count / 3u;
    if (m.base_index % 3u != 0u || (size_t)m.base_index + m.index_count > n_indices) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(err, 4u); return; }
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < n_tri; t += gridDim.x * 256u) {
        float* T = tris + 9u * ((size_t)(m.base_index / 3u) + t);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned vi = (unsigned)m.vertex_offset + indices[m.base_index + 3u * t + c];
            if (vi >= n_vertices) { atomicOr(err, 4u); T[3 * c] = T[3 * c + 1] = T[3 * c + 2] = 0.0f; continue; }
            T[3 * c] = verts[3u * (size_t)vi]; T[3 * c + 1] = verts[3u * (size_t)vi + 1u]; T[3 * c + 2] = verts[3u * (size_t)vi + 2u];
        }
    }
}

// ---- VD_OPT_TRACE_TIGHT_TLAS: a private top level over TIGHT world boxes ------------------------------------------
// The reference seeds every TLAS leaf box with the OBJECT-space mesh box (tlas.rs:39: the fold starts from
// [mesh.min, mesh.max]), so all leaves overlap around the origin and a ray enters most instances near it (145 of 2000 on
// the stress scene).  Which instances a ray ENTERS does not change what it hits: inside an instance only the root's
// children and below are tested (bvh.wgsl:35-76), all within the mesh's root box.  The private top level therefore
// bounds each instance by the eight corners of its BLAS ROOT box under `transform` - no seed - padded by 2e-5 of the
// box's largest coordinate (the rounding of transforming the ray in versus the corners out is ~1e-6 of that) PLUS how far
// `inv_transform`, which is all the walk uses, puts the geometry from where `transform` puts the corners (per axis, from
// T * Tinv - I: see the kernel), and clusters those boxes with the same agglomerative builder.  Used only where it is provably the same geometry: the
// instance's inv_transform must invert its transform (|T * Tinv - I| <= 1e-3 per element) and every corner must be
// finite.  ONE instance that fails either makes vd_trace_prepare_dev decline the option for the whole scene (it keeps
// the scene's own top level; VdTraceAccelInfo.tight_fallback_instances says how many failed): the hits of an instance
// whose inverse is stale lie outside its box, so whether the reference finds them depends on its visit order, and a
// top level with non-finite boxes may have dropped clusters (tlas.rs:87-105 finds no partner for a NaN area) that only
// the reference's own build reproduces.
__global__ __launch_bounds__(256) void tight_boxes_kernel(const VdInstance* __restrict__ inst, unsigned n_inst, const VdMeshInfo* __restrict__ meshes,
                                                          unsigned n_meshes, const VdBvhNode* __restrict__ bvh, unsigned n_bvh,
                                                          const VdTlasNode* __restrict__ scene_tlas, unsigned n_scene_nodes,
                                                          float* __restrict__ boxes, unsigned* __restrict__ n_fallback) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_inst) return;
    const float* T = inst[i].transform;
    const float* V = inst[i].inv_transform;
    const VdMeshInfo m = meshes[min(inst[i].mesh, n_meshes - 1u)];
    bool ok = m.bvh_index < n_bvh;
    float worst = 0.0f;
    float E[3][4];                             // rows 0..2 of T * Tinv - I: where the walk's geometry sits relative to T's
    for (int r = 0; r < 4 && ok; ++r)
        for (int c = 0; c < 4; ++c) {
            float a = 0.0f;
            for (int k = 0; k < 4; ++k) a += T[4 * k + r] * V[4 * c + k];        // (T * Tinv)[r][c], column-major storage
            const float e = fabsf(a - (r == c ? 1.0f : 0.0f));
            if (r < 3) E[r][c] = e;
            worst = fmaxf(worst, e);
        }
    ok = ok && worst <= 1e-3f;                 // false for NaN too
    float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
    float dev[3] = {0.0f, 0.0f, 0.0f};
    if (ok) {
        const VdBvhNode root = bvh[m.bvh_index];
        const float b[2][3] = {{root.min[0], root.min[1], root.min[2]}, {root.max[0], root.max[1], root.max[2]}};
        for (int c = 0; c < 8; ++c) {
            const float px = b[c & 1][0], py = b[(c >> 1) & 1][1], pz = b[(c >> 2) & 1][2];
            float w[3];
            for (int k = 0; k < 3; ++k) {
                w[k] = ((T[k] * px + T[4 + k] * py) + T[8 + k] * pz) + T[12 + k];
                ok = ok && fabsf(w[k]) < 1e30f;       // finite and inside the format's own range (MAX_DIST)
                mn[k] = fminf(mn[k], w[k]); mx[k] = fmaxf(mx[k], w[k]);
            }
            // The walk takes rays into the instance with inv_transform ALONE, so the geometry it sees is Tinv^-1 * p, not
            // T * p: with T * Tinv = I + E that is (I + E)^-1 * (T p) = T p - E (T p) + O(E^2) - off by at most
            // sum_c |E[k][c]| |(T p)_c| + |E[k][3]| along axis k (a float inverse of an instance 2 000 units out leaves
            // E ~ 1e-4 in the translation column; one nudged by 5e-4 without its inverse passes the 1e-3 test too).
            for (int k = 0; k < 3; ++k) dev[k] = fmaxf(dev[k], ((E[k][0] * fabsf(w[0]) + E[k][1] * fabsf(w[1])) + E[k][2] * fabsf(w[2])) + E[k][3]);
        }
    }
    if (ok) {
        float big = 0.0f;
        for (int k = 0; k < 3; ++k) big = fmaxf(big, fmaxf(fabsf(mn[k]), fabsf(mx[k])));
        for (int k = 0; k < 3; ++k) {
            const float pad = (2e-5f * big + 1e-30f) + 1.01f * dev[k];        // rounding of ray-in versus corners-out + the inverse's own error
            mn[k] -= pad; mx[k] += pad;
        }
    } else {
        atomicAdd(n_fallback, 1u);
        const bool own_leaf = i + 1u < n_scene_nodes && scene_tlas[i + 1u].left_right == 0u && scene_tlas[i + 1u].instance_idx == i;
        for (int k = 0; k < 3; ++k) {
            mn[k] = own_leaf ? scene_tlas[i + 1u].min[k] : -1e30f;
            mx[k] = own_leaf ? scene_tlas[i + 1u].max[k] : 1e30f;
        }
    }
    for (int k = 0; k < 3; ++k) { boxes[6u * i + k] = mn[k]; boxes[6u * i + 3 + k] = mx[k]; }
}

// The reference's build ends with one merge too many (tlas.rs:59 `while node_indices > 0`): node 2n has the true root
// 2n - 1 as BOTH children and node 0 is its copy, so a ray that misses everything walks the tree twice.  The private
// top level starts at the true root.
__global__ void tight_root_kernel(VdTlasNode* nodes, unsigned n) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && n >= 2u) nodes[0] = nodes[2u * n - 1u];
}

// ---- the private top level as an LBVH (VD_OPT_TRACE_TIGHT_TLAS = 2): built on all CUs in ~0.1 ms, so it can follow moving
// instances every frame (vd_trace_accel_update_dev); the agglomerative builder (= 1) makes the better tree and takes the
// reference's sequential chain to do it.  Morton codes of the box centres (10 bits per axis of the scene's extent), the
// radix sort the ray binning uses, Karras' binary radix tree (one thread per interior node; equal codes are told apart by
// their position), boxes bottom-up (the second child to arrive at a node goes on).  Nodes: 0 = the root, 1 + j = the
// leaf of the j-th code in sorted order, n + i = interior node i (i >= 1): indices stay below 2n <= 65 536.
__device__ __forceinline__ unsigned ord_of(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float of_ord(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
__global__ __launch_bounds__(256) void lbvh_extent_kernel(const float* __restrict__ boxes, unsigned n, unsigned* __restrict__ ext /*[6]: min xyz, max xyz of the centres*/) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 3; ++k) {
        const float c = 0.5f * (boxes[6u * i + k] + boxes[6u * i + 3 + k]);
        atomicMin(ext + k, ord_of(c)); atomicMax(ext + 3 + k, ord_of(c));
    }
}
__device__ __forceinline__ unsigned spread10(unsigned v) {
    v = (v | (v << 16)) & 0x030000ffu; v = (v | (v << 8)) & 0x0300f00fu; v = (v | (v << 4)) & 0x030c30c3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(256) void lbvh_keys_kernel(const float* __restrict__ boxes, unsigned n, const unsigned* __restrict__ ext,
                                                        unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    unsigned q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = of_ord(ext[k]), hi = of_ord(ext[3 + k]);
        const float c = 0.5f * (boxes[6u * i + k] + boxes[6u * i + 3 + k]);
        const float t = hi > lo ? (c - lo) / (hi - lo) : 0.0f;
        q[k] = (unsigned)fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
    }
    keys[i] = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
    vals[i] = i;
}
// length of the common prefix of the codes at sorted positions i and j (-1 outside the array); equal codes: their positions decide
__device__ __forceinline__ int lbvh_delta(const unsigned* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned a = keys[i], b = keys[j];
    return a != b ? __clz((int)(a ^ b)) : 32 + __clz((int)((unsigned)i ^ (unsigned)j));
}
__global__ __launch_bounds__(256) void lbvh_tree_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ vals, const float* __restrict__ boxes,
                                                        unsigned n, VdTlasNode* __restrict__ nodes, unsigned* __restrict__ parent, unsigned* __restrict__ arrived) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x), N = (int)n;
    if (i >= N) return;
    {   // the leaf of sorted position i
        const unsigned inst = vals[i];
        VdTlasNode lf;
        for (int k = 0; k < 3; ++k) { lf.min[k] = boxes[6u * inst + k]; lf.max[k] = boxes[6u * inst + 3 + k]; }
        lf.left_right = 0u; lf.instance_idx = inst;
        nodes[1 + i] = lf;
    }
    if (i >= N - 1) return;
    // Karras 2012: direction of the node's range, its other end, the split
    const int d = lbvh_delta(keys, N, i, i + 1) - lbvh_delta(keys, N, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, N, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, N, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (lbvh_delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, N, i, j);
    int sp = 0;
    for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1) {
        if (lbvh_delta(keys, N, i, i + (sp + t) * d) > dnode) sp += t;
        if (t == 1) break;
    }
    const int gamma = i + sp * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const unsigned self = i == 0 ? 0u : n + (unsigned)i;
    const unsigned left = lo == gamma ? 1u + (unsigned)gamma : n + (unsigned)gamma;            // gamma >= 1 when interior: the root is nobody's child
    const unsigned right = hi == gamma + 1 ? 2u + (unsigned)gamma : n + (unsigned)gamma + 1u;
    VdTlasNode nd;
    for (int k = 0; k < 3; ++k) { nd.min[k] = 0.0f; nd.max[k] = 0.0f; }
    nd.left_right = left | (right << 16); nd.instance_idx = 0xffffffffu;
    nodes[self] = nd;
    parent[left] = self; parent[right] = self;
    arrived[self] = 0u;
}
__global__ __launch_bounds__(256) void lbvh_fit_kernel(unsigned n, VdTlasNode* nodes, const unsigned* __restrict__ parent, unsigned* arrived) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    unsigned k = 1u + i;                                   // climb from leaf i
    for (;;) {
        const unsigned p = parent[k];
        __threadfence();                                   // my node's box is written before I announce myself
        if (atomicAdd(arrived + p, 1u) == 0u) return;      // the first child to arrive leaves; the second finds both boxes written
        __threadfence();
        const unsigned lr = nodes[p].left_right, l = lr & 0xffffu, r = lr >> 16;
        float mn[3], mx[3];
        for (int q = 0; q < 3; ++q) {
            mn[q] = fminf(__hip_atomic_load(&nodes[l].min[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&nodes[r].min[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            mx[q] = fmaxf(__hip_atomic_load(&nodes[l].max[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&nodes[r].max[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        for (int q = 0; q < 3; ++q) {
            __hip_atomic_store(&nodes[p].min[q], mn[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nodes[p].max[q], mx[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p == 0u) return;
        k = p;
    }
}

// order = the ray ids sorted by ray key (scratch of the context from byte `at`: past the flags and the entry records)
int sort_rays(VdCtx* ctx, const VdTraceScene* sc, const VdRay* d_rays, uint32_t n, const unsigned** out_order, size_t at) {
    const unsigned n_units = (n + kSortUnit - 1u) / kSortUnit;
    const size_t arr = ((size_t)n * 4 + 255) & ~(size_t)255, tab = ((size_t)256 * n_units * 4 + 255) & ~(size_t)255;
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, at + 4 * arr + tab);
    if (rc) return rc;
    char* base = reinterpret_cast<char*>(ctx->scratch) + at;
    unsigned* k[2] = {reinterpret_cast<unsigned*>(base), reinterpret_cast<unsigned*>(base + arr)};
    unsigned* v[2] = {reinterpret_cast<unsigned*>(base + 2 * arr), reinterpret_cast<unsigned*>(base + 3 * arr)};
    unsigned* table = reinterpret_cast<unsigned*>(base + 4 * arr);
    hipLaunchKernelGGL(ray_keys_kernel, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, d_rays, n, sc->tlas_nodes, k[0], v[0]);
    const unsigned blocks = (n_units + kSortWaves - 1u) / kSortWaves;
    for (int pass = 0; pass < 4; ++pass) {
        const int a = pass & 1, b = a ^ 1;
        hipLaunchKernelGGL(rs_count_kernel, dim3(blocks), dim3(64 * kSortWaves), 0, ctx->stream, k[a], n, 8u * pass, n_units, table);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, table, 256u * n_units);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(blocks), dim3(64 * kSortWaves), 0, ctx->stream, k[a], v[a], n, 8u * pass, n_units, table, k[b], v[b]);
    }
    *out_order = v[0];            // four passes: back in buffer 0
    return VD_OK;
}

int launch_trace(VdCtx* ctx, const VdTraceScene* sc, const float* d_tris, const VdRay* d_rays, uint32_t n_rays, VdHit* d_out,
                 uint32_t* d_any = nullptr) {
    // idle waves keep drawing from the ray counter after the last ray: leave it room below 2^32
    if (n_rays > 0xf0000000u) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace: more than 0xf0000000 rays in one call");
    // scratch: [256 B flags and counters][TLAS child pairs, 64 B x 65 536][entry records, 64 B x 65 536][mesh records][de-indexed triangles][ray binning arrays]
    // (pairs and entry records for all 65 536 indices a 16-bit child field can name: a stray index reads a poisoned slot, not beyond)
    const size_t pair_bytes = (size_t)64 * kTlasSlots, irec_bytes = (size_t)64 * kTlasSlots, owner_bytes = (size_t)4 * kTlasSlots;
    const size_t mrec_bytes = (size_t)128 * sc->n_meshes;
    const size_t tris_at = 256 + pair_bytes + irec_bytes + owner_bytes + ((mrec_bytes + 255) & ~(size_t)255);
    // A call that was not given prepared leaves de-indexes them itself when that is cheap next to the walk (one pass over the
    // index buffer, 36 B per triangle into the scratch: 5 us for the stress scene's 131 k triangles, 20 us for the harness
    // scene's 1.2 M) - the plain vd_trace_dev then walks at the prepared rate.  Not for few rays over a big scene, not
    // with the binned / chunked supplies, not with more meshes than one launch's gridDim.y, and only up to 2 Mi triangles
    // (72 MB of the context's grow-only scratch, kept for its lifetime); VD_OPT_TRACE_AUTO_PREPARE = 2 raises that to
    // 16 Mi (576 MB), 0 turns it off.  vd_trace_prepare_dev is the explicit form without a limit.
    const size_t n_tri = sc->n_indices / 3u;
    const bool single = ctx->option(VD_OPT_TRACE_CHUNK, 1) <= 1 &&
                        !(ctx->option(VD_OPT_TRACE_SORT, 0) != 0 && n_rays >= (unsigned)ctx->option(VD_OPT_TRACE_SORT_MIN, 65536));
    const long long auto_opt = ctx->option(VD_OPT_TRACE_AUTO_PREPARE, 1);
    const bool auto_prep = !d_tris && single && auto_opt != 0 && n_tri > 0 && sc->n_vertices > 0 && sc->n_meshes <= 65535u &&
                           n_tri <= ((size_t)1 << (auto_opt >= 2 ? 24 : 21)) && (size_t)n_rays * 8u >= n_tri;
    const size_t tris_bytes = auto_prep ? (((size_t)36 * n_tri + 64 + 255) & ~(size_t)255) : 0;
    const size_t sort_at = tris_at + tris_bytes;
    // fan-out (kFan*): single-ray supply, calls with at least one ray per lane of the grid, scenes with enough instances for
    // a ray to be long.  [256 B control words][best: 8 B per ray][jobs: 48 B x cap]
    const unsigned waves = (unsigned)ctx->num_cus * (unsigned)std::min<long long>(kWavesPerCu, std::max<long long>(1, ctx->option(VD_OPT_TRACE_WAVES, kWavesPerCu)));
    unsigned phases = (unsigned)std::min<long long>(4, std::max<long long>(1, ctx->option(VD_OPT_TRACE_FAN, kFanPhases)));
    if (!single || (size_t)n_rays < (size_t)waves * 64u || sc->n_instances < 64u) phases = 1u;
    // The fan-out's kernels cost ~15 % where no ray lives long enough to fan out (more code, spilled registers, launches that find
    // nothing to do): a call remembers how many jobs it made, and while the last call over the same top level made fewer than one
    // per 64 rays the next 15 calls over it run as one launch with the plain kernels; then the fan-out is tried again.
    const bool fan_default = ctx->option(VD_OPT_TRACE_FAN, -1) < 0;
    const bool fan_same_scene = ctx->fan_tlas == sc->tlas_nodes && ctx->fan_nodes == sc->n_tlas_nodes && ctx->fan_inst == sc->n_instances;
    if (phases > 1u && fan_default && fan_same_scene && ctx->fan_idle_calls != 0u) { phases = 1u; --ctx->fan_idle_calls; }
    unsigned fan_cap = (unsigned)std::min<size_t>((size_t)1 << 21, (size_t)n_rays * 2u);
    if (ctx->option(VD_OPT_TRACE_FAN_SLOTS, -1) >= 0) fan_cap = (unsigned)std::min<long long>(fan_cap, ctx->option(VD_OPT_TRACE_FAN_SLOTS, -1));      // tests: a list that fills up
    const size_t best_bytes = ((size_t)n_rays * 8u + 255) & ~(size_t)255;
    const size_t fan_bytes = phases > 1u ? 256 + best_bytes + (size_t)fan_cap * sizeof(FanJob) : 0;
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, sort_at + fan_bytes);
    if (rc) return rc;
    // one bit per ray: "ran out of its 128 stack entries" (second pass below).  All zero between calls: zeroed when (re)allocated
    // and again after a call that set any, so a call that overflows nowhere pays nothing for it.
    const unsigned ovf_words = (n_rays + 31u) / 32u;
    {
        const void* before = ctx->trace_ovf;
        rc = vd_ensure(ctx, &ctx->trace_ovf, &ctx->trace_ovf_bytes, (size_t)ovf_words * 4u + 4u);
        if (rc) return rc;
        if (ctx->trace_ovf != before || ctx->trace_ovf_dirty) VD_HIP_CHECK(ctx, hipMemsetAsync(ctx->trace_ovf, 0, ctx->trace_ovf_bytes, ctx->stream));
        ctx->trace_ovf_dirty = true;       // until this call has ended cleanly
    }
    unsigned* d_ovf = reinterpret_cast<unsigned*>(ctx->trace_ovf);
    // Binning is OFF by default: measured on the stress scene (tools/ab_trace.py, profiles/r03_ab_trace.log) rays handed
    // out in sorted order are SLOWER (32.0 against 35.7 Mrays/s): the walk is bound by the slowest of a wave's 64 fetches,
    // not by the L1 hit rate, and sorting puts the expensive rays of the dense screen centre side by side in time.
    const bool sorted = ctx->option(VD_OPT_TRACE_SORT, 0) != 0 && n_rays >= (unsigned)ctx->option(VD_OPT_TRACE_SORT_MIN, 65536);
    vd_time_begin(ctx);
    const unsigned* order = nullptr;
    if (sorted) { rc = sort_rays(ctx, sc, d_rays, n_rays, &order, sort_at); if (rc) return rc; }
    unsigned* d_flag = reinterpret_cast<unsigned*>(ctx->scratch);      // after sort_rays: the scratch may have grown
#ifdef VD_TUNING
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_flag, 0, 256, ctx->stream));
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_flag + 16, 0xff, 8, ctx->stream));
#else
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_flag, 0, 64, ctx->stream));
#endif
    unsigned* fan_ctl = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ctx->scratch) + sort_at);      // [0] append, [1 + p] supply counter of launch p, [8 + p] end of launch p's supply
    unsigned long long* fan_best = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(fan_ctl) + 256);
    FanJob* fan_jobs = reinterpret_cast<FanJob*>(reinterpret_cast<char*>(fan_best) + best_bytes);
    if (phases > 1u) {
        VD_HIP_CHECK(ctx, hipMemsetAsync(fan_ctl, 0, 256, ctx->stream));
        if (!d_any) VD_HIP_CHECK(ctx, hipMemsetAsync(fan_best, 0xff, best_bytes, ctx->stream));
    }
    float4* d_pair = reinterpret_cast<float4*>(reinterpret_cast<char*>(ctx->scratch) + 256);
    float4* d_rec = reinterpret_cast<float4*>(reinterpret_cast<char*>(d_pair) + pair_bytes);
    unsigned* d_owner = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(d_rec) + irec_bytes);
    float4* d_mrec = reinterpret_cast<float4*>(reinterpret_cast<char*>(d_owner) + owner_bytes);
    const unsigned yield = (unsigned)std::max<long long>(1, ctx->option(VD_OPT_TRACE_YIELD, kYieldDefault));
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_pair, 0xff, pair_bytes + irec_bytes + owner_bytes, ctx->stream));      // no pair slot carries a tag or has an owner yet, every entry record says "bad"
    const unsigned n_rec = std::min(sc->n_tlas_nodes, kTlasSlots);      // nodes past 65 535 cannot be named by a 16-bit child field
    hipLaunchKernelGGL(pair_owner_kernel, dim3((n_rec + 255u) / 256u), dim3(256), 0, ctx->stream, sc->tlas_nodes, n_rec, d_owner);
    hipLaunchKernelGGL(records_kernel, dim3((std::max(n_rec, sc->n_meshes) + 255u) / 256u), dim3(256), 0, ctx->stream, sc->tlas_nodes,
                       n_rec, sc->instances, sc->n_instances, sc->meshes, sc->n_meshes, sc->bvh_nodes, sc->n_bvh_nodes, d_rec, d_pair, d_mrec, d_owner);
    const unsigned* gate = nullptr;
    if (auto_prep) {
        float* t = reinterpret_cast<float*>(reinterpret_cast<char*>(ctx->scratch) + tris_at);
        hipLaunchKernelGGL(prepare_tris_kernel, dim3(256, sc->n_meshes), dim3(256), 0, ctx->stream, sc->meshes, sc->vertices, sc->indices, sc->n_indices,
                           sc->n_vertices, t, d_flag + 2);
        d_tris = t; gate = d_flag + 2;        // non-zero: some mesh's index range cannot be de-indexed - the indexed kernel takes the call
    }
    Scene s{sc->tlas_nodes, sc->instances, sc->meshes, sc->bvh_nodes, sc->vertices, sc->indices, sc->n_meshes, d_tris, d_rec, d_pair, d_mrec, yield, d_ovf};
    {
        // Default: single rays from one global counter (chunk = 1), one wave per workgroup - the finest balance.  Chunks of
        // consecutive rays per workgroup (a CU-local window of the ray order) lose more to imbalance than they gain in
        // locality: 64 rays per chunk 32.6, 256 rays 16.9 Mrays/s against 35.2 in the same kernel (same log).
        unsigned chunk = (unsigned)ctx->option(VD_OPT_TRACE_CHUNK, 1);
        if (chunk <= 1u && !order) {
            const SceneArgs a{sc->tlas_nodes, sc->instances, sc->meshes, sc->bvh_nodes, sc->vertices, sc->indices, sc->n_meshes, yield, d_rec, d_pair, d_mrec, d_ovf};
            // fan-out: launch 0 hands out the rays; launch p >= 1 the jobs appended before it started (fan_snapshot_kernel);
            // every launch but the last may append; fan_resolve_kernel writes the records of the rays that were fanned out
            for (unsigned ph = 0; ph < phases; ++ph) {
                Fan fan = {};
                if (phases > 1u) {
                    fan.jobs = fan_jobs; fan.append = fan_ctl; fan.cap = fan_cap; fan.best = fan_best; fan.level = ph;
                    fan.below = ph + 1u < phases ? kFanBelow : 0u;
                    if (ph) { fan.begin = fan_ctl + 8 + (ph - 1); fan.end = fan_ctl + 8 + ph; }
                }
                if (ph) hipLaunchKernelGGL(fan_snapshot_kernel, dim3(1), dim3(64), 0, ctx->stream, fan_ctl, fan_cap, fan_ctl + 8 + ph);
                unsigned* next = ph == 0 ? d_flag + 1 : fan_ctl + ph;
#define VD_TRACE_S(K, ...) do { if (phases > 1u) { if (d_any) hipLaunchKernelGGL((K<true, true>), dim3(waves), dim3(64), 0, ctx->stream, __VA_ARGS__); else hipLaunchKernelGGL((K<false, true>), dim3(waves), dim3(64), 0, ctx->stream, __VA_ARGS__); } \
                                else { if (d_any) hipLaunchKernelGGL((K<true, false>), dim3(waves), dim3(64), 0, ctx->stream, __VA_ARGS__); else hipLaunchKernelGGL((K<false, false>), dim3(waves), dim3(64), 0, ctx->stream, __VA_ARGS__); } } while (0)
                if (d_tris) VD_TRACE_S(trace_single_prep_kernel, a, d_rays, n_rays, d_out, d_any, d_flag, next, d_tris, gate, fan);
                if (!d_tris || gate) VD_TRACE_S(trace_single_kernel, a, d_rays, n_rays, d_out, d_any, d_flag, next, gate, fan);
#undef VD_TRACE_S
            }
            if (phases > 1u && !d_any)
                hipLaunchKernelGGL(fan_resolve_kernel, dim3((unsigned)ctx->num_cus * 4u), dim3(256), 0, ctx->stream, fan_jobs, fan_ctl, fan_cap, fan_best, d_out);
        } else {
            const unsigned groups = (unsigned)ctx->num_cus * (kWavesPerCu / kWgWaves);
            chunk = chunk <= 1u ? 64u : ((chunk + 63u) & ~63u);      // binned rays are handed out in chunks (of 64 at least)
            const RaySource src{order, n_rays, chunk, (n_rays + chunk - 1u) / chunk, d_flag + 1};
#define VD_TRACE_C(A, P) hipLaunchKernelGGL((trace_chunk_kernel<A, P>), dim3(groups), dim3(64 * kWgWaves), 0, ctx->stream, s, d_rays, src, d_out, d_any, d_flag)
            if (d_any) { if (d_tris) VD_TRACE_C(true, true); else VD_TRACE_C(true, false); }
            else { if (d_tris) VD_TRACE_C(false, true); else VD_TRACE_C(false, false); }
#undef VD_TRACE_C
        }
    }
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->host_pinned, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (phases > 1u) VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->host_pinned + 1, fan_ctl, 4, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->host_pinned + 2, d_flag + 2, 4, hipMemcpyDeviceToHost, ctx->stream));      // the gate: which of the two kernels walked
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (phases > 1u) {
        ctx->fan_tlas = sc->tlas_nodes; ctx->fan_nodes = sc->n_tlas_nodes; ctx->fan_inst = sc->n_instances;
        ctx->fan_idle_calls = ctx->host_pinned[1] < n_rays / 64u ? 15u : 0u;
    }
    if (ctx->host_pinned[0] & 4u) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace: a TLAS leaf's instance, its mesh's root or the root's children lie outside the scene's buffers");
    if (ctx->host_pinned[0] & 2u) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace: BVH leaf with more than 3 triangles (BvhBuilder never makes one: blas.rs:108)");
    if (ctx->host_pinned[0] & 1u) {
        // Some rays needed more than the 128 entries a lane holds (the reference's own 24-entry stack is unchecked,
        // shaders/utils/stack.wgsl:1-20: it has no answer there at all).  They are walked again, from their start, with the entries
        // beyond 128 in a slab of global memory - same visits, same arithmetic, so the records are the ones an unbounded stack gives
        // (tests/test_gpu_tlas_trace.py holds them against the oracle's 256-entry walk).  Entries per lane grow 1 Ki -> 4 Ki -> ...
        // until nothing overflows; the slab is waves x 64 lanes x entries x 4 B under a 256 MB budget (fewer waves as it deepens),
        // and a stack cannot hold more entries than the scene has nodes.
        const bool prep_walked = d_tris && (!gate || ctx->host_pinned[2] == 0u);
        const size_t node_bound = (size_t)sc->n_tlas_nodes + (size_t)sc->n_bvh_nodes + 2u;
        const size_t list_bytes = ((size_t)n_rays * 4u + 255) & ~(size_t)255;
        const size_t budget = (size_t)256 << 20;
        for (size_t cap = 1024;; cap *= 4) {
            const size_t per_wave = cap * 64u * 4u;
            if (per_wave > ((size_t)2 << 30)) VD_FAIL(ctx, VD_ERR_STACK_OVERFLOW, "vd_trace: a ray's traversal stack needs more than 8 Mi entries");
            const unsigned waves_d = (unsigned)std::max<size_t>(1, std::min<size_t>(waves, budget / per_wave));
            rc = vd_ensure(ctx, &ctx->trace_deep, &ctx->trace_deep_bytes, 256 + list_bytes + (size_t)waves_d * per_wave);
            if (rc) return rc;
            unsigned* ctl = reinterpret_cast<unsigned*>(ctx->trace_deep);                 // [0] listed rays, [1] supply counter
            unsigned* list = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ctl) + 256);
            VD_HIP_CHECK(ctx, hipMemsetAsync(ctl, 0, 256, ctx->stream));
            VD_HIP_CHECK(ctx, hipMemsetAsync(d_flag, 0, 4, ctx->stream));
            hipLaunchKernelGGL(ovf_list_kernel, dim3((ovf_words + 255u) / 256u), dim3(256), 0, ctx->stream, d_ovf, ovf_words, list, ctl);
            Scene sd = s;
            sd.ovf_bits = nullptr;
            sd.deep = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(list) + list_bytes);
            sd.deep_cap = (unsigned)cap;
#define VD_TRACE_D(A, P) hipLaunchKernelGGL((trace_deep_kernel<A, P>), dim3(waves_d), dim3(64), 0, ctx->stream, sd, d_rays, list, ctl, d_out, d_any, d_flag, ctl + 1)
            if (d_any) { if (prep_walked) VD_TRACE_D(true, true); else VD_TRACE_D(true, false); }
            else { if (prep_walked) VD_TRACE_D(false, true); else VD_TRACE_D(false, false); }
#undef VD_TRACE_D
            vd_time_end(ctx);                                                              // vd_last_gpu_ms covers the second pass too
            VD_HIP_CHECK(ctx, hipGetLastError());
            VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->host_pinned, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
            VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->host_pinned[0] & 6u) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace: the scene's buffers are inconsistent (bad leaf or entry met by the second pass)");
            if (!(ctx->host_pinned[0] & 1u)) break;
            if (cap >= node_bound) VD_FAIL(ctx, VD_ERR_STACK_OVERFLOW, "vd_trace: traversal stack deeper than the scene has nodes (cyclic node arrays?)");
        }
        VD_HIP_CHECK(ctx, hipMemsetAsync(d_ovf, 0, (size_t)ovf_words * 4u, ctx->stream));
    }
    ctx->trace_ovf_dirty = false;
    return VD_OK;
}

// The private top level of a prepared scene (VD_OPT_TRACE_TIGHT_TLAS: 1 = agglomerative, 2 = LBVH), from the scene's instances as
// they are now.  Blocks (it reads back how many instances did not qualify); with any, the scene's own top level is walked.
int build_tight_tlas(VdCtx* ctx, VdTraceAccel* a) {
    const unsigned n = a->scene.n_instances;
    hipStream_t st = ctx->stream;
    unsigned* d_fb = reinterpret_cast<unsigned*>(a->boxes + 6 * (size_t)n);
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_fb, 0, 4, st));
    hipLaunchKernelGGL(tight_boxes_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, a->scene.instances, n, a->scene.meshes, a->scene.n_meshes,
                       a->scene.bvh_nodes, a->scene.n_bvh_nodes, a->user_tlas, a->user_n_nodes, a->boxes, d_fb);
    if (a->tight_mode == 2u) {
        const unsigned n_units = (n + kSortUnit - 1u) / kSortUnit;
        const size_t arr = ((size_t)n * 4 + 255) & ~(size_t)255;
        char* base = reinterpret_cast<char*>(a->lbvh);
        unsigned* k[2] = {reinterpret_cast<unsigned*>(base), reinterpret_cast<unsigned*>(base + arr)};
        unsigned* v[2] = {reinterpret_cast<unsigned*>(base + 2 * arr), reinterpret_cast<unsigned*>(base + 3 * arr)};
        unsigned* parent = reinterpret_cast<unsigned*>(base + 4 * arr);                 // 2n + 2 words
        unsigned* arrived = parent + 2 * (size_t)n + 2;                                 // 2n + 2 words
        unsigned* ext = arrived + 2 * (size_t)n + 2;                                    // 8 words
        unsigned* table = ext + 8;                                                      // 256 * n_units words
        VD_HIP_CHECK(ctx, hipMemsetAsync(ext, 0xff, 12, st));
        VD_HIP_CHECK(ctx, hipMemsetAsync(ext + 3, 0, 12, st));
        hipLaunchKernelGGL(lbvh_extent_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, a->boxes, n, ext);
        hipLaunchKernelGGL(lbvh_keys_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, a->boxes, n, ext, k[0], v[0]);
        const unsigned blocks = (n_units + kSortWaves - 1u) / kSortWaves;
        for (int pass = 0; pass < 4; ++pass) {
            const int x = pass & 1, y = x ^ 1;
            hipLaunchKernelGGL(rs_count_kernel, dim3(blocks), dim3(64 * kSortWaves), 0, st, k[x], n, 8u * pass, n_units, table);
            hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, st, table, 256u * n_units);
            hipLaunchKernelGGL(rs_scatter_kernel, dim3(blocks), dim3(64 * kSortWaves), 0, st, k[x], v[x], n, 8u * pass, n_units, table, k[y], v[y]);
        }
        hipLaunchKernelGGL(lbvh_tree_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, k[0], v[0], a->boxes, n, a->tight, parent, arrived);
        hipLaunchKernelGGL(lbvh_fit_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, n, a->tight, parent, arrived);
    } else {
        const int rc = vd_tlas_build_from_boxes(ctx, a->boxes, n, a->tight);
        if (rc) return rc;
        hipLaunchKernelGGL(tight_root_kernel, dim3(1), dim3(64), 0, st, a->tight, n);
    }
    VD_HIP_CHECK(ctx, hipGetLastError());
    unsigned h_fallback = 0;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(&h_fallback, d_fb, 4, hipMemcpyDeviceToHost, st));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(st));
    a->tight_fallbacks = h_fallback;
    if (h_fallback) { a->scene.tlas_nodes = a->user_tlas; a->scene.n_tlas_nodes = a->user_n_nodes; }      // declined: the reference's visit order is the only exact one here
    else { a->scene.tlas_nodes = a->tight; a->scene.n_tlas_nodes = a->tight_mode == 2u ? 2u * n : 2u * n + 1u; }
    return VD_OK;
}

bool scene_ok(const VdTraceScene* s) {
    return s && s->tlas_nodes && s->instances && s->meshes && s->bvh_nodes && s->vertices && s->indices && s->n_meshes &&
           s->n_tlas_nodes && s->n_instances;
}

}  // namespace

extern "C" {

#ifdef VD_TUNING
// tuning build only: {outer iterations, stepping-loop iterations, stepping lanes (64 bit), leaf / entry / TLAS-interior lane-steps} of the last trace call
// tuning build only: the last trace call's timeline in 2 ms slots: out[0..22] waves that ended in the slot, out[23..45] stepping iterations done in it
int vd_debug_trace_timeline(VdCtx* ctx, uint32_t* out /*[46]*/) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx || !out || !ctx->scratch) return VD_ERR_INVALID_ARG;
    unsigned h[64];
    if (hipMemcpy(h, ctx->scratch, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return VD_ERR_HIP;
    for (int k = 0; k < 46; ++k) out[k] = h[18 + k];
    return VD_OK;
}
int vd_debug_trace_counters(VdCtx* ctx, uint64_t* out /*[11]*/) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx || !out || !ctx->scratch) return VD_ERR_INVALID_ARG;
    unsigned h[16];
    if (hipMemcpy(h, ctx->scratch, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return VD_ERR_HIP;
    out[0] = h[4]; out[1] = h[5]; out[2] = (uint64_t)h[6] | ((uint64_t)h[7] << 32); out[3] = h[8]; out[4] = h[9]; out[5] = h[10]; out[6] = h[11]; out[7] = h[12]; out[8] = h[13]; out[9] = h[14]; out[10] = h[15];
    return VD_OK;
}
#endif

int vd_trace_dev(VdCtx* ctx, const VdTraceScene* d_scene, const VdRay* d_rays, uint32_t n_rays, VdHit* d_out) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!scene_ok(d_scene)) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace: incomplete scene");
    if (n_rays == 0) return VD_OK;
    if (!d_rays || !d_out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace: null rays/out");
    return launch_trace(ctx, d_scene, nullptr, d_rays, n_rays, d_out);
}

int vd_trace_any_dev(VdCtx* ctx, const VdTraceScene* d_scene, const VdRay* d_rays, uint32_t n_rays, uint32_t* d_out_hit) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!scene_ok(d_scene)) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace_any: incomplete scene");
    if (n_rays == 0) return VD_OK;
    if (!d_rays || !d_out_hit) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace_any: null rays/out");
    return launch_trace(ctx, d_scene, nullptr, d_rays, n_rays, nullptr, d_out_hit);
}

int vd_trace_prepare_dev(VdCtx* ctx, const VdTraceScene* d_scene, VdTraceAccel** out) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx || !out) return VD_ERR_INVALID_ARG;
    *out = nullptr;
    if (!scene_ok(d_scene) || d_scene->n_indices == 0u || d_scene->n_vertices == 0u) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace_prepare: incomplete scene");
    VdTraceAccel* a = new (std::nothrow) VdTraceAccel();
    if (!a) return VD_ERR_OOM;
    a->scene = *d_scene;
    const size_t n_tri = d_scene->n_indices / 3u;
    if (hipMalloc(reinterpret_cast<void**>(&a->tris), 36 * n_tri + 64) != hipSuccess) { delete a; VD_FAIL(ctx, VD_ERR_OOM, "vd_trace_prepare: triangle array"); }
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, 256);
    if (rc) { (void)hipFree(a->tris); delete a; return rc; }
    unsigned* d_flag = reinterpret_cast<unsigned*>(ctx->scratch);
    (void)hipMemsetAsync(d_flag, 0, 4, ctx->stream);
    (void)hipMemsetAsync(a->tris, 0, 36 * n_tri + 64, ctx->stream);      // index ranges no mesh covers
    for (unsigned m0 = 0; m0 < d_scene->n_meshes; m0 += 65535u)      // gridDim.y carries the mesh: 65 535 per launch
        hipLaunchKernelGGL(prepare_tris_kernel, dim3(256, std::min(d_scene->n_meshes - m0, 65535u)), dim3(256), 0, ctx->stream, d_scene->meshes + m0,
                           d_scene->vertices, d_scene->indices, d_scene->n_indices, d_scene->n_vertices, a->tris, d_flag);
    hipError_t e = hipMemcpyAsync(ctx->host_pinned, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || ctx->host_pinned[0]) {
        (void)hipFree(a->tris); delete a;
        if (e != hipSuccess) VD_FAIL(ctx, VD_ERR_HIP, hipGetErrorString(e));
        VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace_prepare: a mesh's index range or a vertex index lies outside the scene's buffers");
    }
    a->user_tlas = d_scene->tlas_nodes; a->user_n_nodes = d_scene->n_tlas_nodes;
    const long long tight = ctx->option(VD_OPT_TRACE_TIGHT_TLAS, 0);
    if (tight != 0 && d_scene->n_instances >= 2u && d_scene->n_instances <= (tight == 2 ? VD_TLAS_MAX_INSTANCES - 1u : VD_TLAS_MAX_INSTANCES)) {
        const unsigned n = d_scene->n_instances;
        a->tight_mode = tight == 2 ? 2u : 1u;
        const size_t arr = ((size_t)n * 4 + 255) & ~(size_t)255;
        const size_t lbvh_bytes = 4 * arr + 4 * (4 * (size_t)n + 4 + 8 + 256 * (size_t)((n + kSortUnit - 1u) / kSortUnit)) + 256;
        bool good = hipMalloc(reinterpret_cast<void**>(&a->boxes), 24 * (size_t)n + 16) == hipSuccess &&       // (not in the context's scratch: the agglomerative builder lays that out for itself)
                    hipMalloc(reinterpret_cast<void**>(&a->tight), sizeof(VdTlasNode) * (2 * (size_t)n + 1)) == hipSuccess &&
                    (a->tight_mode != 2u || hipMalloc(reinterpret_cast<void**>(&a->lbvh), lbvh_bytes) == hipSuccess);
        if (good) good = hipMemsetAsync(a->tight, 0, sizeof(VdTlasNode) * (2 * (size_t)n + 1), ctx->stream) == hipSuccess;      // slots a builder leaves unused read as leaves nobody reaches
        int rc_t = good ? build_tight_tlas(ctx, a) : VD_ERR_OOM;
        if (rc_t) {
            (void)hipStreamSynchronize(ctx->stream);
            if (a->boxes) (void)hipFree(a->boxes);
            if (a->lbvh) (void)hipFree(a->lbvh);
            if (a->tight) (void)hipFree(a->tight);
            (void)hipFree(a->tris); delete a;
            if (rc_t == VD_ERR_OOM) VD_FAIL(ctx, VD_ERR_OOM, "vd_trace_prepare: the private top level (VD_OPT_TRACE_TIGHT_TLAS) could not be allocated");
            return rc_t;
        }
    }
    *out = a;
    return VD_OK;
}

int vd_trace_accel_info(const VdTraceAccel* accel, VdTraceAccelInfo* out) {
    if (!accel || !out) return VD_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    out->tight_tlas = (accel->tight && accel->scene.tlas_nodes == accel->tight) ? accel->tight_mode : 0u;
    out->n_tlas_nodes = accel->scene.n_tlas_nodes;
    out->tight_fallback_instances = accel->tight_fallbacks;
    out->triangle_bytes = 36ull * (accel->scene.n_indices / 3u);
    out->d_tlas_nodes = accel->scene.tlas_nodes;
    return VD_OK;
}

int vd_trace_accel_update_dev(VdCtx* ctx, VdTraceAccel* accel) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx || !accel) return VD_ERR_INVALID_ARG;
    if (!accel->tight) return VD_OK;          // the scene's own top level is walked: the host refits that one (vd_tlas_refit_dev)
    ctx->fan_forget(accel->tight);
    return build_tight_tlas(ctx, accel);
}

int vd_trace_release(VdCtx* ctx, VdTraceAccel* accel) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx || !accel) return VD_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(ctx->stream);
    if (accel->tight) ctx->fan_forget(accel->tight);
    if (accel->tris) (void)hipFree(accel->tris);
    if (accel->tight) (void)hipFree(accel->tight);
    if (accel->boxes) (void)hipFree(accel->boxes);
    if (accel->lbvh) (void)hipFree(accel->lbvh);
    delete accel;
    return VD_OK;
}

int vd_trace_prepared_dev(VdCtx* ctx, const VdTraceAccel* accel, const VdRay* d_rays, uint32_t n_rays, VdHit* d_out) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!accel) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace_prepared: null accel");
    if (n_rays == 0) return VD_OK;
    if (!d_rays || !d_out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace_prepared: null rays/out");
    return launch_trace(ctx, &accel->scene, accel->tris, d_rays, n_rays, d_out);
}

int vd_trace_any_prepared_dev(VdCtx* ctx, const VdTraceAccel* accel, const VdRay* d_rays, uint32_t n_rays, uint32_t* d_out_hit) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!accel) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace_any_prepared: null accel");
    if (n_rays == 0) return VD_OK;
    if (!d_rays || !d_out_hit) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_trace_any_prepared: null rays/out");
    return launch_trace(ctx, &accel->scene, accel->tris, d_rays, n_rays, nullptr, d_out_hit);
}

int vd_shadow_rays_dev(VdCtx* ctx, const float* d_positions, const float* d_normals, uint32_t n_points, const float* light_position,
                       VdRay* d_rays) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (n_points == 0) return VD_OK;
    if (!d_positions || !d_normals || !light_position || !d_rays) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_shadow_rays: null pointer");
    hipLaunchKernelGGL(shadow_rays_kernel, dim3((n_points + 255) / 256), dim3(256), 0, ctx->stream, d_positions, d_normals, n_points,
                       light_position[0], light_position[1], light_position[2], d_rays);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_primary_rays_dev(VdCtx* ctx, const VdCameraUniform* camera, uint32_t width, uint32_t height, VdRay* d_rays) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!camera) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_primary_rays: null camera");
    const uint64_t n = (uint64_t)width * height;
    if (n == 0) return VD_OK;
    if (!d_rays) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_primary_rays: null rays");
    if (n > 0xffffff00ull) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_primary_rays: width * height does not fit 32 bits");
    Mat4 c2w;
    for (int k = 0; k < 16; ++k) c2w.m[k] = camera->clip_to_world[k];
    hipLaunchKernelGGL(primary_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, c2w, width, height, (unsigned)n, d_rays);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_traverse_iter_dev(VdCtx* ctx, const VdBvhNode* d_nodes, uint32_t n_nodes, const float* d_verts_xyz, const uint32_t* d_indices,
                         const VdRay* d_rays, uint32_t n_rays, float* d_out_dist) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!d_nodes || n_nodes == 0 || !d_verts_xyz || !d_indices) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_traverse_iter: incomplete mesh");
    if (n_rays == 0) return VD_OK;
    if (!d_rays || !d_out_dist) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_traverse_iter: null rays/out");
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, 256);
    if (rc) return rc;
    unsigned* d_flag = reinterpret_cast<unsigned*>(ctx->scratch);
    vd_time_begin(ctx);
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_flag, 0, 4, ctx->stream));
    hipLaunchKernelGGL(traverse_iter_kernel, dim3((n_rays + 63u) / 64u), dim3(64), 0, ctx->stream, d_nodes, d_verts_xyz, d_indices, d_rays,
                       n_rays, d_out_dist, d_flag);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->host_pinned, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->host_pinned[0]) VD_FAIL(ctx, VD_ERR_STACK_OVERFLOW, "vd_traverse_iter: traversal stack (128 entries per ray) exceeded");
    return VD_OK;
}

int vd_traverse_dev(VdCtx* ctx, const VdBvhNode* d_nodes, uint32_t n_nodes, const float* d_verts_xyz, const uint32_t* d_indices,
                    const VdRay* d_rays, uint32_t n_rays, float t0, float* d_out_dist) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!d_nodes || n_nodes == 0 || !d_verts_xyz || !d_indices) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_traverse: incomplete mesh");
    if (n_rays == 0) return VD_OK;
    if (!d_rays || !d_out_dist) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_traverse: null rays/out");
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, 256);
    if (rc) return rc;
    unsigned* d_flag = reinterpret_cast<unsigned*>(ctx->scratch);
    vd_time_begin(ctx);
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_flag, 0, 4, ctx->stream));
    hipLaunchKernelGGL(traverse_rec_kernel, dim3((n_rays + 63u) / 64u), dim3(64), 0, ctx->stream, d_nodes, d_verts_xyz, d_indices, d_rays,
                       n_rays, t0, d_out_dist, d_flag);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->host_pinned, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->host_pinned[0]) VD_FAIL(ctx, VD_ERR_STACK_OVERFLOW, "vd_traverse: traversal stack (128 pending right children per ray) exceeded");
    return VD_OK;
}

int vd_trace(VdCtx* ctx, const VdTraceScene* scene, const VdRay* rays, uint32_t n_rays, VdHit* out) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
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
    rc = launch_trace(ctx, &d, nullptr, reinterpret_cast<const VdRay*>(base + off[6]), n_rays, d_out);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(out, d_out, sz[7], hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VD_OK;
}

int vd_primary_rays(VdCtx* ctx, const VdCameraUniform* camera, uint32_t width, uint32_t height, VdRay* rays) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    const uint64_t n = (uint64_t)width * height;
    if (n && !rays) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_primary_rays: null rays");
    int rc = vd_ensure(ctx, &ctx->stage_out, &ctx->stage_out_bytes, (size_t)n * sizeof(VdRay));
    if (rc) return rc;
    rc = vd_primary_rays_dev(ctx, camera, width, height, reinterpret_cast<VdRay*>(ctx->stage_out));
    if (rc || n == 0) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(rays, ctx->stage_out, (size_t)n * sizeof(VdRay), hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VD_OK;
}

static int traverse_host(VdCtx* ctx, bool recursive, float t0, const VdBvhNode* nodes, uint32_t n_nodes, const float* verts_xyz, uint32_t n_vert,
                         const uint32_t* indices, uint32_t n_tri, const VdRay* rays, uint32_t n_rays, float* out_dist) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!nodes || n_nodes == 0 || !verts_xyz || !indices) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_traverse[_iter]: incomplete mesh");
    if (n_rays == 0) return VD_OK;
    if (!rays || !out_dist) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_traverse[_iter]: null rays/out");
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t sz[5] = {(size_t)n_nodes * sizeof(VdBvhNode), (size_t)n_vert * 12, (size_t)n_tri * 12, (size_t)n_rays * sizeof(VdRay),
                          (size_t)n_rays * 4};
    const void* src[4] = {nodes, verts_xyz, indices, rays};
    size_t off[5], total = 0;
    for (int k = 0; k < 5; ++k) { off[k] = total; total += (sz[k] + 255) & ~(size_t)255; }
    int rc = vd_ensure(ctx, &ctx->stage_in, &ctx->stage_in_bytes, total);
    if (rc) return rc;
    char* base = reinterpret_cast<char*>(ctx->stage_in);
    for (int k = 0; k < 4; ++k)
        if (sz[k]) VD_HIP_CHECK(ctx, hipMemcpyAsync(base + off[k], src[k], sz[k], hipMemcpyHostToDevice, ctx->stream));
    float* d_out = reinterpret_cast<float*>(base + off[4]);
    const VdBvhNode* dn = reinterpret_cast<const VdBvhNode*>(base + off[0]);
    const float* dv = reinterpret_cast<const float*>(base + off[1]);
    const uint32_t* di = reinterpret_cast<const uint32_t*>(base + off[2]);
    const VdRay* dr = reinterpret_cast<const VdRay*>(base + off[3]);
    rc = recursive ? vd_traverse_dev(ctx, dn, n_nodes, dv, di, dr, n_rays, t0, d_out) : vd_traverse_iter_dev(ctx, dn, n_nodes, dv, di, dr, n_rays, d_out);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(out_dist, d_out, sz[4], hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VD_OK;
}

int vd_traverse_iter(VdCtx* ctx, const VdBvhNode* nodes, uint32_t n_nodes, const float* verts_xyz, uint32_t n_vert,
                     const uint32_t* indices, uint32_t n_tri, const VdRay* rays, uint32_t n_rays, float* out_dist) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    return traverse_host(ctx, false, 0.0f, nodes, n_nodes, verts_xyz, n_vert, indices, n_tri, rays, n_rays, out_dist);
}

int vd_traverse(VdCtx* ctx, const VdBvhNode* nodes, uint32_t n_nodes, const float* verts_xyz, uint32_t n_vert,
                const uint32_t* indices, uint32_t n_tri, const VdRay* rays, uint32_t n_rays, float t0, float* out_dist) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    return traverse_host(ctx, true, t0, nodes, n_nodes, verts_xyz, n_vert, indices, n_tri, rays, n_rays, out_dist);
}

}  // extern "C"

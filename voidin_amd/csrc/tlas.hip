// tlas.hip — TLAS leaf boxes, exact agglomerative build and refit on gfx950.
//
// Replaces Tlas::build (reference: crates/bvh/src/tlas.rs:31-105), called once per scene from
// MeshPool::generate_tlas (crates/pools/src/mesh/mod.rs:279-286), and adds refit (SURVEY.md
// §8a T3, not in the reference).
//
//  * leaf boxes (tlas.rs:34-54): one lane per instance, 8 transformed corners folded with the
//    object-space mesh box as seed (bug-compatible), total-order min/max;
//  * build (tlas.rs:56-84): the reference is a sequential chain of ~2.5 N `find_best_match`
//    scans whose tie-breaking depends on the slot order, so the chain itself cannot be
//    reordered.  One 1024-lane workgroup runs the chain (16 of them from 12288 instances on, see
//    "build, several workgroups"); each scan is data-parallel over the
//    active slots, which are kept as a compacted SoA (six float arrays + node ids) so a scan is
//    a pure stream of 24 B per slot; the argmin is a 64-bit {area bits, slot} key reduced by
//    DPP / wave shuffles + per-wave LDS words, which reproduces "strict <, first slot wins";
//  * refit: leaves recomputed, interior boxes by a bottom-up walk with per-node handshake counters
//    (write-through agent-scope stores, no fences; boxes re-read L1/L2-bypassing).
#include "vd_common.hpp"

namespace {

constexpr int kBuildThreads = 1024;

struct Box { float mn[3], mx[3]; };

// tlas.rs:35-44.  glam Mat4::transform_point3: ((X*p.x + Y*p.y) + Z*p.z) + W, no FMA.
__device__ __forceinline__ Box leaf_box(const VdInstance* __restrict__ inst, const VdMeshInfo* __restrict__ meshes,
                                        unsigned n_mesh, unsigned i) {
    const float4* t4 = reinterpret_cast<const float4*>(inst + i);
    const float4 X = t4[0], Y = t4[1], Z = t4[2], W = t4[3];
    const unsigned mesh_id = reinterpret_cast<const unsigned*>(inst + i)[32];
    const VdMeshInfo* m = meshes + min(mesh_id, n_mesh - 1u);
    const float b[2][3] = {{m->min[0], m->min[1], m->min[2]}, {m->max[0], m->max[1], m->max[2]}};
    Box r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.mn[k] = b[0][k]; r.mx[k] = b[1][k]; }   // fold seed: object-space box
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float px = b[(c & 1) == 0][0], py = b[(c & 2) == 0][1], pz = b[(c & 4) == 0][2];
        const float p[3] = {((X.x * px + Y.x * py) + Z.x * pz) + W.x, ((X.y * px + Y.y * py) + Z.y * pz) + W.y,
                            ((X.z * px + Y.z * py) + Z.z * pz) + W.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) { r.mn[k] = vd_min_to(r.mn[k], p[k]); r.mx[k] = vd_max_to(r.mx[k], p[k]); }
    }
    return r;
}

template <typename Node> __device__ __forceinline__ void node_set_children(Node& n, unsigned l, unsigned r);
template <> __device__ __forceinline__ void node_set_children<VdTlasNode>(VdTlasNode& n, unsigned l, unsigned r) {
    n.left_right = l + (r << 16);   // tlas.rs:71
}
template <> __device__ __forceinline__ void node_set_children<VdTlasNodeWide>(VdTlasNodeWide& n, unsigned l, unsigned r) {
    n.left = l; n.right = r; n._pad[0] = n._pad[1] = n._pad[2] = 0u;
}
template <typename Node> __device__ __forceinline__ void node_get_children(const Node& n, unsigned& l, unsigned& r);
template <> __device__ __forceinline__ void node_get_children<VdTlasNode>(const VdTlasNode& n, unsigned& l, unsigned& r) {
    l = n.left_right & 0xffffu; r = n.left_right >> 16;
}
template <> __device__ __forceinline__ void node_get_children<VdTlasNodeWide>(const VdTlasNodeWide& n, unsigned& l, unsigned& r) {
    l = n.left; r = n.right;
}

// Leaves for build: nodes[i+1] = {box, leaf, instance i}; slot arrays seeded in slot order.
template <typename Node>
__global__ __launch_bounds__(256) void tlas_leaves_kernel(const VdInstance* __restrict__ inst, unsigned n,
                                                          const VdMeshInfo* __restrict__ meshes, unsigned n_mesh,
                                                          Node* __restrict__ nodes, float* __restrict__ slot_box /* [6][cap] or null */,
                                                          unsigned* __restrict__ slot_node, unsigned cap, int refit,
                                                          const unsigned* __restrict__ only_if = nullptr) {
    if (only_if && *only_if == 0u) return;                  // second attempt of a build: only when the first gave up
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const unsigned src = refit ? nodes[i + 1].instance_idx : i;
    const Box b = leaf_box(inst, meshes, n_mesh, src);
    Node nd = nodes[i + 1];
    if (!refit) {
        node_set_children(nd, 0u, 0u);
        nd.instance_idx = i;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { nd.min[k] = b.mn[k]; nd.max[k] = b.mx[k]; }
    nodes[i + 1] = nd;
    if (slot_box) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { slot_box[k * cap + i] = b.mn[k]; slot_box[(3 + k) * cap + i] = b.mx[k]; }
        slot_node[i] = i + 1;
    }
}

// {area, slot} -> unsigned-comparable key; areas of real boxes are >= +0, the monotone map also
// orders garbage (negative / NaN-free) input the way `<` does.
__device__ __forceinline__ vd_u64 match_key(float area, unsigned slot) {
    const unsigned k = (unsigned)vd_key(area + 0.0f) ^ 0x80000000u;
    return ((vd_u64)k << 32) | slot;
}

// min over the wave of a 64-bit key.  Within a 16-lane row on the VALU (DPP: quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror), across the four rows through the LDS crossbar - 4 crossbar moves instead of 12: the reduction sits
// on the critical path of every scan.
template <int CTRL> __device__ __forceinline__ vd_u64 dpp_u64(vd_u64 v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, true);
    return ((vd_u64)hi << 32) | lo;
}
__device__ __forceinline__ vd_u64 wave_min_u64(vd_u64 v) {
    vd_u64 o;
    o = dpp_u64<0xB1>(v); v = o < v ? o : v;
    o = dpp_u64<0x4E>(v); v = o < v ? o : v;
    o = dpp_u64<0x141>(v); v = o < v ? o : v;
    o = dpp_u64<0x140>(v); v = o < v ? o : v;
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, off), hi = __shfl_xor((unsigned)(v >> 32), off);
        o = ((vd_u64)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

// tlas.rs:87-105 over the compacted slot arrays; every thread returns the same slot.
// A scan is bound by VALU issue on the one CU that runs the chain (16 k slots x ~50 instructions / 64 lanes per
// clock ~ 5 us), not by the 24 B per slot it streams from L2, so the per-slot arithmetic is what counts.
// FAST (no NaN in any leaf box, checked once; unions of NaN-free boxes are NaN-free): the union extents use
// v_min/v_max instead of the total-order min/max.  Only the sign of a zero extent can differ, the area is then the
// same +0/-0-normalised value, so the argmin is unchanged; the boxes written to the nodes always use the total order.
// Per thread the slots come in increasing order, so `strictly smaller area` alone keeps the first slot.
template <bool FAST>
__device__ __forceinline__ unsigned find_best_match(const float* sb, unsigned cap, unsigned cnt,
                                                    unsigned target, vd_u64* s_red, unsigned call) {
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    vd_u64 best = ~0ull;
    unsigned fbits = 0xffffffffu, fslot = 0;
    if (target < cap) {
        const float t0 = sb[target], t1 = sb[cap + target], t2 = sb[2 * cap + target];
        const float t3 = sb[3 * cap + target], t4 = sb[4 * cap + target], t5 = sb[5 * cap + target];
        // four consecutive slots per lane per step: six independent 16-B loads in flight (issuing the next step's
        // loads before this step's arithmetic was measured: 20 % slower)
        for (unsigned i0 = tid * 4u; i0 < cnt; i0 += kBuildThreads * 4u) {
            const float4 a0 = *reinterpret_cast<const float4*>(sb + i0), a1 = *reinterpret_cast<const float4*>(sb + cap + i0);
            const float4 a2 = *reinterpret_cast<const float4*>(sb + 2 * cap + i0), a3 = *reinterpret_cast<const float4*>(sb + 3 * cap + i0);
            const float4 a4 = *reinterpret_cast<const float4*>(sb + 4 * cap + i0), a5 = *reinterpret_cast<const float4*>(sb + 5 * cap + i0);
            const float mn0[4] = {a0.x, a0.y, a0.z, a0.w}, mn1[4] = {a1.x, a1.y, a1.z, a1.w}, mn2[4] = {a2.x, a2.y, a2.z, a2.w};
            const float mx0[4] = {a3.x, a3.y, a3.z, a3.w}, mx1[4] = {a4.x, a4.y, a4.z, a4.w}, mx2[4] = {a5.x, a5.y, a5.z, a5.w};
            if (FAST) {
                // two slots per instruction where gfx950 has packed fp32 (v_pk_add_f32 / v_pk_mul_f32); min / max stay
                // scalar.  Same operations in the same order as vd_area, no contraction.
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const f32x2 hx = {__builtin_fmaxf(t3, mx0[k]), __builtin_fmaxf(t3, mx0[k + 1])}, lx = {__builtin_fminf(t0, mn0[k]), __builtin_fminf(t0, mn0[k + 1])};
                    const f32x2 hy = {__builtin_fmaxf(t4, mx1[k]), __builtin_fmaxf(t4, mx1[k + 1])}, ly = {__builtin_fminf(t1, mn1[k]), __builtin_fminf(t1, mn1[k + 1])};
                    const f32x2 hz = {__builtin_fmaxf(t5, mx2[k]), __builtin_fmaxf(t5, mx2[k + 1])}, lz = {__builtin_fminf(t2, mn2[k]), __builtin_fminf(t2, mn2[k + 1])};
                    const f32x2 dx = hx - lx, dy = hy - ly, dz = hz - lz;
                    const f32x2 two = {2.0f, 2.0f}, zero = {0.0f, 0.0f};
                    const f32x2 ar = ((dx * dy + dx * dz) + dy * dz) * two + zero;   // -0 -> +0: the bit pattern is then monotone
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned i = i0 + k + h;
                        const float area = ar[h];
                        const bool ok = i < cnt && i != target && area < 1e30f;    // from 1e30, NaN (inf * 0) never passes
                        const unsigned bits = ok ? __float_as_uint(area) : 0xffffffffu;
                        if (bits < fbits) { fbits = bits; fslot = i; }
                    }
                }
            }
            if (!FAST) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned i = i0 + k;
                    const float dx = vd_max_to(t3, mx0[k]) - vd_min_to(t0, mn0[k]);
                    const float dy = vd_max_to(t4, mx1[k]) - vd_min_to(t1, mn1[k]);
                    const float dz = vd_max_to(t5, mx2[k]) - vd_min_to(t2, mn2[k]);
                    const float area = vd_area(dx, dy, dz);
                    if (i < cnt && i != target && area < 1e30f) {   // `surface_area < smallest` from 1e30, NaN never passes
                        const vd_u64 kk = match_key(area, i);
                        best = kk < best ? kk : best;
                    }
                }
            }
        }
    }
    // one barrier per scan: every wave leaves its key in its own word (two-deep by scan parity, so the words of this
    // scan are not rewritten before everybody has read them) and every wave reduces the 16 words itself, in one DPP row
    if (FAST && fbits != 0xffffffffu) best = match_key(__uint_as_float(fbits), fslot);
    best = wave_min_u64(best);
    vd_u64* words = s_red + (call & 1u) * 16u;
    if (lane == 0) words[tid >> 6] = best;
    __syncthreads();
    vd_u64 v = lane < (unsigned)(kBuildThreads / 64) ? words[lane] : ~0ull;
    { vd_u64 o; o = dpp_u64<0xB1>(v); v = o < v ? o : v; o = dpp_u64<0x4E>(v); v = o < v ? o : v;
      o = dpp_u64<0x141>(v); v = o < v ? o : v; o = dpp_u64<0x140>(v); v = o < v ? o : v; }        // lanes 0..15 hold the minimum
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (lo & hi) == 0xffffffffu ? target : lo;
}

// tlas.rs:56-84 — one workgroup runs the whole chain.
template <typename Node, bool FAST>
__device__ __forceinline__ void tlas_build_chain(Node* __restrict__ nodes, unsigned n, float* sb, unsigned* slot_node,
                                                 unsigned cap, vd_u64* s_red) {
    unsigned call = 0;
    unsigned cnt = n, used = n + 1, a = 0;
    unsigned b = find_best_match<FAST>(sb, cap, cnt, a, s_red, call++);
    while (cnt > 0) {
        const unsigned c = find_best_match<FAST>(sb, cap, cnt, b, s_red, call++);
        if (a == c) {
            if (threadIdx.x == 0) {
                const unsigned idx_a = slot_node[a], idx_b = slot_node[b];
                Node nd;
                float u[6];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    u[k] = vd_min_to(sb[k * cap + a], sb[k * cap + b]);
                    u[3 + k] = vd_max_to(sb[(3 + k) * cap + a], sb[(3 + k) * cap + b]);
                    nd.min[k] = u[k];
                    nd.max[k] = u[3 + k];
                }
                node_set_children(nd, idx_a, idx_b);
                nd.instance_idx = 0xffffffffu;
                nodes[used] = nd;
                // node_indices[a] = nodes_used; node_indices[b] = node_indices[cnt - 1]  (in this order)
#pragma unroll
                for (int k = 0; k < 6; ++k) sb[k * cap + a] = u[k];
                slot_node[a] = used;
                const unsigned last = cnt - 1;
#pragma unroll
                for (int k = 0; k < 6; ++k) sb[k * cap + b] = sb[k * cap + last];
                slot_node[b] = slot_node[last];
            }
            used += 1;
            cnt -= 1;
            __syncthreads();
            b = find_best_match<FAST>(sb, cap, cnt, a, s_red, call++);
        } else {
            a = b;
            b = c;
        }
    }
    if (threadIdx.x == 0) nodes[0] = nodes[slot_node[a]];   // tlas.rs:84
}

template <typename Node>
__global__ __launch_bounds__(kBuildThreads) void tlas_build_kernel(Node* __restrict__ nodes, unsigned n,
                                                                   float* sb, unsigned* slot_node,
                                                                   unsigned cap, const unsigned* __restrict__ only_if) {
    if (only_if && *only_if == 0u) return;                  // see tlas_build_impl
    __shared__ vd_u64 s_red[32];   // [2][16] per-wave keys by scan parity
    if (threadIdx.x < 32) s_red[threadIdx.x] = ~0ull;
    int nan = 0;
    for (unsigned i = threadIdx.x; i < n; i += kBuildThreads) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { const float v = sb[q * cap + i]; nan |= v != v; }
    }
    const bool any_nan = __syncthreads_or(nan) != 0;         // also orders the s_red initialisation
    if (any_nan) tlas_build_chain<Node, false>(nodes, n, sb, slot_node, cap, s_red);
    else tlas_build_chain<Node, true>(nodes, n, sb, slot_node, cap, s_red);
}

// ---- build, several workgroups ---------------------------------------------------------------
// The same chain, the scans spread over W co-resident workgroups (one per CU).  Every workgroup runs the identical
// control flow (a, b, c, cnt are functions of the scan results), scans a contiguous W-th of the active slots and
// then takes part in one exchange: it stores its best {area bits, slot, tag} key into its own word of a two-deep ring
// and one wave polls the W words of the scan until all carry the scan's tag - the data is the flag, so an exchange is
// one write-through store and a few polling loads (~0.5 us; `tools/probe_exchange.hip`: an atomic-min + arrival-counter
// exchange costs 1.4 - 2.2 us).  Entry (r % 2, w) is rewritten at scan r + 2, which w reaches only after every workgroup has
// published r + 1, i.e. has finished reading r.  A merge is applied by workgroup 0 alone; the others wait for its
// merge counter.  Slots are shared through memory: all reads and writes of them are agent-scope (L1/L2-bypassing
// loads, write-through stores), no fences.  Spins are bounded: on a timeout every workgroup sets the flag and leaves,
// and the single-workgroup kernel that is queued behind (it returns at once when the flag is clear) redoes the build.
constexpr unsigned kMwMaxGroups = 32;
#ifndef VD_MW_THREADS
#define VD_MW_THREADS 1024
#endif
constexpr int kMwThreads = VD_MW_THREADS;
constexpr unsigned kSpinLimit = 2000000u;   // polls before a workgroup gives up (~1 s; an exchange takes ~1 us)
struct MwShared {
    unsigned long long key[2][kMwMaxGroups];
    unsigned merges;       // merge counter published by workgroup 0
    unsigned fail;
};

// Workgroups are dealt to the 8 XCDs round-robin: the chain uses every 8th workgroup of the grid, i.e. CUs of ONE XCD,
// whose shared L2 then serves the exchange and the slot traffic (correctness does not depend on it: all shared
// accesses are agent-scope).
__device__ __forceinline__ unsigned mw_group() { return blockIdx.x >> 3; }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one scan: returns the best slot for `target` (the same value in every thread of every workgroup), or ~0u on a timeout
template <bool FAST>
__device__ __forceinline__ unsigned mw_find_best_match(float* sb, unsigned cap, unsigned cnt, unsigned target, MwShared* sh,
                                                       unsigned W, unsigned call, vd_u64* s_red, unsigned spin_limit) {
    const unsigned tid = threadIdx.x, lane = tid & 63u, w = mw_group();
    vd_u64 best = ~0ull;
    if (target < cap) {
        float t[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = ld_agent(sb + k * cap + target);
        // contiguous slice of slot PAIRS per workgroup
        const unsigned pairs = (cnt + 1u) >> 1, per = (pairs + W - 1u) / W;
        const unsigned p_lo = w * per, p_hi = min(pairs, p_lo + per);
        for (unsigned pr = p_lo + tid; pr < p_hi; pr += kMwThreads) {
            const unsigned i0 = 2u * pr;
            float v[6][2];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const vd_u64 q = __hip_atomic_load(reinterpret_cast<const vd_u64*>(sb + k * cap + i0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[k][0] = __uint_as_float((unsigned)q); v[k][1] = __uint_as_float((unsigned)(q >> 32));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned i = i0 + h;
                float dx, dy, dz;
                if (FAST) {
                    dx = __builtin_fmaxf(t[3], v[3][h]) - __builtin_fminf(t[0], v[0][h]);
                    dy = __builtin_fmaxf(t[4], v[4][h]) - __builtin_fminf(t[1], v[1][h]);
                    dz = __builtin_fmaxf(t[5], v[5][h]) - __builtin_fminf(t[2], v[2][h]);
                } else {
                    dx = vd_max_to(t[3], v[3][h]) - vd_min_to(t[0], v[0][h]);
                    dy = vd_max_to(t[4], v[4][h]) - vd_min_to(t[1], v[1][h]);
                    dz = vd_max_to(t[5], v[5][h]) - vd_min_to(t[2], v[2][h]);
                }
                const float area = vd_area(dx, dy, dz);
                if (i < cnt && i != target && area < 1e30f) {   // `surface_area < smallest` from 1e30, NaN never passes
                    const vd_u64 kk = match_key(area, i);
                    best = kk < best ? kk : best;
                }
            }
        }
    }
    best = wave_min_u64(best);
    // workgroup minimum: every wave leaves its key in its own word (two-deep by scan parity: no second barrier), wave 0
    // reduces the <= 16 words inside one DPP row - no LDS atomics on one address
    const unsigned ring = call & 1u;
    constexpr unsigned kWaves = kMwThreads / 64;
    static_assert(kWaves <= 16, "one DPP row");
    if (lane == 0) s_red[ring * 16u + (tid >> 6)] = best;
    __syncthreads();
    // exchange: {area bits : 32, slot : 20, tag : 12}; "nothing" = all ones above the tag
    const vd_u64 tag = (vd_u64)(((call >> 1) + 1u) & 0xfffu);
    if (tid < 64u) {
        vd_u64 wg = lane < kWaves ? s_red[ring * 16u + lane] : ~0ull;
        { vd_u64 o; o = dpp_u64<0xB1>(wg); wg = o < wg ? o : wg; o = dpp_u64<0x4E>(wg); wg = o < wg ? o : wg;
          o = dpp_u64<0x141>(wg); wg = o < wg ? o : wg; o = dpp_u64<0x140>(wg); wg = o < wg ? o : wg; }   // lanes 0..15 hold the minimum
        if (tid == 0) {
            const vd_u64 mine = wg;
            const vd_u64 packed = mine == ~0ull ? (~0ull << 12) : ((mine >> 32) << 32 | (mine & 0xfffffull) << 12);
            __hip_atomic_store(&sh->key[ring][w], packed | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        vd_u64 got = ~0ull;
        bool ok = true;
        if (lane < W) {
            unsigned spins = 0;
            for (;;) {
                got = __hip_atomic_load(&sh->key[ring][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((got & 0xfffull) == tag) break;
                if (++spins > spin_limit || __hip_atomic_load(&sh->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
            }
            got >>= 12;
        }
        if (!__all(ok)) {
            if (lane == 0) __hip_atomic_store(&sh->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            got = ~0ull - 1ull;                       // timeout marker
        } else if (W <= 16u) {                          // the W words sit in one DPP row
            vd_u64 o; o = dpp_u64<0xB1>(got); got = o < got ? o : got; o = dpp_u64<0x4E>(got); got = o < got ? o : got;
            o = dpp_u64<0x141>(got); got = o < got ? o : got; o = dpp_u64<0x140>(got); got = o < got ? o : got;
        } else {
            got = wave_min_u64(got);
        }
        if (lane == 0) s_red[32u + ring] = got;         // two-deep: the next scan writes the other word
    }
    __syncthreads();
    const vd_u64 g = s_red[32u + ring];
    if (g == ~0ull - 1ull) return ~0u;
    if ((g >> 20) == 0xffffffffull) return target;     // nobody had a candidate
    return (unsigned)(g & 0xfffffull);
}

template <typename Node, bool FAST>
__device__ __forceinline__ void mw_build_chain(Node* __restrict__ nodes, unsigned n, float* sb, unsigned* slot_node, unsigned cap,
                                               MwShared* sh, unsigned W, vd_u64* s_red, unsigned spin_limit) {
    unsigned call = 0, merges = 0;
    unsigned cnt = n, used = n + 1, a = 0;
    unsigned b = mw_find_best_match<FAST>(sb, cap, cnt, a, sh, W, call++, s_red, spin_limit);
    if (b == ~0u) return;
    while (cnt > 0) {
        const unsigned c = mw_find_best_match<FAST>(sb, cap, cnt, b, sh, W, call++, s_red, spin_limit);
        if (c == ~0u) return;
        if (a == c) {
            merges += 1;
            if (mw_group() == 0u) {
                if (threadIdx.x == 0) {
                    const unsigned idx_a = slot_node[a], idx_b = slot_node[b];
                    Node nd;
                    float u[6];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        u[k] = vd_min_to(ld_agent(sb + k * cap + a), ld_agent(sb + k * cap + b));
                        u[3 + k] = vd_max_to(ld_agent(sb + (3 + k) * cap + a), ld_agent(sb + (3 + k) * cap + b));
                        nd.min[k] = u[k];
                        nd.max[k] = u[3 + k];
                    }
                    node_set_children(nd, idx_a, idx_b);
                    nd.instance_idx = 0xffffffffu;
                    nodes[used] = nd;
                    // node_indices[a] = nodes_used; node_indices[b] = node_indices[cnt - 1]  (in this order)
                    const unsigned last = cnt - 1;
                    float mv[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) st_agent(sb + k * cap + a, u[k]);
                    slot_node[a] = used;
#pragma unroll
                    for (int k = 0; k < 6; ++k) mv[k] = last == a ? u[k] : ld_agent(sb + k * cap + last);   // slot a as just rewritten
#pragma unroll
                    for (int k = 0; k < 6; ++k) st_agent(sb + k * cap + b, mv[k]);
                    slot_node[b] = slot_node[last];
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&sh->merges, merges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (threadIdx.x == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(&sh->merges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != merges) {
                    if (++spins > spin_limit || __hip_atomic_load(&sh->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(&sh->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            used += 1;
            cnt -= 1;
            __syncthreads();
            b = mw_find_best_match<FAST>(sb, cap, cnt, a, sh, W, call++, s_red, spin_limit);
            if (b == ~0u) return;
        } else {
            a = b;
            b = c;
        }
    }
    if (mw_group() == 0u && threadIdx.x == 0) nodes[0] = nodes[slot_node[a]];   // tlas.rs:84
}

template <typename Node>
__global__ __launch_bounds__(kMwThreads) void tlas_build_mw_kernel(Node* __restrict__ nodes, unsigned n, float* sb,
                                                                      unsigned* slot_node, unsigned cap, MwShared* sh, unsigned spin_limit) {
    __shared__ vd_u64 s_red[34];   // [2][16] per-wave keys by scan parity, [32 + parity] the exchanged result
    if (threadIdx.x < 34) s_red[threadIdx.x] = ~0ull;
    int nan = 0;
    for (unsigned i = threadIdx.x; i < n; i += kMwThreads) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { const float v = sb[q * cap + i]; nan |= v != v; }
    }
    const bool any_nan = __syncthreads_or(nan) != 0;         // every workgroup sees the same leaves: same answer
    if ((blockIdx.x & 7u) != 0u) return;                      // see mw_group()
    if (any_nan) mw_build_chain<Node, false>(nodes, n, sb, slot_node, cap, sh, gridDim.x >> 3, s_red, spin_limit);
    else mw_build_chain<Node, true>(nodes, n, sb, slot_node, cap, sh, gridDim.x >> 3, s_red, spin_limit);
}

// ---- refit -------------------------------------------------------------------------------
// The agglomerative tree is tall and thin (32 768 instances of the bench cloud: height 70, and about half the levels
// of the longest path join a cluster with a single leaf), so a refit is one long dependent climb and what counts is
// the number of memory-side round trips per level.
// Launch 1: thread i < n recomputes leaf i+1; the same thread records, for the two children of interior node n+1+i,
// {parent, sibling} and clears the node's handshake word.
template <typename Node>
__global__ __launch_bounds__(256) void tlas_refit_prep_kernel(const VdInstance* __restrict__ inst, unsigned n,
                                                              const VdMeshInfo* __restrict__ meshes, unsigned n_mesh,
                                                              Node* __restrict__ nodes, uint2* __restrict__ up,
                                                              unsigned* __restrict__ arrivals) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const Box b = leaf_box(inst, meshes, n_mesh, nodes[i + 1].instance_idx);
#pragma unroll
    for (int c = 0; c < 3; ++c) { nodes[i + 1].min[c] = b.mn[c]; nodes[i + 1].max[c] = b.mx[c]; }
    const unsigned k = n + 1u + i;
    unsigned l, r;
    node_get_children(nodes[k], l, r);
    up[l] = make_uint2(k, r);
    up[r] = make_uint2(k, l);
    arrivals[k] = 0u;
}

template <typename Node> __device__ __forceinline__ Box load_box_plain(const Node* nodes, unsigned k) {
    Box b;
#pragma unroll
    for (int c = 0; c < 3; ++c) { b.mn[c] = nodes[k].min[c]; b.mx[c] = nodes[k].max[c]; }
    return b;
}
template <typename Node> __device__ __forceinline__ Box load_box_agent(Node* nodes, unsigned k) {
    Box b;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        b.mn[c] = __hip_atomic_load(&nodes[k].min[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b.mx[c] = __hip_atomic_load(&nodes[k].max[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return b;
}
__device__ __forceinline__ Box box_union(const Box& a, const Box& b) {
    Box u;
#pragma unroll
    for (int c = 0; c < 3; ++c) { u.mn[c] = vd_min_to(a.mn[c], b.mn[c]); u.mx[c] = vd_max_to(a.mx[c], b.mx[c]); }
    return u;
}

// Launch 2: climbs start at the interior nodes whose children are both leaves (the lane of the smaller leaf) and
// carry the box of the node just finished in registers.
//  * sibling is a leaf: its box is final since launch 1 - plain load, no handshake, nothing to wait for.  A chain
//    of such levels costs one L2 hit each ({parent, sibling} of the next level is fetched with the sibling's box).
//  * sibling is an interior node: two climbers meet.  No fences (an agent-scope release writes back the XCD's
//    whole L2); boxes travel as write-through agent-scope stores, complete once vmcnt is 0, and are read back L1/L2-
//    bypassing.  The handshake word counts +1 "here" and +2 "my box is readable": a climber that finds the other
//    side already readable (3) goes on after TWO round trips (announce, read) without waiting for its own stores;
//    otherwise it publishes and the later of the two "readable" marks goes on (three round trips, as a plain
//    arrival counter needs every time).  Nobody ever waits for another lane.
template <typename Node>
__global__ __launch_bounds__(256) void tlas_refit_up_kernel(Node* nodes, unsigned n, const uint2* __restrict__ up,
                                                            unsigned* arrivals) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const unsigned c = i + 1u;
    const uint2 u0 = up[c];
    if (u0.y > n || c > u0.y) return;                     // sibling interior: its climber picks this leaf up; or the other leaf's lane
    unsigned k = u0.x;
    Box box = box_union(load_box_plain(nodes, c), load_box_plain(nodes, u0.y));
    for (;;) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            __hip_atomic_store(&nodes[k].min[q], box.mn[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nodes[k].max[q], box.mx[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (k == 2u * n) {                                // tlas.rs:84: nodes[0] is a copy of the last node
#pragma unroll
            for (int q = 0; q < 3; ++q) { nodes[0].min[q] = box.mn[q]; nodes[0].max[q] = box.mx[q]; }
            return;
        }
        const uint2 u = up[k];                            // {parent, sibling}
        const unsigned p = u.x, s = u.y;
        if (s == k) { k = p; continue; }                  // node 2n merges the true root with itself: same box
        if (s <= n) { box = box_union(box, load_box_plain(nodes, s)); k = p; continue; }
        const unsigned v = __hip_atomic_fetch_add(&arrivals[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != 3u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my box has reached memory
            const unsigned w = __hip_atomic_fetch_add(&arrivals[p], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w < 4u) return;                           // the other side is not readable yet: its climber goes on
        }
        box = box_union(box, load_box_agent(nodes, s));
        k = p;
    }
}

template <typename Node>
int tlas_build_impl(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                    Node* d_nodes) {
    // scratch: 6 float slot arrays + slot node ids, capacity n
    const size_t cap = ((size_t)n + 7) & ~(size_t)3;   // multiple of 4 (+ slack): slot arrays are read 16 B at a time
    const size_t need = cap * 7 * 4 + 256 + sizeof(MwShared);
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, need);
    if (rc) return rc;
    float* sb = reinterpret_cast<float*>(ctx->scratch);
    unsigned* slot_node = reinterpret_cast<unsigned*>(sb + 6 * cap);
    MwShared* sh = reinterpret_cast<MwShared*>(reinterpret_cast<char*>(ctx->scratch) + ((cap * 7 * 4 + 255) & ~(size_t)255));
    // several workgroups from a few thousand instances on (below, the exchange costs more than the shorter scans save);
    // the slot field of the exchanged key holds 20 bits.  VD_TLAS_GROUPS = 1 forces the single-workgroup kernel.
    const int env_groups = getenv("VD_TLAS_GROUPS") ? atoi(getenv("VD_TLAS_GROUPS")) : 0;
    const unsigned spin_limit = getenv("VD_TLAS_SPIN_LIMIT") ? (unsigned)atoi(getenv("VD_TLAS_SPIN_LIMIT")) : kSpinLimit;   // tests: 0 forces the fallback
    unsigned groups = env_groups > 0 ? (unsigned)env_groups : (n >= 12288u ? 16u : 1u);
    if (groups > kMwMaxGroups) groups = kMwMaxGroups;
    if (groups > (unsigned)ctx->num_cus) groups = (unsigned)ctx->num_cus;
    if (n >= (1u << 20)) groups = 1u;
    vd_time_begin(ctx);
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_nodes, 0, sizeof(Node) * (2 * (size_t)n + 1), ctx->stream));   // TlasNode::default()
    hipLaunchKernelGGL((tlas_leaves_kernel<Node>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_inst, n, d_meshes,
                       n_mesh, d_nodes, sb, slot_node, (unsigned)cap, 0, (const unsigned*)nullptr);
    if (groups <= 1u) {
        hipLaunchKernelGGL((tlas_build_kernel<Node>), dim3(1), dim3(kBuildThreads), 0, ctx->stream, d_nodes, n, sb, slot_node,
                           (unsigned)cap, (const unsigned*)nullptr);
    } else {
        VD_HIP_CHECK(ctx, hipMemsetAsync(sh, 0, sizeof(MwShared), ctx->stream));
        hipLaunchKernelGGL((tlas_build_mw_kernel<Node>), dim3(groups * 8u), dim3(kMwThreads), 0, ctx->stream, d_nodes, n, sb, slot_node,
                           (unsigned)cap, sh, spin_limit);
        // If the workgroups did not hear from each other in time (not co-resident: spin limit) they set sh->fail and
        // leave; the two launches below then redo the build on one workgroup, and return at once otherwise - decided
        // on the device, so the call stays asynchronous.  Every node a build writes is written again by the redo.
        hipLaunchKernelGGL((tlas_leaves_kernel<Node>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_inst, n, d_meshes,
                           n_mesh, d_nodes, sb, slot_node, (unsigned)cap, 0, (const unsigned*)&sh->fail);
        hipLaunchKernelGGL((tlas_build_kernel<Node>), dim3(1), dim3(kBuildThreads), 0, ctx->stream, d_nodes, n, sb, slot_node,
                           (unsigned)cap, (const unsigned*)&sh->fail);
    }
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

template <typename Node>
int tlas_refit_impl(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                    Node* d_nodes) {
    const size_t total = 2 * (size_t)n + 1;
    const size_t need = total * 12 + 256;
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, need);
    if (rc) return rc;
    uint2* parent = reinterpret_cast<uint2*>(ctx->scratch);                    // {parent, sibling} per node
    unsigned* arrivals = reinterpret_cast<unsigned*>(parent + total);
    vd_time_begin(ctx);
    hipLaunchKernelGGL((tlas_refit_prep_kernel<Node>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_inst, n, d_meshes, n_mesh,
                       d_nodes, parent, arrivals);
    hipLaunchKernelGGL((tlas_refit_up_kernel<Node>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_nodes, n, parent, arrivals);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int check_args(VdCtx* ctx, const void* inst, uint32_t n, const void* meshes, uint32_t n_mesh, const void* nodes, bool wide) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!inst || !meshes || !nodes || n == 0 || n_mesh == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_tlas_*: null pointer or zero count");
    if (!wide && n > VD_TLAS_MAX_INSTANCES)
        VD_FAIL(ctx, VD_ERR_TLAS_OVERFLOW, "vd_tlas_*: n > 32768 does not fit the 16-bit left_right packing (tlas.rs:71); use the _wide variant");
    if (n > 0x3fffffffu) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_tlas_*: n too large");
    return VD_OK;
}

// host-pointer wrapper: stage instances + meshes in, nodes out
template <typename Node, typename Fn>
int tlas_host(VdCtx* ctx, const VdInstance* inst, uint32_t n, const VdMeshInfo* meshes, uint32_t n_mesh, Node* nodes,
              bool upload_nodes, Fn fn) {
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t nb = sizeof(Node) * (2 * (size_t)n + 1);
    int rc = vd_ensure(ctx, &ctx->stage_in, &ctx->stage_in_bytes, (size_t)n * sizeof(VdInstance));
    if (rc) return rc;
    rc = vd_ensure(ctx, &ctx->stage_aux, &ctx->stage_aux_bytes, (size_t)n_mesh * sizeof(VdMeshInfo) + 16);
    if (rc) return rc;
    rc = vd_ensure(ctx, &ctx->stage_out, &ctx->stage_out_bytes, nb);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->stage_in, inst, (size_t)n * sizeof(VdInstance), hipMemcpyHostToDevice, ctx->stream));
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->stage_aux, meshes, (size_t)n_mesh * sizeof(VdMeshInfo), hipMemcpyHostToDevice, ctx->stream));
    if (upload_nodes) VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->stage_out, nodes, nb, hipMemcpyHostToDevice, ctx->stream));
    rc = fn(reinterpret_cast<const VdInstance*>(ctx->stage_in), reinterpret_cast<const VdMeshInfo*>(ctx->stage_aux),
            reinterpret_cast<Node*>(ctx->stage_out));
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(nodes, ctx->stage_out, nb, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VD_OK;
}

}  // namespace

extern "C" {

int vd_tlas_build_dev(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                      VdTlasNode* d_nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, d_inst, n, d_meshes, n_mesh, d_nodes, false);
    return rc ? rc : tlas_build_impl<VdTlasNode>(ctx, d_inst, n, d_meshes, n_mesh, d_nodes);
}
int vd_tlas_build_wide_dev(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                           VdTlasNodeWide* d_nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, d_inst, n, d_meshes, n_mesh, d_nodes, true);
    return rc ? rc : tlas_build_impl<VdTlasNodeWide>(ctx, d_inst, n, d_meshes, n_mesh, d_nodes);
}
int vd_tlas_refit_dev(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                      VdTlasNode* d_nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, d_inst, n, d_meshes, n_mesh, d_nodes, false);
    return rc ? rc : tlas_refit_impl<VdTlasNode>(ctx, d_inst, n, d_meshes, n_mesh, d_nodes);
}
int vd_tlas_refit_wide_dev(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                           VdTlasNodeWide* d_nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, d_inst, n, d_meshes, n_mesh, d_nodes, true);
    return rc ? rc : tlas_refit_impl<VdTlasNodeWide>(ctx, d_inst, n, d_meshes, n_mesh, d_nodes);
}

int vd_tlas_build(VdCtx* ctx, const VdInstance* inst, uint32_t n, const VdMeshInfo* meshes, uint32_t n_mesh, VdTlasNode* nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, inst, n, meshes, n_mesh, nodes, false);
    if (rc) return rc;
    return tlas_host<VdTlasNode>(ctx, inst, n, meshes, n_mesh, nodes, false, [&](const VdInstance* di, const VdMeshInfo* dm, VdTlasNode* dn) {
        return tlas_build_impl<VdTlasNode>(ctx, di, n, dm, n_mesh, dn);
    });
}
int vd_tlas_build_wide(VdCtx* ctx, const VdInstance* inst, uint32_t n, const VdMeshInfo* meshes, uint32_t n_mesh,
                       VdTlasNodeWide* nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, inst, n, meshes, n_mesh, nodes, true);
    if (rc) return rc;
    return tlas_host<VdTlasNodeWide>(ctx, inst, n, meshes, n_mesh, nodes, false, [&](const VdInstance* di, const VdMeshInfo* dm, VdTlasNodeWide* dn) {
        return tlas_build_impl<VdTlasNodeWide>(ctx, di, n, dm, n_mesh, dn);
    });
}
int vd_tlas_refit(VdCtx* ctx, const VdInstance* inst, uint32_t n, const VdMeshInfo* meshes, uint32_t n_mesh, VdTlasNode* nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, inst, n, meshes, n_mesh, nodes, false);
    if (rc) return rc;
    return tlas_host<VdTlasNode>(ctx, inst, n, meshes, n_mesh, nodes, true, [&](const VdInstance* di, const VdMeshInfo* dm, VdTlasNode* dn) {
        return tlas_refit_impl<VdTlasNode>(ctx, di, n, dm, n_mesh, dn);
    });
}

}  // extern "C"

// tlas.hip — TLAS leaf boxes, exact agglomerative build and refit on gfx950.
//
// Replaces Tlas::build (reference: crates/bvh/src/tlas.rs:31-105), called once per scene from
// MeshPool::generate_tlas (crates/pools/src/mesh/mod.rs:279-286), and adds refit (SURVEY.md
// §8a T3, not in the reference).
//
//  * leaf boxes (tlas.rs:34-54): one lane per instance, 8 transformed corners folded with the
//    object-space mesh box as seed (bug-compatible), total-order min/max;
//  * build (tlas.rs:56-84): the reference is a sequential chain of ~3 N `find_best_match` calls whose tie-breaking
//    depends on the slot order, so the chain itself cannot be reordered; what is made cheap is ONE call:
//      - 4096 <= n <= 65536 ("build, indexed"): a call is an exact nearest-neighbour query through a spatial index
//        (inner-corner lower bounds over Morton-ordered groups, pruned against a bound the chain supplies); one
//        512-lane workgroup - four waves answer the call, four answer the call a merge would make next - cost
//        independent of n;
//      - otherwise a scan of all active slots, kept as a compacted SoA (six float arrays + node ids) so that a scan is a
//        pure stream of 24 B per slot, by one 1024-lane workgroup below 4096 instances and by 16 of them above 65536
//        ("build, several workgroups"); the argmin is a 64-bit {area bits, slot} key reduced by DPP / wave shuffles + per-wave LDS
//        words, which reproduces "strict <, first slot wins";
//  * refit: leaves recomputed, interior boxes by a bottom-up walk with per-node handshake counters
//    (write-through agent-scope stores, no fences; boxes re-read L1/L2-bypassing).
#include "vd_common.hpp"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kBuildThreads = 1024;

struct Box { float mn[3], mx[3]; };

// tlas.rs:35-44.  glam Mat4::transform_point3: ((X*p.x + Y*p.y) + Z*p.z) + W, no FMA.
// n_mesh == 0 (internal, vd_tlas_build_from_boxes): `inst` is not an instance array but ready leaf boxes, six floats
// {min xyz, max xyz} per leaf - the same builder then clusters boxes somebody else made (trace.hip's private top level).
__device__ __forceinline__ Box leaf_box(const VdInstance* __restrict__ inst, const VdMeshInfo* __restrict__ meshes,
                                        unsigned n_mesh, unsigned i) {
    if (n_mesh == 0u) {
        const float* B = reinterpret_cast<const float*>(inst) + 6u * (size_t)i;
        Box r;
#pragma unroll
        for (int k = 0; k < 3; ++k) { r.mn[k] = B[k]; r.mx[k] = B[3 + k]; }
        return r;
    }
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

// Wave minimum of a 64-bit key without the LDS crossbar: all-reduce inside each 16-lane row (DPP), then row_bcast:15 folds
// row 0 into 1 and 2 into 3, row_bcast:31 folds rows 0-1 into 2-3, and lane 63 holds the minimum (two readlanes make it
// uniform).  The ds_bpermute steps of wave_min_u64 cost an LDS round trip each, and the chain pays for every one.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ vd_u64 dpp_u64_rows(vd_u64 v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)v, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(v >> 32), (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((vd_u64)hi << 32) | lo;
}
// Minimum of a 32-bit value over the wave, in every lane's scalar copy: six v_min_u32 with a DPP operand (the compiler
// folds the move into the min: lanes without a source keep the identity), the result read from lane 63.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned dpp_min_u32(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned ix_wave_min_u32(unsigned v) {
    v = dpp_min_u32<0x111, 0xf>(v);                                // row_shr:1
    v = dpp_min_u32<0x112, 0xf>(v);                                // row_shr:2
    v = dpp_min_u32<0x114, 0xf>(v);                                // row_shr:4
    v = dpp_min_u32<0x118, 0xf>(v);                                // row_shr:8   -> lane 15 of a row: its minimum
    v = dpp_min_u32<0x142, 0xa>(v);                                // row_bcast:15 into rows 1 and 3
    v = dpp_min_u32<0x143, 0xc>(v);                                // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// Minimum of {area bits, slot} keys: first the area, then the slot among the lanes that hold it - twelve one-instruction
// steps instead of six five-instruction 64-bit ones (this sits on the chain's critical path, twice per scan).
__device__ __forceinline__ vd_u64 ix_wave_min(vd_u64 v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned m_hi = ix_wave_min_u32(hi);
    const unsigned m_lo = ix_wave_min_u32(hi == m_hi ? lo : 0xffffffffu);
    return ((vd_u64)m_hi << 32) | m_lo;
}

// tlas.rs:87-105 over the compacted slot arrays; every thread returns the same slot.
// A scan is bound by VALU issue on the one CU that runs the chain (16 k slots x ~50 instructions / 64 lanes per
// clock ~ 5 us), not by the 24 B per slot it streams from L2, so the per-slot arithmetic is what counts.
// FAST (no NaN in any leaf box, checked once; unions of NaN-free boxes are NaN-free): the union extents use
// v_min/v_max instead of the total-order min/max.  Only the sign of a zero extent can differ, the area is then the
// same +0/-0-normalised value, so the argmin is unchanged; the boxes written to the nodes always use the total order.
// Per thread the slots come in increasing order, so `strictly smaller area` alone keeps the first slot.
template <bool FAST, int THREADS = kBuildThreads>
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
        for (unsigned i0 = tid * 4u; i0 < cnt; i0 += THREADS * 4u) {
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
    best = ix_wave_min(best);
    vd_u64* words = s_red + (call & 1u) * 16u;
    if (lane == 0) words[tid >> 6] = best;
    __syncthreads();
    vd_u64 v = lane < (unsigned)(THREADS / 64) ? words[lane] : ~0ull;
    { vd_u64 o; o = dpp_u64<0xB1>(v); v = o < v ? o : v; o = dpp_u64<0x4E>(v); v = o < v ? o : v;
      o = dpp_u64<0x141>(v); v = o < v ? o : v; o = dpp_u64<0x140>(v); v = o < v ? o : v; }        // lanes 0..15 hold the minimum
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (lo & hi) == 0xffffffffu ? target : lo;
}

// tlas.rs:56-84 — one workgroup runs the whole chain.  Resumable: the indexed build hands over in the middle of the
// chain with {cnt, used, a, b} (b already found); a fresh build starts at {n, n + 1, 0, none}.
struct ChainState { unsigned cnt, used, a, b; bool have_b; };
template <typename Node, bool FAST, int THREADS = kBuildThreads>
__device__ __forceinline__ void tlas_build_chain(Node* __restrict__ nodes, float* sb, unsigned* slot_node,
                                                 unsigned cap, vd_u64* s_red, ChainState st) {
    unsigned call = 0;
    unsigned cnt = st.cnt, used = st.used, a = st.a;
    unsigned b = st.have_b ? st.b : find_best_match<FAST, THREADS>(sb, cap, cnt, a, s_red, call++);
    while (cnt > 0) {
        const unsigned c = find_best_match<FAST, THREADS>(sb, cap, cnt, b, s_red, call++);
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
            b = find_best_match<FAST, THREADS>(sb, cap, cnt, a, s_red, call++);
        } else {
            a = b;
            b = c;
        }
    }
    if (threadIdx.x == 0) nodes[0] = nodes[slot_node[a]];   // tlas.rs:84
}

// `in_lds` (scenes of up to kChainLdsMax instances - the sizes the reference's demos have): the slot arrays move into LDS
// before the chain starts, so a scan reads 24 B per slot at LDS latency instead of from the L2 (1 000 instances: 2 500
// dependent scans of ~1.3 us, most of it the round trip of the loads).
constexpr unsigned kChainLdsMax = 5600u;                 // 28 B per slot: 157 KB of the 160 KB LDS
template <typename Node, int THREADS = kBuildThreads>
__global__ __launch_bounds__(THREADS) void tlas_build_kernel(Node* __restrict__ nodes, unsigned n,
                                                                   float* sb, unsigned* slot_node,
                                                                   unsigned cap, const unsigned* __restrict__ only_if, int in_lds) {
    if (only_if && *only_if == 0u) return;                  // see tlas_build_impl
    extern __shared__ __attribute__((aligned(16))) char chain_lds[];
    __shared__ vd_u64 s_red[32];   // [2][16] per-wave keys by scan parity
    if (threadIdx.x < 32) s_red[threadIdx.x] = ~0ull;
    int nan = 0;
    for (unsigned i = threadIdx.x; i < n; i += THREADS) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { const float v = sb[q * cap + i]; nan |= v != v; }
    }
    const bool any_nan = __syncthreads_or(nan) != 0;         // also orders the s_red initialisation
    const ChainState fresh{n, n + 1u, 0u, 0u, false};
    if (in_lds) {
        float* ls = reinterpret_cast<float*>(chain_lds);
        unsigned* ln = reinterpret_cast<unsigned*>(ls + 6u * cap);
        for (unsigned i = threadIdx.x; i < 6u * cap; i += THREADS) ls[i] = sb[i];      // (the slack past n is read 16 B at a time: copied too)
        for (unsigned i = threadIdx.x; i < n; i += THREADS) ln[i] = slot_node[i];
        __syncthreads();
        if (any_nan) tlas_build_chain<Node, false, THREADS>(nodes, ls, ln, cap, s_red, fresh);
        else tlas_build_chain<Node, true, THREADS>(nodes, ls, ln, cap, s_red, fresh);
        return;
    }
    if (any_nan) tlas_build_chain<Node, false, THREADS>(nodes, sb, slot_node, cap, s_red, fresh);
    else tlas_build_chain<Node, true, THREADS>(nodes, sb, slot_node, cap, s_red, fresh);
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
                                                                      unsigned* slot_node, unsigned cap, MwShared* sh, unsigned spin_limit,
                                                                      const unsigned* __restrict__ only_if) {
    if (only_if && *only_if == 0u) return;                  // queued behind the indexed build: only when that one declined
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

// ---- build, indexed ---------------------------------------------------------------------------
// The chain of tlas.rs:56-84 is sequential (~3 N dependent find_best_match calls), so what can shrink is the cost of
// ONE call.  A call is an exact nearest-neighbour query under the metric area(union(t, o)) with the slot index as the
// tie-break; here it is answered through an index instead of a scan of all cnt clusters:
//
//  * clusters are ENTRIES {box, slot, node} in a fixed order (sorted once by the Morton code of the 6-D point (mn, mx):
//    the reference's leaf boxes all reach back to the object-space mesh box, tlas.rs:39, so two boxes are near when BOTH
//    corners are); 16 consecutive entries form a slice, 16 slices a super-slice;
//  * every group keeps its INNER corner I = {max of the members' mn, min of their mx}.  Every member o has o.mn <= I.mn
//    and o.mx >= I.mx, so the box {min(t.mn, I.mn), max(t.mx, I.mx)} lies inside union(t, o), and the f32 evaluation of
//    Aabb::area (subtract, multiply, add: each monotone under round-to-nearest for non-negative extents) is monotone
//    under inclusion: area of that box <= area(union(t, o)) in f32 for EVERY member.  A group whose lower bound exceeds
//    the union area with any known candidate cannot hold the answer (an equal bound may: the slot index decides);
//  * a merged cluster replaces one of its two parts in that part's entry and only grows, dead entries only loosen a
//    corner: the corners stay valid without updates and are re-tightened every few hundred merges;
//  * the chain supplies the known candidate for free: for c = best(b) it is a (b = best(a) a moment ago), right after a
//    merge it is the chain element before a (never merged while remembered).  Without one, the target's own block of 64
//    entries is scanned first;
//  * slots are an attribute of an entry; idx[b] = idx[cnt-1] relabels one entry, and the stale index a that the
//    reference keeps using when a was the last slot (the merged cluster then sits in slot b and is a candidate of
//    best(a)) falls out of "exclude by slot".
// Precondition: every leaf coordinate finite, |x| < 1e18 (no overflow, hence no inf * 0 = NaN anywhere) and mn <= mx;
// otherwise the plain chain above runs.  tests/cpp/tlas_index_model.cpp restates this algorithm on the CPU and
// tests/test_tlas_index_model.py checks it against the literal oracle, ties, nesting and stale slots included.
//
// One workgroup runs the chain: four waves (one per SIMD) answer a query - super-slice bounds -> slice bounds ->
// surviving entries -> minimum, through wave-private lists and ONE barrier - and four more answer, at the same time, the
// query that follows if this one ends in a merge (ix_query).  When cnt falls to kIxPhase2 the plain scan takes over (few,
// large clusters: every group overlaps every target).
constexpr int kIxGroup = 4;              // waves per question; profiles/r02_tlas_group_sweep.txt has 1 / 2 / 4 with and without helpers
constexpr unsigned kIxSlice = 16u, kIxSuper = 16u, kIxBlock = 64u, kIxDead = 0xffffffffu;
constexpr unsigned kIxMaxInstances = 65536u;         // slice corners of 65536 entries fill 128 KB of the 160 KB LDS
constexpr int kSortThreads = 512;

struct __attribute__((aligned(16))) IxEntry { float mn[3]; unsigned slot; float mx[3]; unsigned node; };   // two 16-byte loads
struct IxCtl { unsigned ok, fallback, done, pad; };      // done: the two-workgroup form finished the build (the one-workgroup form queued behind it returns)
// Mailbox of the two-workgroup form (VD_OPT_TLAS_SPEC = 2): every 8-byte word is {payload : 32, tag : 32} and a message
// is complete when all its words carry the same tag - the data is the flag, no fence, one store instruction per message.
struct IxMail {
    vd_u64 req[16];          // main -> helper: box[6], bound, ea, eb, last, b, epoch, flags
    vd_u64 ans[16];          // helper -> main: key lo, key hi, e, node, box[6]
    unsigned main_xcc;       // XCC id of the main workgroup + 1
    unsigned claimed;        // 0 open, 1 a helper on the same XCC took the role, 2 closed by the main workgroup (nobody came)
};
constexpr unsigned kIxMailQuery = 1u, kIxMailExit = 2u;
constexpr unsigned kIxClaimPolls = 40000u;               // ~20 ms: candidates that are not co-resident by then never will be

__device__ __forceinline__ unsigned ix_spread6(unsigned x) {   // 5 bits -> every sixth bit
    return (x & 1u) | ((x & 2u) << 5) | ((x & 4u) << 10) | ((x & 8u) << 15) | ((x & 16u) << 20);
}

// One workgroup: precondition check, Morton codes, stable LSD radix sort (4 bits a pass, a thread owns a contiguous chunk:
// counts[digit][thread] in LDS, one scan of the flattened matrix per pass), then the entries and the slot -> entry map.
__global__ __launch_bounds__(kSortThreads) void tlas_index_kernel(const float* __restrict__ sb, unsigned cap, unsigned n, unsigned E,
                                                                  unsigned* key0, unsigned* val0, unsigned* key1, unsigned* val1,
                                                                  IxEntry* __restrict__ entries, unsigned* __restrict__ slot_ent, IxCtl* ctl) {
    constexpr unsigned T = kSortThreads, W = kSortThreads / 64;
    __shared__ unsigned s_cnt[16 * kSortThreads];
    __shared__ float s_lo[W][3], s_hi[W][3];
    __shared__ unsigned s_wsum[W];
    const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    int bad = 0;
    for (unsigned i = t; i < n; i += T) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float mn = sb[k * cap + i], mx = sb[(3 + k) * cap + i];
            bad |= !(fabsf(mn) < 1e18f) || !(fabsf(mx) < 1e18f) || !(mn <= mx);
            lo[k] = __builtin_fminf(lo[k], mn); hi[k] = __builtin_fmaxf(hi[k], mx);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = __builtin_fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = __builtin_fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        if (lane == 0) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
    }
    if (__syncthreads_or(bad)) {                          // the fast arithmetic cannot order this input: plain chain
        if (t == 0) { ctl->ok = 0u; ctl->fallback = 1u; ctl->done = 0u; }
        return;
    }
    if (t == 0) { ctl->ok = 1u; ctl->fallback = 0u; ctl->done = 0u; }
    float scale[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float l = s_lo[0][k], h = s_hi[0][k];
        for (unsigned w = 1; w < W; ++w) { l = __builtin_fminf(l, s_lo[w][k]); h = __builtin_fmaxf(h, s_hi[w][k]); }
        lo[k] = l;
        scale[k] = h > l ? 32.0f / (h - l) : 0.0f;
    }
    for (unsigned i = t; i < n; i += T) {
        unsigned c = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float q0 = __builtin_fminf(__builtin_fmaxf((sb[k * cap + i] - lo[k]) * scale[k], 0.0f), 31.0f);
            const float q1 = __builtin_fminf(__builtin_fmaxf((sb[(3 + k) * cap + i] - lo[k]) * scale[k], 0.0f), 31.0f);
            c |= ix_spread6((unsigned)q0) << k;
            c |= ix_spread6((unsigned)q1) << (3 + k);
        }
        key0[i] = c; val0[i] = i;
    }
    __syncthreads();
    const unsigned chunk = (n + T - 1) / T;
    const unsigned begin = min(n, t * chunk), end = min(n, begin + chunk);
    unsigned *kin = key0, *vin = val0, *kout = key1, *vout = val1;
    for (unsigned shift = 0; shift < 32u; shift += 4u) {          // 8 passes: the result is back in key0 / val0
#pragma unroll
        for (unsigned d = 0; d < 16u; ++d) s_cnt[d * T + t] = 0u;
        for (unsigned i = begin; i < end; ++i) s_cnt[((kin[i] >> shift) & 15u) * T + t] += 1u;
        __syncthreads();
        unsigned v[16], run = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { const unsigned c = s_cnt[16u * t + j]; v[j] = run; run += c; }
        unsigned incl = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(incl, off); if (lane >= (unsigned)off) incl += o; }
        if (lane == 63u) s_wsum[wave] = incl;
        __syncthreads();
        unsigned base = incl - run;
        for (unsigned w = 0; w < wave; ++w) base += s_wsum[w];
#pragma unroll
        for (int j = 0; j < 16; ++j) s_cnt[16u * t + j] = base + v[j];
        __syncthreads();
        for (unsigned i = begin; i < end; ++i) {
            const unsigned k = kin[i];
            const unsigned pos = s_cnt[((k >> shift) & 15u) * T + t]++;
            kout[pos] = k; vout[pos] = vin[i];
        }
        __syncthreads();
        unsigned* x = kin; kin = kout; kout = x;
        x = vin; vin = vout; vout = x;
    }
    for (unsigned e = t; e < E; e += T) {
        IxEntry en;
        if (e < n) {
            const unsigned i = vin[e];
#pragma unroll
            for (int k = 0; k < 3; ++k) { en.mn[k] = sb[k * cap + i]; en.mx[k] = sb[(3 + k) * cap + i]; }
            en.slot = i; en.node = i + 1u;
            slot_ent[i] = e;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { en.mn[k] = 1e30f; en.mx[k] = -1e30f; }
            en.slot = kIxDead; en.node = 0u;
        }
        entries[e] = en;
    }
}

struct __attribute__((aligned(16))) IxRec { vd_u64 key; unsigned e, node; float box[8]; };   // three 16-byte LDS accesses
struct IxShared {
    IxRec res[2][8];                                     // by query parity: a wave may start the next query before the others
                                                         // have read this one's result; [4..7]: the helper waves' records
    unsigned work;                                       // slices looked into since the last check (the build declines when pruning fails)
    unsigned role;                                       // two-workgroup form: 1 = this workgroup is the helper
    unsigned msg[2][16];                                 // two-workgroup form: the message wave 0 received (main: the answer; helper: the request), by parity
    vd_u64 red[32];                                      // s_red of the plain chain (phase 2)
};
struct IxHit { vd_u64 key; unsigned e, node, slices; float box[6]; };   // slices: of the CALLING wave (the only field that differs between waves)
// list1 / list2: per-wave survivor lists (every wave owns a quarter of each array)
struct IxLds { IxShared* sh; float4* slice; float4* super; unsigned short* list1; unsigned short* list2; unsigned n_slices, n_super, cap1, cap2; };
struct IxProf { unsigned long long t_bounds, t_entries, t_reduce, t_merge, t_refresh, queries, own, n1, n2; };

// v_max_f32 / v_min_f32 as single instructions: under the precondition no operand is a NaN, so the canonicalising
// v_max x, x the compiler puts in front of every IEEE maxnum / minnum operand (three instructions per min or max) buys
// nothing here, and the chain is bound by the instructions one wave issues per query.
__device__ __forceinline__ float ix_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ix_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ix_union_area(const float (&t)[6], const float (&o)[6]) {
    const float dx = ix_max(t[3], o[3]) - ix_min(t[0], o[0]);
    const float dy = ix_max(t[4], o[4]) - ix_min(t[1], o[1]);
    const float dz = ix_max(t[5], o[5]) - ix_min(t[2], o[2]);
    return vd_area(dx, dy, dz) + 0.0f;
}
__device__ __forceinline__ float ix_lower_bound(const float (&t)[6], const float4 lo, const float4 hi) {
    const float dx = ix_max(t[3], hi.x) - ix_min(t[0], lo.x);
    const float dy = ix_max(t[4], hi.y) - ix_min(t[1], lo.y);
    const float dz = ix_max(t[5], hi.z) - ix_min(t[2], lo.z);
    return vd_area(dx, dy, dz) + 0.0f;
}

// per-lane running minimum over the entries it evaluates
struct IxLane { vd_u64 key; unsigned e; float4 lo, hi; };
__device__ __forceinline__ void ix_eval(IxLane& L, const IxEntry* entries, unsigned e, unsigned t_slot, const float (&tb)[6]) {
    const float4 lo = reinterpret_cast<const float4*>(entries + e)[0], hi = reinterpret_cast<const float4*>(entries + e)[1];
    const unsigned slot = __float_as_uint(lo.w);
    const float o[6] = {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z};
    const float area = ix_union_area(tb, o);
    const bool ok = slot != kIxDead && slot != t_slot && area < 1e30f;   // tlas.rs:88,98: from 1e30, never the target
    const vd_u64 key = ok ? (((vd_u64)__float_as_uint(area) << 32) | slot) : ~0ull;
    if (key < L.key) { L.key = key; L.e = e; L.lo = lo; L.hi = hi; }
}

// survivors among the super-slices into a wave-private list; R rounds of 64 evaluated side by side
template <int R>
__device__ __forceinline__ unsigned ix_supers(const IxLds& L, const float (&tb)[6], float bound, unsigned short* my1, unsigned lane) {
    unsigned n1 = 0;
    for (unsigned base = 0; base < L.n_super; base += 64u * R) {
        float lb[R];
        bool in[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned s = base + 64u * r + lane;
            in[r] = s < L.n_super;
            const unsigned sc = in[r] ? s : 0u;
            lb[r] = ix_lower_bound(tb, L.super[2u * sc], L.super[2u * sc + 1u]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool keep = in[r] && !(lb[r] > bound);      // no bound = NaN: nothing is greater, everything is kept
            const unsigned long long mask = __ballot(keep);
            if (keep) my1[n1 + vd_mbcnt(mask)] = (unsigned short)(base + 64u * r + lane);
            n1 += (unsigned)__popcll(mask);
        }
    }
    return n1;
}

// find_best_match through the index.  Every thread of the workgroup calls it with the same arguments and gets the same
// answer.  bound: union area with a known candidate, NaN = none.  q: running query number (parity of the result slots).
// ONE workgroup barrier per query: every wave works out the surviving super-slices by itself (their corners sit in LDS:
// the same few reads in all four waves), then takes a quarter of their slices, then the entries of the slices ITS
// quarter left over - through wave-private lists, no cross-wave hand-off until the four minima meet.
// Speculation (round 2): when the chain asks c = best(b) it already knows what it will ask next IF c turns out to be a:
// best(a ∪ b), bounded by the previous chain element.  Waves 4-7 answer THAT question meanwhile, on the state the merge
// will leave behind: the entries of a and b are left out (a ∪ b is the target, b dies), and the entry that holds the last
// slot counts as slot b (tlas.rs:75 moves it there) - the only things a merge changes.  The corners of the groups are the
// same for both (a merged box contains every corner its parts contained), so the answer is the one the query after the
// merge would give, and the chain takes it without asking.  About a third of all queries follow a merge.
struct IxSpec { float box[6]; float bound; unsigned ea, eb, last, b; bool valid; };
struct IxPair { IxHit main, spec; };
template <int G, bool SPEC, int ROLE = 0>
__device__ __forceinline__ IxPair ix_query(const IxEntry* entries, const IxLds& L, unsigned q, unsigned t_slot, const float (&tb)[6],
                                           unsigned e_t, float bound, IxProf* prof, const IxSpec& spec) {
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ROLE 1: the workgroup is the remote helper (tlas_build_indexed_kernel<..., REMOTE>): its G waves answer `spec`
    const bool helper = ROLE == 1 || (SPEC && wave >= (unsigned)G);
    const unsigned w4 = ROLE == 1 ? wave : (helper ? wave - (unsigned)G : wave), ri = ROLE == 1 ? wave : (helper ? 4u + w4 : w4);
    IxShared* sh = L.sh;
    const unsigned p = q & 1u;
    long long t0 = 0;
    if (prof) t0 = clock64();
    // this wave's question (uniform per wave)
    float box[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) box[k] = helper ? spec.box[k] : tb[k];
    float bnd = helper ? spec.bound : bound;
    const unsigned not_slot = helper ? 0xfffffffeu : t_slot;          // the target, by slot ...
    const unsigned not_e1 = helper ? spec.ea : 0xffffffffu, not_e2 = helper ? spec.eb : 0xffffffffu;   // ... or by entry
    const unsigned from_slot = helper ? spec.last : 0xfffffffeu, to_slot = spec.b;
    const bool active = !helper || spec.valid;
    IxLane mine{~0ull, 0u, float4{0, 0, 0, 0}, float4{0, 0, 0, 0}};
    unsigned short* my1 = L.list1 + wave * L.cap1;
    unsigned short* my2 = L.list2 + wave * L.cap2;
    unsigned n2 = 0;
    long long t1 = 0;
    // One wave per SIMD and question: nothing hides latency but the wave's own independent instructions, so every stage
    // first issues all its loads and evaluates several items per lane side by side, and only then does the bookkeeping.
    if (active) {
        if (!(bnd == bnd)) {                                  // no candidate known: the target's own block supplies one (main waves only)
            ix_eval(mine, entries, (e_t / kIxBlock) * kIxBlock + lane, t_slot, tb);
            const vd_u64 m = ix_wave_min(mine.key);
            bnd = __uint_as_float((unsigned)(m >> 32));       // 0xffffffff reads back as NaN: still none
            if (prof && tid == 0u) prof->own += 1;
        }
        const unsigned n1 = L.n_super <= 64u ? ix_supers<1>(L, box, bnd, my1, lane)
                          : (L.n_super <= 128u ? ix_supers<2>(L, box, bnd, my1, lane) : ix_supers<4>(L, box, bnd, my1, lane));
        vd_wave_lds_sync();
        // slices of the surviving super-slices; item i is dealt to wave i % 4 of the group (neighbouring slices survive together)
        const unsigned items1 = n1 * kIxSuper;
        auto slices = [&](auto rounds, unsigned base) {
            constexpr int R = decltype(rounds)::value;
            float lb[R];
            bool in[R];
            unsigned sl[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned item = base + (64u * r + lane) * (unsigned)G + w4;
                in[r] = item < items1;
                sl[r] = (unsigned)my1[(in[r] ? item : 0u) / kIxSuper] * kIxSuper + (item % kIxSuper);   // read unconditionally: no branch, the rounds' reads go out together
                in[r] = in[r] && sl[r] < L.n_slices;
                const unsigned sc = in[r] ? sl[r] : 0u;
                lb[r] = ix_lower_bound(box, L.slice[2u * sc], L.slice[2u * sc + 1u]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool keep = in[r] && !(lb[r] > bnd);
                const unsigned long long mask = __ballot(keep);
                if (keep) my2[n2 + vd_mbcnt(mask)] = (unsigned short)sl[r];
                n2 += (unsigned)__popcll(mask);
            }
        };
        // up to 16 surviving super-slices (the usual case) are one round of 64 lanes per wave; more go two rounds at a time
        if (items1 <= 64u * G) { if (items1) slices(std::integral_constant<int, 1>{}, 0u); }
        else if (G < 4 && items1 <= 256u * G) slices(std::integral_constant<int, 4>{}, 0u);
        else for (unsigned base = 0; base < items1; base += 128u * G) slices(std::integral_constant<int, 2>{}, base);
        vd_wave_lds_sync();
        if (prof) t1 = clock64();
        const unsigned items2 = n2 * kIxSlice;
        auto ents = [&](auto rounds, unsigned base) {
            constexpr int R = decltype(rounds)::value;
            float4 lo[R], hi[R];
            unsigned e[R];
            bool in[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned item = base + 64u * r + lane;
                in[r] = item < items2;
                e[r] = (unsigned)my2[(in[r] ? item : 0u) / kIxSlice] * kIxSlice + (item % kIxSlice);
                e[r] = in[r] ? e[r] : 0u;
                lo[r] = reinterpret_cast<const float4*>(entries + e[r])[0];
                hi[r] = reinterpret_cast<const float4*>(entries + e[r])[1];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                unsigned slot = __float_as_uint(lo[r].w);
                const float o[6] = {lo[r].x, lo[r].y, lo[r].z, hi[r].x, hi[r].y, hi[r].z};
                const float area = ix_union_area(box, o);
                const bool ok = in[r] && slot != kIxDead && slot != not_slot && e[r] != not_e1 && e[r] != not_e2 && area < 1e30f;
                slot = slot == from_slot ? to_slot : slot;
                const vd_u64 key = ok ? (((vd_u64)__float_as_uint(area) << 32) | slot) : ~0ull;
                if (key < mine.key) { mine.key = key; mine.e = e[r]; mine.lo = lo[r]; mine.hi = hi[r]; }
            }
        };
        constexpr int kDeep = G == 1 ? 6 : (G == 2 ? 4 : 2);      // loads in flight per lane: fewer waves, deeper rounds
        // one pass whenever the wave's share fits four loads per lane: a second pass is a second L2 round trip, and the
        // helper waves (looser bound: 8.4 slices per wave on average against 7.2) needed one in most of their queries
        if (items2 <= 128u) { if (items2) ents(std::integral_constant<int, 2>{}, 0u); }
        else if (items2 <= 256u) ents(std::integral_constant<int, 4>{}, 0u);
        else for (unsigned base = 0; base < items2; base += 64u * kDeep) ents(std::integral_constant<int, kDeep>{}, base);
    }
    long long t2 = 0;
    if (prof) { t2 = clock64(); }
    const vd_u64 m = ix_wave_min(mine.key);
    float4* rec = reinterpret_cast<float4*>(&sh->res[p][ri]);
    if (m == ~0ull) {
        if (lane == 0u) sh->res[p][ri].key = ~0ull;
    } else if (mine.key == m) {                           // slots are unique: exactly one lane
        rec[0] = float4{__uint_as_float((unsigned)m), __uint_as_float((unsigned)(m >> 32)), __uint_as_float(mine.e), mine.hi.w};
        rec[1] = float4{mine.lo.x, mine.lo.y, mine.lo.z, mine.hi.x};
        rec[2] = float4{mine.hi.y, mine.hi.z, 0.0f, 0.0f};
    }
    __syncthreads();
    // lane w < 8 fetches wave w's whole record; quad 0 holds the answer, quad 1 the speculative one
    const float4* rrec = reinterpret_cast<const float4*>(&sh->res[p][lane & 7u]);
    const float4 r0 = rrec[0], r1 = rrec[1], r2 = rrec[2];
    vd_u64 rkey = ((vd_u64)__float_as_uint(r0.y) << 32) | __float_as_uint(r0.x);
    const unsigned re = __float_as_uint(r0.z), rnode = __float_as_uint(r0.w);
    const float rbox[6] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y};
    vd_u64 best = rkey, o;
    o = dpp_u64<0xB1>(best); best = o < best ? o : best;
    o = dpp_u64<0x4E>(best); best = o < best ? o : best;          // minimum over each quad = over the four waves of a group
    const unsigned long long who = __ballot(rkey == best);
    auto pick = [&](int first, IxHit& h) {
        const int w = first + (int)__builtin_ctz((unsigned)(who >> first) & 0xfu);   // first wave that holds it (ties carry identical records)
        h.key = ((vd_u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(best >> 32), first) << 32) |
                (unsigned)__builtin_amdgcn_readlane((int)(unsigned)best, first);
        h.e = (unsigned)__builtin_amdgcn_readlane((int)re, w); h.node = (unsigned)__builtin_amdgcn_readlane((int)rnode, w);
#pragma unroll
        for (int k = 0; k < 6; ++k) h.box[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rbox[k]), w));
    };
    IxPair r;
    pick(0, r.main);
    pick(4, r.spec);
    r.main.slices = helper ? 0u : n2;                         // of the calling wave: what the decline check adds up
    r.spec.slices = 0u;
    if (prof && (tid == 0u || (tid == 64u * G && active))) {   // thread 0: the answering group; first helper lane: the speculation
        const long long t3 = clock64();
        prof->t_bounds += (unsigned long long)(t1 - t0); prof->t_entries += (unsigned long long)(t2 - t1); prof->t_reduce += (unsigned long long)(t3 - t2);
        prof->queries += 1; prof->n2 += n2;
    }
    return r;
}

// inner corners of all groups from the live entries (a slice per thread, then a super-slice per thread)
__device__ __forceinline__ void ix_refresh(const IxEntry* entries, const IxLds& L) {
    for (unsigned sl = threadIdx.x; sl < L.n_slices; sl += blockDim.x) {
        float4 lo{-1e30f, -1e30f, -1e30f, 0.0f}, hi{1e30f, 1e30f, 1e30f, 0.0f};      // empty group: every bound overflows to +inf
        for (unsigned j = 0; j < kIxSlice; ++j) {
            const float4 a = reinterpret_cast<const float4*>(entries + sl * kIxSlice + j)[0];
            const float4 b = reinterpret_cast<const float4*>(entries + sl * kIxSlice + j)[1];
            if (__float_as_uint(a.w) != kIxDead) {
                lo.x = __builtin_fmaxf(lo.x, a.x); lo.y = __builtin_fmaxf(lo.y, a.y); lo.z = __builtin_fmaxf(lo.z, a.z);
                hi.x = __builtin_fminf(hi.x, b.x); hi.y = __builtin_fminf(hi.y, b.y); hi.z = __builtin_fminf(hi.z, b.z);
            }
        }
        L.slice[2u * sl] = lo; L.slice[2u * sl + 1u] = hi;
    }
    __syncthreads();
    for (unsigned sp = threadIdx.x; sp < L.n_super; sp += blockDim.x) {
        float4 lo{-1e30f, -1e30f, -1e30f, 0.0f}, hi{1e30f, 1e30f, 1e30f, 0.0f};
        for (unsigned j = 0; j < kIxSuper && sp * kIxSuper + j < L.n_slices; ++j) {
            const float4 a = L.slice[2u * (sp * kIxSuper + j)], b = L.slice[2u * (sp * kIxSuper + j) + 1u];
            lo.x = __builtin_fmaxf(lo.x, a.x); lo.y = __builtin_fmaxf(lo.y, a.y); lo.z = __builtin_fmaxf(lo.z, a.z);
            hi.x = __builtin_fminf(hi.x, b.x); hi.y = __builtin_fminf(hi.y, b.y); hi.z = __builtin_fminf(hi.z, b.z);
        }
        L.super[2u * sp] = lo; L.super[2u * sp + 1u] = hi;
    }
    __syncthreads();
}

// wave-private survivor lists: every wave may keep all super-slices, and of the slices its quarter (+ one round of slack)
static __host__ __device__ unsigned ix_cap1(unsigned n_super) { return (n_super + 7u) & ~7u; }
static __host__ __device__ unsigned ix_cap2(unsigned n_slices, unsigned g) { return ((n_slices + g - 1u) / g + 64u + 7u) & ~7u; }
static size_t ix_lds_bytes(unsigned n_slices, unsigned n_super, unsigned g, unsigned waves) {
    return ((sizeof(IxShared) + 15) & ~(size_t)15) + (size_t)n_slices * 32 + (size_t)n_super * 32 + (size_t)ix_cap1(n_super) * 2 * waves +
           (size_t)ix_cap2(n_slices, g) * 2 * waves + 16;            // lists: 2 bytes x waves
}

// ---- the two-workgroup form (VD_OPT_TLAS_SPEC = 2) ----
// The helper waves of the one-workgroup form share the SIMDs of the waves that answer the chain's own question (a query is
// half VALU issue: 2884 -> 3350 cycles, plus ~400 of waiting for the helpers).  Here the speculative question goes to a
// SECOND workgroup on another CU of the same XCC: the main workgroup posts {a U b, bound, ea, eb, last, b} before it
// starts its own query, the helper answers through its own copy of the corner tables (same entries in global memory), and
// the answer is picked up while thread 0 applies the merge.  Same XCC (checked with HW_REG_XCC_ID) = same L2, so the
// entries need no agent-scope traffic: the main workgroup's stores are complete (vmcnt(0) + barrier) before the message
// that follows them is posted, and the helper drops its L1 when it takes a message.  A merge that runs WHILE the helper
// still reads is harmless: the answer is defined on the state the merge leaves, and entry ea / eb are excluded by entry
// index, the relabelled last slot by value - the same result from either side of the merge; the corners only loosen.
__device__ __forceinline__ unsigned ix_xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)); }   // HW_REG_XCC_ID[3:0]
template <int N> __device__ __forceinline__ unsigned ix_lane_pick(const unsigned (&v)[N], unsigned lane) {
    unsigned r = 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) r = lane == (unsigned)i ? v[i] : r;
    return r;
}
// The mailbox lives in the XCC's L2 (both workgroups sit behind it): plain stores (the L1 writes through) and reads that
// are executed BY the L2 - a returning atomic OR with 0 (a load with sc0 may still hit the L1 when a workgroup is not
// split over CUs: tried, the poll never saw the message).  An agent-scope store goes through to memory and, stores
// being counted in vmcnt on gfx950, the first load wait of the query that follows pays for it; an agent-scope load
// fetches from memory (~250 ns per hop: 2900 cycles of waiting per answer, profiles/r04_tlas_two_workgroups.log).
__device__ __forceinline__ void ix_post(vd_u64* words, unsigned lane, unsigned n_words, unsigned payload, unsigned tag) {
    if (lane < n_words) __hip_atomic_store(words + lane, ((vd_u64)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ vd_u64 ix_load_l2(const vd_u64* p) {
    vd_u64 r;
    const vd_u64 zero = 0ull;
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p), "v"(zero) : "memory");
    return r;
}
// one wave: wait for a complete message with tag `want` (or, want == 0, any tag newer than `after`); false on a timeout
__device__ __forceinline__ bool ix_poll(const vd_u64* words, unsigned lane, unsigned n_words, unsigned want, unsigned after, unsigned limit,
                                        unsigned& payload, unsigned& tag_out) {
    unsigned spins = 0;
    for (;;) {
        vd_u64 w = 0ull;
        if (lane < n_words) w = ix_load_l2(words + lane);            // one L2 atomic per word of the message, no more
        const unsigned tag = (unsigned)(w >> 32);
        const unsigned t0 = (unsigned)__builtin_amdgcn_readfirstlane((int)tag);
        const bool same = __all(lane >= n_words || tag == t0);
        if (same && (want ? t0 == want : (t0 != 0u && (int)(t0 - after) > 0))) { payload = (unsigned)w; tag_out = t0; return true; }
        if (++spins > limit) return false;
    }
}

// The helper workgroup: take the newest request, answer it, until told to leave (or nothing arrives for ~a second).
template <int G>
__device__ __forceinline__ void ix_helper_loop(const IxEntry* entries, const IxLds& L, IxMail* mail) {
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    IxShared* sh = L.sh;
    ix_refresh(entries, L);
    unsigned last = 0u, epoch = 0u, q = 0u;
    for (unsigned it = 0;; ++it) {
        unsigned* m = sh->msg[it & 1u];
        if (wave == 0u) {
            unsigned payload = 0u, tg = 0u;
            const bool ok = ix_poll(mail->req, lane, 13u, 0u, last, kSpinLimit, payload, tg);
            if (lane < 13u) m[lane] = payload;
            if (lane == 0u) { m[14] = tg; m[15] = ok ? 1u : 0u; }
            asm volatile("buffer_inv sc1" ::: "memory");           // what was stored before the message: not from this CU's L1
        }
        __syncthreads();
        if (m[15] == 0u || (m[12] & kIxMailExit)) return;
        last = m[14];
        if (m[11] != epoch) { ix_refresh(entries, L); epoch = m[11]; }
        if (m[12] & kIxMailQuery) {
            IxSpec spec;
#pragma unroll
            for (int k = 0; k < 6; ++k) spec.box[k] = __uint_as_float(m[k]);
            spec.bound = __uint_as_float(m[6]); spec.ea = m[7]; spec.eb = m[8]; spec.last = m[9]; spec.b = m[10]; spec.valid = true;
            const float none[6] = {0, 0, 0, 0, 0, 0};
            const IxHit h = ix_query<G, false, 1>(entries, L, q++, 0u, none, 0u, 0.0f, nullptr, spec).main;
            if (wave == 0u) {
                const unsigned v[10] = {(unsigned)h.key, (unsigned)(h.key >> 32), h.e, h.node, __float_as_uint(h.box[0]), __float_as_uint(h.box[1]),
                                        __float_as_uint(h.box[2]), __float_as_uint(h.box[3]), __float_as_uint(h.box[4]), __float_as_uint(h.box[5])};
                ix_post(mail->ans, lane, 10u, ix_lane_pick(v, lane), last);
            }
        }
    }
}

template <typename Node, int G, bool SPEC, bool REMOTE = false>
__global__ __launch_bounds__(64 * G * (SPEC ? 2 : 1)) void tlas_build_indexed_kernel(Node* __restrict__ nodes, unsigned n, IxEntry* entries, unsigned* slot_ent,
                                                                        unsigned E, float* sb, unsigned* slot_node, unsigned cap,
                                                                        IxCtl* ctl, unsigned phase2_cnt, unsigned refresh_every, int profile,
                                                                        IxMail* mail, unsigned spin_limit, int chain_in_lds) {
    constexpr int kThreads = 64 * G * (SPEC ? 2 : 1), kWaves = kThreads / 64;
    constexpr bool spec_on = SPEC;
    static_assert(!(SPEC && REMOTE), "the speculation runs either on this workgroup's helper waves or on the other workgroup");
    if (ctl->ok == 0u) return;                            // precondition failed: the plain chain is queued behind
    if (!REMOTE && ctl->done != 0u) return;               // the two-workgroup form in front of this launch built the tree
    extern __shared__ __attribute__((aligned(16))) char smem[];
    IxLds L;
    L.n_slices = E / kIxSlice;
    L.n_super = (L.n_slices + kIxSuper - 1u) / kIxSuper;
    char* p = smem;
    L.sh = reinterpret_cast<IxShared*>(p); p += (sizeof(IxShared) + 15) & ~(size_t)15;
    L.slice = reinterpret_cast<float4*>(p); p += (size_t)L.n_slices * 32;
    L.super = reinterpret_cast<float4*>(p); p += (size_t)L.n_super * 32;
    L.cap1 = ix_cap1(L.n_super); L.cap2 = ix_cap2(L.n_slices, G);
    L.list1 = reinterpret_cast<unsigned short*>(p); p += (size_t)L.cap1 * 2 * kWaves;
    L.list2 = reinterpret_cast<unsigned short*>(p);
    const unsigned tid = threadIdx.x;
    const unsigned wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 32u) L.sh->red[tid] = ~0ull;
    if (tid < 16u) L.sh->res[tid >> 3][tid & 7u].key = ~0ull;   // records of waves that do not exist stay empty
    if constexpr (REMOTE) {
        // roles: workgroup 0 runs the chain; of the others, the first one that sits on the SAME XCC (= behind the same L2)
        // becomes the helper, the rest leave.  Nobody in time: this launch leaves everything untouched and the
        // one-workgroup form queued behind it runs.
        if (blockIdx.x != 0u) {
            if (tid == 0u) {
                const unsigned me = ix_xcc_id() + 1u;
                unsigned spins = 0u, m;
                while ((m = __hip_atomic_load(&mail->main_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && ++spins <= kIxClaimPolls) {}
                L.sh->role = (m == me && atomicCAS(&mail->claimed, 0u, 1u) == 0u) ? 1u : 0u;
            }
            __syncthreads();
            if (L.sh->role == 1u) ix_helper_loop<G>(entries, L, mail);
            return;
        }
        if (tid == 0u) {
            __hip_atomic_store(&mail->main_xcc, ix_xcc_id() + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0u, c = 0u;
            if (spin_limit != 0u)
                while ((c = __hip_atomic_load(&mail->claimed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && ++spins <= kIxClaimPolls) {}
            if (c == 0u) c = atomicCAS(&mail->claimed, 0u, 2u) == 0u ? 2u : 1u;      // closed - unless a helper got in first
            L.sh->role = c;
        }
        __syncthreads();
        if (L.sh->role != 1u) return;
    }
    IxProf prof_data{0, 0, 0, 0, 0, 0, 0, 0, 0};
    IxProf* prof = profile ? &prof_data : nullptr;
    unsigned long long t_wait = 0ull;                     // REMOTE: cycles wave 0 waited for answers; answers used / timed out
    unsigned n_remote = 0u;
    ix_refresh(entries, L);                               // ends with a barrier

    // chain state: the same values in every thread
    unsigned cnt = n, used = n + 1u, q = 0, since_refresh = 0;
    unsigned a = 0u, ea = slot_ent[0], node_a = 1u, b, eb, node_b;
    float box_a[6], box_b[6], box_prev[6] = {0, 0, 0, 0, 0, 0};
    bool a_stale = false, have_prev = false;
    unsigned e_prev = 0u;
    unsigned el_last = slot_ent[n - 1u];                  // entry of the last slot: used by thread 0 at the next merge
    {
        const float4 lo = reinterpret_cast<const float4*>(entries + ea)[0], hi = reinterpret_cast<const float4*>(entries + ea)[1];
        box_a[0] = lo.x; box_a[1] = lo.y; box_a[2] = lo.z; box_a[3] = hi.x; box_a[4] = hi.y; box_a[5] = hi.z;
    }
    const float kNone = __uint_as_float(0x7fc00000u);
    // Pruning needs slack between the areas: when the targets' bounds keep most slices alive (identical or deeply nested
    // boxes - every union area ties) a query through the index costs more than a plain scan.  The build then declines:
    // it sets the fallback flag and returns, and the plain chain queued behind it starts over from the leaves (it rewrites
    // every node this kernel wrote; the slot arrays were not touched).
    unsigned work = 0;                                    // per wave; the four are added up at a check
    unsigned next_check = 256u, last_check = 0u;
    unsigned seq = 0u, epoch = 0u;                        // REMOTE: requests posted; refreshes done (the helper follows)
    bool remote_ok = REMOTE;                              // false after a timeout: the chain asks its own questions from then on
    auto post_exit = [&]() {
        if constexpr (REMOTE) {
            if (wave_id == 0u) ix_post(mail->req, tid, 13u, tid == 12u ? kIxMailExit : 0u, seq + 1u);
        }
    };
    if (tid == 0u) L.sh->work = 0u;
    auto take = [&](const IxHit& h, unsigned t_slot, unsigned e_t, unsigned node_t, const float (&tb)[6], unsigned& o_slot, unsigned& o_e,
                    unsigned& o_node, float (&o_box)[6]) {
        work += h.slices;
        if (h.key == ~0ull) {                             // nothing: find_best_match returns the target (tlas.rs:89)
            o_slot = t_slot; o_e = e_t; o_node = node_t;
#pragma unroll
            for (int k = 0; k < 6; ++k) o_box[k] = tb[k];
        } else {
            o_slot = (unsigned)h.key; o_e = h.e; o_node = h.node;
#pragma unroll
            for (int k = 0; k < 6; ++k) o_box[k] = h.box[k];
        }
    };
    const IxSpec no_spec{{0, 0, 0, 0, 0, 0}, 0.0f, 0u, 0u, 0u, 0u, false};
    take(ix_query<G, SPEC>(entries, L, q++, a, box_a, ea, kNone, prof, no_spec).main, a, ea, node_a, box_a, b, eb, node_b, box_b);
    while (cnt > phase2_cnt) {
        if (q >= next_check) {                              // every ~256 queries: more than an eighth of all slices per query?
            if ((tid & 63u) == 0u) atomicAdd(&L.sh->work, work);
            __syncthreads();
            const unsigned total = L.sh->work;
            __syncthreads();
            if ((unsigned long long)total > (unsigned long long)(L.n_slices / 8u + 1u) * (q - last_check)) {
                if (tid == 0u) { ctl->fallback = 1u; ctl->ok = 0u; }
                post_exit();
                return;
            }
            if (tid == 0u) L.sh->work = 0u;                 // the next add is 256 queries (and as many barriers) away
            work = 0; last_check = q; next_check = q + 256u;
        }
        float bound = kNone;
        if (!a_stale) bound = ix_union_area(box_b, box_a);                        // a is a live candidate of best(b)
        else if (have_prev && e_prev != eb) bound = ix_union_area(box_b, box_prev);
        unsigned c, ec, node_c;
        float box_c[6];
        // if c comes back as a, the next question is best(a ∪ b) bounded by the previous chain element: prepare it meanwhile
        IxSpec spec;
        spec.valid = (spec_on || remote_ok) && !a_stale && cnt - 1u != a && have_prev && e_prev != eb && e_prev != ea;
        spec.ea = ea; spec.eb = eb; spec.last = cnt - 1u; spec.b = b;
#pragma unroll
        for (int k = 0; k < 3; ++k) { spec.box[k] = ix_min(box_a[k], box_b[k]); spec.box[3 + k] = ix_max(box_a[3 + k], box_b[3 + k]); }
        spec.bound = ix_union_area(spec.box, box_prev);
        if constexpr (REMOTE) {
            if (spec.valid) {                             // out before this workgroup's own query starts
                seq += 1u;
                if (wave_id == 0u) {
                    const unsigned v[13] = {__float_as_uint(spec.box[0]), __float_as_uint(spec.box[1]), __float_as_uint(spec.box[2]),
                                            __float_as_uint(spec.box[3]), __float_as_uint(spec.box[4]), __float_as_uint(spec.box[5]),
                                            __float_as_uint(spec.bound), spec.ea, spec.eb, spec.last, spec.b, epoch, kIxMailQuery};
                    ix_post(mail->req, tid, 13u, ix_lane_pick(v, tid), seq);
                }
            }
        }
        const IxPair hit = ix_query<G, SPEC>(entries, L, q++, b, box_b, eb, bound, prof, spec);
        take(hit.main, b, eb, node_b, box_b, c, ec, node_c, box_c);
        if (a == c) {
            // v_min_f32 / v_max_f32 order -0 below +0 (the ISA's LT_NEG_ZERO compare), so without NaNs (the precondition)
            // the single instruction IS the total-order minimum the nodes are defined with: spec.box is the merged box
            float u[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) u[k] = spec.box[k];
            const unsigned last = cnt - 1u;
            long long tm = 0;
            if (prof) tm = clock64();
            if (tid == 0u) {
                Node nd;
#pragma unroll
                for (int k = 0; k < 3; ++k) { nd.min[k] = u[k]; nd.max[k] = u[3 + k]; }
                node_set_children(nd, node_a, node_b);
                nd.instance_idx = 0xffffffffu;
                nodes[used] = nd;
                // node_indices[a] = nodes_used; node_indices[b] = node_indices[cnt - 1]  (tlas.rs:72-75, in this order)
                float4* ma = reinterpret_cast<float4*>(entries + ea);
                ma[0] = float4{u[0], u[1], u[2], __uint_as_float(last == a ? b : a)};
                ma[1] = float4{u[3], u[4], u[5], __uint_as_float(used)};
                entries[eb].slot = kIxDead;
                if (last == a) slot_ent[b] = ea;                                  // the merged cluster lands in slot b, a goes stale
                else if (last != b) { entries[el_last].slot = b; slot_ent[b] = el_last; }
                if (last >= 1u) el_last = slot_ent[last - 1u];                    // the next merge's last slot; after my own stores (same-thread order)
            }
            if constexpr (REMOTE) {
                if (spec.valid && wave_id == 0u) {        // the helper's answer, awaited while the merge's stores drain
                    unsigned payload = 0u, tg = 0u;
                    long long tw = 0;
                    if (prof) tw = clock64();
                    const bool ok = ix_poll(mail->ans, tid, 10u, seq, 0u, spin_limit, payload, tg);
                    if (prof) t_wait += (unsigned long long)(clock64() - tw);
                    unsigned* m = L.sh->msg[seq & 1u];
                    if (tid < 10u) m[tid] = payload;
                    if (tid == 0u) m[15] = ok ? 1u : 0u;
                }
            }
            if (have_prev && (e_prev == eb || e_prev == ea)) have_prev = false;   // prev is consumed (or names the merged entry)
            a_stale = last == a;
            node_a = used;
#pragma unroll
            for (int k = 0; k < 6; ++k) box_a[k] = u[k];
            used += 1u; cnt -= 1u;
            __syncthreads();                              // thread 0's entry stores are complete and visible to the other waves
            if (prof && tid == 0u) prof->t_merge += (unsigned long long)(clock64() - tm);
            if (refresh_every && ++since_refresh >= refresh_every) {
                if (prof) tm = clock64();
                ix_refresh(entries, L);
                since_refresh = 0u; epoch += 1u;
                if (prof && tid == 0u) prof->t_refresh += (unsigned long long)(clock64() - tm);
            }
            bool answered = spec.valid;
            if constexpr (REMOTE) {
                if (spec.valid) {
                    const unsigned* m = L.sh->msg[seq & 1u];
                    if (m[15] != 0u) {
                        IxHit h;
                        h.key = ((vd_u64)m[1] << 32) | m[0]; h.e = m[2]; h.node = m[3]; h.slices = 0u;
#pragma unroll
                        for (int k = 0; k < 6; ++k) h.box[k] = __uint_as_float(m[4 + k]);
                        take(h, a, ea, node_a, box_a, b, eb, node_b, box_b);
                        n_remote += 1u;
                    } else {                              // the helper did not answer in time: on our own from here on
                        remote_ok = false; answered = false;
                    }
                }
            } else if (spec.valid) {                      // answered while best(b) was being worked out
                take(hit.spec, a, ea, node_a, box_a, b, eb, node_b, box_b);
            }
            if (!answered) {
                const float bnd = have_prev ? ix_union_area(box_a, box_prev) : kNone;
                take(ix_query<G, SPEC>(entries, L, q++, a, box_a, ea, bnd, prof, no_spec).main, a, ea, node_a, box_a, b, eb, node_b, box_b);
            }
        } else {
            have_prev = true; e_prev = ea;
#pragma unroll
            for (int k = 0; k < 6; ++k) { box_prev[k] = box_a[k]; box_a[k] = box_b[k]; box_b[k] = box_c[k]; }
            a = b; ea = eb; node_a = node_b; a_stale = false;
            b = c; eb = ec; node_b = node_c;
        }
    }
    if (prof && (tid == 0u || (SPEC && tid == 64u * G)))
        printf("tlas indexed build n=%u: %llu queries (%llu with an own-block bound), survivors per query: %.1f super-slices, %.1f slices; "
               "cycles per query: bounds %.0f, entries %.0f, reduce+barrier %.0f; per merge %.0f; refresh total %llu\n",
               n, prof->queries, prof->own, (double)prof->n1 / prof->queries, (double)G * prof->n2 / prof->queries, (double)prof->t_bounds / prof->queries,
               (double)prof->t_entries / prof->queries, (double)prof->t_reduce / prof->queries, (double)prof->t_merge / (n - cnt), prof->t_refresh);
    post_exit();
    if (REMOTE && prof && tid == 0u)
        printf("tlas indexed build, two workgroups: %u requests, %u answers used, helper %s; cycles wave 0 waited for an answer: %.0f per answer used\n",
               seq, n_remote, remote_ok ? "alive" : "timed out", n_remote ? (double)t_wait / n_remote : 0.0);
    // ---- hand over to the plain scan: slot arrays from the live entries (+ the stale slot a the chain may still name) ----
    // The corner tables are dead from here on: the slot arrays of the <= phase2_cnt clusters that are left go where they
    // were, in LDS (the host sized the allocation for both), and the ~3 scans per remaining merge read them there.
    __syncthreads();
    auto hand_over = [&](float* hb, unsigned* hn, unsigned hcap) {
        for (unsigned e = tid; e < E; e += kThreads) {
            const float4 lo = reinterpret_cast<const float4*>(entries + e)[0], hi = reinterpret_cast<const float4*>(entries + e)[1];
            const unsigned slot = __float_as_uint(lo.w);
            if (slot != kIxDead) {
                hb[slot] = lo.x; hb[hcap + slot] = lo.y; hb[2 * hcap + slot] = lo.z;
                hb[3 * hcap + slot] = hi.x; hb[4 * hcap + slot] = hi.y; hb[5 * hcap + slot] = hi.z;
                hn[slot] = __float_as_uint(hi.w);
            }
        }
        __syncthreads();
        if (a_stale && tid == 0u) {
#pragma unroll
            for (int k = 0; k < 6; ++k) hb[k * hcap + a] = box_a[k];
            hn[a] = node_a;
        }
        __syncthreads();
        tlas_build_chain<Node, true, kThreads>(nodes, hb, hn, hcap, L.sh->red, ChainState{cnt, used, a, b, true});
    };
    if (chain_in_lds) {
        const unsigned lcap = (phase2_cnt + 8u + 3u) & ~3u;   // slot `cnt` itself can be named (the stale a)
        float* ls = reinterpret_cast<float*>(L.slice);
        hand_over(ls, reinterpret_cast<unsigned*>(ls + 6u * lcap), lcap);
    } else {
        hand_over(sb, slot_node, cap);
    }
    if (REMOTE && tid == 0u) ctl->done = 1u;
}

// ---- refit -------------------------------------------------------------------------------
// The agglomerative tree is tall and thin (32 768 instances of the bench cloud: height 70, and about half the levels
// of the longest path join a cluster with a single leaf), so a refit is one long dependent climb and what counts is
// the number of memory-side round trips per level.
// Launch 1: thread i < n recomputes leaf i+1; the same thread records, for the two children of interior node n+1+i,
// {parent, sibling} and clears the node's handshake word.
// Links live in a refit-only arena of the context that is zeroed when allocated and tagged per launch: an entry
// {parent, sibling, epoch} counts only when its epoch is this launch's, so a node nobody links - the top of a chain -
// is recognised without clearing anything between launches.  (The reference's chain DROPS a cluster whose unions all
// reach 1e30 - an infinite leaf box: find_best_match answers the target itself, tlas.rs:88-104, the cluster is merged with
// itself into a node nothing refers to and its slot goes to the last cluster, tlas.rs:62-79 - so tops other than node 2n
// exist, and node 0 - a COPY of the node the chain ended on, tlas.rs:84 - is not always a copy of node 2n.)
struct RefitRoot { unsigned node, epoch; };        // the interior node whose payload node 0 copies, found by the prep pass
template <typename Node>
__global__ __launch_bounds__(256) void tlas_refit_prep_kernel(const VdInstance* __restrict__ inst, unsigned n,
                                                              const VdMeshInfo* __restrict__ meshes, unsigned n_mesh,
                                                              Node* __restrict__ nodes, uint4* __restrict__ up,
                                                              unsigned* __restrict__ arrivals, RefitRoot* __restrict__ root, unsigned epoch) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const Box b = leaf_box(inst, meshes, n_mesh, nodes[i + 1].instance_idx);
#pragma unroll
    for (int c = 0; c < 3; ++c) { nodes[i + 1].min[c] = b.mn[c]; nodes[i + 1].max[c] = b.mx[c]; }
    const unsigned k = n + 1u + i;
    unsigned l, r, l0, r0;
    node_get_children(nodes[k], l, r);
    node_get_children(nodes[0], l0, r0);
    up[l] = make_uint4(k, r, epoch, 0u);
    up[r] = make_uint4(k, l, epoch, 0u);
    arrivals[k] = 0u;
    // node 0's box follows from its OWN payload, which a refit never touches: a leaf copy -> that instance's leaf box;
    // otherwise the union of its children = the box of the interior node with the same children (node 2n in every
    // ordinary build): the climber that finishes that node writes node 0 as well
    if (l0 != 0u || r0 != 0u) { if (l == l0 && r == r0) *root = RefitRoot{k, epoch}; }
    else if (i == 0u) {
        const Box b0 = leaf_box(inst, meshes, n_mesh, nodes[0].instance_idx);
#pragma unroll
        for (int c = 0; c < 3; ++c) { nodes[0].min[c] = b0.mn[c]; nodes[0].max[c] = b0.mx[c]; }
    }
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
__global__ __launch_bounds__(256) void tlas_refit_up_kernel(Node* nodes, unsigned n, const uint4* __restrict__ up,
                                                            unsigned* arrivals, const RefitRoot* __restrict__ root, unsigned epoch) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const unsigned c = i + 1u;
    const RefitRoot rt = *root;
    const unsigned root_src = rt.epoch == epoch ? rt.node : 0xffffffffu;
    const uint4 u0 = up[c];
    if (u0.z != epoch || u0.y > n || c > u0.y) return;    // no parent; sibling interior: its climber picks this leaf up; or the other leaf's lane
    unsigned k = u0.x;
    uint4 u = up[k];                                      // {parent, sibling, epoch} of the node being finished
    Box box = box_union(load_box_plain(nodes, c), load_box_plain(nodes, u0.y));
    for (;;) {
        // stores count in vmcnt on gfx950, and a wait for a load issued after them waits for them too: the loads a level
        // needs - the next level's link and the sibling's box - are issued together, right after this level's link is in
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            __hip_atomic_store(&nodes[k].min[q], box.mn[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nodes[k].max[q], box.mx[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (k == root_src) {                              // node 0 copies this node's payload: same children, same box
#pragma unroll
            for (int q = 0; q < 3; ++q) { nodes[0].min[q] = box.mn[q]; nodes[0].max[q] = box.mx[q]; }
        }
        if (u.z != epoch) return;                         // nobody linked this node in this launch: the top of a chain (node 2n, or an orphan)
        const unsigned p = u.x, s = u.y;
        const uint4 un = up[p];                           // the next level's link travels with the sibling's box
        if (s == k) { k = p; u = un; continue; }          // a cluster merged with itself (node 2n: the true root): same box
        if (s <= n) { box = box_union(box, load_box_plain(nodes, s)); k = p; u = un; continue; }
        const unsigned v = __hip_atomic_fetch_add(&arrivals[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != 3u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my box has reached memory
            const unsigned w = __hip_atomic_fetch_add(&arrivals[p], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w < 4u) return;                           // the other side is not readable yet: its climber goes on
        }
        box = box_union(box, load_box_agent(nodes, s));
        k = p; u = un;
    }
}

template <typename Node>
int tlas_build_impl(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                    Node* d_nodes) {
    // The indexed build (one workgroup, exact pruning) from a few thousand instances up to what its corner table fits in
    // LDS; above that the scans are spread over several workgroups.  Per-context options (vd_ctx_set_option):
    // VD_OPT_TLAS_INDEX = 0 switches it off (A/B), VD_OPT_TLAS_INDEX_MIN / _PHASE2 / _REFRESH tune it.
    const int env_index = (int)ctx->option(VD_OPT_TLAS_INDEX, 1);
    const unsigned ix_min = (unsigned)ctx->option(VD_OPT_TLAS_INDEX_MIN, 6800);    // below: the chain is faster (from LDS up to 5600 instances; A/B in profiles/NOTEBOOK_r01_r04.md, the script is in the history)
    unsigned phase2 = (unsigned)ctx->option(VD_OPT_TLAS_PHASE2, 4096);     // 2048 until the final scans moved into LDS (round 4): 184.6 -> 182.7 ms at 32 768, 41.1 -> 39.2 at 8192
    const unsigned refresh = (unsigned)ctx->option(VD_OPT_TLAS_REFRESH, 512);      // re-swept in round 4: 256: 183.4 ms at 32 768, 512: 182.4, 1024: 183.1, 2048: 185.5
    if (phase2 < 64u) phase2 = 64u;
    const bool indexed = env_index != 0 && n >= ix_min && n > phase2 && n <= kIxMaxInstances;
    // scratch: 6 float slot arrays + slot node ids (capacity n), the several-workgroup exchange words, and - only when the
    // indexed build runs - the entries, the slot -> entry map, two key / value pairs of the sort and the control words
    const size_t cap = ((size_t)n + 7) & ~(size_t)3;   // multiple of 4 (+ slack): slot arrays are read 16 B at a time
    const size_t off_sh = (cap * 7 * 4 + 255) & ~(size_t)255;
    const unsigned E = (n + kIxBlock - 1u) / kIxBlock * kIxBlock;
    const size_t off_ent = (off_sh + sizeof(MwShared) + 255) & ~(size_t)255;
    const size_t off_map = off_ent + (size_t)E * sizeof(IxEntry);
    const size_t off_sort = off_map + (((size_t)n * 4 + 255) & ~(size_t)255);
    const size_t sort_stride = ((size_t)n * 4 + 255) & ~(size_t)255;
    const size_t off_ctl = off_sort + 4 * sort_stride;
    const size_t need = indexed ? off_ctl + 256 + ((sizeof(IxMail) + 255) & ~(size_t)255) : off_ent;
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, need);
    if (rc) return rc;
    char* base = reinterpret_cast<char*>(ctx->scratch);
    float* sb = reinterpret_cast<float*>(base);
    unsigned* slot_node = reinterpret_cast<unsigned*>(sb + 6 * cap);
    MwShared* sh = reinterpret_cast<MwShared*>(base + off_sh);
    // several workgroups from a few thousand instances on (below, the exchange costs more than the shorter scans save);
    // the slot field of the exchanged key holds 20 bits.  VD_OPT_TLAS_GROUPS = 1 forces the single-workgroup kernel.
    const int env_groups = (int)ctx->option(VD_OPT_TLAS_GROUPS, 0);
    const unsigned spin_limit = (unsigned)ctx->option(VD_OPT_TLAS_SPIN_LIMIT, kSpinLimit);   // tests: 0 forces the fallback
    unsigned groups = env_groups > 0 ? (unsigned)env_groups : (n >= 12288u ? 16u : 1u);
    if (groups > kMwMaxGroups) groups = kMwMaxGroups;
    if (groups > (unsigned)ctx->num_cus) groups = (unsigned)ctx->num_cus;
    if (n >= (1u << 20)) groups = 1u;
    // the single-workgroup chain over LDS-resident slot arrays (28 B per slot) for small scenes
    const int chain_in_lds = n <= kChainLdsMax && ctx->option(VD_OPT_TLAS_CHAIN_LDS, 1) != 0 ? 1 : 0;
    const size_t chain_lds = chain_in_lds ? cap * 28 : 0;
    if (chain_in_lds) {
        constexpr int which = std::is_same<Node, VdTlasNode>::value ? 0 : 1;
        if (!ctx->tlas_chain_lds_opt_in[which]) {
            const int max_lds = (int)((((size_t)kChainLdsMax + 7) & ~(size_t)3) * 28);
            VD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tlas_build_kernel<Node>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
            ctx->tlas_chain_lds_opt_in[which] = true;
        }
    }
    // (fewer waves for small scenes - 256 lanes up to 1024 instances, 512 up to 2048 - were measured: no change; a scan is
    //  ~1 us of dependent instructions whatever the size: target, loads, arithmetic, two reductions around one barrier)
    auto launch_chain = [&](const unsigned* only_if) {
        hipLaunchKernelGGL((tlas_build_kernel<Node>), dim3(1), dim3(kBuildThreads), chain_lds, ctx->stream, d_nodes, n, sb, slot_node, (unsigned)cap, only_if, chain_in_lds);
    };
    vd_time_begin(ctx);
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_nodes, 0, sizeof(Node) * (2 * (size_t)n + 1), ctx->stream));   // TlasNode::default()
    hipLaunchKernelGGL((tlas_leaves_kernel<Node>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_inst, n, d_meshes,
                       n_mesh, d_nodes, sb, slot_node, (unsigned)cap, 0, (const unsigned*)nullptr);
    if (indexed) {
        IxEntry* entries = reinterpret_cast<IxEntry*>(base + off_ent);
        unsigned* slot_ent = reinterpret_cast<unsigned*>(base + off_map);
        unsigned* keys[4];
        for (int k = 0; k < 4; ++k) keys[k] = reinterpret_cast<unsigned*>(base + off_sort + k * sort_stride);
        IxCtl* ctl = reinterpret_cast<IxCtl*>(base + off_ctl);
        const unsigned n_slices = E / kIxSlice, n_super = (n_slices + kIxSuper - 1u) / kIxSuper;
        IxMail* mail = reinterpret_cast<IxMail*>(base + off_ctl + 256);
        const int chain_in_lds_ix = ctx->option(VD_OPT_TLAS_CHAIN_LDS, 1) != 0 && (size_t)phase2 * 28 <= 120000 ? 1 : 0;
        const int spec_mode = (int)ctx->option(VD_OPT_TLAS_SPEC, 1);      // 0 none, 1 helper waves, 2 helper workgroup (then 1 if nobody shows up)
        const bool spec = spec_mode != 0;
        hipLaunchKernelGGL(tlas_index_kernel, dim3(1), dim3(kSortThreads), 0, ctx->stream, sb, (unsigned)cap, n, E, keys[0], keys[1], keys[2], keys[3],
                           entries, slot_ent, ctl);
        auto launch = [&](auto gc, auto sc, auto rc_) -> int {
            constexpr int G = decltype(gc)::value;
            constexpr bool S = decltype(sc)::value, R = decltype(rc_)::value;
            constexpr unsigned waves = G * (S ? 2 : 1);
            // (+ room for the slot arrays of the plain scans the build ends with: they take the corner tables' place)
            const size_t lds_chain = chain_in_lds_ix ? ((sizeof(IxShared) + 15) & ~(size_t)15) + (size_t)((phase2 + 8u + 3u) & ~3u) * 28 : 0;
            const size_t lds = std::max(ix_lds_bytes(n_slices, n_super, G, waves), lds_chain);
            constexpr int which = (std::is_same<Node, VdTlasNode>::value ? 0 : 1) + (S ? 2 : 0) + (R ? 4 : 0);
            if (!ctx->tlas_ix_lds_opt_in[which]) {        // per context (= per device): up to 160 KB of dynamic LDS
                constexpr unsigned max_slices = kIxMaxInstances / kIxSlice;
                VD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tlas_build_indexed_kernel<Node, G, S, R>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      (int)ix_lds_bytes(max_slices, max_slices / kIxSuper, G, waves)));
                ctx->tlas_ix_lds_opt_in[which] = true;
            }
            // two-workgroup form: 32 workgroups, four per XCC - the chain runs on workgroup 0, one of the three others that
            // land on its XCC becomes the helper, everybody else leaves at once
            hipLaunchKernelGGL((tlas_build_indexed_kernel<Node, G, S, R>), dim3(R ? 32 : 1), dim3(64 * waves), lds, ctx->stream, d_nodes, n, entries, slot_ent, E, sb,
                               slot_node, (unsigned)cap, ctl, phase2, refresh, ctx->option(VD_OPT_TLAS_PROFILE, 0) ? 1 : 0, mail, spin_limit, chain_in_lds_ix);
            return 0;
        };
        using std::integral_constant;
        if (spec_mode == 2) {
            VD_HIP_CHECK(ctx, hipMemsetAsync(mail, 0, sizeof(IxMail), ctx->stream));
            const int rrc = launch(integral_constant<int, kIxGroup>{}, std::false_type{}, std::true_type{});
            if (rrc) return rrc;
        }
        // VD_TLAS_SPEC=0: without the helper waves (A/B); behind the two-workgroup form: only if that one found no partner
        const int lrc = spec ? launch(integral_constant<int, kIxGroup>{}, std::true_type{}, std::false_type{})
                             : launch(integral_constant<int, kIxGroup>{}, std::false_type{}, std::false_type{});
        if (lrc) return lrc;
        // Decided on the device: leaf coordinates the fast arithmetic cannot order (NaN, inf, |x| >= 1e18), or boxes that
        // defeat the pruning (all union areas tie).  The indexed kernel then returned early and left the slot arrays as the
        // leaves kernel wrote them; the plain chain runs instead - on 16 workgroups where that pays, with ITS redo behind it.
        if (groups <= 1u) {
            launch_chain((const unsigned*)&ctl->fallback);
        } else {
            VD_HIP_CHECK(ctx, hipMemsetAsync(sh, 0, sizeof(MwShared), ctx->stream));
            hipLaunchKernelGGL((tlas_build_mw_kernel<Node>), dim3(groups * 8u), dim3(kMwThreads), 0, ctx->stream, d_nodes, n, sb, slot_node,
                               (unsigned)cap, sh, spin_limit, (const unsigned*)&ctl->fallback);
            hipLaunchKernelGGL((tlas_leaves_kernel<Node>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_inst, n, d_meshes,
                               n_mesh, d_nodes, sb, slot_node, (unsigned)cap, 0, (const unsigned*)&sh->fail);
            launch_chain((const unsigned*)&sh->fail);
        }
    } else if (groups <= 1u) {
        launch_chain((const unsigned*)nullptr);
    } else {
        VD_HIP_CHECK(ctx, hipMemsetAsync(sh, 0, sizeof(MwShared), ctx->stream));
        hipLaunchKernelGGL((tlas_build_mw_kernel<Node>), dim3(groups * 8u), dim3(kMwThreads), 0, ctx->stream, d_nodes, n, sb, slot_node,
                           (unsigned)cap, sh, spin_limit, (const unsigned*)nullptr);
        // If the workgroups did not hear from each other in time (not co-resident: spin limit) they set sh->fail and
        // leave; the two launches below then redo the build on one workgroup, and return at once otherwise - decided
        // on the device, so the call stays asynchronous.  Every node a build writes is written again by the redo.
        hipLaunchKernelGGL((tlas_leaves_kernel<Node>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_inst, n, d_meshes,
                           n_mesh, d_nodes, sb, slot_node, (unsigned)cap, 0, (const unsigned*)&sh->fail);
        launch_chain((const unsigned*)&sh->fail);
    }
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

template <typename Node>
int tlas_refit_impl(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                    Node* d_nodes) {
    const size_t total = 2 * (size_t)n + 1;
    const size_t need = 256 + total * 16 + total * 4;
    // The arrival words sit behind the link records, so their offset depends on n: after refits at n1 a refit at another
    // n in the same arena would read old arrival words (values up to 4 / 6 = plausible small epochs) as link records.
    // A change of n therefore starts the arena over, like a reallocation or an epoch wrap does.
    if (need > ctx->refit_state_bytes || !ctx->refit_state || ctx->refit_epoch == 0xffffffffu || n != ctx->refit_n) {
        int rc = vd_ensure(ctx, &ctx->refit_state, &ctx->refit_state_bytes, need);
        if (rc) return rc;
        VD_HIP_CHECK(ctx, hipMemsetAsync(ctx->refit_state, 0, ctx->refit_state_bytes, ctx->stream));   // epoch 0 = never written
        ctx->refit_epoch = 0u;
        ctx->refit_n = n;
    }
    const unsigned epoch = ++ctx->refit_epoch;
    char* base = reinterpret_cast<char*>(ctx->refit_state);
    RefitRoot* root = reinterpret_cast<RefitRoot*>(base);
    uint4* parent = reinterpret_cast<uint4*>(base + 256);                      // {parent, sibling, epoch} per node
    unsigned* arrivals = reinterpret_cast<unsigned*>(parent + total);
    vd_time_begin(ctx);
    hipLaunchKernelGGL((tlas_refit_prep_kernel<Node>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_inst, n, d_meshes, n_mesh,
                       d_nodes, parent, arrivals, root, epoch);
    hipLaunchKernelGGL((tlas_refit_up_kernel<Node>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_nodes, n, parent, arrivals, root, epoch);
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

// internal (vd_common.hpp): the agglomerative build of tlas.rs:56-105 over leaf boxes given as six floats each
int vd_tlas_build_from_boxes(VdCtx* ctx, const float* d_boxes, uint32_t n, VdTlasNode* d_nodes) {
    if (!ctx || !d_boxes || !d_nodes || n == 0 || n > VD_TLAS_MAX_INSTANCES) return VD_ERR_INVALID_ARG;
    return tlas_build_impl<VdTlasNode>(ctx, reinterpret_cast<const VdInstance*>(d_boxes), n, nullptr, 0u, d_nodes);
}

extern "C" {

int vd_tlas_build_dev(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                      VdTlasNode* d_nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, d_inst, n, d_meshes, n_mesh, d_nodes, false);
    if (!rc) ctx->fan_forget(d_nodes);     // vd_trace*: what the last walk over this top level found no longer holds
    return rc ? rc : tlas_build_impl<VdTlasNode>(ctx, d_inst, n, d_meshes, n_mesh, d_nodes);
}
int vd_tlas_build_wide_dev(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                           VdTlasNodeWide* d_nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, d_inst, n, d_meshes, n_mesh, d_nodes, true);
    if (!rc) ctx->fan_forget(d_nodes);     // vd_trace*: what the last walk over this top level found no longer holds
    return rc ? rc : tlas_build_impl<VdTlasNodeWide>(ctx, d_inst, n, d_meshes, n_mesh, d_nodes);
}
int vd_tlas_refit_dev(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                      VdTlasNode* d_nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, d_inst, n, d_meshes, n_mesh, d_nodes, false);
    if (!rc) ctx->fan_forget(d_nodes);     // vd_trace*: what the last walk over this top level found no longer holds
    return rc ? rc : tlas_refit_impl<VdTlasNode>(ctx, d_inst, n, d_meshes, n_mesh, d_nodes);
}
int vd_tlas_refit_wide_dev(VdCtx* ctx, const VdInstance* d_inst, uint32_t n, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                           VdTlasNodeWide* d_nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_args(ctx, d_inst, n, d_meshes, n_mesh, d_nodes, true);
    if (!rc) ctx->fan_forget(d_nodes);     // vd_trace*: what the last walk over this top level found no longer holds
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

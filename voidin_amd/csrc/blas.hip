// blas.hip — bit-exact SAH BLAS builder for gfx950.
//
// Replaces BvhBuilder::new(..).build() (reference: crates/bvh/src/blas.rs:51-204), called per
// mesh from MeshPool::add (crates/pools/src/mesh/mod.rs:320-321).  The reference builder is a
// sequential, order-dependent recursion: 21 trial `partition_shuffle`s
// per node, each starting from the arrangement the previous one left, a never-examined element,
// NaN-rejected empty splits and a stale pivot.  The node layout and the permuted index buffer
// must match it bit for bit, so the algorithm is EMULATED, not replaced:
//
//  * one shuffle (blas.rs:168-182) has a closed form: with p the predicate at each position,
//    TL(x) = #true in [0,x), F = x - TL, T = Ttot - TL - p, f_j / t_j the position of the j-th
//    false from the left / true from the right: x is consumed from the left iff x < t_F, its
//    fetch index is x + n - t_F (left) or (n-1-x) + f_{T+1} + 1 (right); the element with fetch
//    index n-1 is the unexamined one `u` and lands on the pivot L = Ttot - p(u); examined trues
//    keep x (left) or go to f_{T+1} (right); examined falses go to t_F - 1 (left) or x - 1
//    (right).  => one shuffle = one prefix count + two rank->position tables + one scatter;
//  * the SAH cost of a trial depends on the arrangement only through `u`: left = trues \ {u},
//    right = falses + {u}.  Bounds therefore come from ONE binning pass per node (8 bins per
//    axis, integer atomic min/max on order-preserving float keys) that skips the <= 21 `u`
//    elements, which are added back per candidate;
//  * children boxes are reduced from the final arrangement, as the reference does.
//
// Phases: A (segments > kMidMax prims): level-synchronous, many workgroups per segment; the 21
// predicates of a level are evaluated once and the arrangement ping-pongs between two 8-byte
// {triangle id, predicate bits} arrays in HBM.  Mid tier (kSmallMax < n <= kMidMax): one workgroup
// per segment, the same rounds with the arrangement in LDS (barriers instead of launches).
// B (<= kSmallMax): one workgroup builds the whole subtree out of LDS - no levels: a wave that has split a
// node goes on with one child and hands the other to an idle wave; nodes of <= 32 prims are split
// several per wave in lane groups of 8 / 16 / 32; the subtree root's shuffles are shared by all waves - and
// restores the reference's pre-order node numbering locally from (start, -count) keys.  C: DFS numbering of
// the (small) top tree on the host while phase B runs, parallel copy-out.
#include "vd_common.hpp"

#include <chrono>
#include <type_traits>

#include <vector>

namespace {

#ifndef VD_SMALL_MAX
#define VD_SMALL_MAX 512
#endif
#ifndef VD_BINEVAL_MIN
#define VD_BINEVAL_MIN 64
#endif
#ifndef VD_LANE_MAX
#define VD_LANE_MAX 8
#endif
// -DVD_ISA_MARKS: comment lines + scheduling barriers at the section boundaries of the small-node path, so that the instructions of
// a section can be counted in the assembly (tools/blas_small_isa.py -> profiles/r04_blas_small_isa.txt); never in a product build
#ifdef VD_ISA_MARKS
#define VD_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; VDMARK " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define VD_MARK(name) do { } while (0)
#endif
constexpr int kSmallMax = VD_SMALL_MAX;  // largest segment built by one wave out of LDS
constexpr int kChunks = kSmallMax / 64;
constexpr int kCand = 21;               // 3 axes x 7 planes (blas.rs:144-145; `bins` hard-coded to 8)
constexpr int kBig = 0x7fffffff;
#ifndef VD_ITEM
#define VD_ITEM 1024
#endif
constexpr int kItem = VD_ITEM;          // phase A: positions per workgroup item (256 lanes x kPer)
constexpr int kPer = kItem / 256;       // positions per lane: position = rel0 + wave * (kItem / 4) + j * 64 + lane
constexpr unsigned kMidEarlyMin = 256;  // mid-tier roots that make an early launch worth it (one per CU)
#ifndef VD_BIN_ITEMS
#define VD_BIN_ITEMS 8
#endif
constexpr int kBinItems = VD_BIN_ITEMS;  // a_bin_kernel: consecutive items per workgroup
constexpr unsigned kFinalChunk = 128u;  // a_boundary_kernel: segments of the level that ended per workgroup
#ifndef VD_MID_MAX
#define VD_MID_MAX 2048
#endif
constexpr int kMidMax = VD_MID_MAX;      // largest segment split by ONE workgroup out of LDS (mid tier); 0 disables the tier
// Mid tier geometry (A/B, profiles/r04_blas_item_isa.txt): 4 positions per lane = 512 lanes per workgroup, four workgroups per
// CU (a 64-register budget) - 27.0 ms per build against 27.7 with 8 positions per lane / 256 lanes: per-lane state halves, and so
// do the serial gathers of the bits / bin / child loops; 2 positions per lane (1024 lanes): 27.9.
#ifndef VD_MID_PER
#define VD_MID_PER 4
#endif
#ifndef VD_MID_WGS
#define VD_MID_WGS 4                    // workgroups per CU the mid kernel's register budget is set for
#endif
constexpr int kMidPer = VD_MID_PER;     // positions per lane: position = wave * 64 * kMidPer + j * 64 + lane
constexpr int kMidThreads = kMidMax > 0 ? kMidMax / kMidPer : 64;
constexpr unsigned kNone = 0xffffffffu;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 24 B, by triangle id.  (32 B with two pad words - two aligned 16-byte loads - until the kernels that move boxes turned out
// HBM-bound: a_child at 5.2 TB/s with 14 % VALU; 24 B: 26.8 -> 26.05 ms per build, -DVD_BOX32 for the A/B.)
#ifdef VD_BOX32
struct TriBox { float mn[3]; float pad0; float mx[3]; float pad1; };
#else
struct TriBox { float mn[3]; float mx[3]; };
#endif
struct TmpNode { float mn[3]; unsigned left_first; float mx[3]; unsigned count; };   // == VdBvhNode layout

enum : unsigned { ERR_DEGENERATE = 1u, ERR_BAD_INDEX = 2u, ERR_INTERNAL = 4u };
constexpr unsigned kInteriorMark = 0x80000000u;   // phase B: TmpNode.count of a split node until the renumber (n stays in the low bits)

__device__ __forceinline__ float pay_c(const u32x4& v, int axis) {
    return __uint_as_float(axis == 0 ? v.y : (axis == 1 ? v.z : v.w));
}

// MAX_DIST-seeded bounds (blas.rs:185-186): `min = min(1e30, ..)`, `max = max(-1e30, ..)`
__device__ __forceinline__ float box_lo(int key) { return vd_unkey(min(key, vd_key(1e30f))); }
__device__ __forceinline__ float box_hi(int key) { return vd_unkey(max(key, vd_key(-1e30f))); }

// Wave-wide min / max / sum entirely on the VALU: inside a 16-lane row by DPP (quad_perm, row_half_mirror, row_mirror),
// across the rows by row_bcast:15 (row r's last lane into row r + 1; rows 1 and 3 take it) and row_bcast:31 (lane 31
// into rows 2 and 3), which leaves the result in lane 63; one v_readlane hands it to everybody.  (The two cross-row steps
// were ds_bpermute round trips through the LDS crossbar, each with its s_waitcnt: 24 reductions per workgroup in
// a_child_kernel = 48 dependent crossbar trips.)
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int CTRL, int ROWS> __device__ __forceinline__ int dpp_rows_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWS, 0xf, false); }
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v)); v = min(v, dpp_i<0x141>(v)); v = min(v, dpp_i<0x140>(v));
    v = min(v, dpp_rows_i<0x142, 0xA>(v, v)); v = min(v, dpp_rows_i<0x143, 0xC>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, dpp_i<0xB1>(v)); v = max(v, dpp_i<0x4E>(v)); v = max(v, dpp_i<0x141>(v)); v = max(v, dpp_i<0x140>(v));
    v = max(v, dpp_rows_i<0x142, 0xA>(v, v)); v = max(v, dpp_rows_i<0x143, 0xC>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned wave_sum_u(unsigned u) {
    int v = (int)u;
    v += dpp_i<0xB1>(v); v += dpp_i<0x4E>(v); v += dpp_i<0x141>(v); v += dpp_i<0x140>(v);
    v += dpp_rows_i<0x142, 0xA>(0, v); v += dpp_rows_i<0x143, 0xC>(0, v);
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}

// min / max over the aligned group of 8 lanes a lane belongs to, on the VALU (DPP) - the LDS crossbar is what the
// shuffles of the trials already saturate.  row_half_mirror pairs lane i with 7 - i, then each quad holds four
// pairs that cover all eight lanes: quad_perm [1,0,3,2] and [2,3,0,1] finish the reduction.
__device__ __forceinline__ int group8_min_i(int v) {
    v = min(v, dpp_i<0x141>(v)); v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v));
    return v;
}
__device__ __forceinline__ int group8_max_i(int v) {
    v = max(v, dpp_i<0x141>(v)); v = max(v, dpp_i<0xB1>(v)); v = max(v, dpp_i<0x4E>(v));
    return v;
}
// groups of gw = 8 / 16 / 32 lanes (wave-uniform gw): row_mirror joins the two 8-groups of a row, one shuffle the two rows
__device__ __forceinline__ int group_min_i(int v, unsigned gw) {
    v = group8_min_i(v);
    if (gw >= 16u) v = min(v, dpp_i<0x140>(v));
    if (gw >= 32u) v = min(v, __shfl_xor(v, 16));
    return v;
}
__device__ __forceinline__ int group_max_i(int v, unsigned gw) {
    v = group8_max_i(v);
    if (gw >= 16u) v = max(v, dpp_i<0x140>(v));
    if (gw >= 32u) v = max(v, __shfl_xor(v, 16));
    return v;
}

// Split plane of candidate c (axis = c / 7, k = c % 7 + 1): glam lerp = min + (max - min) * (k/8)
__device__ __forceinline__ float cand_pos(const float* cbmin, const float* cbmax, int c) {
    const int axis = c / 7, k = c % 7 + 1;
    const float scale = (float)k / 8.0f;
    return cbmin[axis] + (cbmax[axis] - cbmin[axis]) * scale;
}

// ---- precompute: centroid payload + triangle boxes + root box -------------------------------
// A thread takes kPreTris triangles (a workgroup a contiguous run of 256 * kPreTris), keeps the 12 root keys in
// registers, and the reduction is wave (DPP) -> 4 LDS atomics per key -> ONE global atomic per key and workgroup:
// same-address global atomics are serialised chip-wide (~2-5 ns each), and one workgroup per 256 triangles made
// 393 k of them at 8.4 M triangles - most of this kernel's 0.9 ms.
constexpr int kPreTris = 8;
constexpr unsigned kChunk = 256u * kPreTris;      // triangles per workgroup of the precompute and of the final index permute

// One build takes K meshes (vd_bvh_build_batch*; vd_bvh_build* is K = 1).  Their triangles are laid side by side in ONE
// position space - mesh m owns positions [base, base + n_tri) - so every later pass (the level loop over a forest of
// segments, the mid tier, the small subtrees) runs once for the whole batch; only the passes that touch the caller's
// buffers (here, and the copy-out) look the mesh up: workgroup -> chunk of kChunk triangles -> mesh, by bisection over
// the meshes' first chunks.
struct MeshDesc {
    const float* verts; const unsigned* idx_in; unsigned* idx_out; VdBvhNode* out;
    unsigned n_vert, n_tri, base, chunk0;
};
__device__ __forceinline__ unsigned mesh_of_chunk(const MeshDesc* __restrict__ meshes, unsigned n_meshes, unsigned chunk) {
    unsigned lo = 0, hi = n_meshes;              // largest m with meshes[m].chunk0 <= chunk
    while (hi - lo > 1u) { const unsigned mid = (lo + hi) >> 1; if (meshes[mid].chunk0 <= chunk) lo = mid; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(256) void blas_precompute_kernel(const MeshDesc* __restrict__ meshes, unsigned n_meshes,
                                                              unsigned* __restrict__ idx_copy,
                                                              f32x4* __restrict__ cent, TriBox* __restrict__ boxes,
                                                              int* __restrict__ root_keys /*[16] per mesh: box, centroid box*/,
                                                              unsigned* __restrict__ err, unsigned* __restrict__ bad_mesh) {
    __shared__ int s_k[12];
    __shared__ unsigned s_m;
    if (threadIdx.x < 12) s_k[threadIdx.x] = (threadIdx.x % 6) < 3 ? kBig : -kBig - 1;
    if (threadIdx.x == 0) s_m = mesh_of_chunk(meshes, n_meshes, blockIdx.x);
    __syncthreads();
    const unsigned m = s_m;
    const MeshDesc md = meshes[m];
    const float* __restrict__ verts = md.verts;
    const unsigned* __restrict__ idx = md.idx_in;
    const unsigned n_tri = md.n_tri, n_vert = md.n_vert, chunk = blockIdx.x - md.chunk0;
    int k12[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) k12[q] = (q % 6) < 3 ? kBig : -kBig - 1;
    bool bad = false;
#pragma unroll 2
    for (int r = 0; r < kPreTris; ++r) {
        const unsigned t = (chunk * (unsigned)kPreTris + (unsigned)r) * 256u + threadIdx.x;   // coalesced per round
        if (t >= n_tri) break;
        const unsigned g = md.base + t;                     // the triangle's position / id in the batch
        const unsigned i0 = idx[3u * (size_t)t], i1 = idx[3u * (size_t)t + 1], i2 = idx[3u * (size_t)t + 2];
        idx_copy[3u * (size_t)g] = i0; idx_copy[3u * (size_t)g + 1] = i1; idx_copy[3u * (size_t)g + 2] = i2;   // the permute's source (blas.rs:95-100 clones too)
        TriBox bx; float ce[3];
        if (i0 >= n_vert || i1 >= n_vert || i2 >= n_vert) {
            bad = true;
            for (int k = 0; k < 3; ++k) { ce[k] = 0.0f; bx.mn[k] = 0.0f; bx.mx[k] = 0.0f; }
        } else {
        const float* a = verts + 3u * (size_t)i0; const float* b = verts + 3u * (size_t)i1; const float* c = verts + 3u * (size_t)i2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float ak = vd_quiet(a[k]), bk = vd_quiet(b[k]), ck = vd_quiet(c[k]);   // a signalling NaN becomes a quiet one
            ce[k] = ((ak + bk) + ck) / 3.0f;                // blas.rs:80
            // the fold of blas.rs:184-204 starts from +-MAX_DIST and f32::min / max ignore a NaN vertex: a stored box
            // is the seeded fold over its own three vertices - never NaN - and min(1e30, .) is idempotent, so node
            // boxes reduced from these (box_lo / box_hi) equal the reference's fold over all vertices of the node
            bx.mn[k] = vd_min_to(vd_min_to(vd_min_to(1e30f, ak), bk), ck);
            bx.mx[k] = vd_max_to(vd_max_to(vd_max_to(-1e30f, ak), bk), ck);
            k12[k] = min(k12[k], vd_key(bx.mn[k]));
            k12[3 + k] = max(k12[3 + k], vd_key(bx.mx[k]));
            k12[6 + k] = min(k12[6 + k], vd_key_lo(ce[k]));     // a NaN centroid (NaN vertex) drops out of `cb`
            k12[9 + k] = max(k12[9 + k], vd_key_hi(ce[k]));
        }
        }
#ifdef VD_BOX32
        bx.pad0 = bx.pad1 = 0.0f;
#endif
        const f32x4 c4 = {ce[0], ce[1], ce[2], __uint_as_float(g)};   // .w: the triangle's id travels with its centroid
        cent[g] = c4;
        boxes[g] = bx;
    }
    if (bad) { atomicOr(err, ERR_BAD_INDEX); atomicMin(bad_mesh, m); }
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int v = (q % 6) < 3 ? wave_min_i(k12[q]) : wave_max_i(k12[q]);
        if ((threadIdx.x & 63u) == 0u) { if ((q % 6) < 3) atomicMin(&s_k[q], v); else atomicMax(&s_k[q], v); }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        if ((threadIdx.x % 6) < 3) atomicMin(&root_keys[16u * m + threadIdx.x], s_k[threadIdx.x]);
        else atomicMax(&root_keys[16u * m + threadIdx.x], s_k[threadIdx.x]);
    }
}

// Per-triangle data lives in two array sets that phase A ping-pongs: a level reads set `cur` by the position an element
// had when the level started (`pos0`) and its last pass writes set `next` in the order the level left - so the gathers of
// a level stay inside the segment's own window (cache-local once segments are a few MB) and the first pass of the next
// level streams.  cent[].w carries the original triangle id.  A segment that leaves phase A keeps the set its last level
// wrote: `parity` in its root record (bit 0; bit 1: positions were permuted again by the mid tier, see ids32).
struct ArrSet { f32x4* cent; TriBox* boxes; };
// What the 22 shuffles of a level move: the element's pos0 and the predicate bits the current rounds test.
//   Pay4 (n <= 2^25 triangles): 4 bytes = pos0 | the 7 bits of ONE axis << 25; re-derived from bits21[pos0] when the
//        rounds change axis (rounds 7, 14 and the final one) - 28 instead of 44 bytes per triangle and round through HBM;
//   Pay8: 8 bytes = {pos0, all 21 bits} for larger meshes.
struct Pay4 {
    typedef unsigned T;
    static constexpr bool kRefresh = true;
    static __device__ __forceinline__ T make(unsigned pos, unsigned bits21, unsigned axis) { return pos | (((bits21 >> (7u * axis)) & 0x7fu) << 25); }
    static __device__ __forceinline__ unsigned pos(T v) { return v & 0x1ffffffu; }
    static __device__ __forceinline__ unsigned word(T v) { return v; }
    static __host__ __device__ __forceinline__ unsigned shift(unsigned c) { return 25u + c % 7u; }
};
struct Pay8 {
    typedef u32x2 T;
    static constexpr bool kRefresh = false;
    static __device__ __forceinline__ T make(unsigned pos, unsigned bits21, unsigned) { T v = {pos, bits21}; return v; }
    static __device__ __forceinline__ unsigned pos(T v) { return v.x; }
    static __device__ __forceinline__ unsigned word(T v) { return v.y; }
    static __host__ __device__ __forceinline__ unsigned shift(unsigned c) { return c; }
};
constexpr unsigned kPay4Max = 1u << 25;

// Cost of one candidate from binned statistics + the held-out `u` elements (blas.rs:149-155).
// bins: [8][3] keys of the candidate's axis (non-u elements only); u list: payload + box.
// The held-out elements of a node, staged once per node by lanes 0..20 of the evaluating wave: ids, predicate bits, box keys
// and "this id came up before" (a trial's never-examined element can be the same triangle as an earlier trial's).  The 21
// candidates used to walk the list themselves - 21 dependent box fetches from memory per lane, ~15 us per node with one
// wave working (a_eval_kernel: 22 us a level; the mid tier: a tenth of a node's time, the other 15 waves at the barrier).
struct EvalU { int key[kCand][6]; unsigned id[kCand], bits[kCand], dup; };
__device__ __forceinline__ void eval_stage(EvalU& U, const u32x2* u_pay, const TriBox* __restrict__ boxes, unsigned lane) {
    u32x2 v = {0u, 0u};
    if (lane < (unsigned)kCand) {
        v = u_pay[lane];
        const TriBox bx = boxes[v.x];
#pragma unroll
        for (int q = 0; q < 3; ++q) { U.key[lane][q] = vd_key(bx.mn[q]); U.key[lane][3 + q] = vd_key(bx.mx[q]); }
        U.id[lane] = v.x; U.bits[lane] = v.y;
    }
    bool dup = false;
#pragma unroll
    for (int i = 0; i < kCand - 1; ++i) { const unsigned o = (unsigned)__shfl((int)v.x, i); dup |= (unsigned)i < lane && lane < (unsigned)kCand && o == v.x; }
    const unsigned long long m = __ballot(dup);
    if (lane == 0u) U.dup = (unsigned)m;
    vd_wave_lds_sync();
}
struct EvalIn {
    const int* bin_min; const int* bin_max;      // [8][3] for this axis
    const EvalU* u; unsigned own_u;              // own_u = id of this candidate's u
};
__device__ __forceinline__ float eval_candidate(const EvalIn& in, int axis, int k, float pos, unsigned n1, unsigned n) {
    int tmn[3] = {kBig, kBig, kBig}, tmx[3] = {-kBig - 1, -kBig - 1, -kBig - 1};
    int fmn[3] = {kBig, kBig, kBig}, fmx[3] = {-kBig - 1, -kBig - 1, -kBig - 1};
    for (int b = 0; b < 8; ++b) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int lo = in.bin_min[b * 3 + q], hi = in.bin_max[b * 3 + q];
            if (b < k) { tmn[q] = min(tmn[q], lo); tmx[q] = max(tmx[q], hi); }
            else { fmn[q] = min(fmn[q], lo); fmx[q] = max(fmx[q], hi); }
        }
    }
    const unsigned dup = in.u->dup;
    for (int j = 0; j < kCand; ++j) {
        if ((dup >> j) & 1u) continue;
        const unsigned id = in.u->id[j], bits = in.u->bits[j];
        const bool to_left = id != in.own_u && ((bits >> (axis * 7 + k - 1)) & 1u);   // left = examined trues; u itself goes right
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int lo = in.u->key[j][q], hi = in.u->key[j][3 + q];
            if (to_left) { tmn[q] = min(tmn[q], lo); tmx[q] = max(tmx[q], hi); }
            else { fmn[q] = min(fmn[q], lo); fmx[q] = max(fmx[q], hi); }
        }
    }
    const float a1 = vd_area(box_hi(tmx[0]) - box_lo(tmn[0]), box_hi(tmx[1]) - box_lo(tmn[1]), box_hi(tmx[2]) - box_lo(tmn[2]));
    const float a2 = vd_area(box_hi(fmx[0]) - box_lo(fmn[0]), box_hi(fmx[1]) - box_lo(fmn[1]), box_hi(fmx[2]) - box_lo(fmn[2]));
    return a1 * (float)n1 + a2 * (float)(n - n1);
}

// {cost, candidate} -> key; min over keys == "strict <, first candidate wins" (blas.rs:156).
// Rejected candidates (NaN, or cost >= f32::MAX) map to ~0.
__device__ __forceinline__ vd_u64 cost_key(float cost, unsigned c) {
    if (!(cost < 3.40282347e+38f)) return ~0ull;
    const unsigned k = (unsigned)vd_key(cost + 0.0f) ^ 0x80000000u;
    return ((vd_u64)k << 32) | c;
}
__device__ __forceinline__ vd_u64 wave_min_u64(vd_u64 v) {
    auto step = [&](vd_u64 o) { v = o < v ? o : v; };
    auto dpp = [&](auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        const unsigned lo = (unsigned)dpp_i<C>((int)(unsigned)v), hi = (unsigned)dpp_i<C>((int)(unsigned)(v >> 32));
        return ((vd_u64)hi << 32) | lo;
    };
    step(dpp(std::integral_constant<int, 0xB1>{})); step(dpp(std::integral_constant<int, 0x4E>{}));
    step(dpp(std::integral_constant<int, 0x141>{})); step(dpp(std::integral_constant<int, 0x140>{}));
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, off), hi = __shfl_xor((unsigned)(v >> 32), off);
        step(((vd_u64)hi << 32) | lo);
    }
    return v;
}

// =============================================================================================
// Phase B: one wave builds the subtree of one small segment in DFS order, entirely out of LDS.
// =============================================================================================
// Per-wave LDS image of a subtree of N <= kSmallMax triangles, addressed by LOCAL element id
// e in [0, N): centroids and box keys stay put, only the 2-byte permutation moves.
struct SmallRoot { unsigned start, count, top_node, pad; };

#ifndef VD_SUB_WAVES
#define VD_SUB_WAVES 4
#endif
constexpr int kSubWaves = VD_SUB_WAVES;                       // waves of the workgroup that shares one subtree image
struct WaveScratch {                              // per-wave: the node this wave is splitting
    float pos[kCand + 3];
    unsigned ttot[kCand + 3];
    unsigned short u_e[kCand + 3];
    unsigned short u_p[kCand + 3];
    int bin_min[3][8][3], bin_max[3][8][3];       // nodes > 64 prims: box keys of the non-`u` elements by (axis, bin)
    unsigned next_ent;                            // the child this wave goes on with (0: none)
    unsigned short g_ue[8][kCand + 3];            // group path: never-examined element of trial c, per lane group
    unsigned char g_tt[8][kCand + 3];             // group path: trues of trial c | predicate of its u << 7
    unsigned g_first, g_count;                    // group path: the batch of small nodes this wave took
};
// box keys (min xyz, max xyz) of a local element come from the per-triangle boxes in global memory (L2-resident for
// the few hundred triangles of a subtree): keeping them in LDS cost 12 KB of the image and one root per CU
#ifndef VD_BOX_IN_LDS
#define VD_BOX_IN_LDS 0
#endif
struct BoxKeys { int k[6]; };
struct WaveLds {
    float cent[3][kSmallMax];
#if VD_BOX_IN_LDS
    int box[6][kSmallMax];                        // order-preserving keys: min xyz, max xyz
#endif
    unsigned gid[kSmallMax];                      // local element -> triangle id
    unsigned short perm[2][kSmallMax];            // arrangement ping-pong (position -> local element)
    unsigned short falsepos[kSmallMax + 2];       // indexed by ABSOLUTE position s + j: segments are disjoint,
    unsigned short truepos[kSmallMax + 2];        // so waves working on different nodes never collide
    unsigned char uflag[kSmallMax];               // marks the <= 21 never-examined elements of the node being evaluated
    WaveScratch w[kSubWaves];
};
__device__ __forceinline__ BoxKeys box_keys(const WaveLds& L, const TriBox* __restrict__ boxes, unsigned e) {
    BoxKeys r;
#if VD_BOX_IN_LDS
#pragma unroll
    for (int q = 0; q < 6; ++q) r.k[q] = L.box[q][e];
#else
    const TriBox bx = boxes[L.gid[e]];
    r.k[0] = vd_key(bx.mn[0]); r.k[1] = vd_key(bx.mn[1]); r.k[2] = vd_key(bx.mn[2]);
    r.k[3] = vd_key(bx.mx[0]); r.k[4] = vd_key(bx.mx[1]); r.k[5] = vd_key(bx.mx[2]);
#endif
    return r;
}


// One closed-form shuffle of segment [s, s+n) with predicate cent[axis] < pos: reads perm[src],
// writes perm[src^1].  NCH = number of 64-position chunks compiled in (1 = the n <= 64 fast path).
// The seven planes of an axis are nested (blas.rs:146): what trial k left of its pivot passes trial k + 1 as well and
// partition_shuffle walks over it without a swap - trial k + 1 IS the shuffle of the suffix [act, n), act = the previous
// trial's pivot (phase A's rounds rest on the same fact, round_window).  `node_s`, `node_n`: the node; [band, act): what
// the previous trial froze, copied across so that the buffer written holds the whole arrangement; a wave walks over
// ceil((n - act) / 64) chunks instead of all of them.  out_ttot counts the frozen prefix (all trues) as the reference does.
template <int NCH>
__device__ __forceinline__ void wave_shuffle(WaveLds& L, int src, unsigned node_s, unsigned node_n, int axis, float pos, unsigned band, unsigned act,
                                             unsigned& out_ttot, unsigned& out_ue, unsigned& out_up) {
    const unsigned lane = vd_lane();
    for (unsigned x = band + lane; x < act; x += 64u) L.perm[src ^ 1][node_s + x] = L.perm[src][node_s + x];
    const unsigned s = node_s + act, n = node_n - act;
    const unsigned short* pin = L.perm[src] + s;
    unsigned short* pout = L.perm[src ^ 1] + s;
    const float* cen = L.cent[axis];
    unsigned long long masks[NCH];
    unsigned short el[NCH];
    unsigned ttot = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const unsigned x = ch * 64u + lane;
        bool p = false;
        el[ch] = 0;
        if (x < n) { el[ch] = pin[x]; p = cen[el[ch]] < pos; }
        masks[ch] = __ballot(p);
        ttot += (unsigned)__popcll(masks[ch]);
    }
    const unsigned ftot = n - ttot;
    unsigned run = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const unsigned x = ch * 64u + lane;
        if (x < n) {
            const bool p = (masks[ch] >> lane) & 1ull;
            const unsigned tl = run + vd_mbcnt(masks[ch]);
            if (p) L.truepos[s + ttot - tl] = (unsigned short)x;        // (T+1)-th true from the right
            else L.falsepos[s + x - tl + 1u] = (unsigned short)x;       // (F+1)-th false from the left
        }
        run += (unsigned)__popcll(masks[ch]);
    }
    vd_wave_lds_sync();
    run = 0;
    unsigned ue = 0, up = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const unsigned x = ch * 64u + lane;
        bool is_u = false, p = false;
        if (x < n) {
            p = (masks[ch] >> lane) & 1ull;
            const unsigned tl = run + vd_mbcnt(masks[ch]);
            const unsigned F = x - tl, T = ttot - tl - (p ? 1u : 0u);
            const int tF = F == 0u ? (int)n : (F <= ttot ? (int)L.truepos[s + F] : -1);
            const bool left = (int)x < tF;
            const unsigned fj = (T + 1u <= ftot) ? (unsigned)L.falsepos[s + T + 1u] : n;
            const unsigned fetch = left ? x + n - (unsigned)tF : (n - 1u - x) + fj + 1u;
            is_u = fetch == n - 1u;
            unsigned dest;
            if (is_u) dest = ttot - (p ? 1u : 0u);
            else if (left) dest = p ? x : (unsigned)tF - 1u;
            else dest = p ? fj : x - 1u;
            pout[dest] = el[ch];
        }
        const unsigned long long um = __ballot(is_u);
        if (um) {
            const int ul = __builtin_ctzll(um);
            ue = __shfl((unsigned)el[ch], ul);
            up = __shfl(p ? 1u : 0u, ul);
        }
        run += (unsigned)__popcll(masks[ch]);
    }
    vd_wave_lds_sync();
    out_ttot = act + ttot; out_ue = ue; out_up = up;
}

__device__ __forceinline__ void wave_shuffle_any(WaveLds& L, int src, unsigned s, unsigned n, int axis, float pos, unsigned band, unsigned act,
                                                 unsigned& tt, unsigned& ue, unsigned& up) {
    const unsigned nw = n - act;                     // the chunks compiled in follow the window, not the node
    if (nw <= 64u) wave_shuffle<1>(L, src, s, n, axis, pos, band, act, tt, ue, up);
    else if (kChunks >= 2 && nw <= 128u) wave_shuffle<(kChunks >= 2 ? 2 : 1)>(L, src, s, n, axis, pos, band, act, tt, ue, up);
    else if (kChunks >= 4 && nw <= 256u) wave_shuffle<(kChunks >= 4 ? 4 : 1)>(L, src, s, n, axis, pos, band, act, tt, ue, up);
    else wave_shuffle<kChunks>(L, src, s, n, axis, pos, band, act, tt, ue, up);
}

constexpr int kLaneMax = VD_LANE_MAX;   // a subtree root of at most this many prims starts in the group path (8 lanes)
constexpr int kQueue = kSmallMax / 4;   // a BFS level holds at most N/4 splittable nodes
// List capacities that cannot overflow: the maximal nodes of a class are disjoint, and below one of them the class
// nodes form a single chain (two children of the same class do not fit, except 8 -> 4 + 4):
//   4..8 prims: <= 5 nodes per 8 prims -> 5/8 N;  9..16: <= 8 per 9 prims -> 8/9 N;  17..32: <= 16 per 17 -> 16/17 N
constexpr int kSmallList = kSmallMax, kC16 = kSmallMax, kC32 = kSmallMax;

struct WaveQueues {                      // entry = node | start << 10 | count << 20 (never 0: count > 3)
    unsigned work[2 * kQueue];           // wave-wide nodes handed to other waves: append-only, 0 = not written yet
    unsigned small[kSmallList];          // nodes of 4..8 prims: eight per wave
    unsigned pool;                       // next free node pair
    unsigned head, tail;                 // work[head .. tail) is waiting for a wave
    int pending;                         // wave-wide nodes queued or being split
    unsigned s_head;                     // small[s_head .. n_small) is waiting for a lane group
    int s_pending;                       // nodes of <= 32 prims queued or being split
    unsigned c16[kC16], c32[kC32];       // nodes of 9..16 / 17..32 prims: four / two per wave
    unsigned r_cnt[kChunks];             // block-wide shuffle of the subtree root: trues per 64-position chunk
    unsigned r_src, r_axis, r_active, r_ue, r_up;
    float r_pos;
    unsigned h16, t16, h32, t32;
    unsigned n_small, root_left, bad;
#ifdef VD_TUNING
    unsigned cls[48];   // wave-cycles by node size class and step (blas_small_kernel: cls_prof)
#endif
#ifdef VD_PROF_SEL
    unsigned prof[8];   // 0 root node done, 1 wave loop done, 2 lists drained, 3 renumber scan done, 4 group batches, 5 listed nodes, 6 wide nodes
#endif
};

// The same shuffle for the subtree root [0, n), by all waves of the workgroup: wave w owns the 64-position chunks
// w, w + kSubWaves, ...  The root is split while nothing else can run (it is 20 % of a subtree's time on one wave),
// so the other waves serve as helpers: the leader (wave 0, inside its ordinary node body) posts {src, axis, pos} and
// everyone meets at four workgroup barriers per shuffle; r_active = 0 releases the helpers.
constexpr int kOwnChunks = (kChunks + kSubWaves - 1) / kSubWaves;
__device__ __forceinline__ void block_shuffle_part(WaveLds& L, WaveQueues& Q, unsigned n, unsigned wave) {
    const unsigned lane = vd_lane();
    const int src = (int)Q.r_src;
    const unsigned short* pin = L.perm[src];
    unsigned short* pout = L.perm[src ^ 1];
    const float* cen = L.cent[Q.r_axis];
    const float pos = Q.r_pos;
    unsigned long long masks[kOwnChunks];
    unsigned short el[kOwnChunks];
#pragma unroll
    for (int k = 0; k < kOwnChunks; ++k) {
        const unsigned ch = wave + (unsigned)k * kSubWaves, x = ch * 64u + lane;
        bool p = false;
        el[k] = 0;
        if (ch < (unsigned)kChunks && x < n) { el[k] = pin[x]; p = cen[el[k]] < pos; }
        masks[k] = __ballot(p);
        if (lane == 0 && ch < (unsigned)kChunks) Q.r_cnt[ch] = (unsigned)__popcll(masks[k]);
    }
    __syncthreads();                                                     // B1: chunk counts are in
    unsigned ttot = 0, run[kOwnChunks];
#pragma unroll
    for (int k = 0; k < kOwnChunks; ++k) run[k] = 0;
#pragma unroll
    for (int ch = 0; ch < kChunks; ++ch) {
        const unsigned cnt = Q.r_cnt[ch];
#pragma unroll
        for (int k = 0; k < kOwnChunks; ++k) if ((unsigned)ch < wave + (unsigned)k * kSubWaves) run[k] += cnt;
        ttot += cnt;
    }
    const unsigned ftot = n - ttot;
#pragma unroll
    for (int k = 0; k < kOwnChunks; ++k) {
        const unsigned ch = wave + (unsigned)k * kSubWaves, x = ch * 64u + lane;
        if (ch < (unsigned)kChunks && x < n) {
            const bool p = (masks[k] >> lane) & 1ull;
            const unsigned tl = run[k] + vd_mbcnt(masks[k]);
            if (p) L.truepos[ttot - tl] = (unsigned short)x;             // (T+1)-th true from the right
            else L.falsepos[x - tl + 1u] = (unsigned short)x;            // (F+1)-th false from the left
        }
    }
    __syncthreads();                                                     // B2: rank -> position tables are complete
#pragma unroll
    for (int k = 0; k < kOwnChunks; ++k) {
        const unsigned ch = wave + (unsigned)k * kSubWaves, x = ch * 64u + lane;
        bool is_u = false, p = false;
        if (ch < (unsigned)kChunks && x < n) {
            p = (masks[k] >> lane) & 1ull;
            const unsigned tl = run[k] + vd_mbcnt(masks[k]);
            const unsigned F = x - tl, T = ttot - tl - (p ? 1u : 0u);
            const int tF = F == 0u ? (int)n : (F <= ttot ? (int)L.truepos[F] : -1);
            const bool left = (int)x < tF;
            const unsigned fj = (T + 1u <= ftot) ? (unsigned)L.falsepos[T + 1u] : n;
            const unsigned fetch = left ? x + n - (unsigned)tF : (n - 1u - x) + fj + 1u;
            is_u = fetch == n - 1u;
            unsigned dest;
            if (is_u) dest = ttot - (p ? 1u : 0u);
            else if (left) dest = p ? x : (unsigned)tF - 1u;
            else dest = p ? fj : x - 1u;
            pout[dest] = el[k];
            if (is_u) { Q.r_ue = el[k]; Q.r_up = p ? 1u : 0u; }
        }
    }
    __syncthreads();                                                     // B3: the new arrangement and u are in
}
// leader side (wave 0): same out parameters as wave_shuffle
__device__ __forceinline__ void block_shuffle_leader(WaveLds& L, WaveQueues& Q, int src, unsigned n, int axis, float pos,
                                                     unsigned& tt, unsigned& ue, unsigned& up) {
    if (vd_lane() == 0) { Q.r_src = (unsigned)src; Q.r_axis = (unsigned)axis; Q.r_pos = pos; Q.r_active = 1u; }
    __syncthreads();                                                     // B0: the command is posted
    block_shuffle_part(L, Q, n, 0u);
    unsigned t = 0;
#pragma unroll
    for (int ch = 0; ch < kChunks; ++ch) t += Q.r_cnt[ch];
    tt = t; ue = Q.r_ue; up = Q.r_up;
}
__device__ __forceinline__ void block_shuffle_helpers(WaveLds& L, WaveQueues& Q, unsigned n, unsigned wave) {
    for (;;) {
        __syncthreads();                                                 // B0
        if (Q.r_active == 0u) break;
        block_shuffle_part(L, Q, n, wave);
    }
}

// Phase B is a list of independent roots of 4..kSmallMax triangles, a workgroup each, and a root's time grows with its
// size: dispatched largest first, the last workgroups to start are the short ones and the kernel's tail is short
// (longest-processing-time order).  One workgroup: histogram of the counts, suffix sums, scatter.
__global__ __launch_bounds__(1024) void b_order_kernel(const SmallRoot* __restrict__ roots, const unsigned* __restrict__ n_roots_p,
                                                       unsigned* __restrict__ order) {
    __shared__ unsigned s_bin[kSmallMax + 2];
    const unsigned n = *n_roots_p, tid = threadIdx.x;
    for (unsigned b = tid; b < (unsigned)kSmallMax + 2u; b += 1024u) s_bin[b] = 0u;
    __syncthreads();
    for (unsigned i = tid; i < n; i += 1024u) atomicAdd(&s_bin[min(roots[i].count, (unsigned)kSmallMax + 1u)], 1u);
    __syncthreads();
    if (tid == 0) {           // first slot of each count, counting down from the largest
        unsigned run = 0;
        for (int b = kSmallMax + 1; b >= 0; --b) { const unsigned c = s_bin[b]; s_bin[b] = run; run += c; }
    }
    __syncthreads();
    for (unsigned i = tid; i < n; i += 1024u) order[atomicAdd(&s_bin[min(roots[i].count, (unsigned)kSmallMax + 1u)], 1u)] = i;
}

// waves per SIMD the small kernel's register budget is set for: 4: 28.7 ms per build, 5: 27.9, 6: 27.2 (the default until the
// end of round 4), 7: 26.5-27.0, 8: 26.8 (profiles/r04_blas_item_isa.txt, seventh A/B) - the kernel is VALU-issue-bound and
// more resident waves keep the issue slots fuller than the spilled registers cost
#ifndef VD_SMALL_OCC
#define VD_SMALL_OCC 8
#endif
__global__ __launch_bounds__(64 * kSubWaves, VD_SMALL_OCC)
void blas_small_kernel(const SmallRoot* __restrict__ roots, const unsigned* __restrict__ n_roots_p,
                       const unsigned* __restrict__ ids32, ArrSet set0, ArrSet set1,
                       TmpNode* __restrict__ subnodes, unsigned short* __restrict__ submap,
                       unsigned* __restrict__ sub_interior, unsigned* __restrict__ final_ids, unsigned* __restrict__ err,
                       unsigned* __restrict__ dbg_cycles, const unsigned* __restrict__ order, unsigned long long* __restrict__ cls_prof) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    // -DVD_TUNING only (vd_debug_blas_small_classes -> profiles/r06_blas_small_classes.log): wave-cycles by node size class
    // (0: <= 32, the lane-group batches; 1: 33..64; 2: 65..128; 3: 129..256; 4: 257..512) and step (0 setup: centroid bounds,
    // planes, predicate bits; 1 the 21 trials; 2 cost evaluation; 3 the final shuffle; 4 children + hand-over; 5 nodes; 6: batches
    // for class 0), cls_prof[class * 8 + step]; row 5: [40] all waves' lifetimes, [41] idle polling, [42] load, [43] drain + renumber
#ifdef VD_TUNING
#define VD_CLS_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define VD_CLS_ADD(slot, cyc) do { if (vd_lane() == 0u) atomicAdd(&Q.cls[slot], (unsigned)(cyc)); } while (0)      /* in LDS; one flush per workgroup */
#else
#define VD_CLS_T(var) do { } while (0)
#define VD_CLS_ADD(slot, cyc) do { } while (0)
#endif
    const unsigned tid = threadIdx.x, lane = vd_lane();
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds& L = *reinterpret_cast<WaveLds*>(smem);
    WaveQueues& Q = *reinterpret_cast<WaveQueues*>(smem + sizeof(WaveLds));
    WaveScratch& W = L.w[wave];
    if (blockIdx.x >= *n_roots_p) return;
    const unsigned root_i = order ? order[blockIdx.x] : blockIdx.x;     // largest roots first (b_order_kernel)
    const SmallRoot root = roots[root_i];
    const unsigned base = root.start, N = root.count;
    // per-triangle data: position order in the set the segment's last phase-A level wrote (pad & 1), through ids32 when
    // the mid tier permuted the positions again (pad & 2)
    const f32x4* __restrict__ cent = (root.pad & 1u) ? set1.cent : set0.cent;
    const TriBox* __restrict__ boxes = (root.pad & 1u) ? set1.boxes : set0.boxes;
    TmpNode* nodes = subnodes + 2u * (size_t)base;   // disjoint region per root: < 2*N nodes, creation order
    unsigned short* nmap = submap + 2u * (size_t)base;

    for (unsigned x = tid; x < N; x += 64u * kSubWaves) {
        const unsigned id = (root.pad & 2u) ? ids32[base + x] : base + x;
#if VD_BOX_IN_LDS
        const TriBox bx = boxes[id];
#endif
        const f32x4 c4 = cent[id];
        const float ce[3] = {c4.x, c4.y, c4.z};
        L.gid[x] = id;
        L.uflag[x] = 0;
        L.perm[0][x] = (unsigned short)x;
        L.perm[1][x] = (unsigned short)x;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            L.cent[k][x] = ce[k];
#if VD_BOX_IN_LDS
            L.box[k][x] = vd_key(bx.mn[k]);
            L.box[3 + k][x] = vd_key(bx.mx[k]);
#endif
        }
    }
    // Wave-wide nodes ping-pong their own segment between the two perm buffers an even number of
    // times (22) and mirror the result, so both buffers agree outside the node being processed.
    // The subtrees under a root are deep and thin (SAH splits are unbalanced): walked level by level, most levels hold
    // fewer nodes than the workgroup has waves and still cost a full node pass of 22 dependent shuffles plus two
    // barriers.  So there are no levels: a wave that has split a node goes straight on with one child and hands the
    // other to whichever wave is idle (append-only list in LDS); `pending` counts the wave-wide nodes that are queued
    // or being split, and a wave leaves when it finds nothing to take and pending is 0.  Nodes of <= 32 prims go to
    // per-size lists and are split several per wave (try_group), by whichever wave has nothing wave-wide to do.
    for (unsigned x = tid; x < 2u * (unsigned)kQueue; x += 64u * kSubWaves) Q.work[x] = 0u;
    for (unsigned x = tid; x < (unsigned)kSmallList; x += 64u * kSubWaves) Q.small[x] = 0u;
    for (unsigned x = tid; x < (unsigned)kC16; x += 64u * kSubWaves) Q.c16[x] = 0u;
    for (unsigned x = tid; x < (unsigned)kC32; x += 64u * kSubWaves) Q.c32[x] = 0u;
    const unsigned root_ent = 1023u | (N << 20);        // the root entry: node field unused; s = 0, n = N (no child has n = N)
    if (tid == 0) {
        Q.pool = 0; Q.n_small = 0; Q.root_left = kNone; Q.bad = 0; Q.head = 0; Q.tail = 0; Q.pending = 0; Q.s_head = 0;
        Q.s_pending = 0; Q.h16 = 0; Q.t16 = 0; Q.h32 = 0; Q.t32 = 0;
#ifdef VD_TUNING
        for (int q = 0; q < 48; ++q) Q.cls[q] = 0u;
#endif
#ifdef VD_PROF_SEL
        for (int q = 0; q < 8; ++q) Q.prof[q] = 0;
#endif
        if (N > (unsigned)kLaneMax) Q.pending = 1; else { Q.small[0] = root_ent; Q.n_small = 1; Q.s_pending = 1; }
    }
    __syncthreads();
    VD_CLS_T(t_loaded);
    // ---------------- nodes of <= 32 prims: several at a time per wave, one per lane group ----------------
    // A wave takes eight nodes of <= 8 prims, four of 9..16 or two of 17..32 and gives each a group of 8 / 16 / 32
    // lanes: one position per lane, the arrangement kept in registers across the trials and moved through the LDS
    // crossbar exactly as the wave-wide path for <= 64 prims does; the costs of the 21 candidates are evaluated
    // afterwards from the recorded `u` elements, with group reductions on the VALU (DPP).  (One node per wave made the
    // ~60 nodes of 9..175 prims under a root 58 % of this kernel, and one whole sub-subtree of <= 8 prims per lane -
    // the literal loop in registers, a chain 8 -> 7 -> 6 -> 5 -> 4 being five splits of 22 trials in a row - 12 %.)
    // Children of 4..32 prims go to the list of their class; `s_pending` counts the listed nodes not yet split.
    auto push_class = [&](unsigned e2, unsigned cn) {                              // called by one lane
        atomicAdd(&Q.s_pending, 1);
        if (cn <= 8u) { const unsigned k = atomicAdd(&Q.n_small, 1u); if (k < (unsigned)kSmallList) Q.small[k] = e2; else Q.bad = 2; }
        else if (cn <= 16u) { const unsigned k = atomicAdd(&Q.t16, 1u); if (k < (unsigned)kC16) Q.c16[k] = e2; else Q.bad = 2; }
        else { const unsigned k = atomicAdd(&Q.t32, 1u); if (k < (unsigned)kC32) Q.c32[k] = e2; else Q.bad = 2; }
    };
    auto try_group = [&]() -> bool {
        if (lane == 0) {
            unsigned first = 0, count = 0, cls = 0;
            for (int c3 = 2; c3 >= 0 && count == 0u; --c3) {                       // the larger nodes first
                unsigned* hp = c3 == 2 ? &Q.h32 : (c3 == 1 ? &Q.h16 : &Q.s_head);
                unsigned* tp = c3 == 2 ? &Q.t32 : (c3 == 1 ? &Q.t16 : &Q.n_small);
                for (;;) {
                    const unsigned h = *(volatile unsigned*)hp, t = *(volatile unsigned*)tp;
                    if (h >= t) break;
                    const unsigned k = min(8u >> c3, t - h);
                    if (atomicCAS(hp, h, h + k) == h) { first = h; count = k; cls = (unsigned)c3; break; }
                }
            }
            W.g_first = first; W.g_count = count | (cls << 8);
        }
        vd_wave_lds_sync();
        const unsigned first = W.g_first, count = W.g_count & 255u, cls = W.g_count >> 8;
        vd_wave_lds_sync();
        if (count == 0u) return false;
        VD_CLS_T(tg0);
        VD_MARK("setup_begin");
#ifdef VD_PROF_SEL
        if (lane == 0) { atomicAdd(&Q.prof[4], 1u); atomicAdd(&Q.prof[5], count); }
#endif
        const unsigned gw = 8u << cls, gmask = gw == 32u ? 0xffffffffu : (1u << gw) - 1u;
        const unsigned* list = cls == 2u ? Q.c32 : (cls == 1u ? Q.c16 : Q.small);
        const unsigned gl = lane & (gw - 1u), gb = lane & ~(gw - 1u), grp = lane >> (3u + cls);
            unsigned ent = 0u;
            if (grp < count) {
                unsigned spins = 0;
                while ((ent = *(volatile unsigned*)&list[first + grp]) == 0u && ++spins < (1u << 22)) {}   // claimed by its pusher, written next
            }
            const bool have = ent != 0u;
            const unsigned node_id = ent & 1023u, s = (ent >> 10) & 1023u, n = have ? ent >> 20 : 0u;
            const bool valid = gl < n;
            unsigned el = valid ? (unsigned)L.perm[0][s + gl] : 0u;
            float cx = 0.0f, cy = 0.0f, cz = 0.0f;
            int bk[6] = {kBig, kBig, kBig, -kBig - 1, -kBig - 1, -kBig - 1};
            if (valid) { cx = L.cent[0][el]; cy = L.cent[1][el]; cz = L.cent[2][el]; }
            // centroid bounds of the node (blas.rs:139-143)
            float cbmin[3], cbmax[3];
            {
                int kmn[3], kmx[3];
                const float ce3[3] = {cx, cy, cz};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    kmn[k] = group_min_i(valid ? vd_key_lo(ce3[k]) : kBig, gw); kmx[k] = group_max_i(valid ? vd_key_hi(ce3[k]) : -kBig - 1, gw);
                    cbmin[k] = box_lo(kmn[k]); cbmax[k] = box_hi(kmx[k]);
                }
            }
            // the 21 predicates of an element do not depend on where it sits: one bit each, computed once, and the word
            // travels with the element id (two crossbar moves per trial instead of four)
            unsigned pb = 0u;
            if (valid) {
#pragma unroll
                for (int c = 0; c < kCand; ++c) {
                    const float ce = c < 7 ? cx : (c < 14 ? cy : cz);
                    pb |= (ce < cand_pos(cbmin, cbmax, c) ? 1u : 0u) << c;
                }
            }
            VD_MARK("setup_end");
            VD_CLS_T(tg1);
            // one trial (blas.rs:168-182 in closed form, as the wave-wide register path): moves el / pb
            auto trial = [&](int c, bool record) {
                VD_MARK("trial_begin");
                const bool p = valid && ((pb >> c) & 1u);
                const unsigned gm = (unsigned)(__ballot(p) >> gb) & gmask;
                const unsigned ttot = (unsigned)__popc(gm), ftot = n - ttot, tl = (unsigned)__popc(gm & ((1u << gl) - 1u)), x = gl;
                // both rank -> position tables in ONE crossbar move: truepos[k + 1] in lane k < ttot, falsepos[k + 1] in lane
                // ttot + k (ttot + ftot = n <= gw); lanes without an element aim at lane gw - 1, which is free when n < gw
                const int tab = __builtin_amdgcn_ds_permute((int)((gb + (p ? ttot - tl - 1u : (valid ? ttot + x - tl : gw - 1u))) << 2), (int)x);
                const unsigned F = x - tl, T = ttot - tl - (p ? 1u : 0u);
                const int tp_at = __builtin_amdgcn_ds_bpermute((int)((gb + ((F - 1u) & (gw - 1u))) << 2), tab);
                const int fp_at = __builtin_amdgcn_ds_bpermute((int)((gb + ((ttot + T) & (gw - 1u))) << 2), tab);
                unsigned dest = gl;
                bool is_u = false;
                if (valid) {
                    const int tF = F == 0u ? (int)n : (F <= ttot ? tp_at : -1);
                    const bool left = (int)x < tF;
                    const unsigned fj = (T + 1u <= ftot) ? (unsigned)fp_at : n;
                    const unsigned fetch = left ? x + n - (unsigned)tF : (n - 1u - x) + fj + 1u;
                    is_u = fetch == n - 1u;
                    if (is_u) dest = ttot - (p ? 1u : 0u);
                    else if (left) dest = p ? x : (unsigned)tF - 1u;
                    else dest = p ? fj : x - 1u;
                }
                if (record && is_u) {                  // the one never-examined element of the node writes its own record
                    W.g_ue[grp][c] = (unsigned short)el; W.g_tt[grp][c] = (unsigned char)(ttot | ((p ? 1u : 0u) << 7));
                }
                const int da = (int)((gb + dest) << 2);
                const unsigned ep = (unsigned)__builtin_amdgcn_ds_permute(da, (int)(el | (pb << 10)));   // el < 1024, 21 predicate bits
                el = ep & 1023u; pb = ep >> 10;
                VD_MARK("trial_end");
            };
            for (int c = 0; c < kCand; ++c) trial(c, true);                        // blas.rs:144-147
            vd_wave_lds_sync();
            VD_CLS_T(tg2);
            VD_MARK("eval_begin");
            // evaluate (blas.rs:149-161): left = examined trues = {e : p_c(e) and e != u_c}, right = the rest (incl. u_c)
            if (valid) {
                { const BoxKeys bb = box_keys(L, boxes, el);
#pragma unroll
                  for (int q = 0; q < 6; ++q) bk[q] = bb.k[q]; }
            }
            // One LANE per (node, candidate) pair - 21 x count pairs, up to three passes of 64 - walks the node's elements,
            // fetching each element's box keys and predicate word from the lane that holds it (ds_bpermute: the LDS crossbar,
            // not the VALU this kernel is bound by), and the cost is worked out once per pair.  (Candidate by candidate with
            // twelve group reductions each, every lane repeating the cost arithmetic, this evaluation was a third of phase B.)
            unsigned* g_hi = reinterpret_cast<unsigned*>(&W.bin_min[0][0][0]);     // per group: smallest cost key, then the first
            unsigned* g_lo = g_hi + 8;                                             // candidate that has it (the bins are idle here)
            if (lane < 16u) g_hi[lane] = 0xffffffffu;
            vd_wave_lds_sync();
            const unsigned word = el | (pb << 10);
            const unsigned n_pairs = count * (unsigned)kCand;
            unsigned hi3[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, lo3[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, g3[3] = {0u, 0u, 0u};
#pragma unroll
            for (int pass = 0; pass < 3; ++pass) {
                if ((unsigned)pass * 64u >= n_pairs) break;                          // wave-uniform
                const unsigned pidx = (unsigned)pass * 64u + lane;
                const bool pv = pidx < n_pairs;
                const unsigned g = pv ? pidx / (unsigned)kCand : 0u, c = pv ? pidx - g * (unsigned)kCand : 0u;
                const unsigned src0 = g << (3u + cls);
                const unsigned n_g = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src0 << 2), (int)n);
                const unsigned ue = W.g_ue[g][c], tu = W.g_tt[g][c];
                int lk[6] = {kBig, kBig, kBig, -kBig - 1, -kBig - 1, -kBig - 1}, rk[6] = {kBig, kBig, kBig, -kBig - 1, -kBig - 1, -kBig - 1};
                for (unsigned i = 0; i < gw; ++i) {                                  // every lane takes part in the moves
                    VD_MARK("eval_elem_begin");
                    const int from = (int)((src0 + i) << 2);
                    const unsigned w = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)word);
                    int k[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) k[q] = __builtin_amdgcn_ds_bpermute(from, bk[q]);
                    const bool vi = i < n_g;
                    const bool inl = vi && ((w >> (10u + c)) & 1u) && (w & 1023u) != ue, inr = vi && !inl;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        lk[q] = min(lk[q], inl ? k[q] : kBig); lk[3 + q] = max(lk[3 + q], inl ? k[3 + q] : -kBig - 1);
                        rk[q] = min(rk[q], inr ? k[q] : kBig); rk[3 + q] = max(rk[3 + q], inr ? k[3 + q] : -kBig - 1);
                    }
                    VD_MARK("eval_elem_end");
                }
                VD_MARK("cost_begin");
                const unsigned n1 = (tu & 127u) - (tu >> 7);
                const float a1 = vd_area(box_hi(lk[3]) - box_lo(lk[0]), box_hi(lk[4]) - box_lo(lk[1]), box_hi(lk[5]) - box_lo(lk[2]));
                const float a2 = vd_area(box_hi(rk[3]) - box_lo(rk[0]), box_hi(rk[4]) - box_lo(rk[1]), box_hi(rk[5]) - box_lo(rk[2]));
                const vd_u64 kc = pv ? cost_key(a1 * (float)n1 + a2 * (float)(n_g - n1), c) : ~0ull;
                hi3[pass] = (unsigned)(kc >> 32); lo3[pass] = (unsigned)kc; g3[pass] = g;
                if (pv) atomicMin(&g_hi[g], hi3[pass]);
                VD_MARK("cost_end");
            }
            vd_wave_lds_sync();
#pragma unroll
            for (int pass = 0; pass < 3; ++pass)
                if ((unsigned)pass * 64u + lane < n_pairs && hi3[pass] == g_hi[g3[pass]]) atomicMin(&g_lo[g3[pass]], lo3[pass]);
            vd_wave_lds_sync();
            const vd_u64 key = have ? (((vd_u64)g_hi[grp] << 32) | g_lo[grp]) : ~0ull;
            vd_wave_lds_sync();
            const bool rejected = have && key == ~0ull;
            const int best = rejected || !have ? 0 : (int)(unsigned)key;
            const unsigned tb = W.g_tt[grp][best];
            const unsigned Lst = (tb & 127u) - (tb >> 7);                            // stale optimal_pivot (blas.rs:159,165)
            VD_MARK("eval_end");
            VD_CLS_T(tg3);
            trial(best, false);                                                     // blas.rs:164
            VD_CLS_T(tg4);
            VD_MARK("finish_begin");
            if (valid) { L.perm[0][s + gl] = (unsigned short)el; L.perm[1][s + gl] = (unsigned short)el; }
            if (valid) {
                { const BoxKeys bb = box_keys(L, boxes, el);
#pragma unroll
                  for (int q = 0; q < 6; ++q) bk[q] = bb.k[q]; }
            }
            int ck[12];                                                             // children boxes (blas.rs:115-123)
            {
                const bool inl = valid && gl < Lst, inr = valid && !inl;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    ck[q] = inl ? bk[q] : kBig; ck[3 + q] = inl ? bk[3 + q] : -kBig - 1;
                    ck[6 + q] = inr ? bk[q] : kBig; ck[9 + q] = inr ? bk[3 + q] : -kBig - 1;
                }
#pragma unroll
                for (int i2 = 0; i2 < 12; ++i2) ck[i2] = (i2 % 6) < 3 ? group_min_i(ck[i2], gw) : group_max_i(ck[i2], gw);
            }
            if (have && gl == 0u) {
                if (rejected) { Q.bad = 1; atomicSub(&Q.s_pending, 1); }
                else {
                    const unsigned pair = atomicAdd(&Q.pool, 2u);
                    TmpNode ln, rn;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        ln.mn[q] = box_lo(ck[q]); ln.mx[q] = box_hi(ck[3 + q]);
                        rn.mn[q] = box_lo(ck[6 + q]); rn.mx[q] = box_hi(ck[9 + q]);
                    }
                    ln.left_first = base + s; ln.count = Lst;
                    rn.left_first = base + s + Lst; rn.count = n - Lst;
                    nodes[pair] = ln; nodes[pair + 1] = rn;
                    nmap[pair] = (unsigned short)s; nmap[pair + 1u] = (unsigned short)(s + Lst);   // start positions: renumber keys
                    if (ent == root_ent) Q.root_left = pair;                        // N <= kLaneMax: the subtree root itself
                    else { nodes[node_id].left_first = pair; nodes[node_id].count = n | kInteriorMark; }
                    if (Lst > 3u) push_class(pair | (s << 10) | (Lst << 20), Lst);
                    if (n - Lst > 3u) push_class((pair + 1u) | ((s + Lst) << 10) | ((n - Lst) << 20), n - Lst);
                    atomicSub(&Q.s_pending, 1);                                     // after the children were counted
                }
            }
        VD_MARK("finish_end");
#ifdef VD_TUNING
        { VD_CLS_T(tg5); VD_CLS_ADD(0, tg1 - tg0); VD_CLS_ADD(1, tg2 - tg1); VD_CLS_ADD(2, tg3 - tg2); VD_CLS_ADD(3, tg4 - tg3); VD_CLS_ADD(4, tg5 - tg4);
          VD_CLS_ADD(5, count); VD_CLS_ADD(6, 1); }
#endif
        return true;
    };

    unsigned next_ent = (wave == 0u && N > (unsigned)kLaneMax) ? root_ent : 0u;   // wave-uniform; 0 = take one from the list
    unsigned idle_polls = 0;
    const bool root_by_block = N > 128u;                 // the subtree root is shuffled by all waves (block_shuffle_*)
    if (root_by_block && wave != 0u) block_shuffle_helpers(L, Q, N, wave);

    for (;;) {
        {
            unsigned ent = next_ent;
            if (ent == 0u) {
                unsigned got = 0u;
                if (lane == 0) {
                    for (;;) {
                        const unsigned h = *(volatile unsigned*)&Q.head, t = *(volatile unsigned*)&Q.tail;
                        if (h >= t) break;
                        if (atomicCAS(&Q.head, h, h + 1u) == h) {
                            unsigned spins = 0;
                            while ((got = *(volatile unsigned*)&Q.work[h]) == 0u && ++spins < (1u << 22)) {}   // claimed by its pusher, written next
                            break;
                        }
                    }
                }
                ent = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
                if (ent == 0u) {
                    if (try_group()) { idle_polls = 0; continue; }                  // nothing wave-wide to take: split listed nodes meanwhile
                    if (*(volatile int*)&Q.pending <= 0 || *(volatile unsigned*)&Q.bad) break;
                    __builtin_amdgcn_s_sleep(4);
                    if (++idle_polls > (1u << 24)) { if (lane == 0) Q.bad = 2; break; }   // never expected: an error, not a hang
                    continue;
                }
            }
            next_ent = 0u;
            const unsigned node_id = ent & 1023u, s = (ent >> 10) & 1023u, n = ent >> 20;
            const bool is_root = ent == root_ent;
            VD_CLS_T(tw0);
            int cur = 0;
            int kmn[3] = {kBig, kBig, kBig}, kmx[3] = {-kBig - 1, -kBig - 1, -kBig - 1};
            for (unsigned x = lane; x < n; x += 64u) {
                const unsigned e = L.perm[cur][s + x];
#pragma unroll
                for (int k = 0; k < 3; ++k) { const float cv = L.cent[k][e]; kmn[k] = min(kmn[k], vd_key_lo(cv)); kmx[k] = max(kmx[k], vd_key_hi(cv)); }
            }
            float cbmin[3], cbmax[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { cbmin[k] = box_lo(wave_min_i(kmn[k])); cbmax[k] = box_hi(wave_max_i(kmx[k])); }
            if (lane < (unsigned)kCand) W.pos[lane] = cand_pos(cbmin, cbmax, (int)lane);
            vd_wave_lds_sync();
            VD_CLS_T(tw1);
            if (n <= 64u) {
                // One position per lane: the arrangement (element id + the element's 21 predicate bits) stays in registers
                // across the 21 trials and moves through the LDS crossbar (ds_permute / ds_bpermute), the rank ->
                // position tables likewise.  A trial is then three dependent crossbar hops (five moves in all) instead of
                // five dependent LDS accesses.
                const bool valid = lane < n;
                unsigned el = valid ? (unsigned)L.perm[cur][s + lane] : 0u;
                unsigned pb = 0u;                      // the element's 21 predicates: they travel with it
                if (valid) {
                    const float cx = L.cent[0][el], cy = L.cent[1][el], cz = L.cent[2][el];
#pragma unroll
                    for (int c = 0; c < kCand; ++c) pb |= ((c < 7 ? cx : (c < 14 ? cy : cz)) < W.pos[c] ? 1u : 0u) << c;
                }
                for (int c = 0; c < kCand; ++c) {                                     // blas.rs:144-147
                    const bool p = valid && ((pb >> c) & 1u);
                    const unsigned long long mask = __ballot(p);
                    const unsigned ttot = (unsigned)__popcll(mask), ftot = n - ttot, tl = vd_mbcnt(mask), x = lane;
                    // both rank -> position tables in one crossbar move: lane k < ttot receives truepos[k + 1], lane ttot + k
                    // falsepos[k + 1] (0-based here); lanes without an element aim at lane 63, which is free when n < 64
                    const int tab = __builtin_amdgcn_ds_permute((int)((p ? ttot - tl - 1u : (valid ? ttot + x - tl : 63u)) << 2), (int)x);
                    const unsigned F = x - tl, T = ttot - tl - (p ? 1u : 0u);
                    const int tp_at = __builtin_amdgcn_ds_bpermute((int)(((F - 1u) & 63u) << 2), tab);
                    const int fp_at = __builtin_amdgcn_ds_bpermute((int)(((ttot + T) & 63u) << 2), tab);
                    unsigned dest = lane;
                    bool is_u = false;
                    if (valid) {
                        const int tF = F == 0u ? (int)n : (F <= ttot ? tp_at : -1);
                        const bool left = (int)x < tF;
                        const unsigned fj = (T + 1u <= ftot) ? (unsigned)fp_at : n;
                        const unsigned fetch = left ? x + n - (unsigned)tF : (n - 1u - x) + fj + 1u;
                        is_u = fetch == n - 1u;
                        if (is_u) dest = ttot - (p ? 1u : 0u);
                        else if (left) dest = p ? x : (unsigned)tF - 1u;
                        else dest = p ? fj : x - 1u;
                    }
                    if (is_u) { W.u_e[c] = (unsigned short)el; W.u_p[c] = p ? 1 : 0; W.ttot[c] = ttot; }   // the never-examined element records itself
                    const int da = (int)(dest << 2);
                    const unsigned ep = (unsigned)__builtin_amdgcn_ds_permute(da, (int)(el | (pb << 10)));   // el < 1024, 21 predicate bits
                    el = ep & 1023u; pb = ep >> 10;
                }
                cur ^= 1;                                                             // 21 flips
                if (valid) L.perm[cur][s + lane] = (unsigned short)el;
            } else {
                unsigned w_band = 0, w_act = 0;                                       // the window of the wave's own shuffles (wave_shuffle)
                for (int c = 0; c < kCand; ++c) {                                     // blas.rs:144-147
                    unsigned tt, ue, up;
#ifdef VD_NO_B_WINDOWS
                    w_band = 0; w_act = 0;
#else
                    if (c % 7 == 0) { w_band = 0; w_act = 0; }
#endif
                    if (is_root && root_by_block) block_shuffle_leader(L, Q, cur, n, c / 7, W.pos[c], tt, ue, up);
                    else wave_shuffle_any(L, cur, s, n, c / 7, W.pos[c], w_band, w_act, tt, ue, up);
                    w_band = w_act; w_act = tt - up;                                  // this trial's pivot: where the next one on the axis starts
                    cur ^= 1;
                    if (lane == 0) { W.u_e[c] = (unsigned short)ue; W.u_p[c] = (unsigned short)up; W.ttot[c] = tt; }
                }
            }
            vd_wave_lds_sync();
            VD_CLS_T(tw2);
            vd_u64 key = ~0ull;
            if (n > (unsigned)VD_BINEVAL_MIN) {
                // evaluate (blas.rs:149-161) from binned statistics, as phase A does: the cost of a trial depends on the
                // arrangement only through its never-examined element `u` (left = trues \ {u}), so one pass bins the
                // box keys of the non-`u` elements by (axis, number of planes not above the centroid) and each
                // candidate adds the <= 21 `u` elements back on the side its own predicate puts them.  (Walking the
                // node once per candidate, as below, is 56 % of a 350-prim node's time.)
                for (unsigned i = lane; i < 72u; i += 64u) { (&W.bin_min[0][0][0])[i] = kBig; (&W.bin_max[0][0][0])[i] = -kBig - 1; }
                if (lane < (unsigned)kCand) L.uflag[W.u_e[lane]] = 1;
                vd_wave_lds_sync();
                for (unsigned x = lane; x < n; x += 64u) {
                    const unsigned e = L.perm[cur][s + x];
                    if (L.uflag[e]) continue;
                    const BoxKeys ebk = box_keys(L, boxes, e);
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const float ce = L.cent[a][e];
                        int b = 0;
#pragma unroll
                        for (int k = 0; k < 7; ++k) b += !(ce < W.pos[a * 7 + k]);
#pragma unroll
                        for (int q = 0; q < 3; ++q) { atomicMin(&W.bin_min[a][b][q], ebk.k[q]); atomicMax(&W.bin_max[a][b][q], ebk.k[3 + q]); }
                    }
                }
                vd_wave_lds_sync();
                if (lane < (unsigned)kCand) {
                    const int c = (int)lane, a = c / 7, k = c % 7 + 1;
                    int tmn[3] = {kBig, kBig, kBig}, tmx[3] = {-kBig - 1, -kBig - 1, -kBig - 1};
                    int fmn[3] = {kBig, kBig, kBig}, fmx[3] = {-kBig - 1, -kBig - 1, -kBig - 1};
                    for (int b = 0; b < 8; ++b) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            const int lo = W.bin_min[a][b][q], hi = W.bin_max[a][b][q];
                            if (b < k) { tmn[q] = min(tmn[q], lo); tmx[q] = max(tmx[q], hi); }
                            else { fmn[q] = min(fmn[q], lo); fmx[q] = max(fmx[q], hi); }
                        }
                    }
                    const float pos = W.pos[c];
                    const unsigned own_u = W.u_e[c];
                    for (int j = 0; j < kCand; ++j) {
                        const unsigned e = W.u_e[j];
                        bool dup = false;
                        for (int i = 0; i < j; ++i) dup |= W.u_e[i] == e;
                        if (dup) continue;
                        const bool to_left = e != own_u && L.cent[a][e] < pos;   // left = examined trues; u itself goes right
                        const BoxKeys ubk = box_keys(L, boxes, e);
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            const int lo = ubk.k[q], hi = ubk.k[3 + q];
                            if (to_left) { tmn[q] = min(tmn[q], lo); tmx[q] = max(tmx[q], hi); }
                            else { fmn[q] = min(fmn[q], lo); fmx[q] = max(fmx[q], hi); }
                        }
                    }
                    const unsigned n1 = W.ttot[c] - W.u_p[c];
                    const float a1 = vd_area(box_hi(tmx[0]) - box_lo(tmn[0]), box_hi(tmx[1]) - box_lo(tmn[1]), box_hi(tmx[2]) - box_lo(tmn[2]));
                    const float a2 = vd_area(box_hi(fmx[0]) - box_lo(fmn[0]), box_hi(fmx[1]) - box_lo(fmn[1]), box_hi(fmx[2]) - box_lo(fmn[2]));
                    key = cost_key(a1 * (float)n1 + a2 * (float)(n - n1), (unsigned)c);
                }
                vd_wave_lds_sync();
                if (lane < (unsigned)kCand) L.uflag[W.u_e[lane]] = 0;
                vd_wave_lds_sync();
            } else {
            // evaluate (blas.rs:149-161): lane = 3*c + part owns a third of candidate c's elements;
            // left = examined trues = {e : p_c(e) and e != u_c}, right = the rest (incl. u_c)
            {
                const unsigned c = lane / 3u, part = lane - c * 3u;
                int k12[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) k12[i] = (i % 6) < 3 ? kBig : -kBig - 1;
                if (c < (unsigned)kCand) {
                    const float* cen = L.cent[c / 7u];
                    const float pos = W.pos[c];
                    const unsigned ue = W.u_e[c];
                    for (unsigned i = part; i < n; i += 3u) {
                        const unsigned e = L.perm[cur][s + i];
                        const int o = (cen[e] < pos && e != ue) ? 0 : 6;
                        { const BoxKeys kb = box_keys(L, boxes, e);
#pragma unroll
                        for (int q = 0; q < 3; ++q) { k12[o + q] = min(k12[o + q], kb.k[q]); k12[o + 3 + q] = max(k12[o + 3 + q], kb.k[3 + q]); } }
                    }
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const int a1 = __shfl(k12[i], (int)(c * 3u + 1u)), a2 = __shfl(k12[i], (int)(c * 3u + 2u));
                    k12[i] = (i % 6) < 3 ? min(k12[i], min(a1, a2)) : max(k12[i], max(a1, a2));
                }
                if (c < (unsigned)kCand && part == 0u) {
                    const unsigned n1 = W.ttot[c] - W.u_p[c];
                    const float a1 = vd_area(box_hi(k12[3]) - box_lo(k12[0]), box_hi(k12[4]) - box_lo(k12[1]), box_hi(k12[5]) - box_lo(k12[2]));
                    const float a2 = vd_area(box_hi(k12[9]) - box_lo(k12[6]), box_hi(k12[10]) - box_lo(k12[7]), box_hi(k12[11]) - box_lo(k12[8]));
                    key = cost_key(a1 * (float)n1 + a2 * (float)(n - n1), c);
                }
            }
            }
            key = wave_min_u64(key);
            if (key == ~0ull) {
                if (lane == 0) { Q.bad = 1; atomicSub(&Q.pending, 1); }
                if (is_root && root_by_block) { if (lane == 0) Q.r_active = 0u; __syncthreads(); }   // release the helpers
                continue;
            }
            const int best = (int)(unsigned)key;
            const unsigned Lst = W.ttot[best] - W.u_p[best];             // stale optimal_pivot (blas.rs:159,165)
            VD_CLS_T(tw3);
            {
                unsigned tt, ue, up;                                     // blas.rs:164
                if (is_root && root_by_block) {
                    block_shuffle_leader(L, Q, cur, n, best / 7, W.pos[best], tt, ue, up);
                    if (lane == 0) Q.r_active = 0u;
                    __syncthreads();                                     // B0 with r_active = 0: the helpers leave
                } else wave_shuffle_any(L, cur, s, n, best / 7, W.pos[best], 0u, 0u, tt, ue, up);
                cur ^= 1;                                                // 22 flips: back in buffer 0
            }
            VD_CLS_T(tw4);
            int k12[12];                                                 // children boxes (blas.rs:115-123)
#pragma unroll
            for (int i = 0; i < 12; ++i) k12[i] = (i % 6) < 3 ? kBig : -kBig - 1;
            for (unsigned x = lane; x < n; x += 64u) {
                const unsigned short e = L.perm[0][s + x];
                L.perm[1][s + x] = e;                                    // keep both buffers in step
                const int o = x < Lst ? 0 : 6;
                { const BoxKeys kb = box_keys(L, boxes, e);
#pragma unroll
                for (int q = 0; q < 3; ++q) { k12[o + q] = min(k12[o + q], kb.k[q]); k12[o + 3 + q] = max(k12[o + 3 + q], kb.k[3 + q]); } }
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) k12[i] = (i % 6) < 3 ? wave_min_i(k12[i]) : wave_max_i(k12[i]);
            if (lane == 0) {
                const unsigned pair = atomicAdd(&Q.pool, 2u);
                const unsigned cn[2] = {Lst, n - Lst}, cs[2] = {s, s + Lst};
                TmpNode ln, rn;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    ln.mn[q] = box_lo(k12[q]); ln.mx[q] = box_hi(k12[3 + q]);
                    rn.mn[q] = box_lo(k12[6 + q]); rn.mx[q] = box_hi(k12[9 + q]);
                }
                ln.left_first = base + cs[0]; ln.count = cn[0];          // leaf form; interior nodes are patched when split
                rn.left_first = base + cs[1]; rn.count = cn[1];
                nodes[pair] = ln; nodes[pair + 1] = rn;
                nmap[pair] = (unsigned short)cs[0]; nmap[pair + 1u] = (unsigned short)cs[1];   // start positions: renumber keys
                if (is_root) Q.root_left = pair;
                else { nodes[node_id].left_first = pair; nodes[node_id].count = n | kInteriorMark; }   // creation-order link; n kept for the renumber
                unsigned keep = 0u;
                int n_wide = 0;
                for (int side = 0; side < 2; ++side) {
                    if (cn[side] > 3u) {
                        const unsigned e2 = (pair + side) | (cs[side] << 10) | (cn[side] << 20);
                        if (cn[side] > 32u) {
                            n_wide += 1;
                            if (keep == 0u) keep = e2;                      // this wave goes on with it
                            else Q.work[atomicAdd(&Q.tail, 1u)] = e2;       // the other one is for whoever is idle
                        } else push_class(e2, cn[side]);
                    }
                }
                if (n_wide != 1) atomicAdd(&Q.pending, n_wide - 1);         // this node is done, n_wide more exist
#ifdef VD_PROF_SEL
                if (is_root) Q.prof[0] = (unsigned)(__builtin_amdgcn_s_memtime() - t_begin);
                atomicAdd(&Q.prof[6], 1u);
#endif
                W.next_ent = keep;
            }
            vd_wave_lds_sync();
            next_ent = W.next_ent;
#ifdef VD_TUNING
            { VD_CLS_T(tw5); const unsigned cls = n <= 64u ? 1u : (n <= 128u ? 2u : (n <= 256u ? 3u : 4u)), b8 = cls * 8u;
              VD_CLS_ADD(b8 + 0u, tw1 - tw0); VD_CLS_ADD(b8 + 1u, tw2 - tw1); VD_CLS_ADD(b8 + 2u, tw3 - tw2); VD_CLS_ADD(b8 + 3u, tw4 - tw3);
              VD_CLS_ADD(b8 + 4u, tw5 - tw4); VD_CLS_ADD(b8 + 5u, 1); if (is_root && root_by_block) VD_CLS_ADD(b8 + 6u, tw5 - tw0); }
#endif
        }
    }
    VD_CLS_T(t_loop_end);
    __syncthreads();

#ifdef VD_PROF_SEL
    if (tid == 0) Q.prof[1] = (unsigned)(__builtin_amdgcn_s_memtime() - t_begin);
#endif
    // ---------------- drain the lists ----------------
    {
        unsigned idle2 = 0;
        for (;;) {
            if (try_group()) { idle2 = 0; continue; }
            if (*(volatile int*)&Q.s_pending <= 0 || *(volatile unsigned*)&Q.bad) break;
            __builtin_amdgcn_s_sleep(4);
            if (++idle2 > (1u << 24)) { if (lane == 0) Q.bad = 2; break; }
        }
    }
    __threadfence_block();   // node records written by all lanes are re-read below, by this workgroup only (an agent-scope
                             // fence would write back the whole L2 of the XCD, once per subtree)
    __syncthreads();
    if (Q.bad) { if (tid == 0) atomicOr(err, Q.bad == 2u ? ERR_INTERNAL : ERR_DEGENERATE); return; }
#ifdef VD_PROF_SEL
    if (tid == 0) Q.prof[2] = (unsigned)(__builtin_amdgcn_s_memtime() - t_begin);
#endif
    const unsigned pool = Q.pool;
    const unsigned n_interior = pool / 2u;

    // ---- restore the reference's DFS pre-order numbering (blas.rs:110-112,125-126) ----
    // A node covers the positions [s, s + n) of the final arrangement and its children split that range, so pre-order
    // is the order of the keys (s ascending, n descending): rank r(j) = 1 (the subtree root, local pair 0) + the number of
    // interior nodes whose key is smaller - counted by every thread for its own nodes, all at once (two serial passes
    // over the creation order by one thread were 9 % of this kernel).  Work arrays alias the (now dead) centroids.
    unsigned* IK = reinterpret_cast<unsigned*>(&L.cent[0][0]);                     // [2N] key of an interior node, ~0 for a leaf (the centroids: 3 x N floats = exactly IK + R)
    unsigned short* R = reinterpret_cast<unsigned short*>(IK + 2 * kSmallMax);      // [2N]
    const unsigned n_nodes = pool;
    for (unsigned j = tid; j < n_nodes; j += 64u * kSubWaves) {
        const unsigned cnt = nodes[j].count;
        IK[j] = (cnt & kInteriorMark) ? ((unsigned)nmap[j] << 10) | (1023u - (cnt & ~kInteriorMark)) : 0xffffffffu;
    }
    __syncthreads();
    for (unsigned j = tid; j < n_nodes; j += 64u * kSubWaves) {
        const unsigned cnt = nodes[j].count;
        const unsigned key = ((unsigned)nmap[j] << 10) | (1023u - (cnt & ~kInteriorMark));
        unsigned r = 1u;
        for (unsigned k = 0; k < n_nodes; ++k) r += IK[k] < key ? 1u : 0u;
        R[j] = (unsigned short)r;
    }
    __syncthreads();
#ifdef VD_PROF_SEL
    if (tid == 0) Q.prof[3] = (unsigned)(__builtin_amdgcn_s_memtime() - t_begin);
#endif
    // new local index of node j = 2 * r(parent) + side; parent's rank = R[left sibling] - 1
    for (unsigned j = tid; j < n_nodes; j += 64u * kSubWaves) {
        const unsigned rl = R[j & ~1u];                    // rank of the left sibling = r(parent) + 1
        nmap[j] = (unsigned short)(2u * (rl - 1u) + (j & 1u));
        if (IK[j] != 0xffffffffu) { nodes[j].left_first = 2u * R[j]; nodes[j].count = 0u; }   // interior: its own pair in DFS numbering
    }
    for (unsigned x = tid; x < N; x += 64u * kSubWaves) final_ids[base + x] = __float_as_uint(cent[L.gid[L.perm[0][x]]].w);   // the triangle's own id
    if (tid == 0) {
        sub_interior[root_i] = n_interior;
#ifdef VD_PROF_SEL
        if (dbg_cycles) { dbg_cycles[2 * root_i] = (unsigned)(__builtin_amdgcn_s_memtime() - t_begin); dbg_cycles[2 * root_i + 1] = Q.prof[VD_PROF_SEL]; }
#else
        if (dbg_cycles) { dbg_cycles[2 * root_i] = (unsigned)(__builtin_amdgcn_s_memtime() - t_begin); dbg_cycles[2 * root_i + 1] = N; }
#endif
    }
#ifdef VD_TUNING
    { VD_CLS_T(t_end); VD_CLS_ADD(40, t_end - t_begin); VD_CLS_ADD(42, t_loaded - t_begin); VD_CLS_ADD(43, t_end - t_loop_end); VD_CLS_ADD(44, 1); }
    __syncthreads();
    // 64 replicas of the 64 counters (workgroup -> replica by its index): same-address global atomics are serialised chip-wide
    if (tid < 48u && Q.cls[tid] != 0u) atomicAdd(cls_prof + 64u * (blockIdx.x & 63u) + tid, (unsigned long long)Q.cls[tid]);
#endif
}
#undef VD_CLS_T
#undef VD_CLS_ADD

// =============================================================================================
// Phase A: level-synchronous emulation for segments larger than kSmallMax.
// =============================================================================================
struct Seg {
    unsigned start, count, node, item_first;
    unsigned n_items, best, Lst, ttot_cur;
    unsigned act[3], pad_act;         // act[c % 3]: first position round c still has to shuffle (see round_window)
    int cbk[6];                       // centroid-bound keys (min xyz, max xyz), inherited from the parent's a_child pass
    int child_k[24];                  // children: box keys left min/max, right min/max; then centroid keys likewise
    float pos[kCand + 3];
    unsigned ttot[kCand + 3], u_p[kCand + 3];
    u32x2 u_pay[kCand + 1];           // {triangle id, predicate bits} of each trial's never-examined element
    int bin_min[3][8][3], bin_max[3][8][3];
};

struct TopNode {                      // temporary top-tree node
    float mn[3]; unsigned start;
    float mx[3]; unsigned count;
    unsigned kind;                    // 0 leaf, 1 big interior, 2 small root
    unsigned left;                    // tmp id of the left child (right = left + 1) for kind 1
    unsigned small;                   // index into the small-root list for kind 2
    unsigned pad;
};

struct MidRoot { unsigned start, count, node, pad; int cbk[6]; int pad2[2]; };   // a segment the mid tier takes over

struct LevelCtl {                     // device-side counters
    unsigned n_seg, n_seg_next, n_items, n_top, n_small, err, n_mid, arrived;   // arrived: workgroups of a_boundary_kernel that have emitted their children
    unsigned max_count, max_count_next;       // largest segment of this / the next level (the host picks the round kernels by it)
    unsigned active, active_next;             // triangles in the segments of this / the next level
};

// Item geometry: item -> (segment, first relative position, valid count).  Lane order inside an
// item is (wave, j, lane): position = rel0 + wave*256 + j*64 + lane.
struct ItemCtx { unsigned seg, rel0, n_here; };
// The head of a segment record, COPIED into registers once per workgroup (two scalar loads).  Through a `const Seg*` the
// compiler re-read every field after every store that might alias it - the ISA of a_ranks_kernel was a chain of ~15
// dependent loads, each followed by s_waitcnt 0, for a kernel that moves 4 bytes per lane (profiles/r04_blas_item_isa.txt).
struct SegHead { unsigned start, count, node, item_first, n_items, best, Lst, ttot_cur, act[3], pad_act; };
// The kernels read the head THROUGH the record (scalar loads, served by the scalar cache the segment's items share); a
// copy of the 48 bytes into registers cost 8 ms per build as a struct assignment (three 16-byte VECTOR loads per lane, also
// with a readfirstlane'd index) and 16 ms field by field (one s_load_dwordx8 + x2 + x1 batch): -DVD_HEAD_COPY,
// profiles/r04_blas_item_isa.txt.
#ifdef VD_HEAD_COPY
#define VD_HEAD_VIEW(sg, segs, ic) (void)0
#else
#define VD_HEAD_VIEW(sg, segs, ic) sg = reinterpret_cast<const SegHead*>((segs) + (ic).seg)
#endif
static_assert(offsetof(Seg, act) == offsetof(SegHead, act) && offsetof(Seg, ttot_cur) == offsetof(SegHead, ttot_cur), "SegHead mirrors the first words of Seg");
__device__ __forceinline__ bool item_ctx(const Seg* segs, const unsigned* item_seg, const LevelCtl* ctl, ItemCtx& ic, SegHead& h) {
    const unsigned n_items = ctl->n_items, seg = item_seg[blockIdx.x];      // item_seg holds an entry for every workgroup of the grid
    if (blockIdx.x >= n_items) return false;
    ic.seg = seg;
    {   // field by field: a struct assignment is lowered to a memcpy and comes out as vector loads
        const SegHead* g = reinterpret_cast<const SegHead*>(segs + seg);
        h.start = g->start; h.count = g->count; h.node = g->node; h.item_first = g->item_first;
        h.n_items = g->n_items; h.best = g->best; h.Lst = g->Lst; h.ttot_cur = g->ttot_cur;
        h.act[0] = g->act[0]; h.act[1] = g->act[1]; h.act[2] = g->act[2]; h.pad_act = 0u;
    }
    ic.rel0 = (blockIdx.x - h.item_first) * kItem;
    ic.n_here = min((unsigned)kItem, h.count - ic.rel0);
    return true;
}

// The 21 split planes of a segment are fixed before its first trial (blas.rs:142-146), so every predicate of the
// level is evaluated once: bit c of bits21[x] = centroid[axis(c)] < pos[c].  At the start of a level an element's pos0 IS
// its position, so this pass streams: centroid in, bits and the first payload (axis 0) out.
template <typename P>
__global__ __launch_bounds__(256) void a_bits_kernel(Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                     typename P::T* __restrict__ pay, const f32x4* __restrict__ cent,
                                                     unsigned* __restrict__ bits21, unsigned* __restrict__ item_cnt) {
    __shared__ float s_pos[kCand + 3];
    __shared__ unsigned s_w[4];
    ItemCtx ic; SegHead hv; const SegHead* sg = &hv;
    if (!item_ctx(segs, item_seg, ctl, ic, hv)) return;
    VD_HEAD_VIEW(sg, segs, ic);
    // the 21 planes of the segment (blas.rs:142-146) from its centroid bounds; the segment's FIRST item also resets what the
    // level accumulates in the record - child keys, bin keys, the rounds' windows - and leaves the planes there.  (All of
    // that used to be a loop of the single-workgroup boundary kernel: 190 words x 1 859 segments through one CU, 100 us at
    // the widest levels; here it is one store per lane.)
    if (threadIdx.x < (unsigned)kCand) {
        const int c = (int)threadIdx.x, axis = c / 7;
        float cbmin[3] = {0.0f, 0.0f, 0.0f}, cbmax[3] = {0.0f, 0.0f, 0.0f};
        cbmin[axis] = box_lo(segs[ic.seg].cbk[axis]); cbmax[axis] = box_hi(segs[ic.seg].cbk[3 + axis]);      // cand_pos reads this axis only
        s_pos[c] = cand_pos(cbmin, cbmax, c);
    }
    if (blockIdx.x == sg->item_first) {
        Seg& w = segs[ic.seg];
        const unsigned t = threadIdx.x;
        if (t < (unsigned)kCand) w.pos[t] = s_pos[t];
        if (t < 24u) w.child_k[t] = (t % 6u) < 3u ? kBig : -kBig - 1;
        if (t < 144u) { if (t < 72u) (&w.bin_min[0][0][0])[t] = kBig; else (&w.bin_max[0][0][0])[t - 72u] = -kBig - 1; }
        if (t >= 252u && t < 255u) w.act[t - 252u] = 0u;
    }
    if (threadIdx.x < 4u) s_w[threadIdx.x] = 0u;
    __syncthreads();
    unsigned t0 = 0;        // trues of round 0 in this wave
    const unsigned a0 = sg->start + ic.rel0;
    f32x4 cv[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) { const unsigned x = threadIdx.x + 256u * (unsigned)j; cv[j] = cent[a0 + (x < ic.n_here ? x : 0u)]; }   // all loads first (see item_load)
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const unsigned x = threadIdx.x + 256u * (unsigned)j;
        const bool in = x < ic.n_here;
        const f32x4 c = cv[j];
        unsigned bits = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            bits |= (c.x < s_pos[k] ? 1u : 0u) << k;
            bits |= (c.y < s_pos[7 + k] ? 1u : 0u) << (7 + k);
            bits |= (c.z < s_pos[14 + k] ? 1u : 0u) << (14 + k);
        }
        if (in) { bits21[a0 + x] = bits; pay[a0 + x] = P::make(a0 + x, bits, 0u); }
        t0 += (unsigned)__popcll(__ballot(in && (bits & 1u) != 0u));
    }
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = t0;
    __syncthreads();
    if (threadIdx.x == 0) item_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];   // what a_count would find in round 0
}

// predicate mask of this lane group: returns ballot per j (4 per wave)
// The seven planes of an axis grow with k (blas.rs:146), so whatever trial k left of its pivot is also below the plane of
// trial k+1: partition_shuffle (blas.rs:168-182) walks over that prefix without a swap, and trial k+1 is exactly
// partition_shuffle on the suffix [L_k, n) - same never-examined element, same pivot, same arrangement.  Round c of a
// level therefore shuffles only [act, n), act = pivot of the previous trial on the same axis (0 for the first trial of
// an axis and for the final shuffle), and copies the band [band, act) that the previous round froze, so that the
// buffer it writes holds the whole arrangement again.  `r` is the round index 0..21.
struct Window { unsigned band, act; };
__device__ __forceinline__ Window round_window(const SegHead* sg, int r) {
    Window w;
    if (r % 7 == 0) { w.band = w.act = 0u; }
    else { w.act = sg->act[r % 3]; w.band = (r % 7 == 1) ? 0u : sg->act[(r + 2) % 3]; }
    return w;
}
// What the two kernels of a round derive from its index r = 0..21 (21: the final shuffle, c = -1), worked out once on the
// host: the remainders by 7 and 3 were scalar arithmetic at the head of every workgroup of the 700 launches of a build.
struct RoundK { int c; unsigned sh, sh_next, axis_next, i_act, i_band, i_next, whole, second; };
template <typename P>
inline RoundK make_round(int c) {
    const unsigned r = c >= 0 ? (unsigned)c : (unsigned)kCand;
    RoundK k;
    k.c = c; k.sh = P::shift(r); k.sh_next = P::shift(r + 1u); k.axis_next = (r + 1u) / 7u;
    k.i_act = r % 3u; k.i_band = (r + 2u) % 3u; k.i_next = (r + 1u) % 3u; k.whole = r % 7u == 0u ? 1u : 0u; k.second = r % 7u == 1u ? 1u : 0u;
    return k;
}
__device__ __forceinline__ Window round_window(const SegHead* sg, const RoundK& k) {
    Window w;
    if (k.whole) { w.band = w.act = 0u; }
    else { w.act = sg->act[k.i_act]; w.band = k.second ? 0u : sg->act[k.i_band]; }
    return w;
}

// the item's payloads: all loads first, unconditional (a lane past the item's end re-reads the item's first position) - inside
// `if (in window)` every load was its own basic block with its own s_waitcnt 0: four serial round trips per lane
template <typename P>
__device__ __forceinline__ void item_load(const SegHead* sg, const ItemCtx& ic, const typename P::T* __restrict__ pay, typename P::T (&vals)[kPer]) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const typename P::T* __restrict__ base = pay + sg->start + ic.rel0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const unsigned x = wave * (unsigned)(kItem / 4) + j * 64u + lane;
#ifdef VD_OLD_LOAD
        if (x < ic.n_here) vals[j] = base[x];
#else
        vals[j] = base[x < ic.n_here ? x : 0u];
#endif
    }
}
// predicates of the item's positions that lie in the shuffled window (positions below `act` stay out of the ballots)
template <typename P, bool LOAD = true>
__device__ __forceinline__ void item_masks_sh(const SegHead* sg, const ItemCtx& ic, const typename P::T* __restrict__ pay, unsigned sh,
                                              unsigned act, unsigned long long (&masks)[kPer], typename P::T (&vals)[kPer]);
template <typename P, bool LOAD = true>
__device__ __forceinline__ void item_masks(const SegHead* sg, const ItemCtx& ic, const typename P::T* __restrict__ pay, const RoundK& k,
                                           unsigned act, unsigned long long (&masks)[kPer], typename P::T (&vals)[kPer]) {
    item_masks_sh<P, LOAD>(sg, ic, pay, k.c >= 0 ? k.sh : P::shift(sg->best), act, masks, vals);
}
template <typename P, bool LOAD = true>
__device__ __forceinline__ void item_masks(const SegHead* sg, const ItemCtx& ic, const typename P::T* __restrict__ pay, int c,
                                           unsigned act, unsigned long long (&masks)[kPer], typename P::T (&vals)[kPer]) {
    item_masks_sh<P, LOAD>(sg, ic, pay, P::shift(c >= 0 ? (unsigned)c : sg->best), act, masks, vals);
}
template <typename P, bool LOAD>
__device__ __forceinline__ void item_masks_sh(const SegHead* sg, const ItemCtx& ic, const typename P::T* __restrict__ pay, unsigned sh,
                                              unsigned act, unsigned long long (&masks)[kPer], typename P::T (&vals)[kPer]) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (LOAD) item_load<P>(sg, ic, pay, vals);
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const unsigned x = wave * (unsigned)(kItem / 4) + j * 64u + lane;
        const bool in = x < ic.n_here && ic.rel0 + x >= act;
        masks[j] = __ballot(in && ((P::word(vals[j]) >> sh) & 1u));
    }
}

// round step 1: true count of every item.  `refresh` (rounds 7, 14 and the final one, whose window is the whole
// segment): the payload's predicate bits are replaced IN PLACE by those of the axis the coming rounds test.
template <typename P>
__global__ __launch_bounds__(256) void a_count_kernel(const Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                      typename P::T* __restrict__ pay, int c, unsigned* item_cnt,
                                                      const unsigned* __restrict__ bits21, int refresh) {
    __shared__ unsigned s_w[4];
    ItemCtx ic; SegHead hv; const SegHead* sg = &hv;
    if (!item_ctx(segs, item_seg, ctl, ic, hv)) return;
    VD_HEAD_VIEW(sg, segs, ic);
    const Window win = round_window(sg, c >= 0 ? c : kCand);
    if (ic.rel0 + ic.n_here <= win.act) { if (threadIdx.x == 0) item_cnt[blockIdx.x] = 0u; return; }   // wholly frozen
    if (P::kRefresh && refresh) {
        const unsigned axis = (c >= 0 ? (unsigned)c : sg->best) / 7u;
        const unsigned a0 = sg->start + ic.rel0;
        unsigned ps[kPer], nb[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) { const unsigned x = threadIdx.x + 256u * (unsigned)j; ps[j] = P::pos(pay[a0 + (x < ic.n_here ? x : 0u)]); }
#pragma unroll
        for (int j = 0; j < kPer; ++j) nb[j] = bits21[ps[j]];
#pragma unroll
        for (int j = 0; j < kPer; ++j) { const unsigned x = threadIdx.x + 256u * (unsigned)j; if (x < ic.n_here) pay[a0 + x] = P::make(ps[j], nb[j], axis); }
        __syncthreads();       // item_masks re-reads them in another lane order (same workgroup: its own stores are visible)
    }
    unsigned long long masks[kPer]; typename P::T vals[kPer];
    item_masks<P>(sg, ic, pay, c, win.act, masks, vals);
    unsigned t = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) t += (unsigned)__popcll(masks[j]);
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) item_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// round step 2 (single workgroup): exclusive scan of item counts; per-segment Ttot.  Thread t owns the contiguous
// range [t*per, (t+1)*per), per a multiple of 4 (16-byte loads, all in flight at once); two barriers in total.
__global__ __launch_bounds__(1024) void a_scan_kernel(Seg* segs, const LevelCtl* ctl, const unsigned* item_cnt, unsigned* item_pre) {
    __shared__ unsigned s_wave[16];
    const unsigned n = ctl->n_items, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned per = ((n + 1023u) / 1024u + 3u) & ~3u;
    const unsigned lo = min(n, tid * per), hi = min(n, lo + per);
    unsigned sum = 0;
    for (unsigned i = lo; i < hi; i += 4u) {
        if (i + 4u <= hi) { const u32x4 v = *reinterpret_cast<const u32x4*>(item_cnt + i); sum += (v.x + v.y) + (v.z + v.w); }
        else for (unsigned k = i; k < hi; ++k) sum += item_cnt[k];
    }
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off);
        if (lane >= (unsigned)off) incl += t;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    unsigned run = incl - sum, total = 0;
    for (unsigned w = 0; w < 16u; ++w) { if (w < wave) run += s_wave[w]; total += s_wave[w]; }
    for (unsigned i = lo; i < hi; i += 4u) {
        if (i + 4u <= hi) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(item_cnt + i);
            u32x4 o; o.x = run; o.y = run + v.x; o.z = o.y + v.y; o.w = o.z + v.z; run = o.w + v.w;
            *reinterpret_cast<u32x4*>(item_pre + i) = o;
        } else {
            for (unsigned k = i; k < hi; ++k) { item_pre[k] = run; run += item_cnt[k]; }
        }
    }
    if (tid == 0u) item_pre[n] = total;
    __syncthreads();
    __threadfence_block();
    for (unsigned i = tid; i < ctl->n_seg; i += 1024u) {
        Seg& sg = segs[i];
        sg.ttot_cur = item_pre[sg.item_first + sg.n_items] - item_pre[sg.item_first];
    }
}

// An item's prefix inside its segment and the segment's total.  With a_scan: two words of its output.  Without (SF, the
// levels whose segments have <= 1024 items - all but the first few): every thread adds up to four of the segment's item
// counts itself, two wave sums, and the result comes out of the barrier the kernel has anyway - the single-workgroup
// scan between count and ranks (5.9 us, moves nothing) is not launched at those levels.
// (the loads are a step of their own so that a kernel can issue them together with its payload loads: both need only the
// segment's head)
template <bool SF>
__device__ __forceinline__ void prefix_load(const SegHead* sg, const unsigned* __restrict__ item_pre, unsigned (&v)[4]) {
    if (!SF) return;
    const unsigned ni = sg->n_items;
    const unsigned* __restrict__ cn = item_pre + sg->item_first;       // SF: `item_pre` is the count array of the round
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = threadIdx.x + (unsigned)k * 256u;
#ifdef VD_OLD_PREFIX
        v[k] = 0u; if (i < ni) v[k] = cn[i];
#else
        v[k] = cn[i < ni ? i : 0u];                          // unconditional: in flight together
#endif
    }
}
template <bool SF>
__device__ __forceinline__ void prefix_begin(const SegHead* sg, const unsigned (&v)[4], unsigned* s_red) {
    if (!SF) return;
    const unsigned mine = blockIdx.x - sg->item_first, ni = sg->n_items;
    unsigned pa = 0, pb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = threadIdx.x + (unsigned)k * 256u;
        pb += i < ni ? v[k] : 0u; pa += i < mine ? v[k] : 0u;      // mine <= ni
    }
    pa = wave_sum_u(pa); pb = wave_sum_u(pb);
    if ((threadIdx.x & 63u) == 0u) { s_red[threadIdx.x >> 6] = pa; s_red[4u + (threadIdx.x >> 6)] = pb; }
}
template <bool SF>
__device__ __forceinline__ void prefix_end(const SegHead* sg, const unsigned* __restrict__ item_pre, const unsigned* s_red, unsigned& run, unsigned& ttot) {
    if (SF) { run = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]); ttot = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]); }
    else { run = item_pre[blockIdx.x] - item_pre[sg->item_first]; ttot = sg->ttot_cur; }
}

// round step 3: TL per position + rank -> position tables
template <typename P, bool SF>
__global__ __launch_bounds__(256) void a_ranks_kernel(const Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                      const typename P::T* __restrict__ pay, const RoundK rk, const unsigned* item_pre,
                                                      unsigned* __restrict__ falsepos, unsigned* __restrict__ truepos,
                                                      unsigned* __restrict__ cnt_next) {
    __shared__ unsigned s_w[4], s_red[8];
    ItemCtx ic; SegHead hv; const SegHead* sg = &hv;
    if (!item_ctx(segs, item_seg, ctl, ic, hv)) return;
    VD_HEAD_VIEW(sg, segs, ic);
    if (cnt_next && threadIdx.x == 0) cnt_next[blockIdx.x] = 0u;      // a_apply of this round adds the next round's trues up in it
    const Window win = round_window(sg, rk);
    if (ic.rel0 + ic.n_here <= win.act) return;             // wholly frozen
    unsigned long long masks[kPer]; typename P::T vals[kPer];
    unsigned pv[4];
    item_load<P>(sg, ic, pay, vals);
    prefix_load<SF>(sg, item_pre, pv);
    item_masks<P, false>(sg, ic, pay, rk, win.act, masks, vals);
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned t = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) t += (unsigned)__popcll(masks[j]);
    if (lane == 0u) s_w[wave] = t;
    prefix_begin<SF>(sg, pv, s_red);
    __syncthreads();
    unsigned run, ttot;
    prefix_end<SF>(sg, item_pre, s_red, run, ttot);
    for (unsigned w = 0; w < wave; ++w) run += s_w[w];
    // positions and table indices are relative to the window [act, n); the tables of the window start at s + act
    const unsigned s = sg->start + win.act;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const unsigned xr = wave * (unsigned)(kItem / 4) + j * 64u + lane;
        if (xr < ic.n_here && ic.rel0 + xr >= win.act) {
            const unsigned x = ic.rel0 + xr - win.act;
            const bool p = (masks[j] >> lane) & 1ull;
            const unsigned tl = run + vd_mbcnt(masks[j]);
            // Only the elements the shuffle SWAPS are ever looked up and believed (a_apply): the front pointer examines the
            // originals at [0, pivot] in place and the back pointer those right of it, pivot = ttot - p(u) - a true left of
            // ttot - 1 and a false right of ttot + 1 is nobody's t_F / f_{T+1}, and two thirds of the table stores never happen
            // (tests/test_closed_form_shuffle.py holds this form against the literal loop, stale entries included)
#ifdef VD_DENSE_TABLES
            const bool wr = true;
#else
            const bool wr = p ? x + 1u >= ttot : x <= ttot + 1u;
#endif
            if (wr) {
                if (p) truepos[s + (ttot - tl - 1u)] = x;     // index T: (T+1)-th true from the right
                else falsepos[s + (x - tl)] = x;              // index F: (F+1)-th false from the left
            }
        }
        run += (unsigned)__popcll(masks[j]);
    }
}

// Counting into the items elements land in.  Most counted elements land in the item they came from (a false the back
// pointer examined moves one position: partition_shuffle, blas.rs:168-182): those are a ballot and a popcount into a
// wave-uniform register, one atomic per wave at the end.  The others - falses the front pointer threw to the far end,
// `u`, and in mode 2 the trues that fill their holes - land monotonically along a wave within each class, so a wave
// makes a handful of runs of lanes with the same item, and a run's head adds the run.  Called by the whole wave.
__device__ __forceinline__ void count_runs(bool me, unsigned item, unsigned* __restrict__ cnt) {
    const unsigned long long part = __ballot(me);
    if (part == 0ull) return;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull, below = part & lt;
    const unsigned prev_item = __shfl(item, below ? 63 - __clzll((long long)below) : (int)lane);
    const bool head = me && (below == 0ull || prev_item != item);
    const unsigned long long heads = __ballot(head);
    if (head) {
        const unsigned long long above = heads & ~((2ull << lane) - 1ull);
        const unsigned long long upto = above ? (1ull << __builtin_ctzll(above)) - 1ull : ~0ull;
        __hip_atomic_fetch_add(cnt + item, (unsigned)__popcll(part & upto & ~lt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// round step 4: destinations, scatter, `u` - and step 1 of the NEXT round: an element knows where it lands and what the
// next round asks of it, so it adds itself to the true count of the item it lands in (`cnt_next`, zeroed by a_ranks).
//   mode 1: the next round stays on this axis; its window is [pivot, n) = this round's falses and `u`;
//   mode 2: the next round starts an axis: window = the whole segment, and the 4-byte payload swaps its seven bits for
//           that axis' on the way - every position of the segment is rewritten and counted, the frozen prefix too;
//   mode 0: the next round's predicate is not known yet (the final shuffle follows the cost evaluation): a_count runs.
template <typename P, int mode, bool SF>
__global__ __launch_bounds__(256) void a_apply_kernel(Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                      const typename P::T* __restrict__ src, typename P::T* __restrict__ dst, const RoundK rk,
                                                      const unsigned* item_pre, const unsigned* __restrict__ falsepos,
                                                      const unsigned* __restrict__ truepos, unsigned char* __restrict__ is_u_flag,
                                                      const unsigned* __restrict__ bits21, unsigned* __restrict__ cnt_next) {
    __shared__ unsigned s_w[4], s_red[8];
    ItemCtx ic; SegHead hv; const SegHead* sg = &hv;
    if (!item_ctx(segs, item_seg, ctl, ic, hv)) return;
    VD_HEAD_VIEW(sg, segs, ic);
    const int c = rk.c;
    const Window win = round_window(sg, rk);
    const unsigned copy_from = mode == 2 ? 0u : win.band;
    if (ic.rel0 + ic.n_here <= copy_from) return;         // frozen before the previous round: both buffers agree
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned own_cnt = 0;                                  // elements counted into this very item (wave-uniform)
    const unsigned sh_next = rk.sh_next, axis_next = rk.axis_next;
    // Loads first, stores last, nothing conditional in between (see item_load): the item's payloads, in mode 2 the 21 bits
    // of each (every position of the segment is rewritten with the next axis' bits), then the rank tables, then the stores.
    unsigned long long masks[kPer]; typename P::T vals[kPer];
    unsigned pv[4];
    item_load<P>(sg, ic, src, vals);
    prefix_load<SF>(sg, item_pre, pv);
    unsigned nb[kPer];
    if (mode == 2 && P::kRefresh) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) nb[j] = bits21[P::pos(vals[j])];
    }
    // the band the previous round froze: straight copy, so that `dst` holds the whole arrangement
    if (ic.rel0 < win.act && copy_from < win.act) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const unsigned xr = wave * (unsigned)(kItem / 4) + j * 64u + lane, xa = ic.rel0 + xr;
            bool pn = false;
            if (xr < ic.n_here && xa >= copy_from && xa < win.act) {
                typename P::T v = vals[j];
                if (mode == 2) {
                    if (P::kRefresh) v = P::make(P::pos(v), nb[j], axis_next);
                    pn = (P::word(v) >> sh_next) & 1u;
                }
                dst[sg->start + xa] = v;
            }
            if (mode == 2) own_cnt += (unsigned)__popcll(__ballot(pn));
        }
    }
    if (ic.rel0 + ic.n_here <= win.act) {
        if (mode == 2 && own_cnt && lane == 0u) __hip_atomic_fetch_add(cnt_next + blockIdx.x, own_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // predicates and TL are recomputed from the payload (cheaper than a per-position word through HBM)
    item_masks<P, false>(sg, ic, src, rk, win.act, masks, vals);
    unsigned t = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) t += (unsigned)__popcll(masks[j]);
    if (lane == 0u) s_w[wave] = t;
    prefix_begin<SF>(sg, pv, s_red);
    __syncthreads();
    unsigned run, ttot;
    prefix_end<SF>(sg, item_pre, s_red, run, ttot);
    for (unsigned w = 0; w < wave; ++w) run += s_w[w];
    // everything below is partition_shuffle on the window [act, n): positions relative to act
    const unsigned n = sg->count - win.act, s = sg->start + win.act, ftot = n - ttot;
    bool counts[kPer]; unsigned land[kPer];
    bool in[kPer], pp[kPer], need_t[kPer], need_f[kPer];
    unsigned xx[kPer], FF[kPer], TT[kPer], tp[kPer], fp[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const unsigned xr = wave * (unsigned)(kItem / 4) + j * 64u + lane;
        in[j] = xr < ic.n_here && ic.rel0 + xr >= win.act;
        xx[j] = ic.rel0 + xr - win.act;
        pp[j] = (masks[j] >> lane) & 1ull;
        const unsigned tl = run + vd_mbcnt(masks[j]);
        FF[j] = xx[j] - tl; TT[j] = ttot - tl - (pp[j] ? 1u : 0u);
        need_t[j] = in[j] && FF[j] != 0u && FF[j] <= ttot;
        need_f[j] = in[j] && TT[j] + 1u <= ftot;
        run += (unsigned)__popcll(masks[j]);
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {                       // the eight gathers of a lane in flight together (index 0 of the window's tables exists)
        tp[j] = truepos[s + (need_t[j] ? FF[j] - 1u : 0u)];
        fp[j] = falsepos[s + (need_f[j] ? TT[j] : 0u)];
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        counts[j] = false; land[j] = 0u;
        if (in[j]) {
            const unsigned x = xx[j];
            const bool p = pp[j];
            const long long tF = FF[j] == 0u ? (long long)n : (need_t[j] ? (long long)tp[j] : -1ll);
            const unsigned fj = need_f[j] ? fp[j] : n;
#ifdef VD_DENSE_TABLES
            const bool left = (long long)x < tF;
            const bool amb = true;
#else
            // a_ranks writes only the entries the shuffle's swaps look up, so an entry is trusted only where it must be this
            // round's: the front pointer examines the originals at [0, pivot] in place and the back pointer the rest, pivot =
            // ttot or ttot - 1 - everything left of ttot is consumed from the left, everything right of it from the right, and
            // AT ttot the t_F entry decides (its writer sits right of ttot - 1); `u` is within one position of ttot
            const bool left = x < ttot || (x == ttot && (long long)x < tF);
            const bool amb = x + 1u - ttot <= 2u;
#endif
            const unsigned fetch = left ? x + n - (unsigned)tF : (n - 1u - x) + fj + 1u;
            const bool is_u = amb && fetch == n - 1u;
            unsigned dest;
            if (is_u) dest = ttot - (p ? 1u : 0u);
            else if (left) dest = p ? x : (unsigned)tF - 1u;
            else dest = p ? fj : x - 1u;
            typename P::T v = vals[j];
            const unsigned upos = P::pos(v);
            if (mode == 2) {
                if (P::kRefresh) v = P::make(upos, nb[j], axis_next);
                counts[j] = (P::word(v) >> sh_next) & 1u;
            } else if (mode == 1) {
                counts[j] = ((P::word(v) >> sh_next) & 1u) && (is_u || dest >= ttot);   // left of the pivot = frozen for the next round
            }
            land[j] = sg->item_first + (win.act + dest) / (unsigned)kItem;
            dst[s + dest] = v;
            if (is_u && c >= 0) {
                Seg& w = segs[ic.seg];
                // counts as the reference sees them: examined trues of the WHOLE segment = frozen prefix + this window's
                const u32x2 urec = {upos, bits21[upos]};               // the record keeps all 21 bits: the cost evaluation needs them
                w.u_pay[c] = urec; w.u_p[c] = p ? 1u : 0u; w.ttot[c] = win.act + ttot;
                w.act[rk.i_next] = win.act + ttot - (p ? 1u : 0u);   // this trial's pivot: where the next round starts
                is_u_flag[upos] = 1;
            }
        }
    }
    if (mode != 0) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const bool mine = counts[j] && land[j] == blockIdx.x;
            own_cnt += (unsigned)__popcll(__ballot(mine));
            count_runs(counts[j] && !mine, land[j], cnt_next);
        }
        if (own_cnt && lane == 0u) __hip_atomic_fetch_add(cnt_next + blockIdx.x, own_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// binning over the non-u elements (one pass per level)
template <typename P>
__global__ __launch_bounds__(256) void a_bin_kernel(Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                    const typename P::T* __restrict__ pay, const TriBox* __restrict__ boxes,
                                                    const unsigned char* __restrict__ is_u_flag, const unsigned* __restrict__ bits21) {
    // One private copy of the 144 bin keys PER LANE ([entry][lane]: lanes sit in different banks), shared by the four
    // waves: neighbouring elements fall into the same few bins, and same-address LDS atomics serialise - with four copies
    // by quarter wave the 18 atomics of an element were most of this kernel (228 us per level at every level, whatever
    // the gathers cost).  A workgroup covers kBinItems consecutive items and flushes to the segment record only when the
    // segment changes: the flush is 144 same-address global atomics per workgroup (~5 ns each, serialised per address),
    // which at the top levels - one segment, 8 k items - used to be the floor of the kernel.
    __shared__ int s_bins[144][64];                    // entry = (axis * 8 + bin) * 6 + q; q < 3: min keys, q >= 3: max keys
    const unsigned n_items = ctl->n_items;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned cur_seg = kNone;
    auto reset = [&]() { for (unsigned i = threadIdx.x; i < 144u * 64u; i += 256u) (&s_bins[0][0])[i] = ((i >> 6) % 6u) < 3u ? kBig : -kBig - 1; };
    auto flush = [&](unsigned seg) {
        for (unsigned e = wave; e < 144u; e += 4u) {   // a wave reduces the 64 copies of an entry on the VALU
            const bool is_min = (e % 6u) < 3u;
            const int v = is_min ? wave_min_i(s_bins[e][lane]) : wave_max_i(s_bins[e][lane]);
            if (lane == 0u) {
                const unsigned ab = e / 6u, q = e % 6u;      // Seg::bin_min / bin_max are [axis][bin][3]
                if (is_min) { if (v != kBig) atomicMin(&(&segs[seg].bin_min[0][0][0])[ab * 3u + q], v); }
                else if (v != -kBig - 1) atomicMax(&(&segs[seg].bin_max[0][0][0])[ab * 3u + (q - 3u)], v);
            }
        }
    };
    reset();
    if (n_items == 0u) return;
    // A workgroup's life used to be kBinItems times {item -> segment, segment -> head, head -> element loads, LDS atomics}: four
    // dependent round trips per item, 32 per workgroup, with the machine holding one workgroup per slot (98 us per level for a
    // kernel that moves 29 B per element).  Now the descriptors of all its items are fetched together up front (wave-uniform:
    // scalar registers), and the element loads of item k + 1 are in flight while item k's atomics run.
    unsigned seg_k[kBinItems], a0_k[kBinItems], n_k[kBinItems];
#pragma unroll
    for (int k = 0; k < kBinItems; ++k) {
        const unsigned item = blockIdx.x * kBinItems + (unsigned)k;
        seg_k[k] = item_seg[item < n_items ? item : n_items - 1u];
    }
#pragma unroll
    for (int k = 0; k < kBinItems; ++k) {
        const unsigned item = blockIdx.x * kBinItems + (unsigned)k;
        const SegHead* g = reinterpret_cast<const SegHead*>(segs + seg_k[k]);
        const unsigned start = g->start, count = g->count, first = g->item_first;
        const unsigned rel0 = (item - first) * kItem;
        const bool ok = item < n_items && rel0 < count;
        seg_k[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)(ok ? seg_k[k] : kNone));
        a0_k[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)(ok ? start + rel0 : 0u));
        n_k[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)(ok ? min((unsigned)kItem, count - rel0) : 0u));
    }
    struct Loaded { unsigned b21[kPer]; unsigned char u[kPer]; TriBox bx[kPer]; };
    auto load_item = [&](int k, Loaded& d) {
        const unsigned a0 = a0_k[k], n_here = n_k[k];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const unsigned xr = threadIdx.x + 256u * (unsigned)j;
#ifdef VD_BIN_GATHER
            const unsigned ps = P::pos(pay[a0 + (xr < n_here ? xr : 0u)]);
#else
            const unsigned ps = a0 + (xr < n_here ? xr : 0u);       // (n_here == 0: position 0 of the arrays, never used)
#endif
            d.u[j] = is_u_flag[ps]; d.bx[j] = boxes[ps]; d.b21[j] = bits21[ps];
        }
    };
    Loaded buf[2];
    load_item(0, buf[0]);
    __syncthreads();                                   // the reset is done
#pragma unroll
    for (int k = 0; k < kBinItems; ++k) {
        if (k + 1 < kBinItems) load_item(k + 1, buf[(k + 1) & 1]);      // in flight while this item's atomics run
        const unsigned seg = seg_k[k];
        if (seg == kNone) continue;
        if (seg != cur_seg) {
            if (cur_seg != kNone) { __syncthreads(); flush(cur_seg); __syncthreads(); reset(); __syncthreads(); }
            cur_seg = seg;
        }
        const Loaded& d = buf[k & 1];
        const unsigned n_here = n_k[k];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const unsigned xr = threadIdx.x + 256u * (unsigned)j;
            if (xr >= n_here || d.u[j]) continue;
            const TriBox bx = d.bx[j];
            const unsigned b21 = d.b21[j];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int b = 7 - __popc((b21 >> (7 * a)) & 0x7fu);   // bin = number of planes the centroid is not below
                int* row = &s_bins[(a * 8 + b) * 6][lane];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    atomicMin(row + 64 * q, vd_key(bx.mn[q]));
                    atomicMax(row + 64 * (3 + q), vd_key(bx.mx[q]));
                }
            }
        }
    }
    __syncthreads();
    if (cur_seg != kNone) flush(cur_seg);
}

// one wave per segment: 21 costs -> best plane, stale pivot
__global__ __launch_bounds__(64) void a_eval_kernel(Seg* segs, LevelCtl* ctl, const TriBox* __restrict__ boxes) {
    if (blockIdx.x >= ctl->n_seg) return;
    Seg& sg = segs[blockIdx.x];
    const unsigned lane = threadIdx.x;
    __shared__ EvalU s_u;
    eval_stage(s_u, sg.u_pay, boxes, lane);
    vd_u64 key = ~0ull;
    if (lane < (unsigned)kCand) {
        const int c = (int)lane, a = c / 7, k = c % 7 + 1;
        EvalIn in{&sg.bin_min[a][0][0], &sg.bin_max[a][0][0], &s_u, s_u.id[c]};
        const unsigned n1 = sg.ttot[c] - sg.u_p[c];
        key = cost_key(eval_candidate(in, a, k, sg.pos[c], n1, sg.count), (unsigned)c);
    }
    key = wave_min_u64(key);
    if (lane == 0) {
        if (key == ~0ull) { atomicOr(&ctl->err, ERR_DEGENERATE); sg.best = 0; sg.Lst = 1; }
        else { sg.best = (unsigned)key; sg.Lst = sg.ttot[sg.best] - sg.u_p[sg.best]; }
    }
}

// children boxes from the final arrangement (blas.rs:115-123); the same pass writes the per-triangle data out in the
// order the level left (set `next`), which is what the next level - or the tier that takes the segment over - reads
template <typename P>
__global__ __launch_bounds__(256) void a_child_kernel(Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                      const typename P::T* __restrict__ pay, const TriBox* __restrict__ boxes,
                                                      const f32x4* __restrict__ cent, TriBox* __restrict__ boxes_next,
                                                      f32x4* __restrict__ cent_next) {
    __shared__ int s_k[24];
    ItemCtx ic; SegHead hv; const SegHead* sg = &hv;
    if (!item_ctx(segs, item_seg, ctl, ic, hv)) return;
    VD_HEAD_VIEW(sg, segs, ic);
    if (threadIdx.x < 24) s_k[threadIdx.x] = (threadIdx.x % 6) < 3 ? kBig : -kBig - 1;
    __syncthreads();
    // [0,12): vertex boxes of the left / right child (blas.rs:115-123); [12,24): their centroid boxes, which are
    // the next level's `cb` (blas.rs:142) and save that level a pass
    const unsigned Lst = sg->Lst;
    int k24[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) k24[i] = (i % 6) < 3 ? kBig : -kBig - 1;
    // ids first, then the 40 bytes of each (all gathers of a lane in flight together: see item_load), then stores and keys
    const unsigned a0 = sg->start + ic.rel0;
    unsigned ids[kPer];
    TriBox bxs[kPer];
    f32x4 cs[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) { const unsigned xr = threadIdx.x + 256u * (unsigned)j; ids[j] = P::pos(pay[a0 + (xr < ic.n_here ? xr : 0u)]); }
#pragma unroll
    for (int j = 0; j < kPer; ++j) { bxs[j] = boxes[ids[j]]; cs[j] = cent[ids[j]]; }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const unsigned xr = threadIdx.x + 256u * (unsigned)j;
        if (xr < ic.n_here) { boxes_next[a0 + xr] = bxs[j]; cent_next[a0 + xr] = cs[j]; }
    }
#ifndef VD_CHILD24
    // all but ONE item of a segment lie wholly on one side of the pivot: 12 running keys and 12 reductions instead of 24
    // (27.6 ms per build against 27.85 with the 24-key form for every item, -DVD_CHILD24; before the loads of this kernel
    // were issued together the same idea was SLOWER, 3.64 ms of a_child per build against 3.02)
    const bool all_left = ic.rel0 + ic.n_here <= Lst, all_right = ic.rel0 >= Lst;
    if (all_left || all_right) {
        int k12[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) k12[i] = (i % 6) < 3 ? kBig : -kBig - 1;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const unsigned xr = threadIdx.x + 256u * (unsigned)j;
            if (xr >= ic.n_here) continue;
            const TriBox bx = bxs[j];
            const float ce[3] = {cs[j].x, cs[j].y, cs[j].z};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                k12[q] = min(k12[q], vd_key(bx.mn[q]));
                k12[3 + q] = max(k12[3 + q], vd_key(bx.mx[q]));
                k12[6 + q] = min(k12[6 + q], vd_key_lo(ce[q]));
                k12[9 + q] = max(k12[9 + q], vd_key_hi(ce[q]));
            }
        }
        const unsigned o = all_left ? 0u : 6u;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const bool is_min = (i % 6) < 3;
            const int r = is_min ? wave_min_i(k12[i]) : wave_max_i(k12[i]);
            const unsigned slot = (i < 6 ? 0u : 12u) + o + (unsigned)(i % 6);
            if ((threadIdx.x & 63u) == 0u) { if (is_min) atomicMin(&s_k[slot], r); else atomicMax(&s_k[slot], r); }
        }
    } else
#endif
    {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const unsigned xr = threadIdx.x + 256u * (unsigned)j;
        if (xr >= ic.n_here) continue;
        const unsigned x = ic.rel0 + xr;
        const TriBox bx = bxs[j];
        const f32x4 c = cs[j];
        const float ce[3] = {c.x, c.y, c.z};
        const int o = x < Lst ? 0 : 6;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            k24[o + q] = min(k24[o + q], vd_key(bx.mn[q]));
            k24[o + 3 + q] = max(k24[o + 3 + q], vd_key(bx.mx[q]));
            k24[12 + o + q] = min(k24[12 + o + q], vd_key_lo(ce[q]));
            k24[12 + o + 3 + q] = max(k24[12 + o + 3 + q], vd_key_hi(ce[q]));
        }
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const bool is_min = (i % 6) < 3;
        const int r = is_min ? wave_min_i(k24[i]) : wave_max_i(k24[i]);
        if ((threadIdx.x & 63u) == 0u) { if (is_min) atomicMin(&s_k[i], r); else atomicMax(&s_k[i], r); }
    }
    }
    __syncthreads();
    if (threadIdx.x < 24) {
        const bool is_min = (threadIdx.x % 6) < 3;
        if (is_min) atomicMin(&segs[ic.seg].child_k[threadIdx.x], s_k[threadIdx.x]);
        else atomicMax(&segs[ic.seg].child_k[threadIdx.x], s_k[threadIdx.x]);
    }
}

// The level that ended emits its children, kFinalChunk segments per workgroup: one thread per segment reads its record,
// clears the u flags and classifies the two children; the slots they need in the top-node array, the small / mid root lists
// and the next level's segment list are counted in LDS and reserved by ONE global atomic per list and workgroup (the mid
// tier of earlier levels runs on its own stream meanwhile and appends to the same lists), and then everything is written.
// Before: one thread per segment with five returning global atomics each, all on the control words' cache line and in
// divergent code (no wave aggregation) - 90 us at the widest levels - and 21 byte stores to the u flags, each followed by a
// re-read of the record (a byte store may alias anything): 10 us for ONE segment (profiles/r04_blas_boundary.log).
__device__ __forceinline__ void finalize_segments(const Seg* ended, unsigned n_ended, Seg* next, LevelCtl* ctl, unsigned* s_need,
                                                  unsigned* s_base, TopNode* top, SmallRoot* small, unsigned char* is_u_flag, unsigned top_cap,
                                                  unsigned small_cap, MidRoot* mid, unsigned mid_cap, unsigned parity /* the set this level's a_child wrote */) {
    const unsigned tid = threadIdx.x;
    const unsigned i = blockIdx.x * kFinalChunk + tid;
    const bool act = tid < kFinalChunk && i < n_ended;
    if (tid < 6u) s_need[tid] = 0u;                        // top, small, mid, next segments; largest count, triangles of the next level
    __syncthreads();
    int ck[24];
    unsigned sg_start = 0, sg_count = 0, sg_Lst = 0, sg_node = 0, r_top = 0, r_slot[2] = {0u, 0u}, kind[2] = {0u, 0u}, cnt[2] = {0u, 0u};
    if (act) {
        const Seg& sg = ended[i];
        unsigned uid[kCand];
#pragma unroll
        for (int c = 0; c < kCand; ++c) uid[c] = sg.u_pay[c].x;
#pragma unroll
        for (int k = 0; k < 24; ++k) ck[k] = sg.child_k[k];
        sg_start = sg.start; sg_count = sg.count; sg_Lst = sg.Lst; sg_node = sg.node;
#pragma unroll
        for (int c = 0; c < kCand; ++c) is_u_flag[uid[c]] = 0;
        r_top = atomicAdd(&s_need[0], 2u);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            cnt[side] = side == 0 ? sg_Lst : sg_count - sg_Lst;
            if (cnt[side] <= 3u) kind[side] = 0u;                                   // leaf
            else if (cnt[side] <= (unsigned)kSmallMax) { kind[side] = 2u; r_slot[side] = atomicAdd(&s_need[1], 1u); }
            else if (cnt[side] <= (unsigned)kMidMax) { kind[side] = 3u; r_slot[side] = atomicAdd(&s_need[2], 1u); }
            else { kind[side] = 4u; r_slot[side] = atomicAdd(&s_need[3], 1u); atomicMax(&s_need[4], cnt[side]); atomicAdd(&s_need[5], cnt[side]); }
        }
    }
    __syncthreads();
    if (tid < 4u) {
        unsigned* g = tid == 0u ? &ctl->n_top : (tid == 1u ? &ctl->n_small : (tid == 2u ? &ctl->n_mid : &ctl->n_seg_next));
        s_base[tid] = s_need[tid] ? atomicAdd(g, s_need[tid]) : 0u;
    } else if (tid == 4u) { if (s_need[4]) atomicMax(&ctl->max_count_next, s_need[4]); }
    else if (tid == 5u) { if (s_need[5]) atomicAdd(&ctl->active_next, s_need[5]); }
    __syncthreads();
    if (!act) return;
    const unsigned pair = s_base[0] + r_top;
    if (pair + 2u > top_cap) { atomicOr(&ctl->err, 4u); return; }
    top[sg_node].kind = 1u;
    top[sg_node].left = pair;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        TopNode t;
        for (int q = 0; q < 3; ++q) { t.mn[q] = box_lo(ck[side * 6 + q]); t.mx[q] = box_hi(ck[side * 6 + 3 + q]); }
        t.start = side == 0 ? sg_start : sg_start + sg_Lst;
        t.count = cnt[side];
        t.left = 0; t.small = 0; t.pad = parity;
        if (kind[side] == 0u) {
            t.kind = 0u;
        } else if (kind[side] == 2u) {
            t.kind = 2u;
            const unsigned si = s_base[1] + r_slot[side];
            if (si < small_cap) small[si] = SmallRoot{t.start, t.count, pair + side, parity};
            else atomicOr(&ctl->err, 4u);
            t.small = si;
        } else if (kind[side] == 3u) {
            t.kind = 1u;
            const unsigned mi = s_base[2] + r_slot[side];
            if (mi < mid_cap) {
                MidRoot& m = mid[mi];
                m.start = t.start; m.count = t.count; m.node = pair + side; m.pad = parity;
                for (int q = 0; q < 6; ++q) m.cbk[q] = ck[12 + side * 6 + q];
            } else atomicOr(&ctl->err, 4u);
        } else {
            t.kind = 1u;
            Seg& ns = next[s_base[3] + r_slot[side]];
            ns.start = t.start; ns.count = t.count; ns.node = pair + side;
            for (int q = 0; q < 6; ++q) ns.cbk[q] = ck[12 + side * 6 + q];
        }
        top[pair + side] = t;
    }
}

// The boundary between two levels as ONE launch (it was six single-workgroup launches in a dependent chain on 15 levels,
// then one single-workgroup launch that took 160 us at the widest levels: profiles/r04_blas_boundary.log): every workgroup
// emits the children of kFinalChunk segments of the level that ended (finalize_segments); the LAST workgroup to arrive
// then swaps the control words and sets the next level up - items per segment, first item, item -> segment map (the
// records' child / bin keys and planes are reset by a_bits_kernel, on all CUs).  `finalize` = 0 at the first level (its
// segments come from c_root_kernel; one workgroup).
__global__ __launch_bounds__(1024) void a_boundary_kernel(const Seg* ended, Seg* segs, LevelCtl* ctl, TopNode* top, SmallRoot* small,
                                                          unsigned char* is_u_flag, unsigned top_cap, unsigned small_cap, MidRoot* mid,
                                                          unsigned mid_cap, unsigned parity, unsigned* item_seg, int finalize) {
    constexpr unsigned kFirstLds = 8192u;
    __shared__ unsigned s_part[1024], s_first[kFirstLds], s_need[6], s_base[4], s_last;
    const unsigned tid = threadIdx.x;
#ifdef VD_BOUNDARY_PROF
    long long tp0 = clock64(), tp1 = tp0, tp2, tp3, tp4;
    unsigned prof_ended = 0;
#endif
    if (finalize) {
        const unsigned n_ended = ctl->n_seg;
#ifdef VD_BOUNDARY_PROF
        prof_ended = n_ended;
#endif
        finalize_segments(ended, n_ended, segs, ctl, s_need, s_base, top, small, is_u_flag, top_cap, small_cap, mid, mid_cap, parity);
        // the last workgroup to arrive goes on (its own and everybody else's records and counters are complete: release
        // before the ticket, acquire after it)
        if (gridDim.x > 1u) {                              // (a level of <= kFinalChunk segments is one workgroup: nothing to wait for, no fences)
            __threadfence();
            __syncthreads();
            if (tid == 0) {
                const unsigned ticket = __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                s_last = ticket == gridDim.x - 1u ? 1u : 0u;
            }
            __syncthreads();
            if (!s_last) return;
            __threadfence();
        } else __syncthreads();
        if (tid == 0) {
            ctl->arrived = 0u;
            ctl->n_seg = __hip_atomic_load(&ctl->n_seg_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ctl->n_seg_next = 0; ctl->n_items = 0;
            ctl->max_count = __hip_atomic_load(&ctl->max_count_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ctl->max_count_next = 0;
            ctl->active = __hip_atomic_load(&ctl->active_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ctl->active_next = 0;
        }
        __syncthreads();
    }
#ifdef VD_BOUNDARY_PROF
    tp1 = clock64();
#endif
    const unsigned n = ctl->n_seg;
    const unsigned per = (n + 1023u) / 1024u;
    const unsigned lo = min(n, tid * per), hi = min(n, lo + per);
    unsigned sum = 0;
    for (unsigned i = lo; i < hi; ++i) sum += (segs[i].count + kItem - 1) / kItem;      // loads only; n_items is stored with item_first below
    s_part[tid] = sum;
    __syncthreads();
#ifdef VD_BOUNDARY_PROF
    tp2 = clock64();
#endif
    for (unsigned off = 1; off < 1024u; off <<= 1) {
        const unsigned v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    unsigned run = s_part[tid] - sum;
    for (unsigned i = lo; i < hi; ++i) {
        const unsigned ni = (segs[i].count + kItem - 1) / kItem;
        segs[i].item_first = run; segs[i].n_items = ni;
        if (i < kFirstLds) s_first[i] = run;
        run += ni;
    }
    const unsigned n_items = s_part[1023];
    if (tid == 1023u) ctl->n_items = n_items;
    __syncthreads();
#ifdef VD_BOUNDARY_PROF
    tp3 = clock64();
#endif
    // item -> segment: every thread takes items tid, tid + 1024, ... and finds the last segment that starts at or before it
    // (one segment of thousands of items at the first levels, thousands of small ones later: either way a few steps)
    // (the search runs on a copy of the first items in LDS: through the records in memory it was 12 dependent misses per item)
    if (n <= kFirstLds) {
        for (unsigned j = tid; j < n_items; j += 1024u) {
            unsigned a = 0, b = n;
            while (b - a > 1u) { const unsigned m = (a + b) >> 1; if (s_first[m] <= j) a = m; else b = m; }
            item_seg[j] = a;
        }
    } else {
        for (unsigned j = tid; j < n_items; j += 1024u) {
            unsigned a = 0, b = n;
            while (b - a > 1u) { const unsigned m = (a + b) >> 1; if (segs[m].item_first <= j) a = m; else b = m; }
            item_seg[j] = a;
        }
    }
#ifdef VD_BOUNDARY_PROF
    __syncthreads();
    tp4 = clock64();
    if (tid == 0) printf("a_boundary: ended %u, next %u segments, %u items; cycles: finalize %lld, set-up %lld, scan %lld, items %lld\n", prof_ended, n, n_items,
                         tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3);
#endif
}

// =============================================================================================
// Mid tier: one 1024-lane workgroup takes a segment of kSmallMax < n <= kMidMax elements and splits it - and every
// child that is still larger than kSmallMax - with the arrangement resident in LDS: the 22 shuffles of a split are
// 22 x 4 barriers instead of 22 x 4 kernel launches over all of HBM.  Same arithmetic as the a_* kernels
// (closed-form partition_shuffle, held-out `u` elements, binned SAH), same top-tree / small-root outputs.
// =============================================================================================
struct MidNode { unsigned s0, n, node; int cbk[6]; };
struct MidLds {
    u32x2 pay[kMidMax > 0 ? kMidMax : 1];
    unsigned short tpos[kMidMax > 0 ? kMidMax : 1], fpos[kMidMax > 0 ? kMidMax : 1];
    int bin_min[3][8][3], bin_max[3][8][3];
    int bins16[144][16];                            // accumulation: entry = (axis * 8 + bin) * 6 + q, one copy per lane % 16 (same-
                                                    // address LDS atomics serialise, and neighbours fall into the same few bins)
    int child_k[24];
    float pos[kCand + 3];
    u32x2 u_pay[kCand + 1];
    unsigned u_p[kCand + 1], ttot[kCand + 1];
    unsigned wave_cnt[kMidThreads / 64];
    unsigned best, Lst, n_stack, pair;
    MidNode stack[16];
    EvalU eval_u;
};

// one partition_shuffle of the node [s0, s0+n) on predicate bit `cc` (blas.rs:168-182 in closed form);
// c >= 0 records the trial's never-examined element
__device__ __forceinline__ void mid_shuffle(MidLds& L, unsigned s0, unsigned n, unsigned cc, int c) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32x2 v[kMidPer];
    unsigned long long masks[kMidPer];
    unsigned t = 0;
#pragma unroll
    for (int j = 0; j < kMidPer; ++j) {                  // all reads first, unconditional (a lane past the end re-reads position 0): see item_load
        const unsigned x = wave * (64u * (unsigned)kMidPer) + j * 64u + lane;
        v[j] = L.pay[s0 + (x < n ? x : 0u)];
    }
#pragma unroll
    for (int j = 0; j < kMidPer; ++j) {
        const unsigned x = wave * (64u * (unsigned)kMidPer) + j * 64u + lane;
        masks[j] = __ballot(x < n && ((v[j].y >> cc) & 1u));
        t += (unsigned)__popcll(masks[j]);
    }
    if (lane == 0u) L.wave_cnt[wave] = t;
    __syncthreads();
    unsigned before = 0, ttot = 0;
#pragma unroll
    for (unsigned w = 0; w < (unsigned)kMidThreads / 64u; ++w) { const unsigned cw = L.wave_cnt[w]; if (w < wave) before += cw; ttot += cw; }
    unsigned run = before;
#pragma unroll
    for (int j = 0; j < kMidPer; ++j) {
        const unsigned x = wave * (64u * (unsigned)kMidPer) + j * 64u + lane;
        if (x < n) {
            const bool p = (masks[j] >> lane) & 1ull;
            const unsigned tl = run + vd_mbcnt(masks[j]);
            if (p) L.tpos[s0 + (ttot - tl - 1u)] = (unsigned short)x;     // index T: (T+1)-th true from the right
            else L.fpos[s0 + (x - tl)] = (unsigned short)x;               // index F: (F+1)-th false from the left
        }
        run += (unsigned)__popcll(masks[j]);
    }
    __syncthreads();
    const unsigned ftot = n - ttot;
    unsigned dest[kMidPer];
    unsigned short tps[kMidPer], fps[kMidPer];
    run = before;
#pragma unroll
    for (int j = 0; j < kMidPer; ++j) {                  // the sixteen table reads of a lane in flight together
        const unsigned x = wave * (64u * (unsigned)kMidPer) + j * 64u + lane;
        const bool p = (masks[j] >> lane) & 1ull;
        const unsigned tl = run + vd_mbcnt(masks[j]);
        const unsigned F = x - tl, T = ttot - tl - (p ? 1u : 0u);
        tps[j] = L.tpos[s0 + ((x < n && F != 0u && F <= ttot) ? F - 1u : 0u)];
        fps[j] = L.fpos[s0 + ((x < n && T + 1u <= ftot) ? T : 0u)];
        run += (unsigned)__popcll(masks[j]);
    }
    run = before;
#pragma unroll
    for (int j = 0; j < kMidPer; ++j) {
        const unsigned x = wave * (64u * (unsigned)kMidPer) + j * 64u + lane;
        dest[j] = kNone;
        if (x < n) {
            const bool p = (masks[j] >> lane) & 1ull;
            const unsigned tl = run + vd_mbcnt(masks[j]);
            const unsigned F = x - tl, T = ttot - tl - (p ? 1u : 0u);
            const int tF = F == 0u ? (int)n : (F <= ttot ? (int)tps[j] : -1);
            const bool left = (int)x < tF;
            const unsigned fj = (T + 1u <= ftot) ? (unsigned)fps[j] : n;
            const unsigned fetch = left ? x + n - (unsigned)tF : (n - 1u - x) + fj + 1u;
            const bool is_u = fetch == n - 1u;
            unsigned d;
            if (is_u) d = ttot - (p ? 1u : 0u);
            else if (left) d = p ? x : (unsigned)tF - 1u;
            else d = p ? fj : x - 1u;
            dest[j] = d;
            if (is_u && c >= 0) { L.u_pay[c] = v[j]; L.u_p[c] = p ? 1u : 0u; L.ttot[c] = ttot; }
        }
        run += (unsigned)__popcll(masks[j]);
    }
    // (no barrier here: every read of L.pay happened before the first barrier of this shuffle, and the tables read above are
    //  not written again before the barrier below)
#pragma unroll
    for (int j = 0; j < kMidPer; ++j)
        if (dest[j] != kNone) L.pay[s0 + dest[j]] = v[j];
    __syncthreads();
}

__global__ __launch_bounds__(kMidThreads, (VD_MID_WGS * kMidThreads / 64 + 3) / 4) void blas_mid_kernel(const MidRoot* __restrict__ roots, const unsigned* __restrict__ n_roots_p,
                                                               unsigned* __restrict__ ids32, ArrSet set0, ArrSet set1, LevelCtl* ctl, TopNode* top,
                                                               SmallRoot* small, unsigned top_cap, unsigned small_cap, unsigned first) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MidLds& L = *reinterpret_cast<MidLds*>(smem);
    if (first + blockIdx.x >= *n_roots_p) return;
    const MidRoot root = roots[first + blockIdx.x];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // the segment's data sits, in position order, in the set its last phase-A level wrote (root.pad & 1)
    const f32x4* __restrict__ cent = (root.pad & 1u) ? set1.cent : set0.cent;
    const TriBox* __restrict__ boxes = (root.pad & 1u) ? set1.boxes : set0.boxes;
    for (unsigned x = tid; x < root.count; x += kMidThreads) { const u32x2 v = {root.start + x, 0u}; L.pay[x] = v; }
    if (tid == 0) {
        L.n_stack = 1;
        L.stack[0].s0 = 0; L.stack[0].n = root.count; L.stack[0].node = root.node;
        for (int q = 0; q < 6; ++q) L.stack[0].cbk[q] = root.cbk[q];
    }
    __syncthreads();
    while (L.n_stack > 0u) {
        const MidNode nd = L.stack[L.n_stack - 1u];
        __syncthreads();
        const unsigned s0 = nd.s0, n = nd.n;
        if (tid == 0) L.n_stack -= 1u;
        // planes (blas.rs:142-146) and per-node resets
        if (tid < (unsigned)kCand) {
            float cbmin[3], cbmax[3];
            for (int k = 0; k < 3; ++k) { cbmin[k] = box_lo(nd.cbk[k]); cbmax[k] = box_hi(nd.cbk[3 + k]); }
            L.pos[tid] = cand_pos(cbmin, cbmax, (int)tid);
        }
        for (unsigned i2 = tid; i2 < 144u * 16u; i2 += kMidThreads) (&L.bins16[0][0])[i2] = ((i2 >> 4) % 6u) < 3u ? kBig : -kBig - 1;
        if (tid >= 192u && tid < 192u + 24u) L.child_k[tid - 192u] = ((tid - 192u) % 6u) < 3u ? kBig : -kBig - 1;
        __syncthreads();
        // predicate bits of the 21 planes.  (Measured and not kept: 2 / 4 / 8 elements per lane and step with their gathers in
        // flight together, here and in the bin and child loops - the 128-register budget of four workgroups per CU spills
        // 143 registers instead of 36: 28.6 ms per build against 27.9; three or two workgroups per CU: 28.6 / 28.7.)
        for (unsigned x = tid; x < n; x += kMidThreads) {
            u32x2 v = L.pay[s0 + x];
            const f32x4 c = cent[v.x];
            unsigned bits = 0;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                bits |= (c.x < L.pos[k] ? 1u : 0u) << k;
                bits |= (c.y < L.pos[7 + k] ? 1u : 0u) << (7 + k);
                bits |= (c.z < L.pos[14 + k] ? 1u : 0u) << (14 + k);
            }
            v.y = bits;
            L.pay[s0 + x] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < kCand; ++c) mid_shuffle(L, s0, n, (unsigned)c, c);
        // bins over the non-u elements
        for (unsigned x = tid; x < n; x += kMidThreads) {
            const u32x2 v = L.pay[s0 + x];
            bool is_u = false;
#pragma unroll
            for (int c = 0; c < kCand; ++c) is_u |= L.u_pay[c].x == v.x;
            if (is_u) continue;
            const TriBox bx = boxes[v.x];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int b = 7 - __popc((v.y >> (7 * a)) & 0x7fu);
                int* row = &L.bins16[(a * 8 + b) * 6][lane & 15u];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    atomicMin(row + 16 * q, vd_key(bx.mn[q]));
                    atomicMax(row + 16 * (3 + q), vd_key(bx.mx[q]));
                }
            }
        }
        __syncthreads();
        if (tid < 144u) {                              // the 16 copies of an entry -> the [axis][bin][3] arrays eval_candidate reads
            const bool is_min = (tid % 6u) < 3u;
            int v = L.bins16[tid][0];
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) v = is_min ? min(v, L.bins16[tid][k2]) : max(v, L.bins16[tid][k2]);
            if (is_min) (&L.bin_min[0][0][0])[(tid / 6u) * 3u + tid % 6u] = v; else (&L.bin_max[0][0][0])[(tid / 6u) * 3u + (tid % 6u - 3u)] = v;
        }
        __syncthreads();
        if (wave == 0u) {
            eval_stage(L.eval_u, L.u_pay, boxes, lane);
            vd_u64 key = ~0ull;
            if (lane < (unsigned)kCand) {
                const int c = (int)lane, a = c / 7, k = c % 7 + 1;
                EvalIn in{&L.bin_min[a][0][0], &L.bin_max[a][0][0], &L.eval_u, L.eval_u.id[c]};
                const unsigned n1 = L.ttot[c] - L.u_p[c];
                key = cost_key(eval_candidate(in, a, k, L.pos[c], n1, n), (unsigned)c);
            }
            key = wave_min_u64(key);
            if (lane == 0u) {
                if (key == ~0ull) { atomicOr(&ctl->err, ERR_DEGENERATE); L.best = 0; L.Lst = 1; }
                else { L.best = (unsigned)key; L.Lst = L.ttot[L.best] - L.u_p[L.best]; }
            }
        }
        __syncthreads();
        mid_shuffle(L, s0, n, L.best, -1);
        // children: vertex boxes (blas.rs:115-123) and centroid boxes (the children's `cb`)
        const unsigned Lst = L.Lst;
        {
            int k24[24];
#pragma unroll
            for (int i = 0; i < 24; ++i) k24[i] = (i % 6) < 3 ? kBig : -kBig - 1;
            for (unsigned x = tid; x < n; x += kMidThreads) {
                const unsigned id = L.pay[s0 + x].x;
                const TriBox bx = boxes[id];
                const f32x4 c = cent[id];
                const float ce[3] = {c.x, c.y, c.z};
                const int o = x < Lst ? 0 : 6;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    k24[o + q] = min(k24[o + q], vd_key(bx.mn[q]));
                    k24[o + 3 + q] = max(k24[o + 3 + q], vd_key(bx.mx[q]));
                    k24[12 + o + q] = min(k24[12 + o + q], vd_key_lo(ce[q]));
                    k24[12 + o + 3 + q] = max(k24[12 + o + 3 + q], vd_key_hi(ce[q]));
                }
            }
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const bool is_min = (i % 6) < 3;
                const int r = is_min ? wave_min_i(k24[i]) : wave_max_i(k24[i]);
                if (lane == 0u) { if (is_min) atomicMin(&L.child_k[i], r); else atomicMax(&L.child_k[i], r); }
            }
        }
        __syncthreads();
        // emit the two children (as finalize_segment does)
        if (tid == 0) {
            const unsigned pair = atomicAdd(&ctl->n_top, 2u);
            if (pair + 2u > top_cap) { atomicOr(&ctl->err, 4u); L.n_stack = 0; }
            else {
                top[nd.node].kind = 1u;
                top[nd.node].left = pair;
                for (int side = 0; side < 2; ++side) {
                    TopNode t;
                    for (int q = 0; q < 3; ++q) { t.mn[q] = box_lo(L.child_k[side * 6 + q]); t.mx[q] = box_hi(L.child_k[side * 6 + 3 + q]); }
                    const unsigned ls = side == 0 ? s0 : s0 + Lst;
                    t.start = root.start + ls;
                    t.count = side == 0 ? Lst : n - Lst;
                    t.left = 0; t.small = 0; t.pad = (root.pad & 1u) | 2u;       // | 2: positions -> pos0 through ids32
                    if (t.count <= 3u) {
                        t.kind = 0u;
                    } else if (t.count <= (unsigned)kSmallMax) {
                        t.kind = 2u;
                        const unsigned si = atomicAdd(&ctl->n_small, 1u);
                        if (si < small_cap) small[si] = SmallRoot{t.start, t.count, pair + side, (root.pad & 1u) | 2u};
                        else atomicOr(&ctl->err, 4u);
                        t.small = si;
                    } else {
                        t.kind = 1u;
                        MidNode& m = L.stack[L.n_stack++];        // depth <= log2(kMidMax / kSmallMax) + 1 pending nodes
                        m.s0 = ls; m.n = t.count; m.node = pair + side;
                        for (int q = 0; q < 6; ++q) m.cbk[q] = L.child_k[12 + side * 6 + q];
                    }
                    top[pair + side] = t;
                }
            }
        }
        __syncthreads();
    }
    for (unsigned x = tid; x < root.count; x += kMidThreads) ids32[root.start + x] = L.pay[x].x;
}

// =============================================================================================
// Phase C: copy-out.
// =============================================================================================
struct TopOut { unsigned final_index, pair, mesh; };   // pair = final left_first for interior nodes; mesh: whose node array

__global__ void c_top_kernel(const TopNode* top, const TopOut* tout, unsigned n_top, const MeshDesc* __restrict__ meshes) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_top) return;
    const TopNode t = top[i];
    const MeshDesc md = meshes[tout[i].mesh];
    VdBvhNode n;
    if (t.kind == 3u) {                            // tmp slot 2m + 1 mirrors the reference's never-used node 1: all-zero (blas.rs:52,90)
        memset(&n, 0, sizeof(n));
        md.out[1] = n;
        return;
    }
    for (int q = 0; q < 3; ++q) { n.min[q] = t.mn[q]; n.max[q] = t.mx[q]; }
    if (t.kind == 0u) { n.left_first = t.start - md.base; n.count = t.count; }      // leaf: position in the MESH's index buffer
    else { n.left_first = tout[i].pair; n.count = 0u; }
    md.out[tout[i].final_index] = n;
}

__global__ __launch_bounds__(256) void c_sub_kernel(const SmallRoot* roots, const unsigned* sub_interior, const unsigned* root_pair,
                                                    unsigned n_roots, const TmpNode* subnodes, const unsigned short* submap,
                                                    const TopOut* tout, const MeshDesc* __restrict__ meshes) {
    const unsigned r = blockIdx.x;
    if (r >= n_roots) return;
    const MeshDesc md = meshes[tout[roots[r].top_node].mesh];
    const TmpNode* src = subnodes + 2u * (size_t)roots[r].start;
    const unsigned short* nmap = submap + 2u * (size_t)roots[r].start;
    const unsigned n_nodes = 2u * sub_interior[r], off = root_pair[r];
    for (unsigned j = threadIdx.x; j < n_nodes; j += 256u) {
        const TmpNode t = src[j];
        VdBvhNode n;
        for (int q = 0; q < 3; ++q) { n.min[q] = t.mn[q]; n.max[q] = t.mx[q]; }
        n.count = t.count;
        n.left_first = t.count == 0u ? t.left_first + off : t.left_first - md.base;   // interior: local pair -> final pair; leaf: batch position -> mesh position
        md.out[off + nmap[j]] = n;
    }
}

__global__ void c_ids_big_leaves_kernel(const TopNode* top, unsigned n_top, const unsigned* ids32, ArrSet set0, ArrSet set1, unsigned* final_ids) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_top) return;
    const TopNode t = top[i];
    if (t.kind != 0u) return;
    const f32x4* cent = (t.pad & 1u) ? set1.cent : set0.cent;
    for (unsigned k = 0; k < t.count; ++k) {
        const unsigned pos0 = (t.pad & 2u) ? ids32[t.start + k] : t.start + k;
        final_ids[t.start + k] = __float_as_uint(cent[pos0].w);
    }
}

// blas.rs:95-100 per mesh: indices[i] = old_indices[tri_ids[i]]; ids and the copy are batch-wide, the result is the mesh's own buffer
__global__ __launch_bounds__(256) void c_permute_kernel(const unsigned* __restrict__ final_ids, const unsigned* __restrict__ idx_in,
                                                        const MeshDesc* __restrict__ meshes, unsigned n_meshes) {
    __shared__ unsigned s_m;
    if (threadIdx.x == 0) s_m = mesh_of_chunk(meshes, n_meshes, blockIdx.x);
    __syncthreads();
    const MeshDesc md = meshes[s_m];
    const unsigned chunk = blockIdx.x - md.chunk0;
#pragma unroll 2
    for (int r = 0; r < kPreTris; ++r) {
        const unsigned i = (chunk * (unsigned)kPreTris + (unsigned)r) * 256u + threadIdx.x;
        if (i >= md.n_tri) break;
        const unsigned t = final_ids[md.base + i];
        md.idx_out[3u * (size_t)i] = idx_in[3u * (size_t)t];
        md.idx_out[3u * (size_t)i + 1] = idx_in[3u * (size_t)t + 1];
        md.idx_out[3u * (size_t)i + 2] = idx_in[3u * (size_t)t + 2];
    }
}

// one thread per mesh: its root (tmp node 2m; 2m + 1 is the reference's unused slot) and the tier that takes it
__global__ void c_root_kernel(TopNode* top, const int* root_keys, const MeshDesc* __restrict__ meshes, unsigned n_meshes, SmallRoot* small,
                              LevelCtl* ctl, Seg* segs, MidRoot* mid) {
    const unsigned m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_meshes) return;
    if (m == 0u) ctl->n_top = 2u * n_meshes;      // the level loop appends child pairs behind the roots
    const int* rk = root_keys + 16u * m;
    const unsigned n_tri = meshes[m].n_tri, base = meshes[m].base;
    TopNode t;
    for (int q = 0; q < 3; ++q) { t.mn[q] = box_lo(rk[q]); t.mx[q] = box_hi(rk[3 + q]); }
    t.start = base; t.count = n_tri; t.left = 0; t.small = 0; t.pad = 0;
    if (n_tri <= 3u) t.kind = 0u;
    else if (n_tri <= (unsigned)kSmallMax) {
        t.kind = 2u;
        const unsigned si = atomicAdd(&ctl->n_small, 1u);
        small[si] = SmallRoot{base, n_tri, 2u * m, 0u};
        t.small = si;
    } else if (n_tri <= (unsigned)kMidMax) {
        t.kind = 1u;
        MidRoot& r = mid[atomicAdd(&ctl->n_mid, 1u)];
        r.start = base; r.count = n_tri; r.node = 2u * m; r.pad = 0;
        for (int q = 0; q < 6; ++q) r.cbk[q] = rk[6 + q];
    } else {
        t.kind = 1u;
        Seg& sg = segs[atomicAdd(&ctl->n_seg, 1u)];
        sg.start = base; sg.count = n_tri; sg.node = 2u * m;
        for (int q = 0; q < 6; ++q) sg.cbk[q] = rk[6 + q];
        atomicMax(&ctl->max_count, n_tri);
        atomicAdd(&ctl->active, n_tri);
    }
    top[2u * m] = t;
    TopNode z; memset(&z, 0, sizeof(z));
    z.kind = 3u;
    top[2u * m + 1u] = z;
}

struct Arena {
    char* base; size_t off;
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};

// One mesh of a build, host side.  d_out == nullptr: packed - the mesh's nodes follow the previous mesh's in the batch's
// shared node buffer (MeshPool's `bvh_index = bvh_nodes.len()` bookkeeping, mesh/mod.rs:320-345).
struct BuildMesh {
    const float* d_verts; uint32_t n_vert; uint32_t* d_idx; uint32_t n_tri; VdBvhNode* d_out; uint32_t node_cap;
    uint32_t out_n_nodes, out_first;
};

int bvh_build_batch_impl(VdCtx* ctx, BuildMesh* hm, uint32_t K, VdBvhNode* d_packed, uint64_t packed_cap, uint32_t packed_first,
                         uint32_t* out_failed_mesh) {
    size_t T = 0;
    unsigned n_chunks = 0;
    for (uint32_t m = 0; m < K; ++m) { T += hm[m].n_tri; n_chunks += (hm[m].n_tri + kChunk - 1u) / kChunk; }
    if (T > 0x3fffffffu) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_bvh_build: more than 2^30 - 1 triangles in one build");
    const uint32_t n_tri = (uint32_t)T;
    const unsigned seg_cap = (unsigned)(T / kSmallMax + 2);
    const unsigned mid_cap = seg_cap;
    const unsigned item_cap = (unsigned)(T / kItem + seg_cap + 2);
    const unsigned small_cap = (unsigned)(T / 4 + 2);
    const unsigned top_cap = (unsigned)(2 * (size_t)small_cap + 4 * (size_t)seg_cap + 2 * (size_t)K + 64);
    const bool wide_pay = T > kPay4Max || ctx->option(VD_OPT_BLAS_WIDE_PAYLOAD, 0) != 0;
    // ---- scratch layout ----
    Arena probe{nullptr, 0};
    auto layout = [&](Arena& a, bool) {
        struct P { u32x2 *pay0, *pay1; f32x4 *cent, *cent1; TriBox *boxes, *boxes1; unsigned *bits21, *ids32;
                   unsigned *falsepos, *truepos, *final_ids, *stack, *idx_copy;
                   unsigned char* is_u; Seg *seg0, *seg1; MidRoot* mid; unsigned *item_seg, *item_cnt, *item_cnt1, *item_pre; TopNode* top; SmallRoot* small;
                   unsigned* sub_interior; TmpNode* subnodes; unsigned short* submap; LevelCtl* ctl; int* root_keys; TopOut* tout; unsigned* root_pair;
                   MeshDesc* meshes; unsigned* bad_mesh; unsigned long long* cls_prof; } p;
        // payload ping-pong: 4 bytes per triangle up to 2^25 triangles, 8 beyond (allocated for the width in use)
        const size_t pay_words = wide_pay ? T : (T + 1) / 2;
        p.pay0 = a.take<u32x2>(pay_words); p.pay1 = a.take<u32x2>(pay_words);
        p.cent = a.take<f32x4>(T); p.boxes = a.take<TriBox>(T); p.cent1 = a.take<f32x4>(T); p.boxes1 = a.take<TriBox>(T);
        p.bits21 = a.take<unsigned>(T); p.ids32 = a.take<unsigned>(T);
        p.falsepos = a.take<unsigned>(T); p.truepos = a.take<unsigned>(T);
        p.final_ids = a.take<unsigned>(T); p.stack = a.take<unsigned>(T); p.idx_copy = a.take<unsigned>(3 * T);
        p.is_u = a.take<unsigned char>(T);
        p.seg0 = a.take<Seg>(seg_cap); p.seg1 = a.take<Seg>(seg_cap); p.mid = a.take<MidRoot>(mid_cap);
        p.item_seg = a.take<unsigned>(item_cap); p.item_cnt = a.take<unsigned>(item_cap); p.item_cnt1 = a.take<unsigned>(item_cap); p.item_pre = a.take<unsigned>(item_cap + 1);
        p.top = a.take<TopNode>(top_cap); p.small = a.take<SmallRoot>(small_cap); p.sub_interior = a.take<unsigned>(small_cap);
        p.subnodes = a.take<TmpNode>(2 * T + 2); p.submap = a.take<unsigned short>(2 * T + 2); p.ctl = a.take<LevelCtl>(1); p.root_keys = a.take<int>(16 * (size_t)K);
        p.tout = a.take<TopOut>(top_cap); p.root_pair = a.take<unsigned>(small_cap);
        p.meshes = a.take<MeshDesc>(K); p.bad_mesh = a.take<unsigned>(1);
        p.cls_prof = a.take<unsigned long long>(64 * 64);  // written by the -DVD_TUNING build only: 64 replicas of 64 counters
        return p;
    };
    (void)layout(probe, false);
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, probe.off + 256);
    if (rc) return rc;
    Arena arena{reinterpret_cast<char*>(ctx->scratch), 0};
    auto P = layout(arena, true);
    hipStream_t st = ctx->stream;
    VdBvhBuildStats& stats = ctx->bvh_stats;
    stats = VdBvhBuildStats{};
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](float& slot) {
        const auto now = std::chrono::steady_clock::now();
        slot += std::chrono::duration<float, std::milli>(now - t_mark).count();
        t_mark = now;
    };

    vd_time_begin(ctx);
    // pinned staging, first use: [MeshDesc K][root keys 16 K] (phase C lays it out again for its own tables)
    const size_t off_keys = sizeof(MeshDesc) * (size_t)K;
    {
        int rc_s = vd_ensure_host(ctx, off_keys + 64 * (size_t)K + 64);
        if (rc_s) return rc_s;
    }
    MeshDesc* h_desc = reinterpret_cast<MeshDesc*>(ctx->host_stage);
    {
        int* h_keys = reinterpret_cast<int*>(reinterpret_cast<char*>(ctx->host_stage) + off_keys);
        unsigned base = 0, chunk0 = 0;
        for (uint32_t m = 0; m < K; ++m) {
            h_desc[m] = MeshDesc{hm[m].d_verts, hm[m].d_idx, hm[m].d_idx, hm[m].d_out, hm[m].n_vert, hm[m].n_tri, base, chunk0};
            base += hm[m].n_tri; chunk0 += (hm[m].n_tri + kChunk - 1u) / kChunk;
            static const int init[16] = {kBig, kBig, kBig, -kBig - 1, -kBig - 1, -kBig - 1, kBig, kBig, kBig, -kBig - 1, -kBig - 1, -kBig - 1, 0, 0, 0, 0};
            memcpy(h_keys + 16 * (size_t)m, init, sizeof(init));
        }
        VD_HIP_CHECK(ctx, hipMemcpyAsync(P.meshes, h_desc, sizeof(MeshDesc) * (size_t)K, hipMemcpyHostToDevice, st));
        VD_HIP_CHECK(ctx, hipMemcpyAsync(P.root_keys, h_keys, 64 * (size_t)K, hipMemcpyHostToDevice, st));
        VD_HIP_CHECK(ctx, hipMemsetAsync(P.ctl, 0, sizeof(LevelCtl), st));
        VD_HIP_CHECK(ctx, hipMemsetAsync(P.bad_mesh, 0xff, 4, st));
        VD_HIP_CHECK(ctx, hipMemsetAsync(P.is_u, 0, T, st));
    }
#ifdef VD_TUNING
    VD_HIP_CHECK(ctx, hipMemsetAsync(P.cls_prof, 0, 64 * 64 * sizeof(unsigned long long), st));
#endif
    hipLaunchKernelGGL(blas_precompute_kernel, dim3(n_chunks), dim3(256), 0, st, P.meshes, K, P.idx_copy, P.cent, P.boxes, P.root_keys, &P.ctl->err, P.bad_mesh);
    const ArrSet sets[2] = {ArrSet{P.cent, P.boxes}, ArrSet{P.cent1, P.boxes1}};
    hipLaunchKernelGGL(c_root_kernel, dim3((K + 63u) / 64u), dim3(64), 0, st, P.top, P.root_keys, P.meshes, K, P.small, P.ctl, P.seg0, P.mid);
    hipLaunchKernelGGL(a_boundary_kernel, dim3(1), dim3(1024), 0, st, (const Seg*)nullptr, P.seg0, P.ctl, P.top, P.small, P.is_u, top_cap, small_cap, P.mid, mid_cap, 0u,
                       P.item_seg, 0);      // the first level's set-up

    // ---- phase A: level loop ----
    Seg* seg_cur = P.seg0; Seg* seg_next = P.seg1;
    LevelCtl h_ctl;
    unsigned h_bad = kNone;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(&h_ctl, P.ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
    VD_HIP_CHECK(ctx, hipMemcpyAsync(&h_bad, P.bad_mesh, 4, hipMemcpyDeviceToHost, st));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (h_ctl.err & ERR_BAD_INDEX) {
        if (out_failed_mesh) *out_failed_mesh = h_bad;
        snprintf(ctx->err, sizeof(ctx->err), "vd_bvh_build: index >= n_vert (mesh %u of the build)", h_bad);
        return VD_ERR_INVALID_ARG;
    }
    unsigned n_seg = h_ctl.n_seg;
    int levels = 0;
    stats.kernel_launches = 3;
    lap(stats.ms_precompute);
    // one level of phase A; PayT = Pay4 / Pay8 (see there)
    unsigned n_launch = 0;
    auto run_level = [&](auto pay_tag, unsigned n_seg_now, size_t active_bound, Seg* seg_cur, Seg* seg_next, int level, bool scan_free) {
        using PayT = decltype(pay_tag);
        typedef typename PayT::T PT;
        const ArrSet cur = sets[level & 1], nxt = sets[(level + 1) & 1];
        // upper bound of items this level: sum ceil(count/kItem) <= active/kItem + n_seg, `active` = the triangles still in
        // segments of phase A (it only falls from level to level; the last levels of an uneven tree hold a few segments, and a
        // grid of 8 200 workgroups costs 4.9 us to start even if every one of them leaves at once: profiles/r05_blas_issue_cost.log)
        const unsigned items_ub = (unsigned)(std::min<size_t>(T, active_bound) / kItem) + n_seg_now + 1;
        PT* src = reinterpret_cast<PT*>(P.pay0); PT* dst = reinterpret_cast<PT*>(P.pay1);
        unsigned* const cnt[2] = {P.item_cnt, P.item_cnt1};      // round c counts in cnt[c & 1]
        hipLaunchKernelGGL((a_bits_kernel<PayT>), dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, src, cur.cent, P.bits21, cnt[0]);
        for (int c = 0; c <= kCand; ++c) {
            const int cc = c < kCand ? c : -1;       // -1: final re-shuffle with each segment's best plane
            if (c == kCand) {
                hipLaunchKernelGGL((a_bin_kernel<PayT>), dim3((items_ub + kBinItems - 1) / kBinItems), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, src, cur.boxes,
                                   P.is_u, P.bits21);
                hipLaunchKernelGGL(a_eval_kernel, dim3(n_seg_now), dim3(64), 0, st, seg_cur, P.ctl, cur.boxes);
            }
            // the true counts of round c: from a_bits (c = 0), from the a_apply of round c - 1, or - the final shuffle, whose
            // plane a_eval has only just chosen - from a_count, which also puts that axis' bits into the 4-byte payload
            if (c == kCand)
                hipLaunchKernelGGL((a_count_kernel<PayT>), dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, src, cc, cnt[c & 1], P.bits21, 1);
            const int mode = c + 1 >= kCand ? 0 : ((c + 1) % 7 == 0 ? 2 : 1);
            unsigned* const cnt_next = mode ? cnt[(c + 1) & 1] : nullptr;
            // prefixes: from a_scan, or - segments of <= 1024 items - added up by the workgroups themselves (prefix_begin)
            const unsigned* pre = scan_free ? cnt[c & 1] : P.item_pre;
            if (!scan_free) hipLaunchKernelGGL(a_scan_kernel, dim3(1), dim3(1024), 0, st, seg_cur, P.ctl, cnt[c & 1], P.item_pre);
            auto ranks = scan_free ? a_ranks_kernel<PayT, true> : a_ranks_kernel<PayT, false>;
            const RoundK rk = make_round<PayT>(cc);
            hipLaunchKernelGGL(ranks, dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, src, rk, pre, P.falsepos, P.truepos, cnt_next);
            auto apply = scan_free ? (mode == 0 ? a_apply_kernel<PayT, 0, true> : mode == 1 ? a_apply_kernel<PayT, 1, true> : a_apply_kernel<PayT, 2, true>)
                                   : (mode == 0 ? a_apply_kernel<PayT, 0, false> : mode == 1 ? a_apply_kernel<PayT, 1, false> : a_apply_kernel<PayT, 2, false>);
            hipLaunchKernelGGL(apply, dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, src, dst, rk, pre,
                               P.falsepos, P.truepos, P.is_u, P.bits21, cnt_next);
            n_launch += scan_free ? 2u : 3u;
            PT* t = src; src = dst; dst = t;
        }
        // 22 swaps: the arrangement is back in pay0
        hipLaunchKernelGGL((a_child_kernel<PayT>), dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, reinterpret_cast<const PT*>(P.pay0), cur.boxes,
                           cur.cent, nxt.boxes, nxt.cent);
        // this level's children + the next level's set-up: one launch (a_boundary_kernel)
        hipLaunchKernelGGL(a_boundary_kernel, dim3((n_seg_now + kFinalChunk - 1u) / kFinalChunk), dim3(1024), 0, st, seg_cur, seg_next, P.ctl, P.top, P.small, P.is_u,
                           top_cap, small_cap, P.mid, mid_cap, (unsigned)((level + 1) & 1), P.item_seg, 1);
    };
    // The mid tier does not wait for the last levels: segments <= kMidMax go to the mid list as they appear, and after
    // every level the roots listed since the last launch start on the second stream, beside the levels that remain
    // (those stream through HBM, the mid tier computes out of LDS); the rest of the list follows on the main stream.  Disjoint position ranges, shared
    // lists appended by atomics - and no numbering depends on the order in which top nodes or small roots were listed.
    unsigned mid_early = 0;
    struct AuxJoin {          // an error return must not leave the early launch running over scratch the next call reuses
        hipStream_t s = nullptr;
        ~AuxJoin() { if (s) (void)hipStreamSynchronize(s); }
    } aux_join;
    auto launch_mid = [&](hipStream_t on, unsigned first, unsigned count) -> int {
        if (!ctx->mid_lds_opt_in) {   // per context (= per device), not per process
            VD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(blas_mid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MidLds)));
            ctx->mid_lds_opt_in = true;
        }
        hipLaunchKernelGGL(blas_mid_kernel, dim3(count), dim3(kMidThreads), sizeof(MidLds), on, P.mid, &P.ctl->n_mid, P.ids32, sets[0], sets[1],
                           P.ctl, P.top, P.small, top_cap, small_cap, first);
        stats.kernel_launches += 1;
        return VD_OK;
    };
    // The host stays ONE LEVEL AHEAD of the device: level L + 1 is queued - its grids sized by what level L's exact counts
    // allow (at most twice as many segments, none larger than level L's largest) - before the host waits for the control
    // words level L leaves behind, so the device never idles for the round trip (~60 us per level before).  The kernels
    // read the true counts from the control words; the level queued after the last real one finds no items and falls
    // through.  Control words travel into pinned memory (two slots) behind an event each.
    if (n_seg > 0) {
        if (!ctx->lvl_pinned) VD_HIP_CHECK(ctx, hipHostMalloc(&ctx->lvl_pinned, 2 * sizeof(LevelCtl)));
        for (int e = 0; e < 2; ++e)
            if (!ctx->ev_lvl[e]) VD_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_lvl[e], hipEventDisableTiming));
        LevelCtl* h_pin = reinterpret_cast<LevelCtl*>(ctx->lvl_pinned);
        struct MainJoin {        // an error return must not leave queued levels running over scratch the next call reuses
            hipStream_t s = nullptr;
            ~MainJoin() { if (s) (void)hipStreamSynchronize(s); }
        } main_join;
        main_join.s = st;
        auto queue_level = [&](unsigned n_seg_bound, unsigned max_count_bound, size_t active_bound) {
            const bool scan_free = (max_count_bound + kItem - 1) / kItem <= 1024u;
            if (wide_pay) run_level(Pay8{}, n_seg_bound, active_bound, seg_cur, seg_next, levels, scan_free);
            else run_level(Pay4{}, n_seg_bound, active_bound, seg_cur, seg_next, levels, scan_free);
            Seg* t = seg_cur; seg_cur = seg_next; seg_next = t;
            stats.kernel_launches += 5 + n_launch; n_launch = 0;
            levels += 1;
        };
        queue_level(n_seg, h_ctl.max_count, h_ctl.active);   // level 0: exact
        unsigned bound = (unsigned)std::min<size_t>(2 * (size_t)n_seg, seg_cap), max_bound = h_ctl.max_count;
        size_t active_bound = h_ctl.active;
        for (;;) {
            const int slot = (levels - 1) & 1;               // the control words the last queued level leaves
            VD_HIP_CHECK(ctx, hipMemcpyAsync(&h_pin[slot], P.ctl, sizeof(LevelCtl), hipMemcpyDeviceToHost, st));
            VD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_lvl[slot], st));
            if (levels > 4096) VD_FAIL(ctx, VD_ERR_HIP, "vd_bvh_build: level loop did not terminate");
            queue_level(bound, max_bound, active_bound);     // one level ahead
            VD_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_lvl[slot]));
            h_ctl = h_pin[slot];
            if (h_ctl.err & ERR_DEGENERATE)
                VD_FAIL(ctx, VD_ERR_DEGENERATE, "vd_bvh_build: every split candidate rejected (the reference builder crashes on this input)");
            if (h_ctl.err) VD_FAIL(ctx, VD_ERR_HIP, "vd_bvh_build: internal capacity exceeded");
            n_seg = h_ctl.n_seg;
            if (n_seg == 0) { levels -= 1; break; }          // the level just queued is empty
            bound = (unsigned)std::min<size_t>(2 * (size_t)n_seg, seg_cap); max_bound = h_ctl.max_count; active_bound = h_ctl.active;
            if (h_ctl.n_mid >= mid_early + kMidEarlyMin) {   // enough new roots to be worth a launch beside the levels in flight
                if (!ctx->aux_stream) VD_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
                if (!ctx->ev_aux) VD_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming));
                const int rc_m = launch_mid(ctx->aux_stream, mid_early, h_ctl.n_mid - mid_early);    // these roots are complete: the event above is behind their level
                if (rc_m) return rc_m;
                VD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_aux, ctx->aux_stream));
                mid_early = h_ctl.n_mid;
                aux_join.s = ctx->aux_stream;
            }
        }
        main_join.s = nullptr;
    }
    stats.levels_phase_a = (uint32_t)levels;
    lap(stats.ms_phase_a);

    // ---- mid tier: segments of kSmallMax < n <= kMidMax, one workgroup each, down to small roots ----
    if (h_ctl.n_mid) {
        if (h_ctl.n_mid > mid_early) { const int rc_m = launch_mid(st, mid_early, h_ctl.n_mid - mid_early); if (rc_m) return rc_m; }
        if (mid_early) VD_HIP_CHECK(ctx, hipStreamWaitEvent(st, ctx->ev_aux, 0));
        VD_HIP_CHECK(ctx, hipMemcpyAsync(&h_ctl, P.ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
        VD_HIP_CHECK(ctx, hipStreamSynchronize(st));
        if (h_ctl.err & ERR_DEGENERATE)
            VD_FAIL(ctx, VD_ERR_DEGENERATE, "vd_bvh_build: every split candidate rejected (the reference builder crashes on this input)");
        if (h_ctl.err) VD_FAIL(ctx, VD_ERR_HIP, "vd_bvh_build: internal capacity exceeded");
        aux_join.s = nullptr;     // the main stream waited for the early launch and the host for the main stream
    }
    lap(stats.ms_mid);
    // ---- phase B ----
    const unsigned n_small = h_ctl.n_small, n_top = h_ctl.n_top;
    stats.n_top_nodes = n_top; stats.n_small_roots = n_small; stats.n_mid_roots = h_ctl.n_mid;
    if (n_small) {
        hipLaunchKernelGGL(b_order_kernel, dim3(1), dim3(1024), 0, st, P.small, &P.ctl->n_small, P.root_pair);   // root_pair is free until phase C
        hipLaunchKernelGGL(blas_small_kernel, dim3(n_small), dim3(64 * kSubWaves), sizeof(WaveLds) + sizeof(WaveQueues), st, P.small, &P.ctl->n_small, P.ids32,
                           sets[0], sets[1], P.subnodes, P.submap, P.sub_interior, P.final_ids, &P.ctl->err, P.stack, P.root_pair, P.cls_prof);
        ctx->dbg_ptr = P.stack; ctx->dbg_count = 2 * n_small; ctx->dbg_ptr2 = P.cls_prof;
    }
    // ---- phase C: DFS numbering of the top tree on the host ----
    // The top tree is final before phase B starts (the stream was synchronised after the last level / the mid tier), so
    // it is fetched on a second stream and walked WHILE phase B runs; what depends on phase B - how many node pairs
    // each small subtree takes - enters in one linear pass over the recorded pre-order afterwards.  Pinned staging:
    // [TopNode n_top][TopOut n_top][order n_top][sub n_small][root_pair n_small]
    const size_t off_out = sizeof(TopNode) * (size_t)n_top, off_ord = off_out + sizeof(TopOut) * (size_t)n_top;
    const size_t off_sub = off_ord + 4 * (size_t)n_top, off_rp = off_sub + 4 * (size_t)(n_small ? n_small : 1);
    const size_t off_desc = (off_rp + 4 * (size_t)(n_small ? n_small : 1) + 15) & ~(size_t)15;
    int rc_h = vd_ensure_host(ctx, off_desc + sizeof(MeshDesc) * (size_t)K);
    if (rc_h) return rc_h;
    if (!ctx->aux_stream) VD_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    char* hs = reinterpret_cast<char*>(ctx->host_stage);
    TopNode* h_top = reinterpret_cast<TopNode*>(hs);
    TopOut* h_out = reinterpret_cast<TopOut*>(hs + off_out);
    unsigned* h_ord = reinterpret_cast<unsigned*>(hs + off_ord);
    unsigned* h_sub = reinterpret_cast<unsigned*>(hs + off_sub);
    unsigned* h_root_pair = reinterpret_cast<unsigned*>(hs + off_rp);
    h_desc = reinterpret_cast<MeshDesc*>(hs + off_desc);          // the staging may have moved: written again below
    VD_HIP_CHECK(ctx, hipMemcpyAsync(h_top, P.top, sizeof(TopNode) * n_top, hipMemcpyDeviceToHost, ctx->aux_stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->aux_stream));
    // pre-order walk of every mesh's tree, one after the other: an interior node takes the next pair when visited
    // (blas.rs:110-112); h_ord records the visits, mesh boundaries in ord_end
    std::vector<unsigned> ord_end(K);
    unsigned n_ord = 0;
    {
        std::vector<unsigned> stk;
        for (uint32_t m = 0; m < K; ++m) {
            stk.push_back(2u * m);
            while (!stk.empty()) {
                const unsigned v = stk.back(); stk.pop_back();
                const TopNode& t = h_top[v];
                h_out[v].mesh = m;
                if (t.kind == 0u) continue;
                h_ord[n_ord++] = v;
                if (t.kind == 1u) {
                    stk.push_back(t.left + 1);
                    stk.push_back(t.left);
                }
            }
            h_out[2u * m + 1u].mesh = m; h_out[2u * m + 1u].final_index = 1u; h_out[2u * m + 1u].pair = 0u;
            ord_end[m] = n_ord;
        }
    }
    if (n_small) VD_HIP_CHECK(ctx, hipMemcpyAsync(h_sub, P.sub_interior, 4 * (size_t)n_small, hipMemcpyDeviceToHost, st));
    VD_HIP_CHECK(ctx, hipMemcpyAsync(&h_ctl, P.ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (h_ctl.err & ERR_DEGENERATE)
        VD_FAIL(ctx, VD_ERR_DEGENERATE, "vd_bvh_build: every split candidate rejected (the reference builder crashes on this input)");
    if (h_ctl.err & ERR_INTERNAL) VD_FAIL(ctx, VD_ERR_HIP, "vd_bvh_build: phase B work list stalled");
    lap(stats.ms_phase_b);
    uint64_t packed_at = packed_first;
    {
        unsigned k = 0, base = 0, chunk0 = 0;
        for (uint32_t m = 0; m < K; ++m) {
            unsigned pool = 2;
            h_out[2u * m].final_index = 0;
            for (; k < ord_end[m]; ++k) {
                const unsigned v = h_ord[k];
                const TopNode& t = h_top[v];
                const unsigned pair = pool;
                h_out[v].pair = pair;
                if (t.kind == 2u) {
                    h_root_pair[t.small] = pair;
                    pool += 2u * h_sub[t.small];
                } else {
                    pool += 2;
                    h_out[t.left].final_index = pair;
                    h_out[t.left + 1].final_index = pair + 1;
                }
            }
            VdBvhNode* out = hm[m].d_out;
            hm[m].out_n_nodes = pool; hm[m].out_first = 0;
            if (out) {
                if (pool > hm[m].node_cap) {
                    if (out_failed_mesh) *out_failed_mesh = m;
                    snprintf(ctx->err, sizeof(ctx->err), "vd_bvh_build: node_cap too small (mesh %u of the build needs %u nodes)", m, pool);
                    return VD_ERR_INVALID_ARG;
                }
            } else {
                if (packed_at + pool > packed_cap || packed_at + pool > 0xffffffffull) {     // out_first_node / bvh_index are 32-bit
                    if (out_failed_mesh) *out_failed_mesh = m;
                    snprintf(ctx->err, sizeof(ctx->err), "vd_bvh_build_batch: the packed node buffer is full at mesh %u (or its node indices pass 2^32)", m);
                    return VD_ERR_INVALID_ARG;
                }
                out = d_packed + packed_at;
                hm[m].out_first = (uint32_t)packed_at;
                packed_at += pool;
            }
            h_desc[m] = MeshDesc{hm[m].d_verts, hm[m].d_idx, hm[m].d_idx, out, hm[m].n_vert, hm[m].n_tri, base, chunk0};
            base += hm[m].n_tri; chunk0 += (hm[m].n_tri + kChunk - 1u) / kChunk;
        }
    }
    VD_HIP_CHECK(ctx, hipMemcpyAsync(P.meshes, h_desc, sizeof(MeshDesc) * (size_t)K, hipMemcpyHostToDevice, st));      // now with every mesh's node array
    VD_HIP_CHECK(ctx, hipMemcpyAsync(P.tout, h_out, sizeof(TopOut) * n_top, hipMemcpyHostToDevice, st));
    if (n_small) VD_HIP_CHECK(ctx, hipMemcpyAsync(P.root_pair, h_root_pair, 4 * (size_t)n_small, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(c_top_kernel, dim3((n_top + 63) / 64), dim3(64), 0, st, P.top, P.tout, n_top, P.meshes);
    if (n_small) hipLaunchKernelGGL(c_sub_kernel, dim3(n_small), dim3(256), 0, st, P.small, P.sub_interior, P.root_pair, n_small, P.subnodes, P.submap, P.tout, P.meshes);
    hipLaunchKernelGGL(c_ids_big_leaves_kernel, dim3((n_top + 63) / 64), dim3(64), 0, st, P.top, n_top, P.ids32, sets[0], sets[1], P.final_ids);
    hipLaunchKernelGGL(c_permute_kernel, dim3(n_chunks), dim3(256), 0, st, P.final_ids, P.idx_copy, P.meshes, K);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    VD_HIP_CHECK(ctx, hipStreamSynchronize(st));   // the pinned staging is reused by the next build
    stats.kernel_launches += (n_small ? 2 : 0) + 3;
    lap(stats.ms_phase_c);
    (void)n_tri;
    return VD_OK;
}

int check_build_args(VdCtx* ctx, const void* verts, uint32_t n_vert, const void* idx, uint32_t n_tri, const void* out,
                     uint32_t node_cap, const void* out_n) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!verts || !idx || !out || !out_n || n_tri == 0 || n_vert == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_bvh_build: null pointer or zero count");
    if (n_tri > 0x3fffffffu || node_cap < 2u) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_bvh_build: bad sizes");
    return VD_OK;
}

}  // namespace

extern "C" {

int vd_bvh_last_build_stats(const VdCtx* ctx, VdBvhBuildStats* out) {
    if (!ctx || !out) return VD_ERR_INVALID_ARG;
    *out = ctx->bvh_stats;
    return VD_OK;
}

#ifdef VD_TUNING
// Tuning hook (make tuning -> libvoidin_hip_tuning.so; not in the product library): per-subtree {cycles, prims} pairs
// of the last phase B run.
int vd_debug_blas_cycles(VdCtx* ctx, uint32_t* out, uint32_t cap) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx || !ctx->dbg_ptr) return 0;
    const uint32_t n = ctx->dbg_count < cap ? ctx->dbg_count : cap;
    if (hipMemcpy(out, ctx->dbg_ptr, 4 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return (int)n;
}
// ... and the 64 wave-cycle counters of its phase B by node size class and step (blas_small_kernel, cls_prof)
int vd_debug_blas_small_classes(VdCtx* ctx, uint64_t* out64) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx || !ctx->dbg_ptr2 || !out64) return VD_ERR_INVALID_ARG;
    static uint64_t rep[64 * 64];
    if (hipMemcpy(rep, ctx->dbg_ptr2, sizeof(rep), hipMemcpyDeviceToHost) != hipSuccess) return VD_ERR_HIP;
    for (int k = 0; k < 64; ++k) { out64[k] = 0; for (int r = 0; r < 64; ++r) out64[k] += rep[64 * r + k]; }
    return VD_OK;
}
#endif

int vd_bvh_build_dev(VdCtx* ctx, const float* d_verts, uint32_t n_vert, uint32_t* d_idx, uint32_t n_tri, VdBvhNode* d_out,
                     uint32_t node_cap, uint32_t* out_n_nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_build_args(ctx, d_verts, n_vert, d_idx, n_tri, d_out, node_cap, out_n_nodes);
    if (rc) return rc;
    BuildMesh one{d_verts, n_vert, d_idx, n_tri, d_out, node_cap, 0u, 0u};
    rc = bvh_build_batch_impl(ctx, &one, 1u, nullptr, 0, 0u, nullptr);
    if (rc) return rc;
    *out_n_nodes = one.out_n_nodes;
    return VD_OK;
}

int vd_bvh_build(VdCtx* ctx, const float* verts, uint32_t n_vert, uint32_t* idx, uint32_t n_tri, VdBvhNode* out,
                 uint32_t node_cap, uint32_t* out_n_nodes) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    int rc = check_build_args(ctx, verts, n_vert, idx, n_tri, out, node_cap, out_n_nodes);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t vb = (size_t)n_vert * 12, ib = (size_t)n_tri * 12;
    const size_t nb = sizeof(VdBvhNode) * (size_t)node_cap;
    rc = vd_ensure(ctx, &ctx->stage_in, &ctx->stage_in_bytes, vb + 256);
    if (rc) return rc;
    rc = vd_ensure(ctx, &ctx->stage_aux, &ctx->stage_aux_bytes, ib + 256);
    if (rc) return rc;
    rc = vd_ensure(ctx, &ctx->stage_out, &ctx->stage_out_bytes, nb + 256);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->stage_in, verts, vb, hipMemcpyHostToDevice, ctx->stream));
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->stage_aux, idx, ib, hipMemcpyHostToDevice, ctx->stream));
    BuildMesh one{reinterpret_cast<const float*>(ctx->stage_in), n_vert, reinterpret_cast<uint32_t*>(ctx->stage_aux), n_tri,
                  reinterpret_cast<VdBvhNode*>(ctx->stage_out), node_cap, 0u, 0u};
    rc = bvh_build_batch_impl(ctx, &one, 1u, nullptr, 0, 0u, nullptr);
    if (rc) return rc;
    *out_n_nodes = one.out_n_nodes;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(idx, ctx->stage_aux, ib, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipMemcpyAsync(out, ctx->stage_out, sizeof(VdBvhNode) * (size_t)*out_n_nodes, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VD_OK;
}

// K meshes in ONE build (header: "Batched BLAS build").
static int batch_args(VdCtx* ctx, const VdBvhBatchItem* items, uint32_t n_items, const VdBvhNode* packed, uint64_t packed_cap, uint32_t packed_first,
                      std::vector<BuildMesh>& hm) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!items || n_items == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_bvh_build_batch: no items");
    if (n_items > (1u << 24)) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_bvh_build_batch: more than 2^24 meshes");
    hm.resize(n_items);
    uint64_t total = 0;
    for (uint32_t m = 0; m < n_items; ++m) {
        const VdBvhBatchItem& it = items[m];
        if (!it.verts_xyz || !it.indices_inout || it.n_tri == 0 || it.n_vert == 0) {
            snprintf(ctx->err, sizeof(ctx->err), "vd_bvh_build_batch: item %u: null pointer or zero count", m);
            return VD_ERR_INVALID_ARG;
        }
        if (it.out_nodes ? it.node_cap < 2u : (!packed || packed_first > packed_cap)) {
            snprintf(ctx->err, sizeof(ctx->err), "vd_bvh_build_batch: item %u: no room for its nodes (out_nodes with node_cap >= 2, or a packed buffer)", m);
            return VD_ERR_INVALID_ARG;
        }
        total += it.n_tri;
        hm[m] = BuildMesh{it.verts_xyz, it.n_vert, it.indices_inout, it.n_tri, it.out_nodes, it.node_cap, 0u, 0u};
    }
    if (total > 0x3fffffffu) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_bvh_build_batch: more than 2^30 - 1 triangles in one build");
    return VD_OK;
}

int vd_bvh_build_batch_dev(VdCtx* ctx, VdBvhBatchItem* items, uint32_t n_items, VdBvhNode* d_packed_nodes, uint64_t packed_cap,
                           uint32_t packed_first, uint32_t* out_packed_end) {
    VdDeviceGuard vd_guard_(ctx);
    std::vector<BuildMesh> hm;
    int rc = batch_args(ctx, items, n_items, d_packed_nodes, packed_cap, packed_first, hm);
    if (rc) return rc;
    uint32_t failed = 0xffffffffu;
    rc = bvh_build_batch_impl(ctx, hm.data(), n_items, d_packed_nodes, packed_cap, packed_first, &failed);
    uint32_t end = packed_first;
    for (uint32_t m = 0; m < n_items; ++m) {
        items[m].out_n_nodes = rc ? 0u : hm[m].out_n_nodes;
        items[m].out_first_node = rc ? 0u : hm[m].out_first;
        items[m].status = rc ? (failed == 0xffffffffu || failed == m ? rc : VD_OK) : VD_OK;
        if (!rc && !items[m].out_nodes) end = hm[m].out_first + hm[m].out_n_nodes;
    }
    if (out_packed_end) *out_packed_end = end;
    return rc;
}

int vd_bvh_build_batch(VdCtx* ctx, VdBvhBatchItem* items, uint32_t n_items, VdBvhNode* packed_nodes, uint64_t packed_cap,
                       uint32_t packed_first, uint32_t* out_packed_end) {
    VdDeviceGuard vd_guard_(ctx);
    std::vector<BuildMesh> hm;
    int rc = batch_args(ctx, items, n_items, packed_nodes, packed_cap, packed_first, hm);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // staging: [vertices of all meshes][indices of all meshes] in, [nodes: 2 * n_tri per mesh, the reference's own bound] out
    size_t vb = 0, ib = 0, nb = 0;
    for (uint32_t m = 0; m < n_items; ++m) {
        vb += ((size_t)items[m].n_vert * 12 + 255) & ~(size_t)255;
        ib += ((size_t)items[m].n_tri * 12 + 255) & ~(size_t)255;
        nb += sizeof(VdBvhNode) * (2 * (size_t)items[m].n_tri + 2);
    }
    rc = vd_ensure(ctx, &ctx->stage_in, &ctx->stage_in_bytes, vb + 256);
    if (rc) return rc;
    rc = vd_ensure(ctx, &ctx->stage_aux, &ctx->stage_aux_bytes, ib + 256);
    if (rc) return rc;
    rc = vd_ensure(ctx, &ctx->stage_out, &ctx->stage_out_bytes, nb + 256);
    if (rc) return rc;
    char* dv = reinterpret_cast<char*>(ctx->stage_in); char* di = reinterpret_cast<char*>(ctx->stage_aux);
    VdBvhNode* dn = reinterpret_cast<VdBvhNode*>(ctx->stage_out);
    for (uint32_t m = 0; m < n_items; ++m) {
        VD_HIP_CHECK(ctx, hipMemcpyAsync(dv, items[m].verts_xyz, (size_t)items[m].n_vert * 12, hipMemcpyHostToDevice, ctx->stream));
        VD_HIP_CHECK(ctx, hipMemcpyAsync(di, items[m].indices_inout, (size_t)items[m].n_tri * 12, hipMemcpyHostToDevice, ctx->stream));
        hm[m].d_verts = reinterpret_cast<const float*>(dv); hm[m].d_idx = reinterpret_cast<uint32_t*>(di);
        hm[m].d_out = dn; hm[m].node_cap = 2u * items[m].n_tri + 2u;
        dv += ((size_t)items[m].n_vert * 12 + 255) & ~(size_t)255;
        di += ((size_t)items[m].n_tri * 12 + 255) & ~(size_t)255;
        dn += 2 * (size_t)items[m].n_tri + 2;
    }
    uint32_t failed = 0xffffffffu;
    rc = bvh_build_batch_impl(ctx, hm.data(), n_items, nullptr, 0, 0u, &failed);
    uint64_t at = packed_first;
    for (uint32_t m = 0; m < n_items && !rc; ++m) {        // room on the caller's side
        const uint32_t need = hm[m].out_n_nodes;
        if (items[m].out_nodes ? need > items[m].node_cap : (at + need > packed_cap || at + need > 0xffffffffull)) {   // 32-bit node indices
            snprintf(ctx->err, sizeof(ctx->err), "vd_bvh_build_batch: item %u needs %u nodes: node_cap / packed_cap too small", m, need);
            rc = VD_ERR_INVALID_ARG; failed = m;
        }
        if (!items[m].out_nodes) at += need;
    }
    at = packed_first;
    for (uint32_t m = 0; m < n_items; ++m) {
        items[m].status = rc ? (failed == 0xffffffffu || failed == m ? rc : VD_OK) : VD_OK;
        items[m].out_n_nodes = rc ? 0u : hm[m].out_n_nodes;
        items[m].out_first_node = 0u;
        if (rc) continue;
        VdBvhNode* dst = items[m].out_nodes;
        if (!dst) { dst = packed_nodes + at; items[m].out_first_node = (uint32_t)at; at += hm[m].out_n_nodes; }
        VD_HIP_CHECK(ctx, hipMemcpyAsync(items[m].indices_inout, hm[m].d_idx, (size_t)items[m].n_tri * 12, hipMemcpyDeviceToHost, ctx->stream));
        VD_HIP_CHECK(ctx, hipMemcpyAsync(dst, hm[m].d_out, sizeof(VdBvhNode) * (size_t)hm[m].out_n_nodes, hipMemcpyDeviceToHost, ctx->stream));
    }
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (out_packed_end) *out_packed_end = (uint32_t)at;
    return rc;
}

}  // extern "C"

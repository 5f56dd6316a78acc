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
// Phases: A (segments > kSmallMax prims): level-synchronous, many workgroups per segment, the
// arrangement ping-pongs between two 16-B payload arrays in HBM.  B (<= kSmallMax): one wave
// builds the whole subtree in DFS order out of LDS, which yields the reference's pre-order node
// numbering locally.  C: DFS numbering of the (small) top tree on the host, parallel copy-out.
#include "vd_common.hpp"

#include <vector>

namespace {

constexpr int kSmallMax = 512;          // largest segment built by one wave out of LDS
constexpr int kChunks = kSmallMax / 64;
constexpr int kCand = 21;               // 3 axes x 7 planes (blas.rs:144-145; `bins` hard-coded to 8)
constexpr int kBig = 0x7fffffff;
constexpr int kItem = 1024;             // phase A: positions per workgroup item (256 lanes x 4)
constexpr unsigned kNone = 0xffffffffu;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct TriBox { float mn[3]; float pad0; float mx[3]; float pad1; };   // 32 B, by triangle id
struct TmpNode { float mn[3]; unsigned left_first; float mx[3]; unsigned count; };   // == VdBvhNode layout

enum : unsigned { ERR_DEGENERATE = 1u, ERR_BAD_INDEX = 2u };

__device__ __forceinline__ float pay_c(const u32x4& v, int axis) {
    return __uint_as_float(axis == 0 ? v.y : (axis == 1 ? v.z : v.w));
}

// MAX_DIST-seeded bounds (blas.rs:185-186): `min = min(1e30, ..)`, `max = max(-1e30, ..)`
__device__ __forceinline__ float box_lo(int key) { return vd_unkey(min(key, vd_key(1e30f))); }
__device__ __forceinline__ float box_hi(int key) { return vd_unkey(max(key, vd_key(-1e30f))); }

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

// Split plane of candidate c (axis = c / 7, k = c % 7 + 1): glam lerp = min + (max - min) * (k/8)
__device__ __forceinline__ float cand_pos(const float* cbmin, const float* cbmax, int c) {
    const int axis = c / 7, k = c % 7 + 1;
    const float scale = (float)k / 8.0f;
    return cbmin[axis] + (cbmax[axis] - cbmin[axis]) * scale;
}

// ---- precompute: centroid payload + triangle boxes + root box -------------------------------
__global__ __launch_bounds__(256) void blas_precompute_kernel(const float* __restrict__ verts, const unsigned* __restrict__ idx,
                                                              unsigned n_tri, unsigned n_vert, u32x4* __restrict__ payload,
                                                              TriBox* __restrict__ boxes, int* __restrict__ root_keys /*[6]*/,
                                                              unsigned* __restrict__ err) {
    __shared__ int s_k[6];
    if (threadIdx.x < 6) s_k[threadIdx.x] = threadIdx.x < 3 ? kBig : -kBig - 1;
    __syncthreads();
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t < n_tri) {
        const unsigned i0 = idx[3u * (size_t)t], i1 = idx[3u * (size_t)t + 1], i2 = idx[3u * (size_t)t + 2];
        if (i0 >= n_vert || i1 >= n_vert || i2 >= n_vert) {
            atomicOr(err, ERR_BAD_INDEX);
        } else {
            const float* a = verts + 3u * (size_t)i0; const float* b = verts + 3u * (size_t)i1; const float* c = verts + 3u * (size_t)i2;
            TriBox bx; float ce[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ce[k] = ((a[k] + b[k]) + c[k]) / 3.0f;          // blas.rs:80
                bx.mn[k] = vd_min_to(vd_min_to(a[k], b[k]), c[k]);
                bx.mx[k] = vd_max_to(vd_max_to(a[k], b[k]), c[k]);
                atomicMin(&s_k[k], vd_key(bx.mn[k]));
                atomicMax(&s_k[3 + k], vd_key(bx.mx[k]));
            }
            bx.pad0 = bx.pad1 = 0.0f;
            u32x4 p = {t, __float_as_uint(ce[0]), __float_as_uint(ce[1]), __float_as_uint(ce[2])};
            payload[t] = p;
            boxes[t] = bx;
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&root_keys[threadIdx.x], s_k[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&root_keys[threadIdx.x], s_k[threadIdx.x]);
}

// Cost of one candidate from binned statistics + the held-out `u` elements (blas.rs:149-155).
// bins: [8][3] keys of the candidate's axis (non-u elements only); u list: payload + box.
struct EvalIn {
    const int* bin_min; const int* bin_max;      // [8][3] for this axis
    const u32x4* u_pay; const TriBox* boxes; int n_u; unsigned own_u;   // own_u = id of this candidate's u
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
    for (int j = 0; j < in.n_u; ++j) {
        const u32x4 v = in.u_pay[j];
        bool dup = false;
        for (int i = 0; i < j; ++i) dup |= in.u_pay[i].x == v.x;
        if (dup) continue;
        const TriBox bx = in.boxes[v.x];
        const bool to_left = v.x != in.own_u && pay_c(v, axis) < pos;   // left = examined trues; u itself goes right
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int lo = vd_key(bx.mn[q]), hi = vd_key(bx.mx[q]);
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
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, off), hi = __shfl_xor((unsigned)(v >> 32), off);
        const vd_u64 o = ((vd_u64)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

// =============================================================================================
// Phase B: one wave builds the subtree of one small segment in DFS order out of LDS.
// =============================================================================================
struct SmallRoot { unsigned start, count, top_node, pad; };

struct WaveLds {
    u32x4 pay[2][kSmallMax];                // arrangement ping-pong: {id, cx, cy, cz}
    u32x4 u_pay[kCand + 1];
    int bin_min[3][8][3], bin_max[3][8][3];
    float pos[kCand + 3];
    unsigned ttot[kCand + 1], u_p[kCand + 1];
    unsigned short falsepos[kSmallMax + 2];
    unsigned short truepos[kSmallMax + 2];
};

// One closed-form shuffle of segment [s, s+n) with predicate centroid[axis] < pos: reads
// pay[src], writes pay[src^1].  Returns Ttot and u's payload / predicate uniformly.
__device__ __forceinline__ void wave_shuffle(WaveLds& L, int src, unsigned s, unsigned n, int axis, float pos,
                                             unsigned& out_ttot, u32x4& out_u, unsigned& out_up) {
    const unsigned lane = vd_lane();
    const unsigned n_chunks = (n + 63u) >> 6;
    unsigned long long masks[kChunks];
    unsigned ttot = 0;
#pragma unroll
    for (int ch = 0; ch < kChunks; ++ch) {
        masks[ch] = 0ull;
        if ((unsigned)ch < n_chunks) {
            const unsigned x = ch * 64u + lane;
            bool p = false;
            if (x < n) p = pay_c(L.pay[src][s + x], axis) < pos;
            masks[ch] = __ballot(p);
            ttot += (unsigned)__popcll(masks[ch]);
        }
    }
    const unsigned ftot = n - ttot;
    // rank -> position: falsepos[j] = j-th false from the left, truepos[j] = j-th true from the right (1-based)
    unsigned run = 0;
#pragma unroll
    for (int ch = 0; ch < kChunks; ++ch) {
        if ((unsigned)ch < n_chunks) {
            const unsigned x = ch * 64u + lane;
            if (x < n) {
                const bool p = (masks[ch] >> lane) & 1ull;
                const unsigned tl = run + vd_mbcnt(masks[ch]);
                if (p) L.truepos[ttot - tl] = (unsigned short)x;        // T + 1 = ttot - tl - 1 + 1
                else L.falsepos[x - tl + 1u] = (unsigned short)x;       // F + 1
            }
            run += (unsigned)__popcll(masks[ch]);
        }
    }
    vd_wave_lds_sync();
    run = 0;
    u32x4 uv = {0u, 0u, 0u, 0u};
    unsigned up = 0;
#pragma unroll
    for (int ch = 0; ch < kChunks; ++ch) {
        if ((unsigned)ch < n_chunks) {
            const unsigned x = ch * 64u + lane;
            bool is_u = false;
            u32x4 v = {0u, 0u, 0u, 0u};
            bool p = false;
            if (x < n) {
                p = (masks[ch] >> lane) & 1ull;
                const unsigned tl = run + vd_mbcnt(masks[ch]);
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
                v = L.pay[src][s + x];
                L.pay[src ^ 1][s + dest] = v;
            }
            const unsigned long long um = __ballot(is_u);
            if (um) {
                const int ul = __builtin_ctzll(um);
                uv.x = __shfl(v.x, ul); uv.y = __shfl(v.y, ul); uv.z = __shfl(v.z, ul); uv.w = __shfl(v.w, ul);
                up = __shfl(p ? 1u : 0u, ul);
            }
            run += (unsigned)__popcll(masks[ch]);
        }
    }
    vd_wave_lds_sync();
    out_ttot = ttot; out_u = uv; out_up = up;
}

constexpr int kSmallWaves = 2;   // waves (= subtrees) per workgroup: 2 x ~20 KB of LDS
__global__ __launch_bounds__(64 * kSmallWaves) void blas_small_kernel(const SmallRoot* __restrict__ roots, const unsigned* __restrict__ n_roots_p,
                                                         const u32x4* __restrict__ payload, const TriBox* __restrict__ boxes,
                                                         TmpNode* __restrict__ subnodes, unsigned* __restrict__ sub_interior,
                                                         unsigned* __restrict__ final_ids, unsigned* __restrict__ stack_mem,
                                                         unsigned* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane = vd_lane();
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds& L = *reinterpret_cast<WaveLds*>(smem + wave * sizeof(WaveLds));
    const unsigned root_i = blockIdx.x * (unsigned)kSmallWaves + wave;
    if (root_i >= *n_roots_p) return;
    const SmallRoot root = roots[root_i];
    const unsigned base = root.start, N = root.count;
    TmpNode* nodes = subnodes + 2u * (size_t)base;   // disjoint region per root: < 2*N nodes
    unsigned* stack = stack_mem + base;              // disjoint region per root: depth < N

    for (unsigned x = lane; x < N; x += 64u) L.pay[0][x] = payload[base + x];
    vd_wave_lds_sync();
    int cur = 0;
    unsigned pool = 0, n_interior = 0, sp = 0;
    // current node: id (kNone = the subtree root, which lives in the top tree), rel start, count
    unsigned node_id = kNone, s = 0, n = N;
    for (;;) {
        if (n <= 3u) {
            if (lane == 0) nodes[node_id].left_first = base + s;        // leaf: blas.rs:106-109
        } else {
            // ---- centroid bounds (blas.rs:142) and the 21 planes ----
            int kmn[3] = {kBig, kBig, kBig}, kmx[3] = {-kBig - 1, -kBig - 1, -kBig - 1};
            for (unsigned x = lane; x < n; x += 64u) {
                const u32x4 v = L.pay[cur][s + x];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int key = vd_key(pay_c(v, k));
                    kmn[k] = min(kmn[k], key); kmx[k] = max(kmx[k], key);
                }
            }
            float cbmin[3], cbmax[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { cbmin[k] = box_lo(wave_min_i(kmn[k])); cbmax[k] = box_hi(wave_max_i(kmx[k])); }
            if (lane < (unsigned)kCand) L.pos[lane] = cand_pos(cbmin, cbmax, (int)lane);
            for (unsigned i = lane; i < 72u; i += 64u) { (&L.bin_min[0][0][0])[i] = kBig; (&L.bin_max[0][0][0])[i] = -kBig - 1; }
            vd_wave_lds_sync();
            // ---- 21 trial shuffles (blas.rs:144-147) ----
            for (int c = 0; c < kCand; ++c) {
                unsigned tt, up; u32x4 uv;
                wave_shuffle(L, cur, s, n, c / 7, L.pos[c], tt, uv, up);
                cur ^= 1;
                if (lane == 0) { L.u_pay[c] = uv; L.u_p[c] = up; L.ttot[c] = tt; }
            }
            vd_wave_lds_sync();
            // ---- binning pass over the non-u elements ----
            for (unsigned x = lane; x < n; x += 64u) {
                const u32x4 v = L.pay[cur][s + x];
                bool is_u = false;
                for (int c = 0; c < kCand; ++c) is_u |= L.u_pay[c].x == v.x;
                if (is_u) continue;
                const TriBox bx = boxes[v.x];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float ce = pay_c(v, a);
                    int b = 0;
#pragma unroll
                    for (int k = 0; k < 7; ++k) b += !(ce < L.pos[a * 7 + k]);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        atomicMin(&L.bin_min[a][b][q], vd_key(bx.mn[q]));
                        atomicMax(&L.bin_max[a][b][q], vd_key(bx.mx[q]));
                    }
                }
            }
            vd_wave_lds_sync();
            // ---- evaluate: lane c owns candidate c (blas.rs:149-161) ----
            vd_u64 key = ~0ull;
            if (lane < (unsigned)kCand) {
                const int c = (int)lane, a = c / 7, k = c % 7 + 1;
                EvalIn in{&L.bin_min[a][0][0], &L.bin_max[a][0][0], L.u_pay, boxes, kCand, L.u_pay[c].x};
                const unsigned n1 = L.ttot[c] - L.u_p[c];
                key = cost_key(eval_candidate(in, a, k, L.pos[c], n1, n), (unsigned)c);
            }
            key = wave_min_u64(key);
            if (key == ~0ull) {                                          // every candidate rejected: SURVEY.md §8a B7
                if (lane == 0) atomicOr(err, ERR_DEGENERATE);
                return;
            }
            const int best = (int)(unsigned)key;
            const unsigned Lst = L.ttot[best] - L.u_p[best];             // stale optimal_pivot (blas.rs:159,165)
            // ---- final re-shuffle with the best plane, result discarded (blas.rs:164) ----
            {
                unsigned tt, up; u32x4 uv;
                wave_shuffle(L, cur, s, n, best / 7, L.pos[best], tt, uv, up);
                cur ^= 1;
            }
            // ---- children boxes from the actual arrangement (blas.rs:115-123) ----
            int lmn[3] = {kBig, kBig, kBig}, lmx[3] = {-kBig - 1, -kBig - 1, -kBig - 1};
            int rmn[3] = {kBig, kBig, kBig}, rmx[3] = {-kBig - 1, -kBig - 1, -kBig - 1};
            for (unsigned x = lane; x < n; x += 64u) {
                const TriBox bx = boxes[L.pay[cur][s + x].x];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int lo = vd_key(bx.mn[q]), hi = vd_key(bx.mx[q]);
                    if (x < Lst) { lmn[q] = min(lmn[q], lo); lmx[q] = max(lmx[q], hi); }
                    else { rmn[q] = min(rmn[q], lo); rmx[q] = max(rmx[q], hi); }
                }
            }
            TmpNode ln, rn;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                ln.mn[q] = box_lo(wave_min_i(lmn[q])); ln.mx[q] = box_hi(wave_max_i(lmx[q]));
                rn.mn[q] = box_lo(wave_min_i(rmn[q])); rn.mx[q] = box_hi(wave_max_i(rmx[q]));
            }
            const unsigned pair = pool;                                   // blas.rs:110-112, local numbering
            pool += 2; n_interior += 1;
            if (lane == 0) {
                // children keep (rel start, count) until they are visited
                ln.left_first = s; ln.count = Lst;
                rn.left_first = s + Lst; rn.count = n - Lst;
                nodes[pair] = ln; nodes[pair + 1] = rn;
                if (node_id != kNone) { nodes[node_id].left_first = pair; nodes[node_id].count = 0u; }   // blas.rs:112,127
                stack[sp] = pair + 1u;                                    // right after the whole left subtree
            }
            sp += 1;
            node_id = pair; n = Lst;                                      // descend left (s unchanged)
            continue;
        }
        // pop
        if (sp == 0u) break;
        sp -= 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        node_id = __builtin_amdgcn_readfirstlane(stack[sp]);
        s = __builtin_amdgcn_readfirstlane(nodes[node_id].left_first);
        n = __builtin_amdgcn_readfirstlane(nodes[node_id].count);
    }
    for (unsigned x = lane; x < N; x += 64u) final_ids[base + x] = L.pay[cur][x].x;
    if (lane == 0) sub_interior[root_i] = n_interior;
}

// =============================================================================================
// Phase A: level-synchronous emulation for segments larger than kSmallMax.
// =============================================================================================
struct Seg {
    unsigned start, count, node, item_first;
    unsigned n_items, best, Lst, ttot_cur;
    int cbk[6];                       // centroid-bound keys (min xyz, max xyz)
    int child_k[12];                  // children box keys: left min/max, right min/max
    float pos[kCand + 3];
    unsigned ttot[kCand + 3], u_p[kCand + 3];
    u32x4 u_pay[kCand + 1];
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

struct LevelCtl {                     // device-side counters
    unsigned n_seg, n_seg_next, n_items, n_top, n_small, err, pad0, pad1;
};

// one thread per segment: items per segment, centroid keys / bins reset
__global__ void a_seg_begin_kernel(Seg* segs, LevelCtl* ctl) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ctl->n_seg) return;
    Seg& sg = segs[i];
    sg.n_items = (sg.count + kItem - 1) / kItem;
    for (int k = 0; k < 3; ++k) { sg.cbk[k] = kBig; sg.cbk[3 + k] = -kBig - 1; }
    for (int k = 0; k < 3; ++k) { sg.child_k[k] = kBig; sg.child_k[3 + k] = -kBig - 1; sg.child_k[6 + k] = kBig; sg.child_k[9 + k] = -kBig - 1; }
    for (int k = 0; k < 72; ++k) { (&sg.bin_min[0][0][0])[k] = kBig; (&sg.bin_max[0][0][0])[k] = -kBig - 1; }
}

// single workgroup: exclusive scan of n_items over segments -> item_first, total items
__global__ __launch_bounds__(1024) void a_items_scan_kernel(Seg* segs, LevelCtl* ctl) {
    __shared__ unsigned s_part[1024];
    const unsigned n = ctl->n_seg, tid = threadIdx.x;
    const unsigned per = (n + 1023u) / 1024u;
    const unsigned lo = min(n, tid * per), hi = min(n, lo + per);
    unsigned sum = 0;
    for (unsigned i = lo; i < hi; ++i) sum += segs[i].n_items;
    s_part[tid] = sum;
    __syncthreads();
    for (unsigned off = 1; off < 1024u; off <<= 1) {
        const unsigned v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    unsigned run = s_part[tid] - sum;
    for (unsigned i = lo; i < hi; ++i) { segs[i].item_first = run; run += segs[i].n_items; }
    if (tid == 1023u) ctl->n_items = s_part[1023];
}

__global__ void a_items_fill_kernel(const Seg* segs, const LevelCtl* ctl, unsigned* item_seg) {
    const unsigned i = blockIdx.x;
    if (i >= ctl->n_seg) return;
    const Seg& sg = segs[i];
    for (unsigned k = threadIdx.x; k < sg.n_items; k += blockDim.x) item_seg[sg.item_first + k] = i;
}

// Item geometry: item -> (segment, first relative position, valid count).  Lane order inside an
// item is (wave, j, lane): position = rel0 + wave*256 + j*64 + lane.
struct ItemCtx { unsigned seg, rel0, n_here; };
__device__ __forceinline__ bool item_ctx(const Seg* segs, const unsigned* item_seg, const LevelCtl* ctl, ItemCtx& ic, Seg const*& sg) {
    if (blockIdx.x >= ctl->n_items) return false;
    ic.seg = item_seg[blockIdx.x];
    sg = segs + ic.seg;
    ic.rel0 = (blockIdx.x - sg->item_first) * kItem;
    ic.n_here = min((unsigned)kItem, sg->count - ic.rel0);
    return true;
}

// centroid bounds of each segment (blas.rs:142)
__global__ __launch_bounds__(256) void a_cb_kernel(Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                   const u32x4* __restrict__ pay) {
    __shared__ int s_k[6];
    ItemCtx ic; const Seg* sg;
    if (!item_ctx(segs, item_seg, ctl, ic, sg)) return;
    if (threadIdx.x < 6) s_k[threadIdx.x] = threadIdx.x < 3 ? kBig : -kBig - 1;
    __syncthreads();
    int kmn[3] = {kBig, kBig, kBig}, kmx[3] = {-kBig - 1, -kBig - 1, -kBig - 1};
    for (unsigned x = threadIdx.x; x < ic.n_here; x += 256u) {
        const u32x4 v = pay[sg->start + ic.rel0 + x];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int key = vd_key(pay_c(v, k)); kmn[k] = min(kmn[k], key); kmx[k] = max(kmx[k], key); }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = wave_min_i(kmn[k]), b = wave_max_i(kmx[k]);
        if ((threadIdx.x & 63u) == 0u) { atomicMin(&s_k[k], a); atomicMax(&s_k[3 + k], b); }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&segs[ic.seg].cbk[threadIdx.x], s_k[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&segs[ic.seg].cbk[threadIdx.x], s_k[threadIdx.x]);
}

__global__ void a_planes_kernel(Seg* segs, const LevelCtl* ctl) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ctl->n_seg) return;
    Seg& sg = segs[i];
    float cbmin[3], cbmax[3];
    for (int k = 0; k < 3; ++k) { cbmin[k] = box_lo(sg.cbk[k]); cbmax[k] = box_hi(sg.cbk[3 + k]); }
    for (int c = 0; c < kCand; ++c) sg.pos[c] = cand_pos(cbmin, cbmax, c);
}

// predicate mask of this lane group: returns ballot per j (4 per wave)
__device__ __forceinline__ void item_masks(const Seg* sg, const ItemCtx& ic, const u32x4* __restrict__ pay, int c,
                                           unsigned long long (&masks)[4], u32x4 (&vals)[4]) {
    const int cc = c >= 0 ? c : (int)sg->best;
    const int axis = cc / 7;
    const float pos = sg->pos[cc];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned x = wave * 256u + j * 64u + lane;
        bool p = false;
        if (x < ic.n_here) {
            vals[j] = pay[sg->start + ic.rel0 + x];
            p = pay_c(vals[j], axis) < pos;
        }
        masks[j] = __ballot(p);
    }
}

// round step 1: true count of every item
__global__ __launch_bounds__(256) void a_count_kernel(const Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                      const u32x4* __restrict__ pay, int c, unsigned* item_cnt) {
    __shared__ unsigned s_w[4];
    ItemCtx ic; const Seg* sg;
    if (!item_ctx(segs, item_seg, ctl, ic, sg)) return;
    unsigned long long masks[4]; u32x4 vals[4];
    item_masks(sg, ic, pay, c, masks, vals);
    unsigned t = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) t += (unsigned)__popcll(masks[j]);
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) item_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// round step 2 (single workgroup): exclusive scan of item counts; per-segment Ttot
__global__ __launch_bounds__(1024) void a_scan_kernel(Seg* segs, const LevelCtl* ctl, const unsigned* item_cnt, unsigned* item_pre) {
    __shared__ unsigned s_part[1024];
    const unsigned n = ctl->n_items, tid = threadIdx.x;
    const unsigned per = (n + 1023u) / 1024u;
    const unsigned lo = min(n, tid * per), hi = min(n, lo + per);
    unsigned sum = 0;
    for (unsigned i = lo; i < hi; ++i) sum += item_cnt[i];
    s_part[tid] = sum;
    __syncthreads();
    for (unsigned off = 1; off < 1024u; off <<= 1) {
        const unsigned v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    unsigned run = s_part[tid] - sum;
    for (unsigned i = lo; i < hi; ++i) { item_pre[i] = run; run += item_cnt[i]; }
    if (tid == 1023u) item_pre[n] = s_part[1023];
    __syncthreads();
    __threadfence_block();
    for (unsigned i = tid; i < ctl->n_seg; i += 1024u) {
        Seg& sg = segs[i];
        sg.ttot_cur = item_pre[sg.item_first + sg.n_items] - item_pre[sg.item_first];
    }
}

// round step 3: TL per position + rank -> position tables
__global__ __launch_bounds__(256) void a_ranks_kernel(const Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                      const u32x4* __restrict__ pay, int c, const unsigned* item_pre,
                                                      unsigned* __restrict__ tmp, unsigned* __restrict__ falsepos,
                                                      unsigned* __restrict__ truepos) {
    __shared__ unsigned s_w[4];
    ItemCtx ic; const Seg* sg;
    if (!item_ctx(segs, item_seg, ctl, ic, sg)) return;
    unsigned long long masks[4]; u32x4 vals[4];
    item_masks(sg, ic, pay, c, masks, vals);
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned t = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) t += (unsigned)__popcll(masks[j]);
    if (lane == 0u) s_w[wave] = t;
    __syncthreads();
    unsigned run = item_pre[blockIdx.x] - item_pre[sg->item_first];
    for (unsigned w = 0; w < wave; ++w) run += s_w[w];
    const unsigned ttot = sg->ttot_cur, s = sg->start;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned xr = wave * 256u + j * 64u + lane;
        if (xr < ic.n_here) {
            const unsigned x = ic.rel0 + xr;
            const bool p = (masks[j] >> lane) & 1ull;
            const unsigned tl = run + vd_mbcnt(masks[j]);
            tmp[s + x] = (tl << 1) | (p ? 1u : 0u);
            if (p) truepos[s + (ttot - tl - 1u)] = x;     // index T: (T+1)-th true from the right
            else falsepos[s + (x - tl)] = x;              // index F: (F+1)-th false from the left
        }
        run += (unsigned)__popcll(masks[j]);
    }
}

// round step 4: destinations, scatter, `u`
__global__ __launch_bounds__(256) void a_apply_kernel(Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                      const u32x4* __restrict__ src, u32x4* __restrict__ dst, int c,
                                                      const unsigned* __restrict__ tmp, const unsigned* __restrict__ falsepos,
                                                      const unsigned* __restrict__ truepos, unsigned char* __restrict__ is_u_flag) {
    ItemCtx ic; const Seg* sg;
    if (!item_ctx(segs, item_seg, ctl, ic, sg)) return;
    const unsigned n = sg->count, s = sg->start, ttot = sg->ttot_cur, ftot = n - ttot;
    for (unsigned xr = threadIdx.x; xr < ic.n_here; xr += 256u) {
        const unsigned x = ic.rel0 + xr;
        const unsigned tp = tmp[s + x];
        const bool p = tp & 1u;
        const unsigned tl = tp >> 1;
        const unsigned F = x - tl, T = ttot - tl - (p ? 1u : 0u);
        const long long tF = F == 0u ? (long long)n : (F <= ttot ? (long long)truepos[s + F - 1u] : -1ll);
        const bool left = (long long)x < tF;
        const unsigned fj = (T + 1u <= ftot) ? falsepos[s + T] : n;
        const unsigned fetch = left ? x + n - (unsigned)tF : (n - 1u - x) + fj + 1u;
        const bool is_u = fetch == n - 1u;
        unsigned dest;
        if (is_u) dest = ttot - (p ? 1u : 0u);
        else if (left) dest = p ? x : (unsigned)tF - 1u;
        else dest = p ? fj : x - 1u;
        const u32x4 v = src[s + x];
        dst[s + dest] = v;
        if (is_u && c >= 0) {
            Seg& w = segs[ic.seg];
            w.u_pay[c] = v; w.u_p[c] = p ? 1u : 0u; w.ttot[c] = ttot;
            is_u_flag[v.x] = 1;
        }
    }
}

// binning over the non-u elements (one pass per level)
__global__ __launch_bounds__(256) void a_bin_kernel(Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                    const u32x4* __restrict__ pay, const TriBox* __restrict__ boxes,
                                                    const unsigned char* __restrict__ is_u_flag) {
    __shared__ int s_min[3][8][3], s_max[3][8][3];
    __shared__ float s_pos[kCand + 3];
    ItemCtx ic; const Seg* sg;
    if (!item_ctx(segs, item_seg, ctl, ic, sg)) return;
    for (unsigned i = threadIdx.x; i < 72u; i += 256u) { (&s_min[0][0][0])[i] = kBig; (&s_max[0][0][0])[i] = -kBig - 1; }
    if (threadIdx.x < (unsigned)kCand) s_pos[threadIdx.x] = sg->pos[threadIdx.x];
    __syncthreads();
    for (unsigned xr = threadIdx.x; xr < ic.n_here; xr += 256u) {
        const u32x4 v = pay[sg->start + ic.rel0 + xr];
        if (is_u_flag[v.x]) continue;
        const TriBox bx = boxes[v.x];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float ce = pay_c(v, a);
            int b = 0;
#pragma unroll
            for (int k = 0; k < 7; ++k) b += !(ce < s_pos[a * 7 + k]);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                atomicMin(&s_min[a][b][q], vd_key(bx.mn[q]));
                atomicMax(&s_max[a][b][q], vd_key(bx.mx[q]));
            }
        }
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < 72u; i += 256u) {
        const int lo = (&s_min[0][0][0])[i], hi = (&s_max[0][0][0])[i];
        if (lo != kBig) atomicMin(&(&segs[ic.seg].bin_min[0][0][0])[i], lo);
        if (hi != -kBig - 1) atomicMax(&(&segs[ic.seg].bin_max[0][0][0])[i], hi);
    }
}

// one wave per segment: 21 costs -> best plane, stale pivot
__global__ __launch_bounds__(64) void a_eval_kernel(Seg* segs, LevelCtl* ctl, const TriBox* __restrict__ boxes) {
    if (blockIdx.x >= ctl->n_seg) return;
    Seg& sg = segs[blockIdx.x];
    const unsigned lane = threadIdx.x;
    vd_u64 key = ~0ull;
    if (lane < (unsigned)kCand) {
        const int c = (int)lane, a = c / 7, k = c % 7 + 1;
        EvalIn in{&sg.bin_min[a][0][0], &sg.bin_max[a][0][0], sg.u_pay, boxes, kCand, sg.u_pay[c].x};
        const unsigned n1 = sg.ttot[c] - sg.u_p[c];
        key = cost_key(eval_candidate(in, a, k, sg.pos[c], n1, sg.count), (unsigned)c);
    }
    key = wave_min_u64(key);
    if (lane == 0) {
        if (key == ~0ull) { atomicOr(&ctl->err, ERR_DEGENERATE); sg.best = 0; sg.Lst = 1; }
        else { sg.best = (unsigned)key; sg.Lst = sg.ttot[sg.best] - sg.u_p[sg.best]; }
    }
}

// children boxes from the final arrangement (blas.rs:115-123)
__global__ __launch_bounds__(256) void a_child_kernel(Seg* segs, const unsigned* item_seg, const LevelCtl* ctl,
                                                      const u32x4* __restrict__ pay, const TriBox* __restrict__ boxes) {
    __shared__ int s_k[12];
    ItemCtx ic; const Seg* sg;
    if (!item_ctx(segs, item_seg, ctl, ic, sg)) return;
    if (threadIdx.x < 12) s_k[threadIdx.x] = (threadIdx.x % 6) < 3 ? kBig : -kBig - 1;
    __syncthreads();
    int k12[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) k12[i] = (i % 6) < 3 ? kBig : -kBig - 1;
    const unsigned Lst = sg->Lst;
    for (unsigned xr = threadIdx.x; xr < ic.n_here; xr += 256u) {
        const unsigned x = ic.rel0 + xr;
        const TriBox bx = boxes[pay[sg->start + x].x];
        const int o = x < Lst ? 0 : 6;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            k12[o + q] = min(k12[o + q], vd_key(bx.mn[q]));
            k12[o + 3 + q] = max(k12[o + 3 + q], vd_key(bx.mx[q]));
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const bool is_min = (i % 6) < 3;
        const int r = is_min ? wave_min_i(k12[i]) : wave_max_i(k12[i]);
        if ((threadIdx.x & 63u) == 0u) { if (is_min) atomicMin(&s_k[i], r); else atomicMax(&s_k[i], r); }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const bool is_min = (threadIdx.x % 6) < 3;
        if (is_min) atomicMin(&segs[ic.seg].child_k[threadIdx.x], s_k[threadIdx.x]);
        else atomicMax(&segs[ic.seg].child_k[threadIdx.x], s_k[threadIdx.x]);
    }
}

// one thread per segment: emit the two children, classify them, clear u flags
__global__ void a_finalize_kernel(const Seg* segs, Seg* next, LevelCtl* ctl, TopNode* top, SmallRoot* small,
                                  unsigned char* is_u_flag, unsigned top_cap, unsigned small_cap) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ctl->n_seg) return;
    const Seg& sg = segs[i];
    for (int c = 0; c < kCand; ++c) is_u_flag[sg.u_pay[c].x] = 0;
    const unsigned pair = atomicAdd(&ctl->n_top, 2u);
    if (pair + 2u > top_cap) { atomicOr(&ctl->err, 4u); return; }
    top[sg.node].kind = 1u;
    top[sg.node].left = pair;
    for (int side = 0; side < 2; ++side) {
        TopNode t;
        for (int q = 0; q < 3; ++q) { t.mn[q] = box_lo(sg.child_k[side * 6 + q]); t.mx[q] = box_hi(sg.child_k[side * 6 + 3 + q]); }
        t.start = side == 0 ? sg.start : sg.start + sg.Lst;
        t.count = side == 0 ? sg.Lst : sg.count - sg.Lst;
        t.left = 0; t.small = 0; t.pad = 0;
        if (t.count <= 3u) {
            t.kind = 0u;
        } else if (t.count <= (unsigned)kSmallMax) {
            t.kind = 2u;
            const unsigned si = atomicAdd(&ctl->n_small, 1u);
            if (si < small_cap) small[si] = SmallRoot{t.start, t.count, pair + side, 0u};
            else atomicOr(&ctl->err, 4u);
            t.small = si;
        } else {
            t.kind = 1u;
            const unsigned ni = atomicAdd(&ctl->n_seg_next, 1u);
            Seg& ns = next[ni];
            ns.start = t.start; ns.count = t.count; ns.node = pair + side;
        }
        top[pair + side] = t;
    }
}

// =============================================================================================
// Phase C: copy-out.
// =============================================================================================
struct TopOut { unsigned final_index, pair; };   // pair = final left_first for interior nodes

__global__ void c_top_kernel(const TopNode* top, const TopOut* tout, unsigned n_top, VdBvhNode* out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_top || (i == 1u)) return;          // tmp slot 1 is unused (mirrors the reference's node 1)
    const TopNode t = top[i];
    VdBvhNode n;
    for (int q = 0; q < 3; ++q) { n.min[q] = t.mn[q]; n.max[q] = t.mx[q]; }
    if (t.kind == 0u) { n.left_first = t.start; n.count = t.count; }
    else { n.left_first = tout[i].pair; n.count = 0u; }
    out[tout[i].final_index] = n;
}

__global__ __launch_bounds__(256) void c_sub_kernel(const SmallRoot* roots, const unsigned* sub_interior, const unsigned* root_pair,
                                                    unsigned n_roots, const TmpNode* subnodes, VdBvhNode* out) {
    const unsigned r = blockIdx.x;
    if (r >= n_roots) return;
    const TmpNode* src = subnodes + 2u * (size_t)roots[r].start;
    const unsigned n_nodes = 2u * sub_interior[r], off = root_pair[r];
    for (unsigned j = threadIdx.x; j < n_nodes; j += 256u) {
        const TmpNode t = src[j];
        VdBvhNode n;
        for (int q = 0; q < 3; ++q) { n.min[q] = t.mn[q]; n.max[q] = t.mx[q]; }
        n.count = t.count;
        n.left_first = t.count == 0u ? t.left_first + off : t.left_first;   // interior: local pair -> final pair
        out[off + j] = n;
    }
}

__global__ void c_ids_big_leaves_kernel(const TopNode* top, unsigned n_top, const u32x4* pay, unsigned* final_ids) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_top || i == 1u) return;
    const TopNode t = top[i];
    if (t.kind != 0u) return;
    for (unsigned k = 0; k < t.count; ++k) final_ids[t.start + k] = pay[t.start + k].x;
}

__global__ void c_permute_kernel(const unsigned* __restrict__ final_ids, const unsigned* __restrict__ idx_in,
                                 unsigned* __restrict__ idx_out, unsigned n_tri) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tri) return;
    const unsigned t = final_ids[i];                                      // blas.rs:95-100
    idx_out[3u * (size_t)i] = idx_in[3u * (size_t)t];
    idx_out[3u * (size_t)i + 1] = idx_in[3u * (size_t)t + 1];
    idx_out[3u * (size_t)i + 2] = idx_in[3u * (size_t)t + 2];
}

__global__ void c_root_kernel(TopNode* top, const int* root_keys, unsigned n_tri, SmallRoot* small, LevelCtl* ctl, Seg* segs) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    TopNode t;
    for (int q = 0; q < 3; ++q) { t.mn[q] = box_lo(root_keys[q]); t.mx[q] = box_hi(root_keys[3 + q]); }
    t.start = 0; t.count = n_tri; t.left = 0; t.small = 0; t.pad = 0;
    ctl->n_top = 2; ctl->n_small = 0; ctl->n_seg = 0; ctl->n_seg_next = 0; ctl->n_items = 0;
    if (n_tri <= 3u) t.kind = 0u;
    else if (n_tri <= (unsigned)kSmallMax) { t.kind = 2u; small[0] = SmallRoot{0u, n_tri, 0u, 0u}; ctl->n_small = 1; }
    else { t.kind = 1u; segs[0].start = 0; segs[0].count = n_tri; segs[0].node = 0; ctl->n_seg = 1; }
    top[0] = t;
    TopNode z; memset(&z, 0, sizeof(z));
    top[1] = z;
}

__global__ void a_level_swap_kernel(LevelCtl* ctl) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { ctl->n_seg = ctl->n_seg_next; ctl->n_seg_next = 0; ctl->n_items = 0; }
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

int bvh_build_dev_impl(VdCtx* ctx, const float* d_verts, uint32_t n_vert, uint32_t* d_idx, uint32_t n_tri,
                       VdBvhNode* d_out, uint32_t node_cap, uint32_t* out_n_nodes) {
    const size_t T = n_tri;
    const unsigned seg_cap = (unsigned)(T / kSmallMax + 2);
    const unsigned item_cap = (unsigned)(T / kItem + seg_cap + 2);
    const unsigned small_cap = (unsigned)(T / 4 + 2);
    const unsigned top_cap = (unsigned)(2 * (size_t)small_cap + 4 * (size_t)seg_cap + 64);
    // ---- scratch layout ----
    Arena probe{nullptr, 0};
    auto layout = [&](Arena& a, bool) {
        struct P { u32x4 *pay0, *pay1; TriBox* boxes; unsigned *tmp, *falsepos, *truepos, *final_ids, *stack, *idx_copy;
                   unsigned char* is_u; Seg *seg0, *seg1; unsigned *item_seg, *item_cnt, *item_pre; TopNode* top; SmallRoot* small;
                   unsigned* sub_interior; TmpNode* subnodes; LevelCtl* ctl; int* root_keys; TopOut* tout; unsigned* root_pair; } p;
        p.pay0 = a.take<u32x4>(T); p.pay1 = a.take<u32x4>(T); p.boxes = a.take<TriBox>(T);
        p.tmp = a.take<unsigned>(T); p.falsepos = a.take<unsigned>(T); p.truepos = a.take<unsigned>(T);
        p.final_ids = a.take<unsigned>(T); p.stack = a.take<unsigned>(T); p.idx_copy = a.take<unsigned>(3 * T);
        p.is_u = a.take<unsigned char>(T);
        p.seg0 = a.take<Seg>(seg_cap); p.seg1 = a.take<Seg>(seg_cap);
        p.item_seg = a.take<unsigned>(item_cap); p.item_cnt = a.take<unsigned>(item_cap); p.item_pre = a.take<unsigned>(item_cap + 1);
        p.top = a.take<TopNode>(top_cap); p.small = a.take<SmallRoot>(small_cap); p.sub_interior = a.take<unsigned>(small_cap);
        p.subnodes = a.take<TmpNode>(2 * T + 2); p.ctl = a.take<LevelCtl>(1); p.root_keys = a.take<int>(8);
        p.tout = a.take<TopOut>(top_cap); p.root_pair = a.take<unsigned>(small_cap);
        return p;
    };
    (void)layout(probe, false);
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, probe.off + 256);
    if (rc) return rc;
    Arena arena{reinterpret_cast<char*>(ctx->scratch), 0};
    auto P = layout(arena, true);
    hipStream_t st = ctx->stream;

    vd_time_begin(ctx);
    {
        int h_keys[8] = {kBig, kBig, kBig, -kBig - 1, -kBig - 1, -kBig - 1, 0, 0};
        VD_HIP_CHECK(ctx, hipMemcpyAsync(P.root_keys, h_keys, sizeof(h_keys), hipMemcpyHostToDevice, st));
        VD_HIP_CHECK(ctx, hipMemsetAsync(P.ctl, 0, sizeof(LevelCtl), st));
        VD_HIP_CHECK(ctx, hipMemsetAsync(P.is_u, 0, T, st));
        VD_HIP_CHECK(ctx, hipMemcpyAsync(P.idx_copy, d_idx, 3 * T * 4, hipMemcpyDeviceToDevice, st));
    }
    const unsigned tri_blocks = (unsigned)((T + 255) / 256);
    hipLaunchKernelGGL(blas_precompute_kernel, dim3(tri_blocks), dim3(256), 0, st, d_verts, P.idx_copy, n_tri, n_vert, P.pay0,
                       P.boxes, P.root_keys, &P.ctl->err);
    hipLaunchKernelGGL(c_root_kernel, dim3(1), dim3(64), 0, st, P.top, P.root_keys, n_tri, P.small, P.ctl, P.seg0);

    // ---- phase A: level loop ----
    Seg* seg_cur = P.seg0; Seg* seg_next = P.seg1;
    LevelCtl h_ctl;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(&h_ctl, P.ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (h_ctl.err & ERR_BAD_INDEX) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_bvh_build: index >= n_vert");
    unsigned n_seg = h_ctl.n_seg;
    int levels = 0;
    while (n_seg > 0) {
        const unsigned seg_blocks = (n_seg + 63) / 64;
        // upper bound of items this level: sum ceil(count/kItem) <= T/kItem + n_seg
        const unsigned items_ub = (unsigned)(T / kItem) + n_seg + 1;
        hipLaunchKernelGGL(a_seg_begin_kernel, dim3(seg_blocks), dim3(64), 0, st, seg_cur, P.ctl);
        hipLaunchKernelGGL(a_items_scan_kernel, dim3(1), dim3(1024), 0, st, seg_cur, P.ctl);
        hipLaunchKernelGGL(a_items_fill_kernel, dim3(n_seg), dim3(64), 0, st, seg_cur, P.ctl, P.item_seg);
        hipLaunchKernelGGL(a_cb_kernel, dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, P.pay0);
        hipLaunchKernelGGL(a_planes_kernel, dim3(seg_blocks), dim3(64), 0, st, seg_cur, P.ctl);
        u32x4* src = P.pay0; u32x4* dst = P.pay1;
        for (int c = 0; c <= kCand; ++c) {
            const int cc = c < kCand ? c : -1;       // -1: final re-shuffle with each segment's best plane
            if (c == kCand) {
                hipLaunchKernelGGL(a_bin_kernel, dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, src, P.boxes, P.is_u);
                hipLaunchKernelGGL(a_eval_kernel, dim3(n_seg), dim3(64), 0, st, seg_cur, P.ctl, P.boxes);
            }
            hipLaunchKernelGGL(a_count_kernel, dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, src, cc, P.item_cnt);
            hipLaunchKernelGGL(a_scan_kernel, dim3(1), dim3(1024), 0, st, seg_cur, P.ctl, P.item_cnt, P.item_pre);
            hipLaunchKernelGGL(a_ranks_kernel, dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, src, cc, P.item_pre, P.tmp,
                               P.falsepos, P.truepos);
            hipLaunchKernelGGL(a_apply_kernel, dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, src, dst, cc, P.tmp,
                               P.falsepos, P.truepos, P.is_u);
            u32x4* t = src; src = dst; dst = t;
        }
        // 22 swaps: the arrangement is back in pay0
        hipLaunchKernelGGL(a_child_kernel, dim3(items_ub), dim3(256), 0, st, seg_cur, P.item_seg, P.ctl, P.pay0, P.boxes);
        hipLaunchKernelGGL(a_finalize_kernel, dim3(seg_blocks), dim3(64), 0, st, seg_cur, seg_next, P.ctl, P.top, P.small, P.is_u,
                           top_cap, small_cap);
        hipLaunchKernelGGL(a_level_swap_kernel, dim3(1), dim3(64), 0, st, P.ctl);
        VD_HIP_CHECK(ctx, hipMemcpyAsync(&h_ctl, P.ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
        VD_HIP_CHECK(ctx, hipStreamSynchronize(st));
        if (h_ctl.err & ERR_DEGENERATE)
            VD_FAIL(ctx, VD_ERR_DEGENERATE, "vd_bvh_build: every split candidate rejected (the reference builder crashes on this input)");
        if (h_ctl.err) VD_FAIL(ctx, VD_ERR_HIP, "vd_bvh_build: internal capacity exceeded");
        n_seg = h_ctl.n_seg;
        Seg* t = seg_cur; seg_cur = seg_next; seg_next = t;
        if (++levels > 4096) VD_FAIL(ctx, VD_ERR_HIP, "vd_bvh_build: level loop did not terminate");
    }

    // ---- phase B ----
    const unsigned n_small = h_ctl.n_small, n_top = h_ctl.n_top;
    if (n_small) {
        hipLaunchKernelGGL(blas_small_kernel, dim3((n_small + kSmallWaves - 1) / kSmallWaves), dim3(64 * kSmallWaves), kSmallWaves * sizeof(WaveLds), st, P.small, &P.ctl->n_small, P.pay0,
                           P.boxes, P.subnodes, P.sub_interior, P.final_ids, P.stack, &P.ctl->err);
    }
    // ---- phase C: DFS numbering of the top tree on the host ----
    std::vector<TopNode> h_top(n_top);
    std::vector<unsigned> h_sub(n_small ? n_small : 1);
    VD_HIP_CHECK(ctx, hipMemcpyAsync(h_top.data(), P.top, sizeof(TopNode) * n_top, hipMemcpyDeviceToHost, st));
    if (n_small) VD_HIP_CHECK(ctx, hipMemcpyAsync(h_sub.data(), P.sub_interior, 4 * (size_t)n_small, hipMemcpyDeviceToHost, st));
    VD_HIP_CHECK(ctx, hipMemcpyAsync(&h_ctl, P.ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (h_ctl.err & ERR_DEGENERATE)
        VD_FAIL(ctx, VD_ERR_DEGENERATE, "vd_bvh_build: every split candidate rejected (the reference builder crashes on this input)");
    std::vector<TopOut> h_out(n_top);
    std::vector<unsigned> h_root_pair(n_small ? n_small : 1);
    unsigned pool = 2;
    {
        // pre-order walk: an interior node takes the next pair when visited (blas.rs:110-112)
        std::vector<unsigned> stk;
        stk.push_back(0);
        h_out[0].final_index = 0;
        while (!stk.empty()) {
            const unsigned v = stk.back(); stk.pop_back();
            const TopNode& t = h_top[v];
            if (t.kind == 0u) continue;
            const unsigned pair = pool;
            h_out[v].pair = pair;
            if (t.kind == 2u) {
                h_root_pair[t.small] = pair;
                pool += 2u * h_sub[t.small];
            } else {
                pool += 2;
                h_out[t.left].final_index = pair;
                h_out[t.left + 1].final_index = pair + 1;
                stk.push_back(t.left + 1);
                stk.push_back(t.left);
            }
        }
    }
    if (pool > node_cap) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_bvh_build: node_cap too small");
    VD_HIP_CHECK(ctx, hipMemcpyAsync(P.tout, h_out.data(), sizeof(TopOut) * n_top, hipMemcpyHostToDevice, st));
    if (n_small) VD_HIP_CHECK(ctx, hipMemcpyAsync(P.root_pair, h_root_pair.data(), 4 * (size_t)n_small, hipMemcpyHostToDevice, st));
    VD_HIP_CHECK(ctx, hipMemsetAsync(d_out, 0, sizeof(VdBvhNode) * 2, st));   // node 1 stays all-zero (blas.rs:52,90)
    hipLaunchKernelGGL(c_top_kernel, dim3((n_top + 63) / 64), dim3(64), 0, st, P.top, P.tout, n_top, d_out);
    if (n_small) hipLaunchKernelGGL(c_sub_kernel, dim3(n_small), dim3(256), 0, st, P.small, P.sub_interior, P.root_pair, n_small, P.subnodes, d_out);
    hipLaunchKernelGGL(c_ids_big_leaves_kernel, dim3((n_top + 63) / 64), dim3(64), 0, st, P.top, n_top, P.pay0, P.final_ids);
    hipLaunchKernelGGL(c_permute_kernel, dim3(tri_blocks), dim3(256), 0, st, P.final_ids, P.idx_copy, d_idx, n_tri);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    VD_HIP_CHECK(ctx, hipStreamSynchronize(st));   // h_out / h_root_pair go out of scope
    *out_n_nodes = pool;
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

int vd_bvh_build_dev(VdCtx* ctx, const float* d_verts, uint32_t n_vert, uint32_t* d_idx, uint32_t n_tri, VdBvhNode* d_out,
                     uint32_t node_cap, uint32_t* out_n_nodes) {
    int rc = check_build_args(ctx, d_verts, n_vert, d_idx, n_tri, d_out, node_cap, out_n_nodes);
    return rc ? rc : bvh_build_dev_impl(ctx, d_verts, n_vert, d_idx, n_tri, d_out, node_cap, out_n_nodes);
}

int vd_bvh_build(VdCtx* ctx, const float* verts, uint32_t n_vert, uint32_t* idx, uint32_t n_tri, VdBvhNode* out,
                 uint32_t node_cap, uint32_t* out_n_nodes) {
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
    rc = bvh_build_dev_impl(ctx, reinterpret_cast<const float*>(ctx->stage_in), n_vert, reinterpret_cast<uint32_t*>(ctx->stage_aux),
                            n_tri, reinterpret_cast<VdBvhNode*>(ctx->stage_out), node_cap, out_n_nodes);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(idx, ctx->stage_aux, ib, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipMemcpyAsync(out, ctx->stage_out, sizeof(VdBvhNode) * (size_t)*out_n_nodes, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VD_OK;
}

}  // extern "C"

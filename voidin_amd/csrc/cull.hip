// cull.hip — per-instance frustum cull, indirect-draw emission and ordered compaction for
// gfx950 (CDNA4, wave64).
//
// Replaces the `emit_draws` compute pass (reference: shaders/emit_draws.wgsl:13-64,
// shaders/utils/math.wgsl:67-73, dispatched by crates/app/src/pass/visibility.rs:233-254).
//
// Data movement (HBM-bound, no MFMA — fp32 compares and index work):
//   * instances are a 144-byte AoS (not a power of two).  A wave streams 64 consecutive
//     instances = 9216 contiguous bytes as 9 fully coalesced, nontemporal 16-B-per-lane loads,
//     parks them in a wave-private LDS slab, and every lane then reads back its own instance's
//     transform (4 x ds_read_b128 at a 144-B stride: 36-dword stride is conflict-free for b128)
//     and mesh id; the next round's loads are already in flight (register prefetch);
//   * emit path (reference format): the 20-byte commands of a wave (1280 contiguous bytes) go
//     through the same slab and leave as 16-B-per-lane stores;
//   * compact path, large inputs (split form): pass 1 `cull_mask_tiled_kernel` writes only one
//     ballot bit + a compact mesh id per instance, so the read stream runs at ~6.3 TB/s; pass 2 =
//     `mask_scan_kernel` (survivors per 8192-instance chunk, scanned by the last workgroup to
//     arrive) + `expand_mask_u8_kernel` / `expand_mask_kernel` (workgroup c expands chunk c to
//     out[offset[c]...): no ticket, no look-back, every load issued before the first store).
//     (Storing the 20-byte commands from inside the read stream costs ~3x per byte: DESIGN.md §3.1);
//   * compact path, small inputs (fused form): `cull_compact_kernel<ROUNDS>`, one launch, a
//     decoupled look-back over 8-byte {epoch,status,value} granules ranks the tiles while
//     per-round (mesh id | visible) words wait in LDS;
//   * multi-GPU: the same pass 1 / pass 2 pair with the bitmask all-gathered in between
//     (vd_cull_mask_dev / vd_expand_mask_dev, voidin_amd/dist.py).
#include "vd_common.hpp"

#include <math.h>

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kInstBytes = 144;
constexpr int kSlabBytes = kWave * kInstBytes;  // 9216
constexpr int kChunksPerLane = kSlabBytes / (kWave * 16);  // 9



typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct CullCamera {   // the slice of CameraUniform the shader reads (shared.wgsl:13-24)
    float view[16];
    float frustum[4];
    float znear, zfar;
};

struct MeshRec { float mnx, mny, mnz; unsigned index_count; float mxx, mxy, mxz; unsigned base_index; int vertex_offset; };

__device__ __forceinline__ MeshRec load_mesh(const VdMeshInfo* __restrict__ meshes, unsigned mid) {
    const uint4* p = reinterpret_cast<const uint4*>(meshes + mid);
    const uint4 a = p[0], b = p[1];
    MeshRec m;
    m.mnx = __uint_as_float(a.x); m.mny = __uint_as_float(a.y); m.mnz = __uint_as_float(a.z); m.index_count = a.w;
    m.mxx = __uint_as_float(b.x); m.mxy = __uint_as_float(b.y); m.mxz = __uint_as_float(b.z); m.base_index = b.w;
    m.vertex_offset = meshes[mid].vertex_offset;
    return m;
}

__device__ __forceinline__ float len3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

// emit_draws.wgsl:13-33 with the evaluation order of SURVEY.md §8a C2'. T = transform columns.
__device__ __forceinline__ bool is_visible(const CullCamera& cam, const MeshRec& m, const float4 T0,
                                           const float4 T1, const float4 T2, const float4 T3) {
    const float* V = cam.view;
    // center = (mesh.max + mesh.min) / 2
    const float c0x = (m.mxx + m.mnx) / 2.0f, c0y = (m.mxy + m.mny) / 2.0f, c0z = (m.mxz + m.mnz) / 2.0f;
    // rows 0..2 of (view * transform): column j = ((V.c0*Tj.x + V.c1*Tj.y) + V.c2*Tj.z) + V.c3*Tj.w
    float c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float v0 = V[r], v1 = V[4 + r], v2 = V[8 + r], v3 = V[12 + r];
        const float m0 = ((v0 * T0.x + v1 * T0.y) + v2 * T0.z) + v3 * T0.w;
        const float m1 = ((v0 * T1.x + v1 * T1.y) + v2 * T1.z) + v3 * T1.w;
        const float m2 = ((v0 * T2.x + v1 * T2.y) + v2 * T2.z) + v3 * T2.w;
        const float m3 = ((v0 * T3.x + v1 * T3.y) + v2 * T3.z) + v3 * T3.w;
        // (VT * vec4(center, 1)).r
        c[r] = ((m0 * c0x + m1 * c0y) + m2 * c0z) + m3 * 1.0f;
    }
    // extract_scale (math.wgsl:67-73) and max_scale
    const float sx = len3(T0.x, T0.y, T0.z), sy = len3(T1.x, T1.y, T1.z), sz = len3(T2.x, T2.y, T2.z);
    const float max_scale = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz));
    // radius: object-space min/max against the view-space centre — bug-compatible (C2)
    const float d0 = len3(m.mnx - c[0], m.mny - c[1], m.mnz - c[2]);
    const float d1 = len3(m.mxx - c[0], m.mxy - c[1], m.mxz - c[2]);
    const float radius = fmaxf(d0, d1) * max_scale;
    if (c[2] * cam.frustum[1] - fabsf(c[0]) * cam.frustum[0] < -radius) return false;
    if (c[2] * cam.frustum[3] - fabsf(c[1]) * cam.frustum[2] < -radius) return false;
    if (c[2] + radius > cam.znear && c[2] - radius > cam.zfar) return false;
    return true;
}

// Stream the 64 instances starting at `first` into this wave's LDS slab (coalesced 16 B per
// lane), then return this lane's transform + mesh id.  `n_valid` = instances in range (<= 64).
struct LaneInst { float4 T0, T1, T2, T3; unsigned mesh; };

template <bool NT>
__device__ __forceinline__ void slab_fill(const VdInstance* __restrict__ inst, size_t first,
                                          unsigned n_valid, unsigned lane, u32x4 (&regs)[kChunksPerLane]) {
    const u32x4* src = reinterpret_cast<const u32x4*>(inst + first);
    const unsigned n_chunks = n_valid * (kInstBytes / 16);
#pragma unroll
    for (int j = 0; j < kChunksPerLane; ++j) {
        const unsigned c = j * kWave + lane;
        if (c < n_chunks) regs[j] = NT ? __builtin_nontemporal_load(src + c) : src[c];
        else regs[j] = u32x4{0u, 0u, 0u, 0u};
    }
}

__device__ __forceinline__ void slab_store(char* slab, unsigned lane, const u32x4 (&regs)[kChunksPerLane]) {
    u32x4* dst = reinterpret_cast<u32x4*>(slab);
#pragma unroll
    for (int j = 0; j < kChunksPerLane; ++j) dst[j * kWave + lane] = regs[j];
}

__device__ __forceinline__ LaneInst slab_read(const char* slab, unsigned lane) {
    const float4* p = reinterpret_cast<const float4*>(slab + lane * kInstBytes);
    LaneInst li;
    li.T0 = p[0]; li.T1 = p[1]; li.T2 = p[2]; li.T3 = p[3];
    li.mesh = *reinterpret_cast<const unsigned*>(slab + lane * kInstBytes + 128);
    return li;
}

// ------------------------------------------------------------------------------------------
// C1: emit_draws — every slot written (reference format).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock, 4) void emit_draws_kernel(CullCamera cam, const VdMeshInfo* __restrict__ meshes,
                                                            unsigned n_mesh, const VdInstance* __restrict__ inst,
                                                            unsigned n_inst, VdDrawIndexedIndirect* __restrict__ out,
                                                               unsigned n_wave_tiles, unsigned first_instance) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    char* slab = smem + wave * kSlabBytes;
    const unsigned waves_total = gridDim.x * kWavesPerBlock;
    u32x4 regs[kChunksPerLane];

    unsigned wt = blockIdx.x * kWavesPerBlock + wave;
    if (wt < n_wave_tiles) {
        const size_t f0 = (size_t)wt * kWave;
        slab_fill<true>(inst, f0, min(64u, n_inst - (unsigned)f0), lane, regs);
    }
    for (; wt < n_wave_tiles; wt += waves_total) {
        const size_t first = (size_t)wt * kWave;
        const unsigned n_valid = min(64u, n_inst - (unsigned)first);
        slab_store(slab, lane, regs);
        // prefetch the next wave-tile while this one is processed
        const unsigned wn = wt + waves_total;
        if (wn < n_wave_tiles) {
            const size_t fn = (size_t)wn * kWave;
            slab_fill<true>(inst, fn, min(64u, n_inst - (unsigned)fn), lane, regs);
        }
        vd_wave_lds_sync();
        const LaneInst li = slab_read(slab, lane);
        vd_wave_lds_sync();

        const unsigned mid = min(li.mesh, n_mesh - 1u);
        const MeshRec m = load_mesh(meshes, mid);
        const bool vis = is_visible(cam, m, li.T0, li.T1, li.T2, li.T3);

        // emit_draws.wgsl:55-63 — stage the wave's 64 commands (1280 B) and store 16 B per lane
        unsigned* cmd = reinterpret_cast<unsigned*>(slab) + lane * 5u;
        cmd[0] = m.index_count;
        cmd[1] = vis ? 1u : 0u;
        cmd[2] = m.base_index;
        cmd[3] = (unsigned)m.vertex_offset;
        cmd[4] = first_instance + (unsigned)first + lane;
        vd_wave_lds_sync();
        const unsigned n_bytes = n_valid * 20u;
        char* gdst = reinterpret_cast<char*>(out) + first * 20u;  // 1280-B multiples: 16-B aligned
        const uint4* s4 = reinterpret_cast<const uint4*>(slab);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned c = k * kWave + lane;
            const unsigned b = c * 16u;
            if (k == 1 && lane >= 16u) break;
            if (b + 16u <= n_bytes) {
                *reinterpret_cast<uint4*>(gdst + b) = s4[c];
            } else if (b < n_bytes) {  // ragged tail: 20-B records end on a 4-B boundary
                const unsigned* s1 = reinterpret_cast<const unsigned*>(slab + b);
                for (unsigned w = 0; b + 4u * w < n_bytes; ++w) reinterpret_cast<unsigned*>(gdst + b)[w] = s1[w];
            }
        }
        vd_wave_lds_sync();
    }
}

// ------------------------------------------------------------------------------------------
// C1 + C3 fused: cull and emit survivors only, ascending instance order, single pass (used below
// VdCtx::split_min = 2 Mi instances; larger inputs run the split form: cull_mask_tiled_kernel + expand_mask_kernel).
// ------------------------------------------------------------------------------------------
template <int ROUNDS>
__global__ __launch_bounds__(kBlock, 3)
void cull_compact_kernel(CullCamera cam, const VdMeshInfo* __restrict__ meshes, unsigned n_mesh,
                         const VdInstance* __restrict__ inst, unsigned n_inst, VdDrawIndexedIndirect* __restrict__ out,
                         unsigned* __restrict__ out_count, vd_u64* tile_state, vd_u64* ticket_counter,
                         unsigned n_tiles, unsigned first_instance, unsigned* fault_host) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // dynamic LDS: per-wave slabs, then per-round records (mesh id | visible << 31), then scalars
    unsigned* s_rec = reinterpret_cast<unsigned*>(smem + kWavesPerBlock * kSlabBytes);   // [ROUNDS][kBlock]
    unsigned* s_misc = s_rec + ROUNDS * kBlock;   // [0] ticket, [1] epoch / tile_excl, [2..5] wave totals
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    char* slab = smem + wave * kSlabBytes;

    if (threadIdx.x == 0) s_misc[0] = vd_take_ticket(ticket_counter, n_tiles, &s_misc[1]);
    __syncthreads();
    const unsigned tile = s_misc[0], epoch = s_misc[1];
    __syncthreads();
    if (tile >= n_tiles) return;                  // (the ticket word was not at {epoch, 0} when the launch began: never expected)
#ifdef VD_TUNING
    if (ticket_counter[1] == (vd_u64)tile + 1ull) return;   // tests/test_gpu_scan_fault.py: this workgroup "dies" before it publishes anything
#endif
    // The count is written by the LAST tile; until then it holds the error value, stored by the FIRST tile before anything can
    // depend on it (the fence completes the store before this tile's granule - which every other tile's prefix waits for - goes
    // out): a launch that loses its last workgroup leaves the sentinel, not the previous call's count.
    if (tile == 0u && threadIdx.x == 0) { __hip_atomic_store(out_count, VD_SCAN_STUCK, VD_RLX_AGENT); __threadfence(); }
    const size_t tile_first = (size_t)tile * (kBlock * ROUNDS);
    // wave-contiguous ranges keep the output order (wave, round, lane) == instance order
    const size_t wave_first = tile_first + (size_t)wave * (kWave * ROUNDS);
    auto valid_at = [&](size_t f) -> unsigned { return f < n_inst ? (unsigned)min((size_t)64, (size_t)n_inst - f) : 0u; };

    unsigned wave_total = 0;
    u32x4 regs[kChunksPerLane];
    slab_fill<true>(inst, wave_first, valid_at(wave_first), lane, regs);
#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r) {
        const size_t first = wave_first + (size_t)r * kWave;
        const unsigned n_valid = valid_at(first);
        slab_store(slab, lane, regs);
        if (r + 1 < ROUNDS) slab_fill<true>(inst, first + kWave, valid_at(first + kWave), lane, regs);
        vd_wave_lds_sync();
        const LaneInst li = slab_read(slab, lane);
        vd_wave_lds_sync();
        const unsigned mid = min(li.mesh, n_mesh - 1u);
        const MeshRec m = load_mesh(meshes, mid);
        const bool vis = lane < n_valid && is_visible(cam, m, li.T0, li.T1, li.T2, li.T3);
        s_rec[r * kBlock + threadIdx.x] = mid | (vis ? 0x80000000u : 0u);
        wave_total += (unsigned)__popcll(__ballot(vis));
    }

    if (lane == 0) s_misc[2 + wave] = wave_total;
    __syncthreads();
    if (wave == 0) {
        unsigned tile_total = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) tile_total += s_misc[2 + w];
        const unsigned excl = vd_lookback(tile_state, epoch, tile, tile_total);
        if (lane == 0) {
            s_misc[1] = excl;
            if (tile == n_tiles - 1u) *out_count = vd_scan_final_count(tile_state, epoch, excl, tile_total, fault_host);
        }
    }
    __syncthreads();
    unsigned base = s_misc[1];
    if (base == VD_SCAN_STUCK) return;            // a predecessor never published its total: no list (the count says so)
    for (unsigned w = 0; w < wave; ++w) base += s_misc[2 + w];

#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r) {
        const unsigned rec = s_rec[r * kBlock + threadIdx.x];
        const bool vis = (rec >> 31) != 0u;
        const unsigned long long mask = __ballot(vis);
        if (vis) {
            const unsigned mid = rec & 0x7fffffffu;
            const uint4* mp = reinterpret_cast<const uint4*>(meshes + mid);
            unsigned* o = reinterpret_cast<unsigned*>(out + (base + vd_mbcnt(mask)));
            o[0] = mp[0].w;                         // index_count
            o[1] = 1u;
            o[2] = mp[1].w;                         // base_index
            o[3] = (unsigned)meshes[mid].vertex_offset;
            o[4] = first_instance + (unsigned)(wave_first + (size_t)r * kWave) + lane;
        }
        base += (unsigned)__popcll(mask);
    }
}

template <int ROUNDS>
constexpr int compact_lds_bytes() { return kWavesPerBlock * kSlabBytes + ROUNDS * kBlock * 4 + 32; }

// ------------------------------------------------------------------------------------------
// Multi-GPU wire format: cull -> one bit per instance; expand bits -> ordered draw list.
// ------------------------------------------------------------------------------------------
// IdT != void: also write the (clamped) mesh id of every instance as IdT (u8 / u16 / u32 by table
// size) for the expansion pass of the split single-GPU path.
template <typename IdT>
__global__ __launch_bounds__(kBlock, 3) void cull_mask_kernel(CullCamera cam, const VdMeshInfo* __restrict__ meshes,
                                                               unsigned n_mesh, const VdInstance* __restrict__ inst,
                                                               unsigned n_inst, vd_u64* __restrict__ mask,
                                                               IdT* __restrict__ ids_out, unsigned n_wave_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    char* slab = smem + wave * kSlabBytes;
    const unsigned waves_total = gridDim.x * kWavesPerBlock;
    u32x4 regs[kChunksPerLane];
    unsigned wt = blockIdx.x * kWavesPerBlock + wave;
    if (wt < n_wave_tiles) {
        const size_t f0 = (size_t)wt * kWave;
        slab_fill<true>(inst, f0, min(64u, n_inst - (unsigned)f0), lane, regs);
    }
    for (; wt < n_wave_tiles; wt += waves_total) {
        const size_t first = (size_t)wt * kWave;
        const unsigned n_valid = min(64u, n_inst - (unsigned)first);
        slab_store(slab, lane, regs);
        const unsigned wn = wt + waves_total;
        if (wn < n_wave_tiles) {
            const size_t fn = (size_t)wn * kWave;
            slab_fill<true>(inst, fn, min(64u, n_inst - (unsigned)fn), lane, regs);
        }
        vd_wave_lds_sync();
        const LaneInst li = slab_read(slab, lane);
        vd_wave_lds_sync();
        const unsigned mid = min(li.mesh, n_mesh - 1u);
        const MeshRec m = load_mesh(meshes, mid);
        const bool vis = lane < n_valid && is_visible(cam, m, li.T0, li.T1, li.T2, li.T3);
        const unsigned long long b = __ballot(vis);
        if (lane == 0) mask[wt] = b;
        if (ids_out && lane < n_valid) ids_out[first + lane] = (IdT)mid;
    }
}

// ------------------------------------------------------------------------------------------
// Occlusion extension (no reference counterpart — definition in include/voidin_abi.h,
// "Occlusion culling"; pyramid built by hiz.hip).  mask_out = mask_in minus the instances whose bounding sphere lies
// behind the depth pyramid.  Words of mask_in that are 0 cost nothing: their 9 KB of instances are not read, which
// is the common case in the second pass of the two-pass scheme.
// ------------------------------------------------------------------------------------------
struct OccCamera { float view[16]; float p00, p11, p20, p21, p22, p32, znear; };
struct HizView { const float* base; unsigned width, height, n_levels; unsigned off[17]; };

__device__ __forceinline__ bool is_occluded(const OccCamera& cam, const HizView& hz, const MeshRec& m, const float4 T0, const float4 T1,
                                            const float4 T2, const float4 T3) {
    const float* V = cam.view;
    const float c0x = (m.mxx + m.mnx) / 2.0f, c0y = (m.mxy + m.mny) / 2.0f, c0z = (m.mxz + m.mnz) / 2.0f;
    float c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float v0 = V[r], v1 = V[4 + r], v2 = V[8 + r], v3 = V[12 + r];
        const float m0 = ((v0 * T0.x + v1 * T0.y) + v2 * T0.z) + v3 * T0.w;
        const float m1 = ((v0 * T1.x + v1 * T1.y) + v2 * T1.z) + v3 * T1.w;
        const float m2 = ((v0 * T2.x + v1 * T2.y) + v2 * T2.z) + v3 * T2.w;
        const float m3 = ((v0 * T3.x + v1 * T3.y) + v2 * T3.z) + v3 * T3.w;
        c[r] = ((m0 * c0x + m1 * c0y) + m2 * c0z) + m3 * 1.0f;
    }
    const float sx = len3(T0.x, T0.y, T0.z), sy = len3(T1.x, T1.y, T1.z), sz = len3(T2.x, T2.y, T2.z);
    const float max_scale = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz));
    const float r = (len3(m.mxx - m.mnx, m.mxy - m.mny, m.mxz - m.mnz) * 0.5f) * max_scale;
    const float d = -c[2];
    const float dn = d - r;
    if (!(dn > cam.znear)) return false;
    const float rr = r * r, dd = d * d, rd = r * d;
    const float tx = sqrtf((c[0] * c[0] + dd) - rr), ty = sqrtf((c[1] * c[1] + dd) - rr);
    const float dxm = d * tx + c[0] * r, dxp = d * tx - c[0] * r, dym = d * ty + c[1] * r, dyp = d * ty - c[1] * r;
    if (!(dxm > 0.0f && dxp > 0.0f && dym > 0.0f && dyp > 0.0f)) return false;
    const float sx0 = (c[0] * tx - rd) / dxm, sx1 = (c[0] * tx + rd) / dxp;
    const float sy0 = (c[1] * ty - rd) / dym, sy1 = (c[1] * ty + rd) / dyp;
    const float nxa = cam.p00 * sx0 - cam.p20, nxb = cam.p00 * sx1 - cam.p20, nya = cam.p11 * sy0 - cam.p21, nyb = cam.p11 * sy1 - cam.p21;
    const float nx_lo = fminf(nxa, nxb), nx_hi = fmaxf(nxa, nxb), ny_lo = fminf(nya, nyb), ny_hi = fmaxf(nya, nyb);
    const float W = (float)hz.width, H = (float)hz.height;
    const float u0 = (nx_lo * 0.5f + 0.5f) * W - 0.5f, u1 = (nx_hi * 0.5f + 0.5f) * W + 0.5f;
    const float v0 = (0.5f - ny_hi * 0.5f) * H - 0.5f, v1 = (0.5f - ny_lo * 0.5f) * H + 0.5f;
    if (!(u1 >= 0.0f && v1 >= 0.0f && u0 < W && v0 < H)) return false;
    const unsigned x0 = (unsigned)floorf(fmaxf(u0, 0.0f)), x1 = (unsigned)floorf(fminf(u1, W - 1.0f));
    const unsigned y0 = (unsigned)floorf(fmaxf(v0, 0.0f)), y1 = (unsigned)floorf(fminf(v1, H - 1.0f));
    const unsigned span = max(x1 - x0, y1 - y0);
    const unsigned lvl = min(span ? 32u - (unsigned)__clz((int)span) : 0u, hz.n_levels - 1u);
    const float* t = hz.base + hz.off[lvl];
    const unsigned lw = ((hz.width - 1u) >> lvl) + 1u;
    const unsigned ax = x0 >> lvl, bx = x1 >> lvl, ay = y0 >> lvl, by = y1 >> lvl;
    const float h0 = fminf(t[(size_t)ay * lw + ax], t[(size_t)ay * lw + bx]);
    const float h1 = fminf(t[(size_t)by * lw + ax], t[(size_t)by * lw + bx]);
    const float hmin = fminf(h0, h1);
    const float depth = (cam.p32 - cam.p22 * dn) / dn;
    return depth < hmin;
}

__global__ __launch_bounds__(kBlock, 3) void occlusion_mask_kernel(OccCamera cam, HizView hz, const VdMeshInfo* __restrict__ meshes,
                                                                   unsigned n_mesh, const VdInstance* __restrict__ inst, unsigned n_inst,
                                                                   const vd_u64* __restrict__ mask_in, vd_u64* __restrict__ mask_out,
                                                                   unsigned n_wave_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    char* slab = smem + wave * kSlabBytes;
    const unsigned waves_total = gridDim.x * kWavesPerBlock;
    u32x4 regs[kChunksPerLane];
    for (unsigned wt = blockIdx.x * kWavesPerBlock + wave; wt < n_wave_tiles; wt += waves_total) {
        const vd_u64 in = mask_in[wt];                       // wave-uniform
        if (in == 0ull) {
            if (lane == 0) mask_out[wt] = 0ull;
            continue;
        }
        const size_t first = (size_t)wt * kWave;
        const unsigned n_valid = min(64u, n_inst - (unsigned)first);
        slab_fill<false>(inst, first, n_valid, lane, regs);
        slab_store(slab, lane, regs);
        vd_wave_lds_sync();
        const LaneInst li = slab_read(slab, lane);
        vd_wave_lds_sync();
        bool keep = false;
        if (lane < n_valid && ((in >> lane) & 1ull)) {
            const MeshRec m = load_mesh(meshes, min(li.mesh, n_mesh - 1u));
            keep = !is_occluded(cam, hz, m, li.T0, li.T1, li.T2, li.T3);
        }
        const unsigned long long b = __ballot(keep);
        if (lane == 0) mask_out[wt] = b;
    }
}

// Tiled form of pass 1: a wave owns kMaskRounds CONSECUTIVE rounds (1024 instances), keeps their
// mesh ids and ballot words on chip and flushes them once per tile as wide stores, so the read
// stream is interrupted by one 1-KB store per 147 KB read instead of a 64-B store per 9 KB.
constexpr int kMaskRounds = 16;
template <typename IdT>
__global__ __launch_bounds__(kBlock, 3) void cull_mask_tiled_kernel(CullCamera cam, const VdMeshInfo* __restrict__ meshes,
                                                                     unsigned n_mesh, const VdInstance* __restrict__ inst,
                                                                     unsigned n_inst, vd_u64* __restrict__ mask,
                                                                     IdT* __restrict__ ids_out, unsigned n_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr int kIdBytes = kMaskRounds * kWave * (int)sizeof(IdT);
    char* slab = smem + wave * (kSlabBytes + kIdBytes);
    IdT* s_ids = reinterpret_cast<IdT*>(slab + kSlabBytes);
    const unsigned waves_total = gridDim.x * kWavesPerBlock;
    auto valid_at = [&](size_t f) -> unsigned { return f < n_inst ? (unsigned)min((size_t)64, (size_t)n_inst - f) : 0u; };
    u32x4 regs[kChunksPerLane];
    for (unsigned t = blockIdx.x * kWavesPerBlock + wave; t < n_tiles; t += waves_total) {
        const size_t tile_first = (size_t)t * (kWave * kMaskRounds);
        slab_fill<true>(inst, tile_first, valid_at(tile_first), lane, regs);
        // what the id table holds for this tile now: mesh assignment is static in practice (only transforms animate:
        // shaders/compute_update.wgsl), and a row that already matches is not written again - a store interleaved
        // with the read stream costs ~3x its bytes, a load does not.  Always correct: any row that
        // differs (first frame, reallocated scratch, edited instances) is rewritten.
        constexpr int kIdRows = kIdBytes / (kWave * 16);
        u32x4 old_ids[kIdRows];
        const size_t id_base0 = tile_first * sizeof(IdT);
        const bool full_tile = tile_first + (size_t)kWave * kMaskRounds <= (size_t)n_inst;
        if (full_tile) {
#pragma unroll
            for (int r = 0; r < kIdRows; ++r)
                old_ids[r] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(ids_out) + id_base0 + (size_t)r * kWave * 16 + lane * 16u);
        }
        vd_u64 my_word = 0;
#pragma unroll 1
        for (int r = 0; r < kMaskRounds; ++r) {
            const size_t first = tile_first + (size_t)r * kWave;
            const unsigned n_valid = valid_at(first);
            slab_store(slab, lane, regs);
            if (r + 1 < kMaskRounds) slab_fill<true>(inst, first + kWave, valid_at(first + kWave), lane, regs);
            vd_wave_lds_sync();
            const LaneInst li = slab_read(slab, lane);
            vd_wave_lds_sync();
            const unsigned mid = min(li.mesh, n_mesh - 1u);
            const MeshRec m = load_mesh(meshes, mid);
            const bool vis = lane < n_valid && is_visible(cam, m, li.T0, li.T1, li.T2, li.T3);
            const unsigned long long b = __ballot(vis);
            if (lane == (unsigned)r) my_word = b;
            s_ids[r * kWave + lane] = (IdT)mid;
        }
        vd_wave_lds_sync();
        // flush: ballot words (lane r holds round r) and the tile's ids as 16-B stores
        const size_t w0 = (size_t)t * kMaskRounds;
        const size_t n_words = ((size_t)n_inst + 63) / 64;
        if (lane < (unsigned)kMaskRounds && w0 + lane < n_words) mask[w0 + lane] = my_word;
        const size_t id_base = tile_first * sizeof(IdT);                 // bytes; tile_first % 1024 == 0 -> 16-B aligned
        const size_t id_end = min((size_t)n_inst, tile_first + (size_t)kWave * kMaskRounds) * sizeof(IdT);
        char* gids = reinterpret_cast<char*>(ids_out);
        if (full_tile) {
#pragma unroll
            for (int r = 0; r < kIdRows; ++r) {
                const unsigned b0 = (unsigned)r * kWave * 16u + lane * 16u;
                const u32x4 nv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(s_ids) + b0);
                const bool diff = nv.x != old_ids[r].x || nv.y != old_ids[r].y || nv.z != old_ids[r].z || nv.w != old_ids[r].w;
                if (__any(diff)) *reinterpret_cast<u32x4*>(gids + id_base + b0) = nv;
            }
        } else {
            for (unsigned b0 = lane * 16u; b0 < (unsigned)kIdBytes; b0 += kWave * 16u) {
                if (id_base + b0 + 16u <= id_end) {
                    *reinterpret_cast<u32x4*>(gids + id_base + b0) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(s_ids) + b0);
                } else {
                    for (unsigned q = 0; q < 16u && id_base + b0 + q < id_end; ++q) gids[id_base + b0 + q] = reinterpret_cast<const char*>(s_ids)[b0 + q];
                }
            }
        }
        vd_wave_lds_sync();
    }
}

constexpr int kExpandGroup = 4;                      // mask words staged and stored as one contiguous run
constexpr int kExpandWords = 32;                     // mask words (64 instances each) per wave
constexpr int kChunkWords = kWavesPerBlock * kExpandWords;   // per workgroup: 128 words = 8192 instances

// Pass 2a of the split form: survivors per 8192-instance chunk, then (last workgroup to finish) their exclusive
// scan in place and the total.  Keeping the scan out of pass 2b leaves that kernel without tickets, look-back or
// any other load that depends on another workgroup: under a saturated store stream every dependent load costs
// microseconds (on gfx950 loads and stores share vmcnt and the same queue), and 2b had four of them in a chain.
//
// Hand-off between workgroups without fences (an agent-scope release writes back the whole L2 of the XCD, ~100 ns per
// workgroup while the previous frame's command list is still dirty in it) and without a data race: a chunk's count
// travels as ONE naturally aligned 8-byte {launch epoch, count} word, written by one agent-scope atomic store and
// read by agent-scope atomic loads - the datum is its own flag, as in the look-back granules of vd_common.hpp.  The
// arrival counter only ELECTS the workgroup that scans; that workgroup accepts an entry when its tag is this
// launch's epoch (and polls the few that are still in flight), so nothing is inferred from the order of accesses to
// different addresses.  The epoch word is read at the start of every workgroup and advanced by the elected one after
// everybody has arrived, i.e. it is stable for the whole launch; launches on one stream are ordered by the stream.
constexpr int kScanBlock = 1024;                     // 16 waves: the last workgroup's scan is one round trip even at 80 M
struct ScanState { unsigned done, epoch, pad[2]; };  // followed by one vd_u64 entry per chunk: {epoch : 32 | value : 32}
__global__ __launch_bounds__(kScanBlock) void mask_scan_kernel(const vd_u64* __restrict__ mask, unsigned n_words, unsigned n_chunks,
                                                           vd_u64* chunk_entry, ScanState* state,
                                                           unsigned* __restrict__ out_count) {
    constexpr int kScanWaves = kScanBlock / kWave;
    __shared__ unsigned s_last, s_wave_sum[kScanWaves];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned ep = __hip_atomic_load(&state->epoch, VD_RLX_AGENT);
    const vd_u64 tag = (vd_u64)ep << 32;
    for (unsigned c = blockIdx.x * kScanWaves + wave; c < n_chunks; c += gridDim.x * kScanWaves) {
        const unsigned w = c * kChunkWords + lane;
        unsigned v = (w < n_words ? (unsigned)__popcll(mask[w]) : 0u) + (w + 64u < n_words ? (unsigned)__popcll(mask[w + 64u]) : 0u);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) __hip_atomic_store(&chunk_entry[c], tag | v, VD_RLX_AGENT);   // write-through: the datum is its own flag
    }
    // not needed for correctness (entries are self-validating): arriving only after this workgroup's stores have
    // completed means the elected workgroup almost never has to poll
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&state->done, 1u, VD_RLX_AGENT) == gridDim.x - 1u;
    __syncthreads();
    if (!s_last) return;
    // thread t scans the contiguous range [t*per, (t+1)*per): all its loads are independent (one round trip) and
    // read at agent scope (other XCDs wrote the counts)
    const unsigned per = (n_chunks + kScanBlock - 1) / kScanBlock;
    const unsigned begin = min(threadIdx.x * per, n_chunks), end = min(begin + per, n_chunks);
    // Bounded like every other cross-workgroup wait of the library: if the header and the entries ever disagree (a
    // launch on this context faulted mid-kernel, a stray write hit the state) the count becomes the sentinel 0xffffffff
    // - which the expansion pass and every consumer of *out_count treat as "no list" (it exceeds any instance count) -
    // instead of a stream that never finishes.
    bool stuck = false;
    auto entry = [&](unsigned c) -> unsigned {
        vd_u64 e = __hip_atomic_load(&chunk_entry[c], VD_RLX_AGENT);
        unsigned spins = 0;
        while ((e >> 32) != (vd_u64)ep) {                    // still in flight: its writer has arrived, the store lands shortly
            if (++spins > (1u << 22)) { stuck = true; return 0u; }
            __builtin_amdgcn_s_sleep(1);
            e = __hip_atomic_load(&chunk_entry[c], VD_RLX_AGENT);
        }
        return (unsigned)e;
    };
    unsigned sum = 0;
    for (unsigned c = begin; c < end; ++c) sum += entry(c);
    const bool any_stuck = __syncthreads_or(stuck ? 1 : 0) != 0;
    unsigned incl = sum;                                   // inclusive scan of the per-thread sums
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const unsigned t = __shfl_up(incl, off);
        if (lane >= (unsigned)off) incl += t;
    }
    if (lane == kWave - 1u) s_wave_sum[wave] = incl;
    __syncthreads();
    unsigned run = incl - sum;
    for (unsigned w = 0; w < wave; ++w) run += s_wave_sum[w];
    for (unsigned c = begin; c < end; ++c) {
        if (any_stuck) { chunk_entry[c] = tag | 0xffffffffu; continue; }   // pass 2b leaves such a chunk alone
        const unsigned v = entry(c);
        chunk_entry[c] = tag | run;                        // read by pass 2b, a later launch on this stream
        run += v;
    }
    if (threadIdx.x == kScanBlock - 1u) {
        *out_count = any_stuck ? 0xffffffffu : run;
        __hip_atomic_store(&state->done, 0u, VD_RLX_AGENT);        // re-armed for the next launch on this stream
        __hip_atomic_store(&state->epoch, ep + 1u, VD_RLX_AGENT);  // every workgroup of this launch has read it (all arrived)
    }
}

// Pass 2b: workgroup c expands the 128 mask words of chunk c to out[chunk_offset[c] ...); word w belongs to shard
// w / wps and holds the instances shard*shard_size + 64*(w % wps) + bit.  Every load is issued before the first
// store (one round trip per workgroup).
// General form (any id width, any shard size); expand_mask_u8_kernel below is the tuned common case.
//   TAB:  the mesh table fits the LDS copy (no global loads in the store loop).
template <typename IdT, bool TAB>
__global__ __launch_bounds__(kBlock) void expand_mask_kernel(const vd_u64* __restrict__ mask, unsigned n_words, unsigned wps,
                                                             unsigned shard_size, unsigned n_total, unsigned first_instance,
                                                             const IdT* __restrict__ mesh_ids,
                                                             const VdMeshInfo* __restrict__ meshes, unsigned n_mesh,
                                                             VdDrawIndexedIndirect* __restrict__ out,
                                                             const vd_u64* __restrict__ chunk_entry) {
    constexpr int kGroups = kExpandWords / kExpandGroup;
    constexpr unsigned kTab = TAB ? 512 : 1;              // mesh tables up to 512 entries are served from LDS
    __shared__ unsigned s_tab[kTab][3];                   // {index_count, base_index, vertex_offset}
    constexpr int kStageBytes = kExpandGroup * 1280 + 32;
    __shared__ __attribute__((aligned(16))) char s_stage[kWavesPerBlock][kStageBytes];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned chunk = blockIdx.x;
    const unsigned cw0 = chunk * kChunkWords;
    const unsigned w0 = __builtin_amdgcn_readfirstlane(cw0 + wave * kExpandWords);   // wave-uniform: scalar index math
    unsigned base = (unsigned)chunk_entry[chunk];
    if (base == 0xffffffffu) return;                       // the scan gave up (mask_scan_kernel): no list
    // survivors of the chunk's earlier waves: lane l looks at words cw0 + l and cw0 + 64 + l
    unsigned before = 0;
    if (lane < wave * kExpandWords && cw0 + lane < n_words) before = (unsigned)__popcll(mask[cw0 + lane]);
    if (lane + 64u < wave * kExpandWords && cw0 + 64u + lane < n_words) before += (unsigned)__popcll(mask[cw0 + 64u + lane]);
    // lane l < 32 holds mask word w0 + l
    vd_u64 my_word = 0;
    if (lane < (unsigned)kExpandWords && w0 + lane < n_words) my_word = mask[w0 + lane];
    // first instance of each word of group g (one division per group)
    auto group_first = [&](unsigned wg, unsigned (&f)[kExpandGroup]) {
        unsigned shard = wg / wps, r = wg - shard * wps;
#pragma unroll
        for (int q = 0; q < kExpandGroup; ++q) {
            f[q] = shard * shard_size + 64u * r;
            if (++r >= wps) { r = 0u; ++shard; }
        }
    };
    unsigned ids[kExpandWords];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        const unsigned wg = w0 + g * kExpandGroup;
        unsigned f[kExpandGroup];
        group_first(wg, f);
#pragma unroll
        for (int q = 0; q < kExpandGroup; ++q) {
            const unsigned idx = f[q] + lane;
            ids[g * kExpandGroup + q] = (wg + q < n_words && idx < n_total) ? (unsigned)mesh_ids[idx] : 0u;
        }
    }
    if (TAB)
        for (unsigned i = threadIdx.x; i < n_mesh; i += kBlock) {
            s_tab[i][0] = meshes[i].index_count; s_tab[i][1] = meshes[i].base_index; s_tab[i][2] = (unsigned)meshes[i].vertex_offset;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    base += before;
    __syncthreads();
    // survivors of kExpandGroup mask words are staged in LDS at the destination's 16-B phase and leave as
    // 16-B-per-lane stores in one contiguous run (this kernel is write-dominated: 20 B out per ~1 B in)
    char* stage = s_stage[wave];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        const unsigned wg = w0 + g * kExpandGroup;
        vd_u64 m[kExpandGroup];
        unsigned cnt = 0;
#pragma unroll
        for (int q = 0; q < kExpandGroup; ++q) {
            const int k = g * kExpandGroup + q;
            const unsigned lo = __shfl((unsigned)my_word, k), hi = __shfl((unsigned)(my_word >> 32), k);
            m[q] = ((vd_u64)hi << 32) | lo;
            cnt += (unsigned)__popcll(m[q]);
        }
        if (cnt == 0u) continue;
        char* gbase = reinterpret_cast<char*>(out + base);
        const unsigned shift = (unsigned)(reinterpret_cast<uintptr_t>(gbase) & 15u);
        unsigned run = 0;
        unsigned first[kExpandGroup];
        group_first(wg, first);
#pragma unroll
        for (int q = 0; q < kExpandGroup; ++q) {
            unsigned mid = ids[g * kExpandGroup + q];
            if ((m[q] >> lane) & 1ull) {
                mid = min(mid, n_mesh - 1u);
                unsigned* o = reinterpret_cast<unsigned*>(stage + shift + 20u * (run + vd_mbcnt(m[q])));
                if (TAB) { o[0] = s_tab[mid][0]; o[2] = s_tab[mid][1]; o[3] = s_tab[mid][2]; }
                else { o[0] = meshes[mid].index_count; o[2] = meshes[mid].base_index; o[3] = (unsigned)meshes[mid].vertex_offset; }
                o[1] = 1u;
                o[4] = first_instance + first[q] + lane;
            }
            run += (unsigned)__popcll(m[q]);
        }
        vd_wave_lds_sync();
        const unsigned total = shift + 20u * cnt;
        char* g16 = gbase - shift;
        for (unsigned b0 = lane * 16u; b0 < total; b0 += kWave * 16u) {
            if (b0 >= shift && b0 + 16u <= total) {
                *reinterpret_cast<u32x4*>(g16 + b0) = *reinterpret_cast<const u32x4*>(stage + b0);
            } else {
                const unsigned lo_b = b0 > shift ? b0 : shift, hi_b = b0 + 16u < total ? b0 + 16u : total;
                for (unsigned b = lo_b; b < hi_b; b += 4u)
                    *reinterpret_cast<unsigned*>(g16 + b) = *reinterpret_cast<const unsigned*>(stage + b);
            }
        }
        vd_wave_lds_sync();
        base += cnt;
    }
}

// Pass 2b, fast path: 1-byte ids, mesh table <= 256 entries, every group of 4 mask words inside one shard and the id
// table 4-byte aligned.  This kernel is bound by instruction issue (a wave64 instruction takes 4 cycles), not by LDS
// or store bandwidth, so the per-survivor instruction count is what is tuned here:
//   * all index math that is uniform across the wave runs on the scalar unit (w0 through readfirstlane, shard walk
//     by increments instead of a division per word);
//   * the mask word is the exec mask of the survivor branch (inverse ballot), no per-lane bit test;
//   * a group's 256 ids are one dword per lane, redistributed with ds_bpermute;
//   * the LDS mesh table holds ready-made {index_count, 1, base_index, vertex_offset} rows: one ds_read_b128;
//   * DIRECT: a command leaves as one 16-byte + one 4-byte store at a 20-byte lane stride (L2 merges the lines);
//     otherwise it is staged in LDS at the destination's 16-B phase and leaves in 16-B-per-lane runs.
// The dword that holds the last valid id may extend past n_total: an aligned dword that contains one valid byte
// never crosses a page, and the extra bytes belong to instances whose mask bit is 0.
template <bool DIRECT, int PF>
__global__ __launch_bounds__(kBlock) void expand_mask_u8_kernel(const vd_u64* __restrict__ mask, unsigned n_words, unsigned wps,
                                                                unsigned shard_size, unsigned n_total, unsigned first_instance,
                                                                const unsigned char* __restrict__ mesh_ids,
                                                                const VdMeshInfo* __restrict__ meshes, unsigned n_mesh,
                                                                VdDrawIndexedIndirect* __restrict__ out,
                                                                const vd_u64* __restrict__ chunk_entry) {
    constexpr int kGroups = kExpandWords / kExpandGroup;
    __shared__ __attribute__((aligned(16))) unsigned s_tab[256][4];
    constexpr int kStageBytes = DIRECT ? 16 : kExpandGroup * 1280 + 32;
    __shared__ __attribute__((aligned(16))) char s_stage[kWavesPerBlock][kStageBytes];
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned chunk = blockIdx.x;
    const unsigned cw0 = chunk * kChunkWords;
    const unsigned w0 = cw0 + wave * kExpandWords;
    unsigned base = (unsigned)chunk_entry[chunk];
    if (base == 0xffffffffu) return;                       // the scan gave up (mask_scan_kernel): no list
    // survivors of the chunk's earlier waves: lane l looks at words cw0 + l and cw0 + 64 + l
    unsigned before = 0;
    if (lane < wave * kExpandWords && cw0 + lane < n_words) before = (unsigned)__popcll(mask[cw0 + lane]);
    if (lane + 64u < wave * kExpandWords && cw0 + 64u + lane < n_words) before += (unsigned)__popcll(mask[cw0 + 64u + lane]);
    vd_u64 my_word = 0;                                   // lane l < 32 holds mask word w0 + l
    if (lane < (unsigned)kExpandWords && w0 + lane < n_words) my_word = mask[w0 + lane];
    // ids: lanes 16q..16q+15 of register g hold the 64 ids of the group's word q, four per lane (one dword); the
    // shard walk advances by increments (wps >= 4), one division per wave
    unsigned ids[kGroups];
    {
        const unsigned wl = w0 + (lane >> 4);
        unsigned shard = wl / wps, r = wl - shard * wps;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            const unsigned i0 = shard * shard_size + 64u * r + 4u * (lane & 15u);
            ids[g] = (wl + g * kExpandGroup < n_words && i0 < n_total) ? *reinterpret_cast<const unsigned*>(mesh_ids + i0) : 0u;
            r += kExpandGroup;
            if (r >= wps) { r -= wps; ++shard; }
        }
    }
    unsigned s_shard = w0 / wps, s_r = w0 - s_shard * wps;   // scalar walk over the wave's words
    unsigned word_first = s_shard * shard_size + 64u * s_r;
    // PF > 0: touch the mask words and ids of chunk c + 8 PF.  Workgroups are dealt to the 8 XCDs round-robin, so that
    // chunk will be expanded on this XCD and finds its inputs in this L2: behind a write-saturated L2 a load MISS
    // waits for an eviction (tens of microseconds), and these misses are nobody's critical path.
    unsigned pf = 0;
    if (PF > 0) {
        const unsigned pw = (chunk + 8u * PF) * kChunkWords + (threadIdx.x >> 1);
        if (pw < n_words) {
            const unsigned ps = pw / wps;
            const unsigned pi = ps * shard_size + 64u * (pw - ps * wps) + 32u * (threadIdx.x & 1u);
            if (pi + 32u <= n_total) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(mesh_ids + pi), b = *reinterpret_cast<const u32x4*>(mesh_ids + pi + 16u);
                pf = a.x ^ a.w ^ b.x ^ b.w;
            }
            if ((threadIdx.x & 1u) == 0u) pf ^= (unsigned)mask[pw];
        }
    }
    if (threadIdx.x < n_mesh) {
        const VdMeshInfo mi = meshes[threadIdx.x];
        s_tab[threadIdx.x][0] = mi.index_count; s_tab[threadIdx.x][1] = 1u;
        s_tab[threadIdx.x][2] = mi.base_index;  s_tab[threadIdx.x][3] = (unsigned)mi.vertex_offset;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    base = __builtin_amdgcn_readfirstlane(base + before);
    __syncthreads();
    const unsigned max_mid = n_mesh - 1u;
    const unsigned inst_lane = first_instance + lane;
    const unsigned id_shift = 8u * (lane & 3u);
    char* stage = s_stage[wave];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        vd_u64 m[kExpandGroup];
        unsigned cnt = 0;
#pragma unroll
        for (int q = 0; q < kExpandGroup; ++q) {
            const int k = g * kExpandGroup + q;
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)my_word, k), hi = __builtin_amdgcn_readlane((unsigned)(my_word >> 32), k);
            m[q] = ((vd_u64)hi << 32) | lo;
            cnt += (unsigned)__popcll(m[q]);
        }
        unsigned wf[kExpandGroup];                         // first instance of each word of the group
#pragma unroll
        for (int q = 0; q < kExpandGroup; ++q) {
            wf[q] = word_first;
            if (++s_r >= wps) { s_r = 0u; ++s_shard; }
            word_first = s_shard * shard_size + 64u * s_r;
        }
        if (cnt == 0u) continue;
        if (DIRECT) {
            unsigned run = base;
#pragma unroll
            for (int q = 0; q < kExpandGroup; ++q) {
                const unsigned v = (unsigned)__shfl((int)ids[g], q * 16 + (int)(lane >> 2));
                const unsigned mid = min((v >> id_shift) & 0xffu, max_mid);
                if (__builtin_amdgcn_inverse_ballot_w64(m[q])) {
                    unsigned* o = reinterpret_cast<unsigned*>(out + (run + vd_mbcnt(m[q])));
                    typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
                    *reinterpret_cast<u32x4_a4*>(o) = *reinterpret_cast<const u32x4*>(s_tab[mid]);
                    o[4] = inst_lane + wf[q];
                }
                run += (unsigned)__popcll(m[q]);
            }
        } else {
            char* gbase = reinterpret_cast<char*>(out + base);
            const unsigned shift = (unsigned)(reinterpret_cast<uintptr_t>(gbase) & 15u);
            unsigned run = 0;
#pragma unroll
            for (int q = 0; q < kExpandGroup; ++q) {
                const unsigned v = (unsigned)__shfl((int)ids[g], q * 16 + (int)(lane >> 2));
                const unsigned mid = min((v >> id_shift) & 0xffu, max_mid);
                if (__builtin_amdgcn_inverse_ballot_w64(m[q])) {
                    unsigned* o = reinterpret_cast<unsigned*>(stage + shift + 20u * (run + vd_mbcnt(m[q])));
                    const u32x4 c = *reinterpret_cast<const u32x4*>(s_tab[mid]);
                    o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
                    o[4] = inst_lane + wf[q];
                }
                run += (unsigned)__popcll(m[q]);
            }
            vd_wave_lds_sync();
            const unsigned total = shift + 20u * cnt;
            char* g16 = gbase - shift;
            for (unsigned b0 = lane * 16u; b0 < total; b0 += kWave * 16u) {
                if (b0 >= shift && b0 + 16u <= total) {
                    *reinterpret_cast<u32x4*>(g16 + b0) = *reinterpret_cast<const u32x4*>(stage + b0);
                } else {
                    const unsigned lo_b = b0 > shift ? b0 : shift, hi_b = b0 + 16u < total ? b0 + 16u : total;
                    for (unsigned b = lo_b; b < hi_b; b += 4u)
                        *reinterpret_cast<unsigned*>(g16 + b) = *reinterpret_cast<const unsigned*>(stage + b);
                }
            }
            vd_wave_lds_sync();
        }
        base += cnt;
    }
    if (PF > 0 && pf == 0x9e3779b9u && n_mesh == 0u) out[0].instance_count = pf;   // never true: keeps the prefetch loads
}

// Reference-format emission (C1: every slot written, shaders/emit_draws.wgsl:49-63) from pass 1's bits and ids: the
// split form of vd_cull_emit for large inputs.  A lane owns 4 consecutive instances = 80 contiguous bytes = five
// aligned 16-byte stores; no scan is needed (slot = instance index).
template <typename IdT>
__global__ __launch_bounds__(kBlock) void emit_from_mask_kernel(const vd_u64* __restrict__ mask, const IdT* __restrict__ mesh_ids,
                                                               const VdMeshInfo* __restrict__ meshes, unsigned n_mesh,
                                                               unsigned n_inst, unsigned first_instance,
                                                               VdDrawIndexedIndirect* __restrict__ out) {
    constexpr unsigned kTab = 512;
    __shared__ __attribute__((aligned(16))) unsigned s_tab[kTab][4];
    const bool tab = n_mesh <= kTab;
    if (tab)
        for (unsigned i = threadIdx.x; i < n_mesh; i += kBlock) {
            s_tab[i][0] = meshes[i].index_count; s_tab[i][1] = 0u;
            s_tab[i][2] = meshes[i].base_index;  s_tab[i][3] = (unsigned)meshes[i].vertex_offset;
        }
    __syncthreads();
    const unsigned n_quads = (n_inst + 3u) / 4u;
    for (unsigned q = blockIdx.x * kBlock + threadIdx.x; q < n_quads; q += gridDim.x * kBlock) {
        const unsigned i0 = q * 4u;
        const unsigned bits = (unsigned)(mask[i0 >> 6] >> (i0 & 63u)) & 15u;
        unsigned mid[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) mid[k] = i0 + k < n_inst ? min((unsigned)mesh_ids[i0 + k], n_mesh - 1u) : 0u;
        unsigned w[20];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32x4 c;
            if (tab) c = *reinterpret_cast<const u32x4*>(s_tab[mid[k]]);
            else { c.x = meshes[mid[k]].index_count; c.z = meshes[mid[k]].base_index; c.w = (unsigned)meshes[mid[k]].vertex_offset; }
            w[5 * k] = c.x; w[5 * k + 1] = (bits >> k) & 1u; w[5 * k + 2] = c.z; w[5 * k + 3] = c.w; w[5 * k + 4] = first_instance + i0 + k;
        }
        unsigned* o = reinterpret_cast<unsigned*>(out + i0);
        if (i0 + 4u <= n_inst) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                u32x4 v; v.x = w[4 * k]; v.y = w[4 * k + 1]; v.z = w[4 * k + 2]; v.w = w[4 * k + 3];
                *reinterpret_cast<u32x4*>(o + 4 * k) = v;
            }
        } else {
            for (unsigned k = 0; k < 5u * (n_inst - i0); ++k) o[k] = w[k];
        }
    }
}

// The same for 1-byte ids and a table of <= 256 meshes, with the access pattern of expand_mask_u8_kernel's direct
// form (a lane per instance: one 16-byte + one 4-byte store at a 20-byte lane stride, ids as one dword per lane
// redistributed with ds_bpermute), which runs at the store ceiling of the part.
__global__ __launch_bounds__(kBlock) void emit_all_u8_kernel(const vd_u64* __restrict__ mask, unsigned n_words, unsigned n_inst,
                                                             unsigned first_instance, const unsigned char* __restrict__ mesh_ids,
                                                             const VdMeshInfo* __restrict__ meshes, unsigned n_mesh,
                                                             VdDrawIndexedIndirect* __restrict__ out) {
    constexpr int kGroups = kExpandWords / kExpandGroup;
    __shared__ __attribute__((aligned(16))) unsigned s_tab[256][4];
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned w0 = (blockIdx.x * kWavesPerBlock + wave) * kExpandWords;
    vd_u64 my_word = 0;                                   // lane l < 32 holds mask word w0 + l
    if (lane < (unsigned)kExpandWords && w0 + lane < n_words) my_word = mask[w0 + lane];
    unsigned ids[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        const unsigned i0 = 64u * (w0 + g * kExpandGroup + (lane >> 4)) + 4u * (lane & 15u);
        ids[g] = i0 < n_inst ? *reinterpret_cast<const unsigned*>(mesh_ids + i0) : 0u;   // aligned dword with >= 1 valid byte
    }
    if (threadIdx.x < n_mesh) {
        const VdMeshInfo mi = meshes[threadIdx.x];
        s_tab[threadIdx.x][0] = mi.index_count; s_tab[threadIdx.x][1] = 0u;
        s_tab[threadIdx.x][2] = mi.base_index;  s_tab[threadIdx.x][3] = (unsigned)mi.vertex_offset;
    }
    __syncthreads();
    const unsigned max_mid = n_mesh - 1u;
    const unsigned id_shift = 8u * (lane & 3u);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
#pragma unroll
        for (int q = 0; q < kExpandGroup; ++q) {
            const int k = g * kExpandGroup + q;
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)my_word, k), hi = __builtin_amdgcn_readlane((unsigned)(my_word >> 32), k);
            const vd_u64 m = ((vd_u64)hi << 32) | lo;
            const unsigned v = (unsigned)__shfl((int)ids[g], q * 16 + (int)(lane >> 2));
            const unsigned mid = min((v >> id_shift) & 0xffu, max_mid);
            const unsigned i = 64u * (w0 + (unsigned)k) + lane;
            if (i < n_inst) {
                u32x4 c = *reinterpret_cast<const u32x4*>(s_tab[mid]);
                c.y = (unsigned)(m >> lane) & 1u;
                unsigned* o = reinterpret_cast<unsigned*>(out + i);
                typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
                *reinterpret_cast<u32x4_a4*>(o) = c;
                o[4] = first_instance + i;
            }
        }
    }
}

// Indices-only wire format: the set bits of a shard mask as an ascending list of global instance
// indices (4 B per survivor on the wire instead of 1 bit per instance: smaller below 1 survivor in 32), and the
// commands rebuilt from such a list.  Workgroup c owns mask chunk c as in pass 2b; no inter-workgroup dependency.
__global__ __launch_bounds__(kBlock) void mask_to_indices_kernel(const vd_u64* __restrict__ mask, unsigned n_words, unsigned first_instance,
                                                                 unsigned* __restrict__ out, const vd_u64* __restrict__ chunk_entry) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned cw0 = blockIdx.x * kChunkWords;
    const unsigned w0 = cw0 + wave * kExpandWords;
    unsigned base = (unsigned)chunk_entry[blockIdx.x];
    if (base == 0xffffffffu) return;                       // the scan gave up (mask_scan_kernel): no list
    unsigned before = 0;
    if (lane < wave * kExpandWords && cw0 + lane < n_words) before = (unsigned)__popcll(mask[cw0 + lane]);
    if (lane + 64u < wave * kExpandWords && cw0 + 64u + lane < n_words) before += (unsigned)__popcll(mask[cw0 + 64u + lane]);
    vd_u64 my_word = 0;                                   // lane l < 32 holds mask word w0 + l
    if (lane < (unsigned)kExpandWords && w0 + lane < n_words) my_word = mask[w0 + lane];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    base = __builtin_amdgcn_readfirstlane(base + before);
    const unsigned inst_lane = first_instance + 64u * w0 + lane;
#pragma unroll
    for (int k = 0; k < kExpandWords; ++k) {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)my_word, k), hi = __builtin_amdgcn_readlane((unsigned)(my_word >> 32), k);
        const vd_u64 m = ((vd_u64)hi << 32) | lo;
        if (__builtin_amdgcn_inverse_ballot_w64(m)) out[base + vd_mbcnt(m)] = inst_lane + 64u * (unsigned)k;
        base += (unsigned)__popcll(m);
    }
}

template <typename IdT>
__global__ __launch_bounds__(kBlock) void indices_to_draws_kernel(const unsigned* __restrict__ indices, unsigned n_indices,
                                                                  const IdT* __restrict__ mesh_ids, unsigned n_total,
                                                                  const VdMeshInfo* __restrict__ meshes, unsigned n_mesh,
                                                                  VdDrawIndexedIndirect* __restrict__ out) {
    constexpr unsigned kTab = 512;
    __shared__ __attribute__((aligned(16))) unsigned s_tab[kTab][4];     // ready-made {index_count, 1, base_index, vertex_offset}
    const bool tab = n_mesh <= kTab;
    if (tab)
        for (unsigned i = threadIdx.x; i < n_mesh; i += kBlock) {
            s_tab[i][0] = meshes[i].index_count; s_tab[i][1] = 1u;
            s_tab[i][2] = meshes[i].base_index;  s_tab[i][3] = (unsigned)meshes[i].vertex_offset;
        }
    __syncthreads();
    for (unsigned k = blockIdx.x * kBlock + threadIdx.x; k < n_indices; k += gridDim.x * kBlock) {
        const unsigned i = indices[k];
        const unsigned mid = min((unsigned)mesh_ids[min(i, n_total - 1u)], n_mesh - 1u);
        u32x4 c;
        if (tab) c = *reinterpret_cast<const u32x4*>(s_tab[mid]);
        else { c.x = meshes[mid].index_count; c.y = 1u; c.z = meshes[mid].base_index; c.w = (unsigned)meshes[mid].vertex_offset; }
        unsigned* o = reinterpret_cast<unsigned*>(out + k);
        typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
        *reinterpret_cast<u32x4_a4*>(o) = c;
        o[4] = i;
    }
}

// Host side of pass 2 (shared by vd_cull_compact* and vd_expand_mask_dev).
// Pass 2a: per-chunk survivor counts -> exclusive offsets (ctx->expand_state) and the total (*d_out_count).
static int launch_mask_scan(VdCtx* ctx, const vd_u64* d_mask, unsigned n_words, unsigned* d_out_count, vd_u64** out_entries) {
    const unsigned n_chunks = (n_words + kChunkWords - 1) / kChunkWords;
    const size_t need = sizeof(ScanState) + (size_t)n_chunks * 8;
    if (need > ctx->expand_state_bytes || !ctx->expand_state) {
        int rc = vd_ensure(ctx, &ctx->expand_state, &ctx->expand_state_bytes, need);
        if (rc) return rc;
        // entries zeroed = tagged with epoch 0; the first launch runs in epoch 1
        VD_HIP_CHECK(ctx, hipMemsetAsync(ctx->expand_state, 0, ctx->expand_state_bytes, ctx->stream));
        VD_HIP_CHECK(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(reinterpret_cast<char*>(ctx->expand_state) + offsetof(ScanState, epoch)), 1, 1, ctx->stream));
    }
    ScanState* state = reinterpret_cast<ScanState*>(ctx->expand_state);
    vd_u64* entries = reinterpret_cast<vd_u64*>(reinterpret_cast<char*>(ctx->expand_state) + sizeof(ScanState));
    unsigned sblocks = (n_chunks + 15u) / 16u;                                // one wave per chunk, grid-stride beyond 2 per CU
    if (sblocks > (unsigned)ctx->num_cus * 2u) sblocks = (unsigned)ctx->num_cus * 2u;
    hipLaunchKernelGGL(mask_scan_kernel, dim3(sblocks), dim3(kScanBlock), 0, ctx->stream, d_mask, n_words, n_chunks, entries, state,
                       d_out_count);
    *out_entries = entries;
    return VD_OK;
}

static int launch_expand(VdCtx* ctx, const vd_u64* d_mask, unsigned n_words, unsigned wps, unsigned shard_size,
                         unsigned n_total, unsigned first_instance, const void* d_ids, unsigned id_bytes,
                         const VdMeshInfo* d_meshes, unsigned n_mesh, VdDrawIndexedIndirect* d_out, unsigned* d_out_count) {
    const unsigned n_chunks = (n_words + kChunkWords - 1) / kChunkWords;
    vd_u64* offsets;
    int rc_scan = launch_mask_scan(ctx, d_mask, n_words, d_out_count, &offsets);
    if (rc_scan) return rc_scan;
    const bool one_shard = wps >= n_words;
    const bool tab = n_mesh <= 512u;
    // fast path: 1-byte ids that can be fetched as aligned dwords (every word's first instance is a multiple of 4)
    const bool fast = id_bytes == 1u && n_mesh <= 256u && (one_shard || (shard_size % 4u == 0u && wps >= (unsigned)kExpandGroup)) &&
                      (reinterpret_cast<uintptr_t>(d_ids) & 3u) == 0u;
    if (fast) {
        // up to ~250 MB of commands (the Infinity Cache absorbs them) the direct form is at the write ceiling; past
        // that the L2 merges fewer of its 4-byte pieces in time and the LDS-staged 16-byte runs win (A/B: -71 / -74)
        const bool direct = ctx->cull_variant == -74 || (ctx->cull_variant > -71 && n_total <= (12u << 20));
#define VD_EXPAND_U8(D, P)                                                                                                \
        hipLaunchKernelGGL((expand_mask_u8_kernel<D, P>), dim3(n_chunks), dim3(kBlock), 0, ctx->stream, d_mask, n_words, wps,  \
                           shard_size, n_total, first_instance, reinterpret_cast<const unsigned char*>(d_ids), d_meshes,    \
                           n_mesh, d_out, offsets)
        if (direct) VD_EXPAND_U8(true, 0);
        else if (ctx->cull_variant == -81) VD_EXPAND_U8(false, 0);          // A/B: without the same-XCD prefetch
        else VD_EXPAND_U8(false, 32);
#undef VD_EXPAND_U8
        return VD_OK;
    }
#define VD_EXPAND(IdT, T)                                                                                               \
    hipLaunchKernelGGL((expand_mask_kernel<IdT, T>), dim3(n_chunks), dim3(kBlock), 0, ctx->stream, d_mask, n_words, wps,  \
                       shard_size, n_total, first_instance, reinterpret_cast<const IdT*>(d_ids), d_meshes, n_mesh, d_out,    \
                       offsets)
    if (id_bytes == 1u) { if (tab) VD_EXPAND(unsigned char, true); else VD_EXPAND(unsigned char, false); }
    else if (id_bytes == 2u) { if (tab) VD_EXPAND(unsigned short, true); else VD_EXPAND(unsigned short, false); }
    else { if (tab) VD_EXPAND(unsigned, true); else VD_EXPAND(unsigned, false); }
#undef VD_EXPAND
    return VD_OK;
}

// Zero-fill out[count..n) so the unchanged multi_draw_indexed_indirect(buf, 0, N) consumer
// (visibility.rs:188-192) sees instance_count = 0 in the tail.  The tail starts on a 4-byte boundary (count x 20 B): up to
// three single dwords bring it to a 16-byte one, the body leaves as nontemporal 16-byte stores (the `dist small` cloud pads
// 126 MB per step), the last few dwords singly again.  A count beyond n is the scan's error value (VD_SCAN_STUCK: no list was
// written): then the WHOLE buffer is zeroed, so the consumer draws nothing instead of a mix of this frame's and the last
// frame's commands.
__global__ __launch_bounds__(kBlock) void pad_tail_kernel(VdDrawIndexedIndirect* __restrict__ out,
                                                          const unsigned* __restrict__ count, unsigned n) {
    const unsigned c = *count;
    const size_t begin = (c > n ? (size_t)0 : (size_t)c) * 5u, end = (size_t)n * 5u;
    unsigned* o = reinterpret_cast<unsigned*>(out);
    if (begin >= end) return;
    const size_t gid = (size_t)blockIdx.x * kBlock + threadIdx.x, gsz = (size_t)gridDim.x * kBlock;
    const size_t mis = (reinterpret_cast<uintptr_t>(o + begin) >> 2) & 3u;            // dwords past a 16-byte boundary
    const size_t head = min(end - begin, (4u - mis) & 3u);
    const size_t quads = (end - begin - head) >> 2;
    const size_t tail_first = begin + head + quads * 4u;
    if (gid < head) o[begin + gid] = 0u;
    if (gid < end - tail_first) o[tail_first + gid] = 0u;
    u32x4* q = reinterpret_cast<u32x4*>(o + begin + head);
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (size_t i = gid; i < quads; i += gsz) __builtin_nontemporal_store(z, q + i);
}

// ------------------------------------------------------------------------------------------
// C3 alone: ordered compaction of an existing command buffer (pure function of C1's output).
// ------------------------------------------------------------------------------------------
constexpr int kCompactPerThread = 8;
constexpr int kCompactTile = kBlock * kCompactPerThread;

__global__ __launch_bounds__(kBlock) void compact_draws_kernel(const VdDrawIndexedIndirect* __restrict__ in, unsigned n,
                                                               VdDrawIndexedIndirect* __restrict__ out,
                                                               unsigned* __restrict__ out_count, vd_u64* tile_state,
                                                               vd_u64* ticket_counter, unsigned n_tiles, unsigned* fault_host) {
    __shared__ unsigned s_ticket, s_epoch, s_wave_total[kWavesPerBlock], s_tile_excl;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_ticket = vd_take_ticket(ticket_counter, n_tiles, &s_epoch);
    __syncthreads();
    const unsigned tile = s_ticket, epoch = s_epoch;
    if (tile >= n_tiles) return;                  // (the ticket word was not at {epoch, 0} when the launch began: never expected)
    if (tile == 0u && threadIdx.x == 0) { __hip_atomic_store(out_count, VD_SCAN_STUCK, VD_RLX_AGENT); __threadfence(); }   // see cull_compact_kernel
    const size_t wave_first = (size_t)tile * kCompactTile + (size_t)wave * (kWave * kCompactPerThread);
    unsigned long long masks[kCompactPerThread];
    unsigned wave_total = 0;
    const unsigned* in32 = reinterpret_cast<const unsigned*>(in);
#pragma unroll
    for (int r = 0; r < kCompactPerThread; ++r) {
        const size_t i = wave_first + (size_t)r * kWave + lane;
        const bool keep = i < n && in32[i * 5u + 1u] == 1u;
        masks[r] = __ballot(keep);
        wave_total += (unsigned)__popcll(masks[r]);
    }
    if (lane == 0) s_wave_total[wave] = wave_total;
    __syncthreads();
    if (wave == 0) {
        unsigned tile_total = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) tile_total += s_wave_total[w];
        const unsigned excl = vd_lookback(tile_state, epoch, tile, tile_total);
        if (lane == 0) {
            s_tile_excl = excl;
            if (tile == n_tiles - 1u) *out_count = vd_scan_final_count(tile_state, epoch, excl, tile_total, fault_host);
        }
    }
    __syncthreads();
    unsigned base = s_tile_excl;
    if (base == VD_SCAN_STUCK) return;
    for (unsigned w = 0; w < wave; ++w) base += s_wave_total[w];
    unsigned* out32 = reinterpret_cast<unsigned*>(out);
#pragma unroll
    for (int r = 0; r < kCompactPerThread; ++r) {
        const unsigned long long mask = masks[r];
        if ((mask >> lane) & 1ull) {
            const size_t i = wave_first + (size_t)r * kWave + lane;
            const size_t d = (size_t)(base + vd_mbcnt(mask)) * 5u;
#pragma unroll
            for (int k = 0; k < 5; ++k) out32[d + k] = in32[i * 5u + k];
        }
        base += (unsigned)__popcll(mask);
    }
}

// compute_update.wgsl:10-28 — FOUR lanes per listed instance, one matrix column (16 B) each, so a wave's load is 16
// contiguous 64-byte pieces instead of 64 pieces of 16 bytes in 64 different lines (10 M instances with
// fix_inverse: 0.685 -> 0.571 ms; transform only: 0.630 -> 0.604 ms - a read-modify-write stream over 144-byte records).  rotz * transform acts on each column
// separately; inv_transform * rotz(-angle) needs all four columns, which the quad trades through DPP.  Same operation
// order as the oracle's mat_mul_cm (products by the rotation's zeros and ones included: they matter for inf / NaN);
// (c, s) for both signs come from the host.
struct RotZ { float c_pos, s_pos, c_neg, s_neg; };
template <int SRC> __device__ __forceinline__ float quad_bcast(float v) {          // value of lane SRC of this lane's quad
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), SRC * 0x55, 0xf, 0xf, true));
}
__global__ __launch_bounds__(256) void compute_update_kernel(const unsigned* __restrict__ indices, unsigned n_indices,
                                                             VdInstance* __restrict__ inst, unsigned n_inst, RotZ rz, int fix_inverse) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x, k = t >> 2, col = t & 3u;
    const unsigned idx = k < n_indices ? indices[k] : 0xffffffffu;
    const bool live = idx < n_inst;                         // out-of-range ids are dropped; the whole quad agrees
    float4* t4 = reinterpret_cast<float4*>(inst[live ? idx : 0u].transform);
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live) v = t4[col];
    const float t14 = quad_bcast<3>(v.z);                   // transform[3][2]
    const bool pos = t14 > -15.0f;
    const float c = pos ? rz.c_pos : rz.c_neg, s = pos ? rz.s_pos : rz.s_neg;
    // column j of R * T, R = columns (c, s, 0, 0), (-s, c, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)
    float4 o;
    o.x = ((c * v.x + -s * v.y) + 0.0f * v.z) + 0.0f * v.w;
    o.y = ((s * v.x + c * v.y) + 0.0f * v.z) + 0.0f * v.w;
    o.z = ((0.0f * v.x + 0.0f * v.y) + 1.0f * v.z) + 0.0f * v.w;
    o.w = ((0.0f * v.x + 0.0f * v.y) + 0.0f * v.z) + 1.0f * v.w;
    if (live) t4[col] = o;
    if (fix_inverse) {
        float4* i4 = reinterpret_cast<float4*>(inst[live ? idx : 0u].inv_transform);
        float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (live) w = i4[col];
        // all four columns of inv_transform: I0 .. I3
        const float4 I0 = make_float4(quad_bcast<0>(w.x), quad_bcast<0>(w.y), quad_bcast<0>(w.z), quad_bcast<0>(w.w));
        const float4 I1 = make_float4(quad_bcast<1>(w.x), quad_bcast<1>(w.y), quad_bcast<1>(w.z), quad_bcast<1>(w.w));
        const float4 I2 = make_float4(quad_bcast<2>(w.x), quad_bcast<2>(w.y), quad_bcast<2>(w.z), quad_bcast<2>(w.w));
        const float4 I3 = make_float4(quad_bcast<3>(w.x), quad_bcast<3>(w.y), quad_bcast<3>(w.z), quad_bcast<3>(w.w));
        // column `col` of rotz(-angle): (c, -s, 0, 0), (s, c, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)
        const float b0 = col == 0u ? c : (col == 1u ? s : 0.0f);
        const float b1 = col == 0u ? -s : (col == 1u ? c : 0.0f);
        const float b2 = col == 2u ? 1.0f : 0.0f, b3 = col == 3u ? 1.0f : 0.0f;
        float4 r;
        r.x = ((I0.x * b0 + I1.x * b1) + I2.x * b2) + I3.x * b3;
        r.y = ((I0.y * b0 + I1.y * b1) + I2.y * b2) + I3.y * b3;
        r.z = ((I0.z * b0 + I1.z * b1) + I2.z * b2) + I3.z * b3;
        r.w = ((I0.w * b0 + I1.w * b1) + I2.w * b2) + I3.w * b3;
        if (live) i4[col] = r;
    }
}

CullCamera make_cam(const VdCameraUniform* c) {
    CullCamera k;
    memcpy(k.view, c->view, sizeof(k.view));
    memcpy(k.frustum, c->frustum, sizeof(k.frustum));
    k.znear = c->znear;
    k.zfar = c->zfar;
    return k;
}

}  // namespace

extern "C" {

int vd_cull_emit_dev(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                     const VdInstance* d_instances, uint32_t n_inst, VdDrawIndexedIndirect* d_out) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    return vd_cull_emit_shard_dev(ctx, camera, d_meshes, n_mesh, d_instances, n_inst, 0u, d_out);
}

// Pass 1 of the split forms: instances -> one bit + a compact mesh id each, in ctx scratch.
static int launch_mask_pass(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                            const VdInstance* d_instances, uint32_t n_inst, vd_u64** out_mask, void** out_ids, unsigned* out_id_bytes) {
    const unsigned n_words = (n_inst + 63u) / 64u;
    const unsigned id_bytes = n_mesh <= 256u ? 1u : (n_mesh <= 65536u ? 2u : 4u);
    const size_t need = (size_t)n_words * 8 + (size_t)n_inst * id_bytes + 512;
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, need);
    if (rc) return rc;
    vd_u64* d_mask = reinterpret_cast<vd_u64*>(ctx->scratch);
    void* d_ids = reinterpret_cast<char*>(ctx->scratch) + (((size_t)n_words * 8 + 255) & ~(size_t)255);
    vd_time_begin(ctx);
    const unsigned n_wave_tiles = n_words;
    unsigned blocks = (n_wave_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned cap = (unsigned)ctx->num_cus * 4u;
    if (blocks > cap) blocks = cap;
#define VD_SPLIT(IdT)                                                                                              \
    do {                                                                                                         \
        if (ctx->cull_variant == -70) {                                                                          \
            hipLaunchKernelGGL(cull_mask_kernel<IdT>, dim3(blocks), dim3(kBlock), kWavesPerBlock * kSlabBytes,   \
                               ctx->stream, make_cam(camera), d_meshes, n_mesh, d_instances, n_inst, d_mask,     \
                               reinterpret_cast<IdT*>(d_ids), n_wave_tiles);                                     \
        } else {                                                                                                 \
            const unsigned n_mt = (n_inst + kWave * kMaskRounds - 1) / (kWave * kMaskRounds);                    \
            unsigned mb = (n_mt + kWavesPerBlock - 1) / kWavesPerBlock;                                          \
            if (mb > (unsigned)ctx->num_cus * 3u) mb = (unsigned)ctx->num_cus * 3u;                              \
            hipLaunchKernelGGL(cull_mask_tiled_kernel<IdT>, dim3(mb), dim3(kBlock),                              \
                               kWavesPerBlock * (kSlabBytes + kMaskRounds * kWave * (int)sizeof(IdT)),           \
                               ctx->stream, make_cam(camera), d_meshes, n_mesh, d_instances, n_inst, d_mask,     \
                               reinterpret_cast<IdT*>(d_ids), n_mt);                                             \
        }                                                                                                        \
    } while (0)
    if (id_bytes == 1u) VD_SPLIT(unsigned char);
    else if (id_bytes == 2u) VD_SPLIT(unsigned short);
    else VD_SPLIT(unsigned);
#undef VD_SPLIT
    vd_time_mid(ctx);
    *out_mask = d_mask; *out_ids = d_ids; *out_id_bytes = id_bytes;
    return VD_OK;
}

int vd_cull_emit_shard_dev(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                           const VdInstance* d_instances, uint32_t n_inst, uint32_t first_instance,
                           VdDrawIndexedIndirect* d_out) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!camera || !d_meshes || n_mesh == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_emit: null camera/meshes or n_mesh == 0");
    if (n_inst == 0) return VD_OK;
    if (!d_instances || !d_out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_emit: null instances/out");
    if (ctx->cull_variant <= 0 && ctx->cull_variant != -80 && n_inst >= ctx->split_min) {
        // split form, as for the compacted list: the 20-byte stores leave the read stream
        vd_u64* d_mask; void* d_ids; unsigned id_bytes;
        int rc = launch_mask_pass(ctx, camera, d_meshes, n_mesh, d_instances, n_inst, &d_mask, &d_ids, &id_bytes);
        if (rc) return rc;
        const unsigned quads = (n_inst + 3u) / 4u;
        unsigned eb = (quads + kBlock - 1) / kBlock;
        if (eb > (unsigned)ctx->num_cus * 16u) eb = (unsigned)ctx->num_cus * 16u;
#define VD_EMIT(IdT) hipLaunchKernelGGL(emit_from_mask_kernel<IdT>, dim3(eb), dim3(kBlock), 0, ctx->stream, d_mask,               \
                                        reinterpret_cast<const IdT*>(d_ids), d_meshes, n_mesh, n_inst, first_instance, d_out)
        if (id_bytes == 1u) {
            const unsigned n_words = (n_inst + 63u) / 64u;
            hipLaunchKernelGGL(emit_all_u8_kernel, dim3((n_words + kChunkWords - 1) / kChunkWords), dim3(kBlock), 0, ctx->stream, d_mask,
                               n_words, n_inst, first_instance, reinterpret_cast<const unsigned char*>(d_ids), d_meshes, n_mesh, d_out);
        } else if (id_bytes == 2u) VD_EMIT(unsigned short); else VD_EMIT(unsigned);
#undef VD_EMIT
        vd_time_end(ctx);
        VD_HIP_CHECK(ctx, hipGetLastError());
        return VD_OK;
    }
    const unsigned n_wave_tiles = (n_inst + kWave - 1) / kWave;
    unsigned blocks = (n_wave_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned cap = (unsigned)ctx->num_cus * 4u;   // 4 x 36 KB LDS slabs per CU
    if (blocks > cap) blocks = cap;
    vd_time_begin(ctx);
    hipLaunchKernelGGL(emit_draws_kernel, dim3(blocks), dim3(kBlock), kWavesPerBlock * kSlabBytes, ctx->stream,
                       make_cam(camera), d_meshes, n_mesh, d_instances, n_inst, d_out, n_wave_tiles, first_instance);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_cull_compact_dev(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                        const VdInstance* d_instances, uint32_t n_inst, VdDrawIndexedIndirect* d_out,
                        uint32_t* d_out_count, int pad_tail) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    return vd_cull_compact_shard_dev(ctx, camera, d_meshes, n_mesh, d_instances, n_inst, 0u, d_out, d_out_count, pad_tail);
}

int vd_cull_compact_shard_dev(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                              const VdInstance* d_instances, uint32_t n_inst, uint32_t first_instance,
                              VdDrawIndexedIndirect* d_out, uint32_t* d_out_count, int pad_tail) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!camera || !d_meshes || n_mesh == 0 || !d_out_count)
        VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_compact: null camera/meshes/count or n_mesh == 0");
    if (n_inst == 0) {
        VD_HIP_CHECK(ctx, hipMemsetAsync(d_out_count, 0, 4, ctx->stream));
        return VD_OK;
    }
    if (!d_instances || !d_out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_compact: null instances/out");
    int variant = ctx->cull_variant;
    vd_u64* ticket; vd_u64* states;
    int rc = vd_scan_check_fault(ctx);       // an EARLIER launch's scan gave up: said once, here
    if (rc) return rc;
    if ((variant <= 0) && n_inst >= ctx->split_min) {
        // Split form (default for large inputs): pass 1 streams the instances and writes only one bit
        // + a compact mesh id per instance (reads run at ~6.4 TB/s when no 20-byte commands are stored
        // in the same kernel); pass 2 expands the bits into the ordered command list.  Mixing the
        // command stores into the read stream costs more than the 1-5 B/instance round trip
        // (A/B: profiles/).
        vd_u64* d_mask; void* d_ids; unsigned id_bytes;
        rc = launch_mask_pass(ctx, camera, d_meshes, n_mesh, d_instances, n_inst, &d_mask, &d_ids, &id_bytes);
        if (rc) return rc;
        const unsigned n_words = (n_inst + 63u) / 64u;
        rc = launch_expand(ctx, d_mask, n_words, n_words, n_inst, n_inst, first_instance, d_ids, id_bytes, d_meshes, n_mesh,
                           d_out, d_out_count);
        if (rc) return rc;
        vd_time_end(ctx);
        if (pad_tail) {
            unsigned pblocks = (unsigned)ctx->num_cus * 4u;
            hipLaunchKernelGGL(pad_tail_kernel, dim3(pblocks), dim3(kBlock), 0, ctx->stream, d_out, d_out_count, n_inst);
        }
        VD_HIP_CHECK(ctx, hipGetLastError());
        return VD_OK;
    }
#define VD_LAUNCH_COMPACT(R)                                                                                     \
    do {                                                                                                         \
        const unsigned n_tiles = (n_inst + kBlock * (R) - 1) / (kBlock * (R));                                   \
        rc = vd_scan_scratch(ctx, n_tiles, &ticket, &states, true);                                              \
        if (rc) return rc;                                                                                       \
        hipLaunchKernelGGL((cull_compact_kernel<R>), dim3(n_tiles), dim3(kBlock), (compact_lds_bytes<R>()),      \
                           ctx->stream, make_cam(camera), d_meshes, n_mesh, d_instances, n_inst, d_out,          \
                           d_out_count, states, ticket, n_tiles, first_instance, vd_scan_fault_word(ctx));       \
    } while (0)
    // fused form: tile size grows with n so that ticket + two barriers + look-back amortise while
    // small inputs still spread over the chip (a 100 k-instance scene in 1024-instance tiles is 98
    // workgroups on 256 CUs; thresholds from profiles/r04_ab_cull_small.log).  variant > 0 forces a
    // tile size (tools/ab_cull.py).
    const int rounds = variant > 0 ? variant : (n_inst >= (4u << 20) ? 32 : (n_inst >= (5u << 18) ? 16 : (n_inst >= 600000u ? 8 : (n_inst >= 192000u ? 4 : (n_inst >= 48000u ? 2 : 1)))));
    switch (rounds) {
        case 1: VD_LAUNCH_COMPACT(1); break;
        case 2: VD_LAUNCH_COMPACT(2); break;
        case 4: VD_LAUNCH_COMPACT(4); break;
        case 8: VD_LAUNCH_COMPACT(8); break;
        case 16: VD_LAUNCH_COMPACT(16); break;
        default: VD_LAUNCH_COMPACT(32); break;
    }
#undef VD_LAUNCH_COMPACT
    vd_time_end(ctx);
    if (pad_tail) {
        unsigned blocks = (unsigned)ctx->num_cus * 4u;
        hipLaunchKernelGGL(pad_tail_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, d_out, d_out_count, n_inst);
    }
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_cull_mask_dev(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                     const VdInstance* d_instances, uint32_t n_inst, uint64_t* d_mask) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!camera || !d_meshes || n_mesh == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_mask: null camera/meshes or n_mesh == 0");
    if (n_inst == 0) return VD_OK;
    if (!d_instances || !d_mask) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_mask: null instances/mask");
    const unsigned n_wave_tiles = (n_inst + kWave - 1) / kWave;
    unsigned blocks = (n_wave_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned cap = (unsigned)ctx->num_cus * 4u;
    if (blocks > cap) blocks = cap;
    vd_time_begin(ctx);
    hipLaunchKernelGGL(cull_mask_kernel<unsigned>, dim3(blocks), dim3(kBlock), kWavesPerBlock * kSlabBytes, ctx->stream, make_cam(camera),
                       d_meshes, n_mesh, d_instances, n_inst, reinterpret_cast<vd_u64*>(d_mask), (unsigned*)nullptr, n_wave_tiles);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_occlusion_mask_dev(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                          const VdInstance* d_instances, uint32_t n_inst, const float* d_pyramid, uint32_t width, uint32_t height,
                          const uint64_t* d_mask_in, uint64_t* d_mask_out) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!camera || !d_meshes || n_mesh == 0 || !d_pyramid) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_occlusion_mask: null camera/meshes/pyramid or n_mesh == 0");
    VdHizLayout L;
    if (vd_hiz_layout(width, height, &L)) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_occlusion_mask: bad pyramid size");
    if (!(camera->projection[11] == -1.0f && camera->projection[15] == 0.0f))
        VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_occlusion_mask: projection is not a right-handed perspective matrix (projection[11] == -1, [15] == 0)");
    if (n_inst == 0) return VD_OK;
    if (!d_instances || !d_mask_in || !d_mask_out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_occlusion_mask: null instances/masks");
    OccCamera oc;
    for (int k = 0; k < 16; ++k) oc.view[k] = camera->view[k];
    const float* P = camera->projection;
    oc.p00 = P[0]; oc.p11 = P[5]; oc.p20 = P[8]; oc.p21 = P[9]; oc.p22 = P[10]; oc.p32 = P[14]; oc.znear = camera->znear;
    HizView hz;
    hz.base = d_pyramid; hz.width = width; hz.height = height; hz.n_levels = L.n_levels;
    for (int k = 0; k < 17; ++k) hz.off[k] = L.level_offset[k];
    const unsigned n_wave_tiles = (n_inst + kWave - 1) / kWave;
    unsigned blocks = (n_wave_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned cap = (unsigned)ctx->num_cus * 8u;
    if (blocks > cap) blocks = cap;
    vd_time_begin(ctx);
    hipLaunchKernelGGL(occlusion_mask_kernel, dim3(blocks), dim3(kBlock), kWavesPerBlock * kSlabBytes, ctx->stream, oc, hz, d_meshes, n_mesh,
                       d_instances, n_inst, reinterpret_cast<const vd_u64*>(d_mask_in), reinterpret_cast<vd_u64*>(d_mask_out), n_wave_tiles);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_expand_mask_dev(VdCtx* ctx, const uint64_t* d_mask, uint32_t n_total, uint32_t shard_size, const void* d_mesh_ids,
                       uint32_t id_bytes, const VdMeshInfo* d_meshes, uint32_t n_mesh, VdDrawIndexedIndirect* d_out,
                       uint32_t* d_out_count) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!d_out_count || !d_meshes || n_mesh == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_expand_mask: null count/meshes");
    if (n_total == 0) {
        VD_HIP_CHECK(ctx, hipMemsetAsync(d_out_count, 0, 4, ctx->stream));
        return VD_OK;
    }
    if (!d_mask || !d_mesh_ids || !d_out || shard_size == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_expand_mask: null mask/ids/out or shard_size == 0");
    if (id_bytes != 1u && id_bytes != 2u && id_bytes != 4u) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_expand_mask: id_bytes must be 1, 2 or 4");
    const unsigned n_shards = (n_total + shard_size - 1) / shard_size;
    const unsigned wps = (shard_size + 63u) / 64u;
    const unsigned n_words = n_shards * wps;   // padding bits (beyond a shard's / the scene's end) are 0 by construction
    vd_time_begin(ctx);
    int rc = launch_expand(ctx, reinterpret_cast<const vd_u64*>(d_mask), n_words, wps, shard_size, n_total, 0u, d_mesh_ids,
                           id_bytes, d_meshes, n_mesh, d_out, d_out_count);
    if (rc) return rc;
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_mask_to_indices_dev(VdCtx* ctx, const uint64_t* d_mask, uint32_t n_inst, uint32_t first_instance, uint32_t* d_out_indices,
                           uint32_t* d_out_count) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!d_out_count) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_mask_to_indices: null count");
    if (n_inst == 0) {
        VD_HIP_CHECK(ctx, hipMemsetAsync(d_out_count, 0, 4, ctx->stream));
        return VD_OK;
    }
    if (!d_mask || !d_out_indices) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_mask_to_indices: null mask/out");
    const unsigned n_words = (n_inst + 63u) / 64u;
    vd_time_begin(ctx);
    vd_u64* offsets;
    int rc = launch_mask_scan(ctx, reinterpret_cast<const vd_u64*>(d_mask), n_words, d_out_count, &offsets);
    if (rc) return rc;
    hipLaunchKernelGGL(mask_to_indices_kernel, dim3((n_words + kChunkWords - 1) / kChunkWords), dim3(kBlock), 0, ctx->stream,
                       reinterpret_cast<const vd_u64*>(d_mask), n_words, first_instance, d_out_indices, offsets);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_indices_to_draws_dev(VdCtx* ctx, const uint32_t* d_indices, uint32_t n_indices, const void* d_mesh_ids, uint32_t id_bytes,
                            uint32_t n_total, const VdMeshInfo* d_meshes, uint32_t n_mesh, VdDrawIndexedIndirect* d_out) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!d_meshes || n_mesh == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_indices_to_draws: null meshes or n_mesh == 0");
    if (n_indices == 0) return VD_OK;
    if (!d_indices || !d_mesh_ids || !d_out || n_total == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_indices_to_draws: null indices/ids/out or n_total == 0");
    if (id_bytes != 1u && id_bytes != 2u && id_bytes != 4u) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_indices_to_draws: id_bytes must be 1, 2 or 4");
    unsigned blocks = (n_indices + kBlock - 1) / kBlock;
    if (blocks > (unsigned)ctx->num_cus * 16u) blocks = (unsigned)ctx->num_cus * 16u;
    vd_time_begin(ctx);
#define VD_I2D(IdT) hipLaunchKernelGGL(indices_to_draws_kernel<IdT>, dim3(blocks), dim3(kBlock), 0, ctx->stream, d_indices, n_indices,   \
                                       reinterpret_cast<const IdT*>(d_mesh_ids), n_total, d_meshes, n_mesh, d_out)
    if (id_bytes == 1u) VD_I2D(unsigned char); else if (id_bytes == 2u) VD_I2D(unsigned short); else VD_I2D(unsigned);
#undef VD_I2D
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_compact_draws_dev(VdCtx* ctx, const VdDrawIndexedIndirect* d_in, uint32_t n, VdDrawIndexedIndirect* d_out,
                         uint32_t* d_out_count) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!d_out_count) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_compact_draws: null count");
    if (n == 0) {
        VD_HIP_CHECK(ctx, hipMemsetAsync(d_out_count, 0, 4, ctx->stream));
        return VD_OK;
    }
    if (!d_in || !d_out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_compact_draws: null in/out");
    const unsigned n_tiles = (n + kCompactTile - 1) / kCompactTile;
    vd_u64* ticket; vd_u64* states;
    int rc = vd_scan_check_fault(ctx);
    if (rc) return rc;
    rc = vd_scan_scratch(ctx, n_tiles, &ticket, &states, true);
    if (rc) return rc;
    hipLaunchKernelGGL(compact_draws_kernel, dim3(n_tiles), dim3(kBlock), 0, ctx->stream, d_in, n, d_out, d_out_count,
                       states, ticket, n_tiles, vd_scan_fault_word(ctx));
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_compute_update_dev(VdCtx* ctx, const uint32_t* d_indices, uint32_t n_indices, VdInstance* d_instances,
                          uint32_t n_instances, float time, float dt, int fix_inverse) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (n_indices == 0) return VD_OK;
    if (!d_indices || !d_instances) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_compute_update: null indices/instances");
    const float speed = 2.0f * sinf(time * 0.5f);          // compute_update.wgsl:20
    const float a_pos = (speed * 1.0f) * dt, a_neg = (speed * -1.0f) * dt;
    RotZ rz{cosf(a_pos), sinf(a_pos), cosf(a_neg), sinf(a_neg)};
    vd_time_begin(ctx);
    hipLaunchKernelGGL(compute_update_kernel, dim3((unsigned)(((size_t)n_indices * 4u + 255u) / 256u)), dim3(256), 0, ctx->stream, d_indices, n_indices,
                       d_instances, n_instances, rz, fix_inverse);
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

// ---- host-pointer variants: stage through ctx-owned device buffers ----------------------
static int stage_cull_inputs(VdCtx* ctx, const VdMeshInfo* meshes, uint32_t n_mesh, const VdInstance* instances,
                             uint32_t n_inst, VdMeshInfo** d_meshes, VdInstance** d_inst, VdDrawIndexedIndirect** d_out) {
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    int rc = vd_ensure(ctx, &ctx->stage_in, &ctx->stage_in_bytes, (size_t)n_inst * sizeof(VdInstance));
    if (rc) return rc;
    rc = vd_ensure(ctx, &ctx->stage_aux, &ctx->stage_aux_bytes, (size_t)n_mesh * sizeof(VdMeshInfo) + 16);
    if (rc) return rc;
    rc = vd_ensure(ctx, &ctx->stage_out, &ctx->stage_out_bytes, (size_t)n_inst * sizeof(VdDrawIndexedIndirect) + 16);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->stage_in, instances, (size_t)n_inst * sizeof(VdInstance), hipMemcpyHostToDevice, ctx->stream));
    VD_HIP_CHECK(ctx, hipMemcpyAsync(reinterpret_cast<char*>(ctx->stage_aux) + 16, meshes, (size_t)n_mesh * sizeof(VdMeshInfo),
                                     hipMemcpyHostToDevice, ctx->stream));
    *d_inst = reinterpret_cast<VdInstance*>(ctx->stage_in);
    *d_meshes = reinterpret_cast<VdMeshInfo*>(reinterpret_cast<char*>(ctx->stage_aux) + 16);
    *d_out = reinterpret_cast<VdDrawIndexedIndirect*>(ctx->stage_out);
    return VD_OK;
}

int vd_cull_emit(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* meshes, uint32_t n_mesh,
                 const VdInstance* instances, uint32_t n_inst, VdDrawIndexedIndirect* out) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!camera || !meshes || n_mesh == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_emit: null camera/meshes or n_mesh == 0");
    if (n_inst == 0) return VD_OK;
    if (!instances || !out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_emit: null instances/out");
    VdMeshInfo* dm; VdInstance* di; VdDrawIndexedIndirect* dout;
    int rc = stage_cull_inputs(ctx, meshes, n_mesh, instances, n_inst, &dm, &di, &dout);
    if (rc) return rc;
    rc = vd_cull_emit_dev(ctx, camera, dm, n_mesh, di, n_inst, dout);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(out, dout, (size_t)n_inst * sizeof(VdDrawIndexedIndirect), hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VD_OK;
}

int vd_cull_compact(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* meshes, uint32_t n_mesh,
                    const VdInstance* instances, uint32_t n_inst, VdDrawIndexedIndirect* out, uint32_t* out_count,
                    int pad_tail) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!camera || !meshes || n_mesh == 0 || !out_count)
        VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_compact: null camera/meshes/count or n_mesh == 0");
    *out_count = 0;
    if (n_inst == 0) return VD_OK;
    if (!instances || !out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_compact: null instances/out");
    VdMeshInfo* dm; VdInstance* di; VdDrawIndexedIndirect* dout;
    int rc = stage_cull_inputs(ctx, meshes, n_mesh, instances, n_inst, &dm, &di, &dout);
    if (rc) return rc;
    uint32_t* d_count = reinterpret_cast<uint32_t*>(ctx->stage_aux);
    rc = vd_cull_compact_dev(ctx, camera, dm, n_mesh, di, n_inst, dout, d_count, pad_tail);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->host_pinned, d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out_count = ctx->host_pinned[0];
    // a cross-workgroup wait of the scan timed out (vd_common.hpp): the fault word is up and the count is 0 - or the launch lost its
    // LAST workgroup and the count still holds the value the first one pre-stored, VD_SCAN_STUCK.  No list was written either way.
    if (*out_count > n_inst || ctx->host_pinned[kScanFaultWord] != 0u) {
        *out_count = 0;
        ctx->host_pinned[kScanFaultWord] = 0u;
        if (ctx->scan_state) (void)hipMemsetAsync(ctx->scan_state, 0, ctx->scan_state_bytes, ctx->stream);   // whatever state the launch left: start over
        VD_FAIL(ctx, VD_ERR_HIP, "vd_cull_compact: the compaction scan gave up waiting for a workgroup");
    }
    const size_t n_copy = pad_tail ? n_inst : *out_count;
    if (n_copy) {
        VD_HIP_CHECK(ctx, hipMemcpyAsync(out, dout, n_copy * sizeof(VdDrawIndexedIndirect), hipMemcpyDeviceToHost, ctx->stream));
        VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return VD_OK;
}

#ifdef VD_TUNING
// Tuning / test hook (libvoidin_hip_tuning.so only): the workgroup that draws ticket `tile` of the next fused
// cull + compaction launches leaves before it publishes anything - what a workgroup lost to a fault looks like to the
// others.  tile < 0 clears it.  tests/test_gpu_scan_fault.py.
int vd_debug_scan_fault(VdCtx* ctx, int tile) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx) return VD_ERR_INVALID_ARG;
    vd_u64* ticket; vd_u64* states;
    int rc = vd_scan_scratch(ctx, 1u << 16, &ticket, &states, false);   // makes sure the arena exists (and is large enough for the test's launches)
    if (rc) return rc;
    const vd_u64 v = tile < 0 ? 0ull : (vd_u64)tile + 1ull;
    VD_HIP_CHECK(ctx, hipMemcpyAsync(ticket + 1, &v, 8, hipMemcpyHostToDevice, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VD_OK;
}
#endif

}  // extern "C"

// cull.hip — per-instance frustum cull, indirect-draw emission and ordered compaction for
// gfx950 (CDNA4, wave64).
//
// Replaces the `emit_draws` compute pass (reference: shaders/emit_draws.wgsl:13-64,
// shaders/utils/math.wgsl:67-73, dispatched by crates/app/src/pass/visibility.rs:233-254).
//
// Data movement (HBM-bound, no MFMA — fp32 compares and index work):
//   * instances are a 144-byte AoS (not a power of two).  A wave streams 64 consecutive
//     instances = 9216 contiguous bytes as 9 fully coalesced 16-B-per-lane loads, parks them in
//     a wave-private LDS slab, and every lane then reads back its own instance's transform
//     (4 x ds_read_b128 at a 144-B stride: 36-dword stride is conflict-free for b128) and mesh id;
//   * emit path: the 20-byte commands of a wave (1280 contiguous bytes) go through the same
//     slab and leave as 16-B-per-lane stores;
//   * compact path: wave ballot + mbcnt rank the survivors, a 4-wave LDS scan ranks the waves,
//     and a single-pass decoupled look-back over 8-byte {status,value} granules ranks the
//     tiles, so instances are read exactly once and only survivors are written.
#include "vd_common.hpp"

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kInstBytes = 144;
constexpr int kSlabBytes = kWave * kInstBytes;  // 9216
constexpr int kChunksPerLane = kSlabBytes / (kWave * 16);  // 9
constexpr int kRounds = 8;                       // rounds of 64 instances per wave per tile
constexpr int kTileInst = kBlock * kRounds;      // 2048 instances per tile
constexpr int kCompactLds = kWavesPerBlock * kSlabBytes + kRounds * kBlock * 4 + 32;

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

__device__ __forceinline__ void slab_fill(char* slab, const VdInstance* __restrict__ inst, size_t first,
                                          unsigned n_valid, unsigned lane, u32x4 (&regs)[kChunksPerLane]) {
    const u32x4* src = reinterpret_cast<const u32x4*>(inst + first);
    const unsigned n_chunks = n_valid * (kInstBytes / 16);
#pragma unroll
    for (int j = 0; j < kChunksPerLane; ++j) {
        const unsigned c = j * kWave + lane;
        regs[j] = c < n_chunks ? __builtin_nontemporal_load(src + c) : u32x4{0u, 0u, 0u, 0u};
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
        slab_fill(slab, inst, f0, min(64u, n_inst - (unsigned)f0), lane, regs);
    }
    for (; wt < n_wave_tiles; wt += waves_total) {
        const size_t first = (size_t)wt * kWave;
        const unsigned n_valid = min(64u, n_inst - (unsigned)first);
        slab_store(slab, lane, regs);
        // prefetch the next wave-tile while this one is processed
        const unsigned wn = wt + waves_total;
        if (wn < n_wave_tiles) {
            const size_t fn = (size_t)wn * kWave;
            slab_fill(slab, inst, fn, min(64u, n_inst - (unsigned)fn), lane, regs);
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
// C1 + C3 fused: cull and emit survivors only, ascending instance order, single pass.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock, 4) void cull_compact_kernel(CullCamera cam, const VdMeshInfo* __restrict__ meshes,
                                                                 unsigned n_mesh, const VdInstance* __restrict__ inst,
                                                                 unsigned n_inst, VdDrawIndexedIndirect* __restrict__ out,
                                                                 unsigned* __restrict__ out_count, vd_u64* tile_state,
                                                                 unsigned* ticket_counter, unsigned n_tiles,
                                                                 unsigned first_instance) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // dynamic LDS: 4 wave slabs, then per-round records (mesh id | visible << 31), then scalars
    unsigned* s_rec = reinterpret_cast<unsigned*>(smem + kWavesPerBlock * kSlabBytes);   // [kRounds][kBlock]
    unsigned* s_misc = s_rec + kRounds * kBlock;   // [0] ticket, [1] tile_excl, [2..5] wave totals
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    char* slab = smem + wave * kSlabBytes;

    if (threadIdx.x == 0) s_misc[0] = atomicAdd(ticket_counter, 1u);
    __syncthreads();
    const unsigned tile = s_misc[0];
    const size_t tile_first = (size_t)tile * kTileInst;
    // wave-contiguous ranges keep the output order (wave, round, lane) == instance order
    const size_t wave_first = tile_first + (size_t)wave * (kWave * kRounds);

    unsigned wave_total = 0;
    u32x4 regs[kChunksPerLane];
    {
        const unsigned nv = wave_first < n_inst ? (unsigned)min((size_t)64, (size_t)n_inst - wave_first) : 0u;
        slab_fill(slab, inst, wave_first, nv, lane, regs);
    }
#pragma unroll 1
    for (int r = 0; r < kRounds; ++r) {
        const size_t first = wave_first + (size_t)r * kWave;
        const unsigned n_valid = first < n_inst ? (unsigned)min((size_t)64, (size_t)n_inst - first) : 0u;
        slab_store(slab, lane, regs);
        if (r + 1 < kRounds) {
            const size_t fn = first + kWave;
            const unsigned nv = fn < n_inst ? (unsigned)min((size_t)64, (size_t)n_inst - fn) : 0u;
            slab_fill(slab, inst, fn, nv, lane, regs);
        }
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
        const unsigned excl = vd_lookback(tile_state, tile, tile_total);
        if (lane == 0) {
            s_misc[1] = excl;
            if (tile == n_tiles - 1u) *out_count = excl + tile_total;
        }
    }
    __syncthreads();
    unsigned base = s_misc[1];
    for (unsigned w = 0; w < wave; ++w) base += s_misc[2 + w];

#pragma unroll 2
    for (int r = 0; r < kRounds; ++r) {
        const unsigned rec = s_rec[r * kBlock + threadIdx.x];
        const bool vis = (rec >> 31) != 0u;
        const unsigned long long mask = __ballot(vis);
        if (vis) {
            const unsigned mid = rec & 0x7fffffffu;
            const unsigned dst = base + vd_mbcnt(mask);
            unsigned* o = reinterpret_cast<unsigned*>(out + dst);
            const uint4* mp = reinterpret_cast<const uint4*>(meshes + mid);
            o[0] = mp[0].w;                         // index_count
            o[1] = 1u;
            o[2] = mp[1].w;                         // base_index
            o[3] = (unsigned)meshes[mid].vertex_offset;
            o[4] = first_instance + (unsigned)(wave_first + (size_t)r * kWave) + lane;
        }
        base += (unsigned)__popcll(mask);
    }
}

// Zero-fill out[count..n) so the unchanged multi_draw_indexed_indirect(buf, 0, N) consumer
// (visibility.rs:188-192) sees instance_count = 0 in the tail.
__global__ __launch_bounds__(kBlock) void pad_tail_kernel(VdDrawIndexedIndirect* __restrict__ out,
                                                          const unsigned* __restrict__ count, unsigned n) {
    const size_t begin = (size_t)(*count) * 5u, end = (size_t)n * 5u;
    unsigned* o = reinterpret_cast<unsigned*>(out);
    for (size_t i = begin + (size_t)blockIdx.x * kBlock + threadIdx.x; i < end; i += (size_t)gridDim.x * kBlock) o[i] = 0u;
}

// ------------------------------------------------------------------------------------------
// C3 alone: ordered compaction of an existing command buffer (pure function of C1's output).
// ------------------------------------------------------------------------------------------
constexpr int kCompactPerThread = 8;
constexpr int kCompactTile = kBlock * kCompactPerThread;

__global__ __launch_bounds__(kBlock) void compact_draws_kernel(const VdDrawIndexedIndirect* __restrict__ in, unsigned n,
                                                               VdDrawIndexedIndirect* __restrict__ out,
                                                               unsigned* __restrict__ out_count, vd_u64* tile_state,
                                                               unsigned* ticket_counter, unsigned n_tiles) {
    __shared__ unsigned s_ticket, s_wave_total[kWavesPerBlock], s_tile_excl;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket_counter, 1u);
    __syncthreads();
    const unsigned tile = s_ticket;
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
        const unsigned excl = vd_lookback(tile_state, tile, tile_total);
        if (lane == 0) {
            s_tile_excl = excl;
            if (tile == n_tiles - 1u) *out_count = excl + tile_total;
        }
    }
    __syncthreads();
    unsigned base = s_tile_excl;
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

CullCamera make_cam(const VdCameraUniform* c) {
    CullCamera k;
    memcpy(k.view, c->view, sizeof(k.view));
    memcpy(k.frustum, c->frustum, sizeof(k.frustum));
    k.znear = c->znear;
    k.zfar = c->zfar;
    return k;
}

// scratch layout for the scans: [0,16) ticket counter (+pad), [16, 16+8*n_tiles) tile states
int scan_scratch(VdCtx* ctx, unsigned n_tiles, unsigned** ticket, vd_u64** states) {
    const size_t need = 16 + (size_t)n_tiles * 8;
    const size_t need16 = (need + 15) & ~(size_t)15;
    int rc = vd_ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, need16);
    if (rc) return rc;
    VD_HIP_CHECK(ctx, hipMemsetAsync(ctx->scratch, 0, need16, ctx->stream));
    *ticket = reinterpret_cast<unsigned*>(ctx->scratch);
    *states = reinterpret_cast<vd_u64*>(reinterpret_cast<char*>(ctx->scratch) + 16);
    return VD_OK;
}

}  // namespace

extern "C" {

int vd_cull_emit_dev(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                     const VdInstance* d_instances, uint32_t n_inst, VdDrawIndexedIndirect* d_out) {
    return vd_cull_emit_shard_dev(ctx, camera, d_meshes, n_mesh, d_instances, n_inst, 0u, d_out);
}

int vd_cull_emit_shard_dev(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                           const VdInstance* d_instances, uint32_t n_inst, uint32_t first_instance,
                           VdDrawIndexedIndirect* d_out) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!camera || !d_meshes || n_mesh == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_emit: null camera/meshes or n_mesh == 0");
    if (n_inst == 0) return VD_OK;
    if (!d_instances || !d_out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_emit: null instances/out");
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
    return vd_cull_compact_shard_dev(ctx, camera, d_meshes, n_mesh, d_instances, n_inst, 0u, d_out, d_out_count, pad_tail);
}

int vd_cull_compact_shard_dev(VdCtx* ctx, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                              const VdInstance* d_instances, uint32_t n_inst, uint32_t first_instance,
                              VdDrawIndexedIndirect* d_out, uint32_t* d_out_count, int pad_tail) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!camera || !d_meshes || n_mesh == 0 || !d_out_count)
        VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_compact: null camera/meshes/count or n_mesh == 0");
    if (n_inst == 0) {
        VD_HIP_CHECK(ctx, hipMemsetAsync(d_out_count, 0, 4, ctx->stream));
        return VD_OK;
    }
    if (!d_instances || !d_out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_cull_compact: null instances/out");
    const unsigned n_tiles = (n_inst + kTileInst - 1) / kTileInst;
    unsigned* ticket; vd_u64* states;
    vd_time_begin(ctx);
    int rc = scan_scratch(ctx, n_tiles, &ticket, &states);
    if (rc) return rc;
    hipLaunchKernelGGL(cull_compact_kernel, dim3(n_tiles), dim3(kBlock), kCompactLds, ctx->stream,
                       make_cam(camera), d_meshes, n_mesh, d_instances, n_inst, d_out, d_out_count, states, ticket, n_tiles, first_instance);
    if (pad_tail) {
        unsigned blocks = (unsigned)ctx->num_cus * 4u;
        hipLaunchKernelGGL(pad_tail_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, d_out, d_out_count, n_inst);
    }
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_compact_draws_dev(VdCtx* ctx, const VdDrawIndexedIndirect* d_in, uint32_t n, VdDrawIndexedIndirect* d_out,
                         uint32_t* d_out_count) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!d_out_count) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_compact_draws: null count");
    if (n == 0) {
        VD_HIP_CHECK(ctx, hipMemsetAsync(d_out_count, 0, 4, ctx->stream));
        return VD_OK;
    }
    if (!d_in || !d_out) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_compact_draws: null in/out");
    const unsigned n_tiles = (n + kCompactTile - 1) / kCompactTile;
    unsigned* ticket; vd_u64* states;
    vd_time_begin(ctx);
    int rc = scan_scratch(ctx, n_tiles, &ticket, &states);
    if (rc) return rc;
    hipLaunchKernelGGL(compact_draws_kernel, dim3(n_tiles), dim3(kBlock), 0, ctx->stream, d_in, n, d_out, d_out_count,
                       states, ticket, n_tiles);
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
    const size_t n_copy = pad_tail ? n_inst : *out_count;
    if (n_copy) {
        VD_HIP_CHECK(ctx, hipMemcpyAsync(out, dout, n_copy * sizeof(VdDrawIndexedIndirect), hipMemcpyDeviceToHost, ctx->stream));
        VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return VD_OK;
}

}  // extern "C"

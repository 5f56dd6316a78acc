// dist.hip — the sharded cull's exchange behind the C ABI: RCCL over xGMI, on the context's stream.
//
// NEW (BASELINE.json configs[3]): the reference is single-GPU (one wgpu::Device,
// crates/app/src/app.rs:108-118).  Instances are independent (shaders/emit_draws.wgsl:38-63 touches only slot i), so
// rank r owns the contiguous shard [r*S, min(N, (r+1)*S)), S = ceil(N / world), and every rank ends a step with the
// ordered compacted draw list of the WHOLE scene, bit-identical to vd_cull_compact on one GPU.
//
//   vd_dist_step_full_dev   cull_mask (own shard -> 1 bit per instance) -> ncclAllGather of the masks -> expand_mask
//                           (all shards -> the whole list).  Three enqueues on ONE stream, no host round trip: xGMI is
//                           ~40x slower than HBM, so the wire carries a bit per instance, not a 20-byte command.
//   vd_dist_step_draws_dev  the literal exchange: compact the own shard, all-gather the counts, then an exact-size direct
//                           exchange of the 20-byte commands (grouped ncclSend / ncclRecv, one xGMI link per peer).
//   vd_dist_step_indices_dev  the same exchange with 4-byte survivor indices on the wire (SURVEY.md 8e's option):
//                           cull_mask -> mask_to_indices -> counts -> exact-size exchange -> indices_to_draws.
//
// RCCL is bound at run time (dlopen), not at link time: a host process usually has a copy loaded already (rccl-sys in a
// Rust host, torch's bundled librccl in the Python tests) and a second copy under another soname would bring its own
// global state.  Order: $VD_RCCL_LIB when the host sets it (an explicit choice overrides everything - it is how the
// tests bind their test double, tests/cpp/fake_rccl.cpp), else the copy already in the process, else the ROCm install.
#include "vd_common.hpp"

#include <dlfcn.h>
#include <new>
#include <rccl/rccl.h>   // types and prototypes only; nothing links against librccl
#include <stdlib.h>

namespace {

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    char where[256] = {0};
};

RcclApi g_rccl;   // process-wide: one RCCL per process

const char* load_rccl() {   // nullptr = ok, else what went wrong
    if (g_rccl.lib) return nullptr;
    static char msg[512];
    const char* env = getenv("VD_RCCL_LIB");
    struct Cand { const char* name; int flags; };
    const Cand cands[] = {
        {env, RTLD_NOW | RTLD_LOCAL},                 // the host's explicit choice, if any
        {"librccl.so", RTLD_NOW | RTLD_NOLOAD},       // a copy the host process already holds (torch, rccl-sys)
        {"librccl.so.1", RTLD_NOW | RTLD_NOLOAD},
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL},
        {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL},
        {"librccl.so", RTLD_NOW | RTLD_LOCAL},
    };
    void* h = nullptr;
    for (const Cand& c : cands) {
        if (!c.name || !*c.name) continue;
        h = dlopen(c.name, c.flags);
        if (h) { snprintf(g_rccl.where, sizeof(g_rccl.where), "%s%s", c.name, (c.flags & RTLD_NOLOAD) ? " (already in the process)" : ""); break; }
    }
    if (!h) { snprintf(msg, sizeof(msg), "RCCL not found (tried VD_RCCL_LIB, librccl.so[.1], /opt/rocm/lib): %s", dlerror()); return msg; }
#define VD_SYM(field, sym)                                                                              \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym));                             \
    if (!g_rccl.field) { snprintf(msg, sizeof(msg), "RCCL at %s lacks %s", g_rccl.where, sym); dlclose(h); return msg; }
    VD_SYM(GetUniqueId, "ncclGetUniqueId") VD_SYM(CommInitRank, "ncclCommInitRank") VD_SYM(CommDestroy, "ncclCommDestroy")
    VD_SYM(AllGather, "ncclAllGather") VD_SYM(Send, "ncclSend") VD_SYM(Recv, "ncclRecv") VD_SYM(GroupStart, "ncclGroupStart")
    VD_SYM(GroupEnd, "ncclGroupEnd") VD_SYM(GetErrorString, "ncclGetErrorString") VD_SYM(GetVersion, "ncclGetVersion")
#undef VD_SYM
    g_rccl.lib = h;
    return nullptr;
}

// instance -> mesh table rows of one shard: min(u32 instance.mesh, n_mesh - 1) (the clamp of the emit path,
// cull.hip), `W` bytes per row; rows beyond n_local are 0
template <typename IdT>
__global__ void mesh_ids_kernel(const VdInstance* __restrict__ inst, unsigned n_local, unsigned rows, unsigned n_mesh, IdT* __restrict__ out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    out[i] = i < n_local ? (IdT)min(inst[i].mesh, n_mesh - 1u) : (IdT)0;
}

__global__ void set_u32_kernel(unsigned* p, unsigned v) { *p = v; }

}  // namespace

struct VdDist {
    VdCtx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    // scene (vd_dist_set_scene_dev)
    uint32_t n_total = 0, shard = 0, first = 0, n_local = 0, wps = 0, id_bytes = 0, n_mesh = 0;
    uint64_t* d_mask = nullptr;       // wps words: own shard, zero padded
    uint64_t* d_mask_all = nullptr;   // wps * world words
    void* d_mesh_ids = nullptr;       // shard * world rows of id_bytes
    VdDrawIndexedIndirect* d_local = nullptr;   // draws mode: own shard's compacted commands (shard slots)
    uint32_t* d_idx = nullptr;        // indices mode: own shard's survivor indices (shard slots), made on first use
    uint32_t* d_idx_all = nullptr;    // indices mode: all survivors' indices (shard * world slots)
    uint32_t* d_counts = nullptr;     // draws / indices mode: [world] survivor counts + [1] own count
    uint32_t* h_counts = nullptr;     // pinned mirror
};

#define VD_RCCL_CHECK(d, call)                                                                          \
    do {                                                                                                \
        ncclResult_t r_ = (call);                                                                       \
        if (r_ != ncclSuccess) {                                                                        \
            snprintf((d)->ctx->err, sizeof((d)->ctx->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call, \
                     g_rccl.GetErrorString(r_));                                                        \
            return VD_ERR_COMM;                                                                         \
        }                                                                                               \
    } while (0)

static void dist_free_scene(VdDist* d) {
    if (d->d_mask) (void)hipFree(d->d_mask);
    if (d->d_mask_all) (void)hipFree(d->d_mask_all);
    if (d->d_mesh_ids) (void)hipFree(d->d_mesh_ids);
    if (d->d_local) (void)hipFree(d->d_local);
    if (d->d_idx) (void)hipFree(d->d_idx);
    if (d->d_idx_all) (void)hipFree(d->d_idx_all);
    d->d_mask = d->d_mask_all = nullptr; d->d_mesh_ids = nullptr; d->d_local = nullptr; d->d_idx = d->d_idx_all = nullptr;
    d->n_total = 0;
}

extern "C" {

int vd_dist_unique_id(void* out_id) {
    if (!out_id) return VD_ERR_INVALID_ARG;
    if (load_rccl()) return VD_ERR_COMM;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return VD_ERR_COMM;
    static_assert(sizeof(ncclUniqueId) == VD_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(out_id, &id, sizeof(id));
    return VD_OK;
}

int vd_dist_create(VdCtx* ctx, const void* unique_id, int rank, int world, VdDist** out) {
    VdDeviceGuard vd_guard_(ctx);   // the communicator binds to the CURRENT device: make it the context's
    if (!ctx || !out) return VD_ERR_INVALID_ARG;
    *out = nullptr;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_create: bad rank / world / id");
    if (const char* why = load_rccl()) VD_FAIL(ctx, VD_ERR_COMM, why);
    VdDist* d = new (std::nothrow) VdDist();
    if (!d) return VD_ERR_OOM;
    d->ctx = ctx; d->rank = rank; d->world = world;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&d->comm, world, id, rank);
    if (r != ncclSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "ncclCommInitRank(rank %d of %d, %s) -> %s", rank, world, g_rccl.where, g_rccl.GetErrorString(r));
        delete d;
        return VD_ERR_COMM;
    }
    if (hipMalloc(reinterpret_cast<void**>(&d->d_counts), 4 * (size_t)(world + 4)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&d->h_counts), 4 * (size_t)(world + 4)) != hipSuccess) {
        vd_dist_destroy(d);
        VD_FAIL(ctx, VD_ERR_OOM, "vd_dist_create: count buffers");
    }
    *out = d;
    return VD_OK;
}

int vd_dist_destroy(VdDist* d) {
    if (!d) return VD_ERR_INVALID_ARG;
    VdDeviceGuard vd_guard_(d->ctx);
    (void)hipStreamSynchronize(d->ctx->stream);
    dist_free_scene(d);
    if (d->d_counts) (void)hipFree(d->d_counts);
    if (d->h_counts) (void)hipHostFree(d->h_counts);
    if (d->comm) (void)g_rccl.CommDestroy(d->comm);
    delete d;
    return VD_OK;
}

int vd_dist_info(const VdDist* d, VdDistInfo* out) {
    if (!d || !out) return VD_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    out->rank = d->rank; out->world = d->world;
    out->n_total = d->n_total; out->shard_size = d->shard; out->first_instance = d->first; out->n_local = d->n_local;
    out->mask_words_per_shard = d->wps; out->id_bytes = d->id_bytes;
    out->d_mask = d->d_mask; out->d_mask_all = d->d_mask_all; out->d_mesh_ids = d->d_mesh_ids;
    int v = 0;
    if (g_rccl.GetVersion && g_rccl.GetVersion(&v) == ncclSuccess) out->rccl_version = v;
    snprintf(out->rccl_library, sizeof(out->rccl_library), "%s", g_rccl.where);
    return VD_OK;
}

int vd_dist_allgather_dev(VdDist* d, const void* d_send, void* d_recv, uint64_t bytes_per_rank) {
    if (!d) return VD_ERR_INVALID_ARG;
    VdDeviceGuard vd_guard_(d->ctx);
    if (!d_send || !d_recv) VD_FAIL(d->ctx, VD_ERR_INVALID_ARG, "vd_dist_allgather: null buffer");
    if (bytes_per_rank == 0) return VD_OK;
    VD_RCCL_CHECK(d, g_rccl.AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclUint8, d->comm, d->ctx->stream));
    return VD_OK;
}

int vd_dist_set_scene_dev(VdDist* d, const VdInstance* d_shard_instances, uint32_t n_local, uint32_t n_total, uint32_t n_mesh) {
    if (!d) return VD_ERR_INVALID_ARG;
    VdCtx* ctx = d->ctx;
    VdDeviceGuard vd_guard_(ctx);
    if (n_total == 0 || n_mesh == 0) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_set_scene: empty scene or mesh table");
    const uint32_t S = (uint32_t)(((uint64_t)n_total + (uint64_t)d->world - 1) / (uint64_t)d->world);
    const uint64_t lo64 = (uint64_t)d->rank * S, hi64 = lo64 + S;
    const uint32_t lo = (uint32_t)(lo64 < n_total ? lo64 : n_total), hi = (uint32_t)(hi64 < n_total ? hi64 : n_total);
    if (n_local != hi - lo) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_set_scene: n_local is not this rank's shard [r*S, min(N, (r+1)*S)), S = ceil(N / world)");
    if (n_local && !d_shard_instances) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_set_scene: null instances");
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    dist_free_scene(d);
    d->n_total = n_total; d->shard = S; d->first = lo; d->n_local = n_local; d->n_mesh = n_mesh;
    d->wps = (S + 63u) / 64u;
    d->id_bytes = n_mesh <= 256u ? 1u : (n_mesh <= 65536u ? 2u : 4u);     // the width rule of launch_mask_pass (cull.hip)
    const size_t rows = (size_t)S * (size_t)d->world;
    if (hipMalloc(reinterpret_cast<void**>(&d->d_mask), 8 * (size_t)d->wps) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d->d_mask_all), 8 * (size_t)d->wps * (size_t)d->world) != hipSuccess ||
        hipMalloc(&d->d_mesh_ids, rows * d->id_bytes + 16) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d->d_local), sizeof(VdDrawIndexedIndirect) * (size_t)(S ? S : 1)) != hipSuccess) {
        dist_free_scene(d);
        VD_FAIL(ctx, VD_ERR_OOM, "vd_dist_set_scene: scene buffers");
    }
    VD_HIP_CHECK(ctx, hipMemsetAsync(d->d_mask, 0, 8 * (size_t)d->wps, ctx->stream));   // words past ceil(n_local / 64) stay 0
    VD_HIP_CHECK(ctx, hipMemsetAsync(d->d_mask_all, 0, 8 * (size_t)d->wps * (size_t)d->world, ctx->stream));
    // own rows of the replicated instance -> mesh table, in place at row rank * S; then the all-gather (once per scene:
    // mesh assignment is static, only transforms animate - shaders/compute_update.wgsl:10-28)
    char* own = reinterpret_cast<char*>(d->d_mesh_ids) + (size_t)d->rank * S * d->id_bytes;
    const dim3 grid((S + 255u) / 256u), block(256);
    if (d->id_bytes == 1u) hipLaunchKernelGGL(mesh_ids_kernel<unsigned char>, grid, block, 0, ctx->stream, d_shard_instances, n_local, S, n_mesh, reinterpret_cast<unsigned char*>(own));
    else if (d->id_bytes == 2u) hipLaunchKernelGGL(mesh_ids_kernel<unsigned short>, grid, block, 0, ctx->stream, d_shard_instances, n_local, S, n_mesh, reinterpret_cast<unsigned short*>(own));
    else hipLaunchKernelGGL(mesh_ids_kernel<unsigned>, grid, block, 0, ctx->stream, d_shard_instances, n_local, S, n_mesh, reinterpret_cast<unsigned*>(own));
    VD_HIP_CHECK(ctx, hipGetLastError());
    VD_RCCL_CHECK(d, g_rccl.AllGather(own, d->d_mesh_ids, (size_t)S * d->id_bytes, ncclUint8, d->comm, ctx->stream));
    return VD_OK;
}

int vd_dist_step_full_dev(VdDist* d, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                          const VdInstance* d_shard_instances, VdDrawIndexedIndirect* d_out, uint32_t* d_out_count) {
    if (!d) return VD_ERR_INVALID_ARG;
    VdCtx* ctx = d->ctx;
    VdDeviceGuard vd_guard_(ctx);
    if (!d->n_total) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_step: vd_dist_set_scene_dev first");
    if (n_mesh != d->n_mesh) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_step: n_mesh differs from the scene's");
    int rc = VD_OK;
    if (d->n_local) rc = vd_cull_mask_dev(ctx, camera, d_meshes, n_mesh, d_shard_instances, d->n_local, d->d_mask);
    if (rc) return rc;
    VD_RCCL_CHECK(d, g_rccl.AllGather(d->d_mask, d->d_mask_all, (size_t)d->wps, ncclUint64, d->comm, ctx->stream));
    return vd_expand_mask_dev(ctx, d->d_mask_all, d->n_total, d->shard, d->d_mesh_ids, d->id_bytes, d_meshes, n_mesh, d_out, d_out_count);
}

// The variable-size leg shared by the draws and the indices steps: all-gather the survivor counts (d_own -> d_counts,
// read back: the sizes are data dependent, one host round trip), then the exact-size direct exchange - every rank's
// `rec`-byte records go straight to their final offset in every peer's `d_dst` (one xGMI link per peer), the own
// records by a device copy.  *out_total = survivors of the whole scene.
static int exchange_records(VdDist* d, const void* d_own_records, size_t rec, void* d_dst, uint64_t* out_total) {
    VdCtx* ctx = d->ctx;
    uint32_t* d_own = d->d_counts + d->world;
    VD_RCCL_CHECK(d, g_rccl.AllGather(d_own, d->d_counts, 1, ncclUint32, d->comm, ctx->stream));
    VD_HIP_CHECK(ctx, hipMemcpyAsync(d->h_counts, d->d_counts, 4 * (size_t)d->world, hipMemcpyDeviceToHost, ctx->stream));
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t off = 0, my_off = 0;
    for (int q = 0; q < d->world; ++q) {
        if (q == d->rank) my_off = off;
        if (d->h_counts[q] > d->shard) VD_FAIL(ctx, VD_ERR_COMM, "vd_dist_step: a gathered count exceeds the shard size");
        off += d->h_counts[q];
    }
    const uint64_t total = off;
    if (total > d->n_total) VD_FAIL(ctx, VD_ERR_COMM, "vd_dist_step: gathered counts exceed the scene");
    const uint32_t mine = d->h_counts[d->rank];
    char* dst = static_cast<char*>(d_dst);
    if (mine) VD_HIP_CHECK(ctx, hipMemcpyAsync(dst + rec * my_off, d_own_records, rec * mine, hipMemcpyDeviceToDevice, ctx->stream));
    if (d->world > 1) {
        // a failed ncclSend / ncclRecv must not leave the group open on this thread (every later collective would be
        // queued into it and never issued): remember the first error, still issue the operations of the other peers
        // (they are served instead of left waiting), ALWAYS close the group, then report
        VD_RCCL_CHECK(d, g_rccl.GroupStart());
        ncclResult_t first = ncclSuccess;
        const char* what = "";
        uint64_t o = 0;
        for (int q = 0; q < d->world; ++q) {
            const uint32_t cq = d->h_counts[q];
            if (q != d->rank) {
                if (mine) {
                    const ncclResult_t r = g_rccl.Send(d_own_records, rec * mine, ncclUint8, q, d->comm, ctx->stream);
                    if (r != ncclSuccess && first == ncclSuccess) { first = r; what = "ncclSend"; }
                }
                if (cq) {
                    const ncclResult_t r = g_rccl.Recv(dst + rec * o, rec * cq, ncclUint8, q, d->comm, ctx->stream);
                    if (r != ncclSuccess && first == ncclSuccess) { first = r; what = "ncclRecv"; }
                }
            }
            o += cq;
        }
        const ncclResult_t end = g_rccl.GroupEnd();
        if (first != ncclSuccess || end != ncclSuccess) {
            snprintf(ctx->err, sizeof(ctx->err), "vd_dist_step: %s -> %s (group closed; peers of a failed exchange time out in RCCL)",
                     first != ncclSuccess ? what : "ncclGroupEnd", g_rccl.GetErrorString(first != ncclSuccess ? first : end));
            return VD_ERR_COMM;
        }
    }
    *out_total = total;
    return VD_OK;
}

int vd_dist_step_draws_dev(VdDist* d, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                           const VdInstance* d_shard_instances, VdDrawIndexedIndirect* d_out, uint32_t* d_out_count) {
    if (!d) return VD_ERR_INVALID_ARG;
    VdCtx* ctx = d->ctx;
    VdDeviceGuard vd_guard_(ctx);
    if (!d->n_total) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_step: vd_dist_set_scene_dev first");
    if (n_mesh != d->n_mesh) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_step: n_mesh differs from the scene's");
    if (!d_out || !d_out_count) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_step_draws: null output");
    uint32_t* d_own = d->d_counts + d->world;
    int rc = VD_OK;
    if (d->n_local) rc = vd_cull_compact_shard_dev(ctx, camera, d_meshes, n_mesh, d_shard_instances, d->n_local, d->first, d->d_local, d_own, 0);
    else hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, ctx->stream, d_own, 0u);
    if (rc) return rc;
    uint64_t total = 0;
    rc = exchange_records(d, d->d_local, sizeof(VdDrawIndexedIndirect), d_out, &total);
    if (rc) return rc;
    hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, ctx->stream, d_out_count, (unsigned)total);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

int vd_dist_step_indices_dev(VdDist* d, const VdCameraUniform* camera, const VdMeshInfo* d_meshes, uint32_t n_mesh,
                             const VdInstance* d_shard_instances, VdDrawIndexedIndirect* d_out, uint32_t* d_out_count) {
    if (!d) return VD_ERR_INVALID_ARG;
    VdCtx* ctx = d->ctx;
    VdDeviceGuard vd_guard_(ctx);
    if (!d->n_total) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_step: vd_dist_set_scene_dev first");
    if (n_mesh != d->n_mesh) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_step: n_mesh differs from the scene's");
    if (!d_out || !d_out_count) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_dist_step_indices: null output");
    if (!d->d_idx) {   // first use of this mode on the scene: 4 B per shard slot + 4 B per scene slot
        const size_t rows = (size_t)d->shard * (size_t)d->world;
        if (hipMalloc(reinterpret_cast<void**>(&d->d_idx), 4 * (size_t)(d->shard ? d->shard : 1) + 16) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&d->d_idx_all), 4 * (rows ? rows : 1) + 16) != hipSuccess) {
            if (d->d_idx) (void)hipFree(d->d_idx);
            d->d_idx = d->d_idx_all = nullptr;
            VD_FAIL(ctx, VD_ERR_OOM, "vd_dist_step_indices: index buffers");
        }
    }
    uint32_t* d_own = d->d_counts + d->world;
    int rc = VD_OK;
    if (d->n_local) {
        rc = vd_cull_mask_dev(ctx, camera, d_meshes, n_mesh, d_shard_instances, d->n_local, d->d_mask);
        if (!rc) rc = vd_mask_to_indices_dev(ctx, d->d_mask, d->n_local, d->first, d->d_idx, d_own);
    } else hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, ctx->stream, d_own, 0u);
    if (rc) return rc;
    uint64_t total = 0;
    rc = exchange_records(d, d->d_idx, sizeof(uint32_t), d->d_idx_all, &total);
    if (rc) return rc;
    if (total) rc = vd_indices_to_draws_dev(ctx, d->d_idx_all, (uint32_t)total, d->d_mesh_ids, d->id_bytes, d->shard * (uint32_t)d->world, d_meshes, n_mesh, d_out);
    if (rc) return rc;
    hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, ctx->stream, d_out_count, (unsigned)total);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

}  // extern "C"

// ctx.hip — context, scratch arenas and instrumentation of libvoidin_hip.so.
// Replaces the wgpu device/queue pair the reference passes pull from `World`
// (crates/app/src/app.rs:108-118) and the wgpu_profiler scopes (visibility.rs:50,243-245).
#include "vd_common.hpp"

#include <new>
#include <stdlib.h>

int vd_ensure(VdCtx* ctx, void** buf, size_t* cur, size_t need) {
    if (need <= *cur && *buf) return VD_OK;
    size_t cap = *cur ? *cur : (size_t)1 << 16;
    while (cap < need) cap *= 2;
    if (*buf) {
        VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        VD_HIP_CHECK(ctx, hipFree(*buf));
        *buf = nullptr;
        *cur = 0;
    }
    hipError_t e = hipMalloc(buf, cap);
    if (e != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "hipMalloc(%zu) -> %s", cap, hipGetErrorString(e));
        *buf = nullptr;
        return e == hipErrorOutOfMemory ? VD_ERR_OOM : VD_ERR_HIP;
    }
    *cur = cap;
    return VD_OK;
}

// grow-only pinned host staging
int vd_ensure_host(VdCtx* ctx, size_t need) {
    if (need <= ctx->host_stage_bytes && ctx->host_stage) return VD_OK;
    size_t cap = ctx->host_stage_bytes ? ctx->host_stage_bytes : (size_t)1 << 16;
    while (cap < need) cap *= 2;
    if (ctx->host_stage) {
        VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        VD_HIP_CHECK(ctx, hipHostFree(ctx->host_stage));
        ctx->host_stage = nullptr; ctx->host_stage_bytes = 0;
    }
    hipError_t e = hipHostMalloc(&ctx->host_stage, cap);
    if (e != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "hipHostMalloc(%zu) -> %s", cap, hipGetErrorString(e));
        ctx->host_stage = nullptr;
        return VD_ERR_OOM;
    }
    ctx->host_stage_bytes = cap;
    return VD_OK;
}

int vd_scan_check_fault(VdCtx* ctx) {
    if (!ctx->host_pinned || ctx->host_pinned[kScanFaultWord] == 0u) return VD_OK;
    (void)hipStreamSynchronize(ctx->stream);
    ctx->host_pinned[kScanFaultWord] = 0u;
    if (ctx->scan_state) (void)hipMemsetAsync(ctx->scan_state, 0, ctx->scan_state_bytes, ctx->stream);      // whatever state the launch left: start over
    VD_FAIL(ctx, VD_ERR_HIP, "an earlier compaction launch on this context gave up waiting for a workgroup (bounded look-back scan): its "
                             "count was 0 and no list was written; the scan state has been reset");
}

int vd_scan_scratch(VdCtx* ctx, unsigned n_tiles, unsigned long long** ticket, unsigned long long** states, bool start_timer) {
    const size_t need = (32 + (size_t)n_tiles * 8 + 15) & ~(size_t)15;
    const bool periodic = (++ctx->scan_launches & ((1ull << 28) - 1)) == 0;   // epoch field is 30 bits: never let it lap
    if (need > ctx->scan_state_bytes || !ctx->scan_state || periodic) {
        int rc = vd_ensure(ctx, &ctx->scan_state, &ctx->scan_state_bytes, need);
        if (rc) return rc;
        VD_HIP_CHECK(ctx, hipMemsetAsync(ctx->scan_state, 0, ctx->scan_state_bytes, ctx->stream));
    }
    *ticket = reinterpret_cast<unsigned long long*>(ctx->scan_state);
    *states = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ctx->scan_state) + 32);      // [2] = the stuck word (vd_common.hpp)
    if (start_timer) vd_time_begin(ctx);   // vd_last_gpu_ms brackets the scan kernel itself
    return VD_OK;
}

extern "C" {

const char* vd_version(void) { return "voidin_hip 0.1.0 (gfx950)"; }

int vd_ctx_create(int device, VdCtx** out_ctx) {
    if (!out_ctx) return VD_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return VD_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return VD_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VD_ERR_NO_DEVICE;  // code objects are gfx950-only
    VdCtx* ctx = new (std::nothrow) VdCtx();
    if (!ctx) return VD_ERR_OOM;
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess ||
        hipEventCreate(&ctx->ev_mid) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&ctx->host_pinned), 64 * sizeof(uint32_t)) != hipSuccess) {
        vd_ctx_destroy(ctx);
        return VD_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    memset(ctx->host_pinned, 0, 64 * sizeof(uint32_t));
    {   // the scan kernels raise host_pinned[kScanFaultWord] themselves (system-scope store): they need its device address
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, ctx->host_pinned, 0) != hipSuccess || !dp) { vd_ctx_destroy(ctx); return VD_ERR_HIP; }
        ctx->fault_dev = reinterpret_cast<unsigned*>(dp) + kScanFaultWord;
    }
    for (int o = 0; o < VD_OPT_COUNT_; ++o) ctx->opt[o] = -1;
    *out_ctx = ctx;
    return VD_OK;
}

int vd_ctx_destroy(VdCtx* ctx) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->scan_state) (void)hipFree(ctx->scan_state);
    if (ctx->expand_state) (void)hipFree(ctx->expand_state);
    if (ctx->trace_ovf) (void)hipFree(ctx->trace_ovf);
    if (ctx->trace_deep) (void)hipFree(ctx->trace_deep);
    if (ctx->refit_state) (void)hipFree(ctx->refit_state);
    if (ctx->stage_in) (void)hipFree(ctx->stage_in);
    if (ctx->stage_out) (void)hipFree(ctx->stage_out);
    if (ctx->stage_aux) (void)hipFree(ctx->stage_aux);
    if (ctx->host_pinned) (void)hipHostFree(ctx->host_pinned);
    if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
    if (ctx->lvl_pinned) (void)hipHostFree(ctx->lvl_pinned);
    for (int e = 0; e < 2; ++e) if (ctx->ev_lvl[e]) (void)hipEventDestroy(ctx->ev_lvl[e]);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->ev_mid) (void)hipEventDestroy(ctx->ev_mid);
    if (ctx->ev_aux) (void)hipEventDestroy(ctx->ev_aux);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return VD_OK;
}

int vd_ctx_set_stream(VdCtx* ctx, void* hip_stream) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);   // NULL = the HIP default stream
    return VD_OK;
}

int vd_ctx_reset_stream(VdCtx* ctx) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return VD_OK;
}

int vd_ctx_synchronize(VdCtx* ctx) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return VD_OK;
}

const char* vd_last_error(const VdCtx* ctx) { return ctx ? ctx->err : "null ctx"; }

// Tuning hook (not part of the reference boundary): pick the cull_compact kernel variant.
struct VdExternalBuffer { hipExternalMemory_t mem; void* ptr; };

int vd_import_external_buffer(VdCtx* ctx, int opaque_fd, uint64_t size_bytes, VdExternalBuffer** out_handle, void** out_device_ptr) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (opaque_fd < 0 || size_bytes == 0 || !out_handle || !out_device_ptr) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_import_external_buffer: bad fd/size/out pointer");
    *out_handle = nullptr; *out_device_ptr = nullptr;
    hipExternalMemoryHandleDesc hd;
    memset(&hd, 0, sizeof(hd));
    hd.type = hipExternalMemoryHandleTypeOpaqueFd;
    hd.handle.fd = opaque_fd;
    hd.size = size_bytes;
    hipExternalMemory_t mem;
    VD_HIP_CHECK(ctx, hipImportExternalMemory(&mem, &hd));
    hipExternalMemoryBufferDesc bd;
    memset(&bd, 0, sizeof(bd));
    bd.offset = 0; bd.size = size_bytes;
    void* ptr = nullptr;
    hipError_t e = hipExternalMemoryGetMappedBuffer(&ptr, mem, &bd);
    if (e != hipSuccess) {
        (void)hipDestroyExternalMemory(mem);
        snprintf(ctx->err, sizeof(ctx->err), "hipExternalMemoryGetMappedBuffer -> %s", hipGetErrorString(e));
        return VD_ERR_HIP;
    }
    VdExternalBuffer* h = new (std::nothrow) VdExternalBuffer{mem, ptr};
    if (!h) { (void)hipDestroyExternalMemory(mem); return VD_ERR_OOM; }
    *out_handle = h; *out_device_ptr = ptr;
    return VD_OK;
}

int vd_release_external_buffer(VdCtx* ctx, VdExternalBuffer* handle) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx || !handle) return VD_ERR_INVALID_ARG;
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    hipError_t e = hipDestroyExternalMemory(handle->mem);
    delete handle;
    if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "hipDestroyExternalMemory -> %s", hipGetErrorString(e)); return VD_ERR_HIP; }
    return VD_OK;
}

// The other half of the hand-off: the renderer's frame is one queue.submit (crates/app/src/app.rs:334-348); a Vulkan semaphore
// exported as an opaque fd (VK_KHR_external_semaphore_fd; binary or timeline) orders that submit against the HIP stream
// WITHOUT a CPU wait on either side: wait -> vd_cull_*_dev -> signal are all enqueued on the context's stream.
struct VdExternalSemaphore { hipExternalSemaphore_t sem; int timeline; };

int vd_import_external_semaphore(VdCtx* ctx, int opaque_fd, int is_timeline, VdExternalSemaphore** out_handle) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (opaque_fd < 0 || !out_handle) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_import_external_semaphore: bad fd / out pointer");
    *out_handle = nullptr;
    hipExternalSemaphoreHandleDesc hd;
    memset(&hd, 0, sizeof(hd));
    hd.type = is_timeline ? hipExternalSemaphoreHandleTypeTimelineSemaphoreFd : hipExternalSemaphoreHandleTypeOpaqueFd;
    hd.handle.fd = opaque_fd;
    hipExternalSemaphore_t sem = nullptr;
    VD_HIP_CHECK(ctx, hipImportExternalSemaphore(&sem, &hd));
    VdExternalSemaphore* h = new (std::nothrow) VdExternalSemaphore{sem, is_timeline ? 1 : 0};
    if (!h) { (void)hipDestroyExternalSemaphore(sem); return VD_ERR_OOM; }
    *out_handle = h;
    return VD_OK;
}

int vd_wait_external_semaphore_async(VdCtx* ctx, VdExternalSemaphore* handle, uint64_t value) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!handle) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_wait_external_semaphore_async: null handle");
    hipExternalSemaphoreWaitParams wp;
    memset(&wp, 0, sizeof(wp));
    wp.params.fence.value = handle->timeline ? value : 0ull;       // a binary semaphore carries no value
    VD_HIP_CHECK(ctx, hipWaitExternalSemaphoresAsync(&handle->sem, &wp, 1, ctx->stream));
    return VD_OK;
}

int vd_signal_external_semaphore_async(VdCtx* ctx, VdExternalSemaphore* handle, uint64_t value) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!handle) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_signal_external_semaphore_async: null handle");
    hipExternalSemaphoreSignalParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.params.fence.value = handle->timeline ? value : 0ull;
    VD_HIP_CHECK(ctx, hipSignalExternalSemaphoresAsync(&handle->sem, &sp, 1, ctx->stream));
    return VD_OK;
}

int vd_release_external_semaphore(VdCtx* ctx, VdExternalSemaphore* handle) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx || !handle) return VD_ERR_INVALID_ARG;
    VD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));          // waits / signals still queued on the stream refer to it
    hipError_t e = hipDestroyExternalSemaphore(handle->sem);
    delete handle;
    if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "hipDestroyExternalSemaphore -> %s", hipGetErrorString(e)); return VD_ERR_HIP; }
    return VD_OK;
}

// Ordering through a shared word and through the host (include/voidin_abi.h, "Ordering that WORKS on this platform").
int vd_wait_value32_async(VdCtx* ctx, const uint32_t* d_word, uint32_t value) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!d_word || (reinterpret_cast<uintptr_t>(d_word) & 3u)) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_wait_value32_async: null or misaligned word");
    VD_HIP_CHECK(ctx, hipStreamWaitValue32(ctx->stream, const_cast<uint32_t*>(d_word), value, hipStreamWaitValueGte, 0xffffffffu));
    return VD_OK;
}

int vd_write_value32_async(VdCtx* ctx, uint32_t* d_word, uint32_t value) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!d_word || (reinterpret_cast<uintptr_t>(d_word) & 3u)) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_write_value32_async: null or misaligned word");
    VD_HIP_CHECK(ctx, hipStreamWriteValue32(ctx->stream, d_word, value, 0));
    return VD_OK;
}

int vd_host_callback_async(VdCtx* ctx, VdHostFn fn, void* user) {
    VdDeviceGuard vd_guard_(ctx);
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (!fn) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_host_callback_async: null function");
    VD_HIP_CHECK(ctx, hipLaunchHostFunc(ctx->stream, fn, user));
    return VD_OK;
}

int vd_ctx_set_option(VdCtx* ctx, int option, int64_t value) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    if (option <= 0 || option >= VD_OPT_COUNT_) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_ctx_set_option: unknown option");
    ctx->opt[option] = value < 0 ? -1 : (long long)value;
    if (option == VD_OPT_CULL_SPLIT_MIN) ctx->split_min = value < 0 ? (2u << 20) : (unsigned)value;
    if (option == VD_OPT_CULL_VARIANT) ctx->cull_variant = (int)value;     // variants are small signed ids, 0 = default
    return VD_OK;
}

int vd_ctx_set_timing(VdCtx* ctx, int enabled) {
    if (!ctx) return VD_ERR_INVALID_ARG;
    ctx->timing_enabled = enabled != 0;
    ctx->timed = false;
    return VD_OK;
}

float vd_last_gpu_ms_stage(VdCtx* ctx, int stage) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx || !ctx->timed || !ctx->timed_mid || stage < 0 || stage > 1) return -1.0f;
    if (hipEventSynchronize(ctx->ev_stop) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    const hipError_t e = stage == 0 ? hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_mid) : hipEventElapsedTime(&ms, ctx->ev_mid, ctx->ev_stop);
    return e == hipSuccess ? ms : -1.0f;
}

float vd_last_gpu_ms(VdCtx* ctx) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx || !ctx->timed) return -1.0f;
    if (hipEventSynchronize(ctx->ev_stop) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop) != hipSuccess) return -1.0f;
    return ms;
}

}  // extern "C"

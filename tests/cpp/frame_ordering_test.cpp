// GPU check of the ordering that works on this platform (include/voidin_abi.h "Ordering that WORKS on this
// platform"): vd_wait_value32_async / vd_write_value32_async / vd_host_callback_async, through the same import path a Vulkan
// allocation takes.  An fd-exported allocation stands for the renderer's draw buffer (as in external_buffer_test.cpp); its
// last 64 bytes hold the frame word.  Per frame f:
//   "renderer" (a second context, standing in for the Vulkan queue: its stream writes through the EXPORTER's own mapping):
//        ... uploads ..., then write_value32(word, f)
//   HIP side (the C ABI's view of the buffer through vd_import_external_buffer):
//        wait_value32(word, f) -> vd_compact_draws_dev into the imported buffer -> host_callback(frame done)
// Checked for three frames: the HIP side's stream does NOT run before the renderer's write of that frame (the callback has
// not fired 150 ms after everything was queued), does run after it, and the list in the shared buffer is the right one.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "voidin_abi.h"

#define SKIP(msg) do { std::printf("SKIP: %s\n", msg); return 0; } while (0)
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, ctx ? vd_last_error(ctx) : ""); return 1; } } while (0)

static std::atomic<unsigned> g_done{0};
static void frame_done(void* user) { g_done.store((unsigned)(uintptr_t)user, std::memory_order_release); }

int main() {
    VdCtx* ctx = nullptr;
    CHECK(vd_ctx_create(0, &ctx) == VD_OK);
    VdCtx* renderer = nullptr;
    CHECK(vd_ctx_create(0, &renderer) == VD_OK);
    // argument validation first
    CHECK(vd_wait_value32_async(ctx, nullptr, 1) == VD_ERR_INVALID_ARG);
    CHECK(vd_write_value32_async(ctx, reinterpret_cast<uint32_t*>(uintptr_t(0x1002)), 1) == VD_ERR_INVALID_ARG);      // misaligned
    CHECK(vd_host_callback_async(ctx, nullptr, nullptr) == VD_ERR_INVALID_ARG);
    CHECK(vd_wait_value32_async(nullptr, nullptr, 1) == VD_ERR_INVALID_ARG);

    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) SKIP("no VMM granularity");
    const uint32_t n = 100000;
    const size_t size = ((20u * (size_t)n + 64 + gran - 1) / gran) * gran;
    hipMemGenericAllocationHandle_t alloc;
    if (hipMemCreate(&alloc, size, &prop, 0) != hipSuccess) SKIP("hipMemCreate with an exportable handle is not supported here");
    int fd = -1;
    if (hipMemExportToShareableHandle(&fd, alloc, hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess || fd < 0) SKIP("export to fd not supported");
    void* own = nullptr;                                   // the exporter's ("renderer's") own view
    CHECK(hipMemAddressReserve(&own, size, 0, nullptr, 0) == hipSuccess);
    CHECK(hipMemMap(own, size, 0, alloc, 0) == hipSuccess);
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof(acc));
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK(hipMemSetAccess(own, size, &acc, 1) == hipSuccess);
    CHECK(hipMemset(own, 0, size) == hipSuccess && hipDeviceSynchronize() == hipSuccess);
    VdExternalBuffer* h = nullptr; void* imported = nullptr;
    if (vd_import_external_buffer(ctx, fd, size, &h, &imported) != VD_OK) { std::printf("SKIP: import refused: %s\n", vd_last_error(ctx)); return 0; }
    uint32_t* word_hip = reinterpret_cast<uint32_t*>(static_cast<char*>(imported) + size - 64);      // the same 4 bytes,
    uint32_t* word_renderer = reinterpret_cast<uint32_t*>(static_cast<char*>(own) + size - 64);      // seen through the two mappings

    std::vector<VdDrawIndexedIndirect> in(n);
    VdDrawIndexedIndirect* d_in = nullptr; uint32_t* d_count = nullptr;
    CHECK(hipMalloc(&d_in, n * sizeof(in[0])) == hipSuccess && hipMalloc(&d_count, 16) == hipSuccess);
    for (uint32_t frame = 1; frame <= 3; ++frame) {
        for (uint32_t i = 0; i < n; ++i) in[i] = VdDrawIndexedIndirect{36u + frame, ((i + frame) % 3u) ? 1u : 0u, i * 3u, (int32_t)i, i};
        // the renderer's upload of this frame (its own stream; finished before it writes the frame word)
        CHECK(hipMemcpy(d_in, in.data(), n * sizeof(in[0]), hipMemcpyHostToDevice) == hipSuccess);
        // HIP side: everything queued at once, nothing waits on the host
        int rc = vd_wait_value32_async(ctx, word_hip, frame);
        if (rc != VD_OK) { std::printf("SKIP: stream value wait refused: %s\n", vd_last_error(ctx)); return 0; }
        CHECK(vd_compact_draws_dev(ctx, d_in, n, static_cast<VdDrawIndexedIndirect*>(imported), d_count) == VD_OK);
        CHECK(vd_host_callback_async(ctx, frame_done, reinterpret_cast<void*>(uintptr_t(frame))) == VD_OK);
        std::this_thread::sleep_for(std::chrono::milliseconds(150));
        if (g_done.load(std::memory_order_acquire) == frame) {
            std::fprintf(stderr, "FAILED: frame %u ran before the renderer wrote its frame word\n", frame);
            return 1;
        }
        // the renderer's queue reaches the end of its submit
        CHECK(vd_write_value32_async(renderer, word_renderer, frame) == VD_OK);
        const auto t0 = std::chrono::steady_clock::now();
        while (g_done.load(std::memory_order_acquire) != frame) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) { std::fprintf(stderr, "FAILED: frame %u never completed after the write\n", frame); return 1; }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        // the consumer's view: the renderer reads the list through its own mapping
        uint32_t count = 0;
        CHECK(hipMemcpy(&count, d_count, 4, hipMemcpyDeviceToHost) == hipSuccess);
        std::vector<VdDrawIndexedIndirect> got(count);
        CHECK(hipMemcpy(got.data(), own, count * sizeof(got[0]), hipMemcpyDeviceToHost) == hipSuccess);
        uint32_t k = 0;
        for (uint32_t i = 0; i < n; ++i)
            if (in[i].instance_count == 1u) { CHECK(k < count && std::memcmp(&got[k], &in[i], sizeof(in[i])) == 0); ++k; }
        CHECK(k == count);
    }
    CHECK(vd_ctx_synchronize(ctx) == VD_OK && vd_ctx_synchronize(renderer) == VD_OK);
    CHECK(vd_release_external_buffer(ctx, h) == VD_OK);
    vd_ctx_destroy(renderer);
    std::printf("frame_ordering_test OK (3 frames: the cull's stream held on a word of the imported buffer until the other queue wrote it, completion by host callback)\n");
    vd_ctx_destroy(ctx);
    return 0;
}

// GPU check of vd_import_external_buffer: an allocation exported as a file descriptor (here by HIP's own VMM API,
// standing in for the Vulkan allocation behind wgpu's draw_cmd_buffer) is mapped through the C ABI, written through
// the imported pointer, and read back through the exporter's own mapping.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "voidin_abi.h"

#define SKIP(msg) do { std::printf("SKIP: %s\n", msg); return 0; } while (0)
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    VdCtx* ctx = nullptr;
    CHECK(vd_ctx_create(0, &ctx) == VD_OK);
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) SKIP("no VMM granularity");
    const size_t size = ((20u * 100000u + gran - 1) / gran) * gran;      // 100 k draw commands, rounded up
    hipMemGenericAllocationHandle_t alloc;
    if (hipMemCreate(&alloc, size, &prop, 0) != hipSuccess) SKIP("hipMemCreate with an exportable handle is not supported here");
    int fd = -1;
    if (hipMemExportToShareableHandle(&fd, alloc, hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess || fd < 0) SKIP("export to fd not supported");
    // the exporter's own view
    void* own = nullptr;
    CHECK(hipMemAddressReserve(&own, size, 0, nullptr, 0) == hipSuccess);
    CHECK(hipMemMap(own, size, 0, alloc, 0) == hipSuccess);
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof(acc));
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK(hipMemSetAccess(own, size, &acc, 1) == hipSuccess);
    CHECK(hipMemset(own, 0, size) == hipSuccess);
    // the C ABI's view
    VdExternalBuffer* h = nullptr; void* imported = nullptr;
    const int rc = vd_import_external_buffer(ctx, fd, size, &h, &imported);
    if (rc != VD_OK) { std::printf("SKIP: import refused: %s\n", vd_last_error(ctx)); return 0; }
    CHECK(imported != nullptr);
    // write a command list through the imported pointer with the product's own kernel (standalone compaction of a
    // synthetic command buffer), read it back through the exporter's mapping
    const uint32_t n = 100000;
    std::vector<VdDrawIndexedIndirect> in(n);
    for (uint32_t i = 0; i < n; ++i) in[i] = VdDrawIndexedIndirect{36u + i % 7u, (i % 3u) ? 1u : 0u, i * 3u, (int32_t)i, i};
    VdDrawIndexedIndirect* d_in = nullptr; uint32_t* d_count = nullptr;
    CHECK(hipMalloc(&d_in, n * sizeof(in[0])) == hipSuccess && hipMalloc(&d_count, 16) == hipSuccess);
    CHECK(hipMemcpy(d_in, in.data(), n * sizeof(in[0]), hipMemcpyHostToDevice) == hipSuccess);
    CHECK(vd_compact_draws_dev(ctx, d_in, n, (VdDrawIndexedIndirect*)imported, d_count) == VD_OK);
    CHECK(vd_ctx_synchronize(ctx) == VD_OK);
    uint32_t count = 0;
    CHECK(hipMemcpy(&count, d_count, 4, hipMemcpyDeviceToHost) == hipSuccess);
    std::vector<VdDrawIndexedIndirect> got(count);
    CHECK(hipMemcpy(got.data(), own, count * sizeof(got[0]), hipMemcpyDeviceToHost) == hipSuccess);
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (in[i].instance_count == 1u) { CHECK(k < count && std::memcmp(&got[k], &in[i], sizeof(in[i])) == 0); ++k; }
    CHECK(k == count);
    CHECK(vd_release_external_buffer(ctx, h) == VD_OK);
    std::printf("external_buffer_test OK (%u commands through an imported %zu-byte allocation)\n", count, size);
    return 0;
}

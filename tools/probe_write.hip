// Write-bandwidth probe: how fast can gfx950 absorb a pure store stream, by footprint and pattern?
// build: hipcc --offload-arch=gfx950 -O3 -o build/probe_write tools/probe_write.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// (a) aligned fill, grid-stride, 16 B per lane
__global__ __launch_bounds__(256) void fill16(u32x4* out, size_t n16) {
    const u32x4 v = {1u, 2u, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}
// (b) aligned fill, one 32 KB chunk per workgroup (non-persistent), 16 B per lane
__global__ __launch_bounds__(256) void fill16_chunk(u32x4* out, size_t n16) {
    const u32x4 v = {1u, 2u, 3u, 4u};
    const size_t base = (size_t)blockIdx.x * 2048;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const size_t i = base + k * 256 + threadIdx.x; if (i < n16) out[i] = v; }
}
// (c) as (b) but each wave owns a contiguous 8 KB run (like expand: wave-contiguous runs)
__global__ __launch_bounds__(256) void fill16_waverun(u32x4* out, size_t n16, int off16) {
    const u32x4 v = {1u, 2u, 3u, 4u};
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = ((size_t)blockIdx.x * 4 + wave) * 2560 + off16;   // 40 KB per wave
#pragma unroll 8
    for (int k = 0; k < 40; ++k) { const size_t i = base + k * 64 + lane; if (i < n16) out[i] = v; }
}
// (d) 20-byte records: dwordx4 + dword per lane at a 20-byte stride
__global__ __launch_bounds__(256) void fill20(unsigned* out, size_t nrec) {
    typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
    const u32x4 v = {1u, 2u, 3u, 4u};
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = ((size_t)blockIdx.x * 4 + wave) * 2048;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) { const size_t r = base + k * 64 + lane; if (r < nrec) { unsigned* o = out + r * 5; *reinterpret_cast<u32x4_a4*>(o) = v; o[4] = (unsigned)r; } }
}
// (e) nontemporal aligned fill, grid-stride
__global__ __launch_bounds__(256) void fill16_nt(u32x4* out, size_t n16) {
    const u32x4 v = {1u, 2u, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, &out[i]);
}
template <typename F> static float timeit(F f, int iters = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int i = 0; i < iters + 3; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (i >= 3) t.push_back(ms); }
    std::sort(t.begin(), t.end()); return t[t.size() / 2];
}
int main() {
    const size_t sizes[] = {191ull << 20, 1530ull << 20};
    for (size_t bytes : sizes) {
        void* d; if (hipMalloc(&d, bytes + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
        const size_t n16 = bytes / 16, nrec = bytes / 20;
        auto rep = [&](const char* name, float ms) { printf("%5zu MB %-28s %8.1f us  %7.0f GB/s\n", bytes >> 20, name, ms * 1e3, bytes / (ms * 1e-3) / 1e9); };
        rep("fill16 grid-stride 1024 blk", timeit([&] { fill16<<<1024, 256>>>((u32x4*)d, n16); }));
        rep("fill16 grid-stride 4096 blk", timeit([&] { fill16<<<4096, 256>>>((u32x4*)d, n16); }));
        rep("fill16 nontemporal 1024 blk", timeit([&] { fill16_nt<<<1024, 256>>>((u32x4*)d, n16); }));
        rep("fill16 32KB chunk per block", timeit([&] { fill16_chunk<<<(unsigned)((n16 + 2047) / 2048), 256>>>((u32x4*)d, n16); }));
        rep("fill16 wave runs 40KB +0", timeit([&] { fill16_waverun<<<(unsigned)((n16 + 10239) / 10240), 256>>>((u32x4*)d, n16, 0); }));
        rep("fill16 wave runs 40KB +16B", timeit([&] { fill16_waverun<<<(unsigned)((n16 + 10239) / 10240), 256>>>((u32x4*)d, n16, 1); }));
        rep("fill20 records x4+x1", timeit([&] { fill20<<<(unsigned)((nrec + 8191) / 8192), 256>>>((unsigned*)d, nrec); }));
        hipFree(d);
    }
    return 0;
}

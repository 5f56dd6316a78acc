// Round-trip latencies that bound a dependent climb (TLAS refit): one lane, N dependent operations each.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_latency.hip -o build/probe_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N = 2000;

__global__ void chase_plain(const unsigned* next, unsigned* out) {
    unsigned k = 0;
    for (int i = 0; i < N; ++i) k = next[k];
    *out = k;
}
__global__ void chase_agent(unsigned* next, unsigned* out) {
    unsigned k = 0;
    for (int i = 0; i < N; ++i) k = __hip_atomic_load(&next[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *out = k;
}
__global__ void chase_atomic_add(unsigned* ctr, unsigned* out) {
    unsigned k = 0;
    for (int i = 0; i < N; ++i) k = __hip_atomic_fetch_add(&ctr[(k & 1023u) * 64u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + i;
    *out = k;
}
__global__ void store_wait(unsigned* buf, unsigned* out) {
    for (int i = 0; i < N; ++i) {
        __hip_atomic_store(&buf[(i & 1023) * 64], (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *out = 1;
}
__global__ void store_wait_plain(unsigned* buf, unsigned* out) {
    for (int i = 0; i < N; ++i) {
        buf[(i & 1023) * 64] = (unsigned)i;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *out = 1;
}
__global__ void chase_wg_scope(unsigned* next, unsigned* out) {
    unsigned k = 0;
    for (int i = 0; i < N; ++i) k = __hip_atomic_load(&next[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    *out = k;
}

template <typename F> float time_it(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    const int M = 1 << 16;                 // 64 K entries, stride 64 B apart via permutation: 4 MB > L1, fits one L2
    std::vector<unsigned> h(M * 16);
    unsigned x = 12345;
    std::vector<unsigned> perm(M);
    for (int i = 0; i < M; ++i) perm[i] = i;
    for (int i = M - 1; i > 0; --i) { x = x * 1664525u + 1013904223u; int j = x % (i + 1); std::swap(perm[i], perm[j]); }
    for (int i = 0; i < M; ++i) h[(size_t)perm[i] * 16] = perm[(i + 1) % M] * 16;
    unsigned *d_next, *d_ctr, *d_out;
    hipMalloc(&d_next, h.size() * 4); hipMalloc(&d_ctr, 1024 * 64 * 4); hipMalloc(&d_out, 64);
    hipMemcpy(d_next, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d_ctr, 0, 1024 * 64 * 4);
    printf("per operation, one lane, %d dependent ops:\n", N);
    printf("  plain load, 4 MB working set      %7.1f ns\n", time_it([&] { chase_plain<<<1, 1>>>(d_next, d_out); }) * 1e6 / N);
    printf("  workgroup-scope atomic load        %7.1f ns\n", time_it([&] { chase_wg_scope<<<1, 1>>>(d_next, d_out); }) * 1e6 / N);
    printf("  agent-scope atomic load            %7.1f ns\n", time_it([&] { chase_agent<<<1, 1>>>(d_next, d_out); }) * 1e6 / N);
    printf("  agent-scope fetch_add (returning)  %7.1f ns\n", time_it([&] { chase_atomic_add<<<1, 1>>>(d_ctr, d_out); }) * 1e6 / N);
    printf("  agent-scope store + vmcnt(0)       %7.1f ns\n", time_it([&] { store_wait<<<1, 1>>>(d_ctr, d_out); }) * 1e6 / N);
    printf("  plain store + vmcnt(0)             %7.1f ns\n", time_it([&] { store_wait_plain<<<1, 1>>>(d_ctr, d_out); }) * 1e6 / N);
    return 0;
}

// What a wave pays for 64 divergent 64-byte fetches (the traversal's node step), and whether the shape of the loads
// matters: (a) every lane reads the four 16-byte quarters of ITS line (4 x global_load_dwordx4, 64 lines per instruction);
// (b) quad-cooperative: in instruction j the four lanes of a quad read the four quarters of the line of the quad's lane j
// (4 instructions as well, but 16 lines per instruction, each covered by one quad).  Dependent chains (the next line index
// depends on the data), 28 waves per CU like the traversal kernel, working sets that sit in L2 / MALL / HBM.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_gather.hip -o tools/probe_gather ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int N = 1000;

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int J> __device__ __forceinline__ unsigned quad_bcast(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, true);
}

__global__ __launch_bounds__(64, 7) void own_lines(const float4* __restrict__ buf, unsigned mask, unsigned* out) {
    unsigned st = mix(blockIdx.x * 64u + threadIdx.x + 1u);
    float acc = 0.f;
    for (int i = 0; i < N; ++i) {
        const float4* p = buf + 4u * (size_t)(st & mask);
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += (a.x + b.y) + (c.z + d.w);
        st = mix(st + __float_as_uint(a.x));
    }
    out[blockIdx.x * 64u + threadIdx.x] = st + __float_as_uint(acc);
}

__global__ __launch_bounds__(64, 7) void quad_lines(const float4* __restrict__ buf, unsigned mask, unsigned* out) {
    unsigned st = mix(blockIdx.x * 64u + threadIdx.x + 1u);
    const unsigned q = threadIdx.x & 3u;
    float acc = 0.f;
    for (int i = 0; i < N; ++i) {
        const unsigned mine = st & mask;
        const float4 r0 = buf[4u * (size_t)quad_bcast<0>(mine) + q];
        const float4 r1 = buf[4u * (size_t)quad_bcast<1>(mine) + q];
        const float4 r2 = buf[4u * (size_t)quad_bcast<2>(mine) + q];
        const float4 r3 = buf[4u * (size_t)quad_bcast<3>(mine) + q];
        acc += (r0.x + r1.y) + (r2.z + r3.w);
        const float own = q == 0u ? r0.x : q == 1u ? r1.x : q == 2u ? r2.x : r3.x;   // a word of this lane's own line
        st = mix(st + __float_as_uint(own));
    }
    out[blockIdx.x * 64u + threadIdx.x] = st + __float_as_uint(acc);
}

// the same 64 bytes per lane as two 32-byte halves in two separate lines (the TLAS step: two children, two nodes)
__global__ __launch_bounds__(64, 7) void own_two_halves(const float4* __restrict__ buf, unsigned mask, unsigned* out) {
    unsigned st = mix(blockIdx.x * 64u + threadIdx.x + 1u);
    float acc = 0.f;
    for (int i = 0; i < N; ++i) {
        const float4* p = buf + 4u * (size_t)(st & mask);
        const float4* p2 = buf + 4u * (size_t)(mix(st) & mask) + 2;
        const float4 a = p[0], b = p[1], c = p2[0], d = p2[1];
        acc += (a.x + b.y) + (c.z + d.w);
        st = mix(st + __float_as_uint(a.x));
    }
    out[blockIdx.x * 64u + threadIdx.x] = st + __float_as_uint(acc);
}

int main() {
    int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount, waves = cus * 28;
    const double clk = prop.clockRate * 1e3;
    unsigned* d_out; hipMalloc(&d_out, (size_t)waves * 64 * 4);
    const size_t max_lines = (size_t)1 << 24;                      // 1 GB of 64-byte lines
    float4* d_buf; hipMalloc(&d_buf, max_lines * 64);
    { std::vector<float> h(max_lines * 16 / 64); for (auto& v : h) v = (float)rand() / RAND_MAX;      // a 16 MB pattern, repeated
      for (size_t o = 0; o < max_lines * 64; o += h.size() * 4) hipMemcpy((char*)d_buf + o, h.data(), h.size() * 4, hipMemcpyHostToDevice); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs, %d waves (28 per CU), %d dependent steps per lane, clock %.0f MHz\n", cus, waves, N, clk / 1e6);
    printf("%-10s %-18s %10s %14s %16s\n", "set", "kernel", "ms", "us/wave-step", "lane-steps/clk/CU");
    for (unsigned lg : {14u, 17u, 20u, 24u}) {                       // 1 MB, 8 MB, 64 MB, 1 GB
        const unsigned mask = (1u << lg) - 1u;
        for (int k = 0; k < 3; ++k) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (k == 0) hipLaunchKernelGGL(own_lines, dim3(waves), dim3(64), 0, 0, d_buf, mask, d_out);
                else if (k == 1) hipLaunchKernelGGL(quad_lines, dim3(waves), dim3(64), 0, 0, d_buf, mask, d_out);
                else hipLaunchKernelGGL(own_two_halves, dim3(waves), dim3(64), 0, 0, d_buf, mask, d_out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            char set[32]; snprintf(set, sizeof set, "%u MB", (1u << lg) / 16384u);
            printf("%-10s %-18s %10.3f %14.3f %16.4f\n", set, k == 0 ? "own 4 x 16 B" : k == 1 ? "quad-cooperative" : "own 2 x 32 B, 2 lines", best,
                   best * 1e3 / N, (double)waves * 64 * N / (best * 1e-3) / clk / cus);
        }
    }
    return 0;
}

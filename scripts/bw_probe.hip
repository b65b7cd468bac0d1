// Development aid: what HBM bandwidth do shade_kernel-like access patterns reach on this device?  Streams of 16-byte elements,
// one per lane, read from K arrays and written to K arrays (a) densely, (b) through an index list with gaps (every
// `keep`-th of 8 slots dropped, as a path queue looks after a few bounces).  Prints GB/s; compare with roofline.achieved of
// shade_kernel in bench.py's line.   hipcc --offload-arch=gfx950 -O3 -o build/bw_probe scripts/bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

template <int K>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ in, float4* __restrict__ out, const uint32_t* __restrict__ idx,
                                                     uint32_t n, size_t stride) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = idx ? idx[i] : i;
    float4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = in[k * stride + s];
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k].x += 1.0f; out[k * stride + s] = v[k]; }
}

int main() {
    const uint32_t slots = 1u << 26;
    constexpr int K = 5;
    float4 *in, *out; uint32_t* idx;
    CHECK(hipMalloc(&in, sizeof(float4) * (size_t)slots * K));
    CHECK(hipMalloc(&out, sizeof(float4) * (size_t)slots * K));
    CHECK(hipMalloc(&idx, sizeof(uint32_t) * (size_t)slots));
    CHECK(hipMemset(in, 0, sizeof(float4) * (size_t)slots * K));
    CHECK(hipMemset(out, 0, sizeof(float4) * (size_t)slots * K));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 4; ++mode) {
        // mode 0: dense, no index; 1: index = identity; 2: one of 4 slots dropped; 3: random half of the slots dropped
        std::vector<uint32_t> h;
        uint32_t n = slots;
        if (mode > 0) {
            h.reserve(slots);
            uint32_t lcg = 12345u;
            for (uint32_t s = 0; s < slots; ++s) {
                lcg = lcg * 1664525u + 1013904223u;
                const bool keep = mode == 1 || (mode == 2 && (s & 3u) != 3u) || (mode == 3 && (lcg >> 16 & 1u));
                if (keep) h.push_back(s);
            }
            n = (uint32_t)h.size();
            CHECK(hipMemcpy(idx, h.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
        }
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(stream_kernel<K>, dim3((n + 255u) / 256u), dim3(256), 0, 0, in, out, mode ? idx : nullptr, n, (size_t)slots);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = (double)n * (K * 32.0 + (mode ? 4.0 : 0.0));
        std::printf("mode %d: %u elements x (%d x 16 B in + %d x 16 B out): %.3f ms, %.0f GB/s of useful bytes\n", mode, n, K, K, best, bytes / best / 1e6);
    }
    return 0;
}

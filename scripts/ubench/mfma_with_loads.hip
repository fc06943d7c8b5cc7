// micro-benchmark: fp32 MFMA throughput of the whole chip while every wave ALSO streams 16-byte loads from an L2 / Infinity-Cache
// resident buffer at GEMM-like rates (NL loads of 1 KiB per wave per 32 MFMAs; nothing depends on them but a final sum).
// Separates "the memory path costs issue slots / stalls in MY kernel" from "MFMA + memory traffic together exceed the power
// budget and the chip clocks down".
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ inline float rnd(unsigned s) {
    s ^= s >> 16; s *= 0x7feb352dU; s ^= s >> 15; s *= 0x846ca68bU; s ^= s >> 16;
    return (float)(int)(s & 0xffffff) * (1.0f / 8388608.0f) - 1.0f;
}
template <int NL>
__global__ __launch_bounds__(512) void k(float* out, const f32x4* __restrict__ src, unsigned mask, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[16], b[16];
    for (int e = 0; e < 16; ++e) { a[e] = rnd(threadIdx.x * 131 + e * 7 + blockIdx.x * 977); b[e] = rnd(threadIdx.x * 257 + e * 13 + blockIdx.x * 31 + 5); }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    unsigned pos = (blockIdx.x * 4099u + (threadIdx.x >> 6) * 523u) * 64u + (threadIdx.x & 63);
    for (int n = 0; n < iters; ++n) {
        f32x4 v[NL > 0 ? NL : 1];
#pragma unroll
        for (int q = 0; q < NL; ++q) v[q] = src[(pos + q * 64u * 131u) & mask];
        pos += 64u * 1543u;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * 4 + i) & 15], b[(u * 4 + i + 5) & 15], acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NL; ++q) sum += v[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][n & 15] *= 0.5f;
    }
    float s = sum[0] + sum[1] + sum[2] + sum[3];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NL>
void run(const f32x4* src, unsigned mask, const char* name) {
    const int wgs = 1024, threads = 512, iters = 2000;
    float* out;
    hipMalloc(&out, (size_t)wgs * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NL>), dim3(wgs), dim3(threads), 0, 0, out, src, mask, iters);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<NL>), dim3(wgs), dim3(threads), 0, 0, out, src, mask, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)wgs * (threads / 64) * iters * 32.0 * 2.0 * 32 * 32 * 2;
        const double bytes = (double)wgs * (threads / 64) * iters * NL * 1024.0;
        printf("%-34s %2d loads / 32 MFMAs: %8.3f ms -> %7.1f TFLOP/s, %6.2f TB/s of loads (%4.1f B/clk/CU at 2.4 GHz)\n", name, NL, ms, flop / ms * 1e-9,
               bytes / ms * 1e-9, bytes / ms * 1e-9 * 1e12 / 256 / 2.4e9);
    }
    (void)hipFree(out);
}
int main() {
    const size_t n16 = (size_t)1 << 24;   // 256 MiB of float4 at most
    f32x4* src;
    hipMalloc(&src, n16 * 16);
    hipMemset(src, 0x3c, n16 * 16);
    const unsigned m_l2 = (1u << 17) - 1;    // 2 MiB: L2-resident in every XCD
    const unsigned m_mall = (1u << 23) - 1;  // 128 MiB: Infinity-Cache resident, mostly L2 misses
    run<0>(src, m_l2, "no loads");
    run<3>(src, m_l2, "L2-resident source");
    run<6>(src, m_l2, "L2-resident source");
    run<12>(src, m_l2, "L2-resident source");
    run<6>(src, m_mall, "Infinity-Cache-resident source");
    run<12>(src, m_mall, "Infinity-Cache-resident source");
    return 0;
}

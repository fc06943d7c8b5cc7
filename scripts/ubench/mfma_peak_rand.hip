// micro-benchmark: sustained fp32 MFMA throughput with RANDOM, changing operands (power-realistic), against the constant-operand
// loop of mfma_peak.hip: the chip clocks to its power budget, so the "157 TFLOP/s" roof is data-dependent.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ inline float rnd(unsigned s) {
    s ^= s >> 16; s *= 0x7feb352dU; s ^= s >> 15; s *= 0x846ca68bU; s ^= s >> 16;
    return (float)(int)(s & 0xffffff) * (1.0f / 8388608.0f) - 1.0f;   // [-1, 1)
}
template <bool RANDOM>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[16], b[16];
    for (int e = 0; e < 16; ++e) {
        a[e] = RANDOM ? rnd(threadIdx.x * 131 + e * 7 + blockIdx.x * 977) : 1.0f;
        b[e] = RANDOM ? rnd(threadIdx.x * 257 + e * 13 + blockIdx.x * 31 + 5) : 2.0f;
    }
    for (int n = 0; n < iters; ++n) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * 4 + i) & 15], b[(u * 4 + i + 5) & 15], acc[i], 0, 0, 0);
        if (RANDOM) {   // keep the accumulators bounded and changing (cheap VALU, every 32 MFMAs)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][n & 15] *= 0.5f;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <bool RANDOM>
void run(int wgs, int threads, int iters, const char* name) {
    float* out;
    hipMalloc(&out, (size_t)wgs * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<RANDOM>), dim3(wgs), dim3(threads), 0, 0, out, iters);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<RANDOM>), dim3(wgs), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)wgs * (threads / 64) * iters * 32.0 * 2.0 * 32 * 32 * 2;
        printf("%-26s %5d WGs x %4d threads, %6d x 32 MFMAs per wave: %8.3f ms  -> %8.1f TFLOP/s\n", name, wgs, threads, iters, ms, flop / ms * 1e-9);
    }
    (void)hipFree(out);
}
int main() {
    run<false>(1024, 512, 400, "constant operands");
    run<true>(1024, 512, 400, "random operands");
    run<true>(1024, 512, 4000, "random operands (long)");
    run<false>(1024, 512, 4000, "constant operands (long)");
    return 0;
}

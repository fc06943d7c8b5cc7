// micro-benchmark: SUSTAINED whole-chip MFMA throughput (wall clock, HIP events) — what the "157.3 / 2500 TFLOP/s" roofs are
// worth under a millisecond-long all-CU load (clock management included).  1024 workgroups x 512 threads, pure MFMA loops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.25f * i + r;
    const float x = a + threadIdx.x * 1e-3f, w = b;
    bf16x8 xb, wb;
    for (int e = 0; e < 8; ++e) { xb[e] = (__bf16)(x + e); wb[e] = (__bf16)(w - e); }
    for (int n = 0; n < iters; ++n) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, xb, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w + i, x + 0.5f * i, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <bool BF>
void run(int wgs, int threads, int iters, const char* name) {
    float* out;
    hipMalloc(&out, (size_t)wgs * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BF>), dim3(wgs), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<BF>), dim3(wgs), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)wgs * (threads / 64) * iters * 32.0 * (BF ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2);
        printf("%-22s %5d WGs x %4d threads, %6d x 32 MFMAs per wave: %8.3f ms  -> %8.1f TFLOP/s\n", name, wgs, threads, iters, ms, flop / ms * 1e-9);
    }
    hipFree(out);
}
int main() {
    run<false>(1024, 512, 400, "f32 32x32x2");
    run<false>(1024, 512, 4000, "f32 32x32x2 (long)");
    run<false>(256, 256, 1600, "f32, 1 wave/SIMD");
    run<true>(1024, 512, 800, "bf16 32x32x16");
    run<true>(1024, 512, 8000, "bf16 32x32x16 (long)");
    return 0;
}

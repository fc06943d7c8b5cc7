// micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 for dependent / independent chains at 1..4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int N>
__global__ void k(float* out, long long* clk, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.25f * i + r;
    float x = a + threadIdx.x * 1e-3f, w = b;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w + i, x + 0.5f * i, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    asm volatile("s_nop 0" ::"v"(s));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) clk[blockIdx.x] = (long long)(t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int N>
void run(int threads, const char* name) {
    float* out; long long* clk;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&clk, 1024 * 8);
    hipLaunchKernelGGL((k<NACC, N>), dim3(256), dim3(threads), 0, 0, out, clk, 1.f, 2.f);
    hipLaunchKernelGGL((k<NACC, N>), dim3(256), dim3(threads), 0, 0, out, clk, 1.f, 2.f);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
    printf("%-28s threads/WG=%4d (waves/SIMD=%d): %7.1f clk per MFMA per wave (block avg %.0f clk for %d MFMAs)\n", name, threads, threads / 256,
           s / 256 / (N * NACC), s / 256, N * NACC);
    hipFree(out); hipFree(clk);
}
int main() {
    run<1, 64>(256, "dependent chain"); run<1, 64>(512, "dependent chain"); run<1, 64>(1024, "dependent chain");
    run<2, 32>(256, "2 independent acc"); run<2, 32>(512, "2 independent acc");
    run<4, 16>(256, "4 independent acc"); run<4, 16>(512, "4 independent acc");
    return 0;
}

// micro-benchmark: when do the waves of a workgroup start?  228 workgroups (one per CU, like the UNet GEMM at B=1) of
// NW waves; every wave stamps s_memtime at entry.  Reported: per-workgroup spread (last wave start - first wave start)
// and the chip-wide spread, for different VGPR budgets, LDS sizes and workgroup sizes.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int NW, int NV>
__global__ __launch_bounds__(64 * NW) void k_stamp(long long* out, float* sink) {
    extern __shared__ float sm[];
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    // keep NV VGPRs live so the allocation is real
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * NW + (threadIdx.x >> 6)] = (long long)t;
    if (s == -1.f) { sm[threadIdx.x] = s; sink[0] = sm[0]; }
}

template <int NW, int NV>
void run(int nwg, int lds) {
    long long* d; float* sink;
    hipMalloc(&d, sizeof(long long) * nwg * NW); hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stamp<NW, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<long long> h(nwg * NW);
    double wg_spread = 0, chip = 0, half = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; ++r) {
        hipLaunchKernelGGL((k_stamp<NW, NV>), dim3(nwg), dim3(64 * NW), lds, 0, d, sink);
        hipDeviceSynchronize();
        if (r < 2) continue;
        hipMemcpy(h.data(), d, sizeof(long long) * nwg * NW, hipMemcpyDeviceToHost);
        long long lo = *std::min_element(h.begin(), h.end()), hi = *std::max_element(h.begin(), h.end());
        chip += (double)(hi - lo);
        double sp = 0, hf = 0;
        for (int g = 0; g < nwg; ++g) {
            auto b = h.begin() + g * NW;
            sp += (double)(*std::max_element(b, b + NW) - *std::min_element(b, b + NW));
            if (NW >= 8) {   // second half of the waves vs the first half
                double a0 = 0, a1 = 0;
                for (int w = 0; w < NW / 2; ++w) { a0 += (double)b[w]; a1 += (double)b[NW / 2 + w]; }
                hf += (a1 - a0) / (NW / 2);
            }
        }
        wg_spread += sp / nwg; half += hf / nwg;
    }
    printf("%3d WGs x %d waves, %3d VGPRs live, %6d B LDS: in-WG spread %6.0f clk, waves[NW/2..) - waves[0..NW/2) %6.0f clk, chip-wide spread %6.0f clk\n",
           nwg, NW, NV, lds, wg_spread / reps, half / reps, chip / reps);
    hipFree(d); hipFree(sink);
}

int main() {
    run<8, 8>(228, 0);
    run<8, 8>(228, 42 * 1024);
    run<8, 8>(228, 140 * 1024);
    run<8, 100>(228, 42 * 1024);
    run<8, 200>(228, 42 * 1024);
    run<4, 100>(228, 42 * 1024);
    run<4, 100>(456, 42 * 1024);
    run<8, 100>(114, 42 * 1024);
    run<8, 100>(456, 42 * 1024);
    run<16, 60>(228, 42 * 1024);
    return 0;
}

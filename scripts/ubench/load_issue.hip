// micro-benchmark: how long does a 512-thread workgroup (one per CU) need to issue and receive N coalesced loads per wave,
// as dword vs dwordx4, from an L2/MALL-resident buffer?  (228 workgroups, like the UNet GEMMs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
template <int N, int W>
__global__ __launch_bounds__(512) void k(const float* src, float* out, long long* clk, int rowstride) {
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64 << 20, 0x00020000);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    float acc = 0.f;
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int base = (blockIdx.x * 8 + w) * N * rowstride;
    if (W == 1) {
        float v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, l * 4, (base + i * rowstride) * 4, 0));
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
#pragma unroll
        for (int i = 0; i < N; ++i) acc += v[i];
    } else {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v[N / 4];
#pragma unroll
        for (int i = 0; i < N / 4; ++i) v[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, l * 16, (base + i * rowstride * 4) * 4, 0));
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
#pragma unroll
        for (int i = 0; i < N / 4; ++i) acc += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    asm volatile("s_nop 0" ::"v"(acc));
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    if (l == 0) { clk[(blockIdx.x * 8 + w) * 2] = t1 - t0; clk[(blockIdx.x * 8 + w) * 2 + 1] = t2 - t0; }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int N, int W>
void run(const char* name, float* src, float* out, long long* clk, int blocks) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<N, W>), dim3(blocks), dim3(512), 0, 0, src, out, clk, 64);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 16);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0, mb = 0;
    for (int i = 0; i < blocks * 8; ++i) { a += h[2 * i]; b += h[2 * i + 1]; if (h[2 * i + 1] > mb) mb = h[2 * i + 1]; }
    printf("%-10s N=%3d elems/lane (%s): issue %7.0f clk, data back %7.0f clk (max %7.0f)  [%d WGs]\n", name, N, W == 1 ? "dword  " : "dwordx4", a / blocks / 8, b / blocks / 8, mb, blocks);
}
int main() {
    float *src, *out; long long* clk;
    hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20); hipMalloc(&out, 1 << 22); hipMalloc(&clk, 1 << 20);
    run<24, 1>("1x1", src, out, clk, 228); run<24, 4>("1x1", src, out, clk, 228);
    run<72, 1>("conv", src, out, clk, 228); run<72, 4>("conv", src, out, clk, 228);
    run<72, 1>("conv", src, out, clk, 16); run<72, 4>("conv", src, out, clk, 16);
    return 0;
}

// micro-benchmark: per-node cost of a hipGraph chain of dependent kernels with the UNet GEMM's launch geometry
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char b[1024]; };
__global__ void k_trivial(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ __launch_bounds__(512) void k_empty(int* p) { extern __shared__ float sm[]; if (threadIdx.x == 0) sm[0] = 1.f; __syncthreads(); if (sm[0] < 0) p[1] = 1; }
__global__ __launch_bounds__(512) void k_big(const Big a, int* p) { extern __shared__ float sm[]; if (threadIdx.x == 0) sm[0] = a.b[5]; __syncthreads(); if (sm[0] < -100) p[1] = 1; }
__global__ __launch_bounds__(512) void k_touch(const float* src, float* dst, int n) {   // reads 100 KB/WG, writes 4 KB/WG
    extern __shared__ float sm[];
    float acc = 0.f;
    const float* s = src + (size_t)blockIdx.x * 25600;
    for (int i = threadIdx.x; i < 25600; i += 512) acc += s[i];
    sm[threadIdx.x] = acc; __syncthreads();
    dst[(size_t)blockIdx.x * 1024 + threadIdx.x] = sm[threadIdx.x ^ 1] + acc;
}
// like the conv: 74 KB of "weights" shared by the 38 WGs with the same blockIdx.y + 26 KB private "x"
__global__ __launch_bounds__(512) void k_conv_like(const float* wsrc, const float* xsrc, float* dst) {
    extern __shared__ float sm[];
    float acc = 0.f;
    const float* wp = wsrc + (size_t)blockIdx.y * 18944;
    for (int i = threadIdx.x; i < 18944; i += 512) acc += wp[i];
    const float* xp = xsrc + (size_t)(blockIdx.x + 19 * blockIdx.z) * 6656;
    for (int i = threadIdx.x; i < 6656; i += 512) acc += xp[i];
    sm[threadIdx.x] = acc; __syncthreads();
    dst[(size_t)(blockIdx.x + 19 * (blockIdx.y + 6 * blockIdx.z)) * 1024 + threadIdx.x] = sm[threadIdx.x ^ 1] + acc;
}
__global__ __launch_bounds__(512) void k_same(const float* src, float* dst) {   // every WG reads the SAME 100 KB
    extern __shared__ float sm[];
    float acc = 0.f;
    for (int i = threadIdx.x; i < 25600; i += 512) acc += src[i];
    sm[threadIdx.x] = acc; __syncthreads();
    dst[(size_t)blockIdx.x * 1024 + threadIdx.x] = sm[threadIdx.x ^ 1] + acc;
}
template <typename F>
double bench(const char* name, int nodes, F launch) {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int i = 0; i < 3; ++i) launch(s);
    hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nodes; ++i) launch(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    const int reps = 50;
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps / nodes;
    printf("%-44s %6.2f us per node (graph of %d nodes)\n", name, us, nodes);
    // eager for comparison
    t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps * nodes; ++r) launch(s);
    hipStreamSynchronize(s);
    us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps / nodes;
    printf("%-44s %6.2f us per launch (eager stream)\n", "", us);
    return us;
}
int main() {
    int* p; hipMalloc(&p, 64); hipMemset(p, 0, 64);
    float *src, *dst; hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20); hipMalloc(&dst, 8 << 20);
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big big; big.b[5] = 3;
    bench("trivial <<<1,1>>>", 46, [&](hipStream_t s) { hipLaunchKernelGGL(k_trivial, dim3(1), dim3(1), 0, s, p); });
    bench("empty <<<228,512>>> 40 KB LDS", 46, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(19, 6, 2), dim3(512), 40960, s, p); });
    bench("empty <<<114,512>>> 40 KB LDS", 46, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(19, 6, 1), dim3(512), 40960, s, p); });
    bench("empty <<<228,512>>> 40 KB LDS, 1 KB kernarg", 46, [&](hipStream_t s) { hipLaunchKernelGGL(k_big, dim3(19, 6, 2), dim3(512), 40960, s, big, p); });
    bench("empty <<<912,512>>> 74 KB LDS", 46, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(19, 24, 2), dim3(512), 75776, s, p); });
    bench("empty <<<228,256>>> 8 KB LDS", 46, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(19, 6, 2), dim3(256), 8192, s, p); });
    bench("touch 100KB/WG <<<228,512>>>", 46, [&](hipStream_t s) { hipLaunchKernelGGL(k_touch, dim3(228), dim3(512), 2048, s, src, dst, 0); });
    bench("same 100KB for all WGs <<<228,512>>>", 46, [&](hipStream_t s) { hipLaunchKernelGGL(k_same, dim3(228), dim3(512), 2048, s, src, dst); });
    bench("conv-like 74KB shared/6 + 26KB private", 46, [&](hipStream_t s) { hipLaunchKernelGGL(k_conv_like, dim3(19, 6, 2), dim3(512), 2048, s, src, src + (8 << 20), dst); });
    return 0;
}

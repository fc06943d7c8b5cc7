// micro-benchmark: the floor of a hipGraph node when the graph cycles through MANY DIFFERENT, LARGE kernels (the engine's step: 24 launches of 10 kernels,
// 4.4-5.1 us per node with every workgroup returning at entry) against one small kernel repeated (flag_chain.hip: 1.57 us per node).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Pad { int v[48]; };   // a 192-byte argument block behind the leading scalars (the engine's kernels take 200-400 bytes)

// a large body (about 30 KB of code, > 100 VGPRs) behind an early exit
template <int ID>
__global__ __launch_bounds__(512) void k_fat(const float* x, float* y, int go, int n, const Pad p) {
    extern __shared__ float sm[];
    if (go <= 0) return;
    float acc[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) acc[i] = x[(threadIdx.x + i * 512 + ID) % n];
#pragma unroll
    for (int r = 0; r < 24; ++r) {
#pragma unroll
        for (int i = 0; i < 96; ++i) acc[i] = fmaf(acc[i], acc[(i + r + 1) % 96], (float)(ID + r + p.v[r]));
        sm[threadIdx.x] = acc[r]; __syncthreads(); acc[r] += sm[threadIdx.x ^ 1]; __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += acc[i];
    y[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k_small(const float* x, float* y, int go, int n, const Pad p) { if (go <= 0) return; y[threadIdx.x] = x[0] + p.v[0]; }

typedef void (*KF)(const float*, float*, int, int, const Pad);
template <int... I> static void fill(KF* t, std::integer_sequence<int, I...>) { ((t[I] = k_fat<I>), ...); }

static double run(KF* tab, int nk, int n, int lds, int K) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *x, *y; CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&y, 1 << 20));
    Pad p = {};
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int q = 0; q < K; ++q) hipLaunchKernelGGL(tab[q % nk], dim3(n), dim3(512), lds, s, x, y, 0, 1 << 18, p);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    const int reps = 20;
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps / K;
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(x)); CK(hipFree(y)); CK(hipStreamDestroy(s));
    return us;
}
int main() {
    KF fat[24], small[1] = {k_small};
    fill(fat, std::make_integer_sequence<int, 24>());
    for (int i = 0; i < 24; ++i) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fat[i]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_small), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int n : {38, 228})
        for (int lds : {40 * 1024, 150 * 1024}) {
            printf("%3d workgroups, %3d KB LDS: one small kernel %5.2f | one large kernel %5.2f | 4 large kernels in turn %5.2f | 24 large kernels in turn %5.2f us per node\n", n, lds / 1024,
                   run(small, 1, n, lds, 240), run(fat, 1, n, lds, 240), run(fat, 4, n, lds, 240), run(fat, 24, n, lds, 240));
            fflush(stdout);
        }
    return 0;
}

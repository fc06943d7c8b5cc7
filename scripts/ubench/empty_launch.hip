// How long does an EMPTY kernel take as a function of grid size, block size and dynamic LDS?  (round 4: battn_kernel with every
// instruction knocked out still took 48 us at 1152 workgroups x 512 threads x 80 KB.)   hipcc --offload-arch=gfx950 -O3 empty_launch.hip -o empty_launch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
template <int NT> __global__ __launch_bounds__(NT) void empty_lb(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
int main() {
    hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {256, 576, 1152, 1920, 4096};
    const int blocks[] = {256, 512};
    const int ldss[] = {0, 16384, 40960, 80384, 81920, 120000};
    for (int b : blocks) for (int l : ldss) for (int g : grids) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(b), l, s, (int*)nullptr);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(b), l, s, (int*)nullptr);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("block %3d lds %6d grid %5d: %7.2f us per launch\n", b, l, g, ms * 1000 / 20);
    }
    return 0;
}

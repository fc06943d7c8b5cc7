// mfma_f16_chain.hip — does a chain of v_mfma_f32_32x32x16_f16 that accumulate into the same registers lose updates when ANOTHER kernel's waves share the SIMD?
// (round 4, DESIGN.md 8.4 / profiles/r04i_attn_split_hazard.txt: the engine's split-fp16 kernels were bit-stable alone and not next to other clip groups.)
//
// Victim kernel: every wave runs ITER steps; per step it converts fresh fp32 values to fp16 operands with VALU instructions (as the engine does) and issues MFMAs in
// one of the issue orders below.  Operands are small integers, so every product and every partial sum is exact: the result of a wave is known in closed form and ANY
// deviation is a lost or doubled update.  Aggressor kernel (other stream, same CUs): fp32 MFMA loop / fp16 MFMA loop / global-memory stream / none.
//   order 0: one accumulator, MFMAs back to back                     acc, acc, acc, ...
//   order 1: two accumulators alternating (one other MFMA between)   a, b, a, b, ...
//   order 2: three accumulators in rotation (two between)            a, b, c, a, b, c, ...
//   order 3: one accumulator, 16 idle slots between the MFMAs
//   order 4: fgemm's FAILING variant: three column tiles per step, each: convert its B operand, fence, (cross_j, main_j, cross_j), fence — six accumulators
//   order 5: the same without the fences
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_chain scripts/ubench/mfma_f16_chain.hip && /tmp/mfma_f16_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 4096;

template <int ORDER>
__global__ __launch_bounds__(64) void victim(const float* __restrict__ src, float* __restrict__ out) {
    const int l = threadIdx.x;
    f32x16 a, b, c, d, e, f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a[r] = 0.f; b[r] = 0.f; c[r] = 0.f; d[r] = 0.f; e[r] = 0.f; f[r] = 0.f; }
    float x = src[l & 7];   // 1.0 (read from memory so that nothing folds)
    for (int it = 0; it < ITER; ++it) {
        // fresh operands every step, written by VALU conversions: A rows all (it & 3) + 1, B columns all 1  ->  every element gains 16 * ((it & 3) + 1)
        const float va = x * (float)((it & 3) + 1);
        f16x8 fa, fb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)va; fb[i] = (_Float16)x; }
        if (ORDER == 0) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a, 0, 0, 0);
        } else if (ORDER == 1) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, b, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a, 0, 0, 0);
        } else if (ORDER == 2) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, b, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c, 0, 0, 0);
        } else if (ORDER == 4 || ORDER == 5) {
            // one MFMA per accumulator and step in total (a, b, c main; the cross accumulators add into a / b / c at the end via d, e, f): 3 x 3 MFMAs
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f16x8 fj;
                const float vj = x * (float)(j + 1) * (1.0f / (float)(j + 1));   // fresh VALU work per column tile (value 1)
#pragma unroll
                for (int i = 0; i < 8; ++i) fj[i] = (_Float16)vj;
                if (ORDER == 4) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7\n\ts_nop 7"); __builtin_amdgcn_sched_barrier(0); }
                f32x16& xacc = j == 0 ? d : (j == 1 ? e : f);
                f32x16& macc = j == 0 ? a : (j == 1 ? b : c);
                xacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fj, xacc, 0, 0, 0);
                macc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fj, macc, 0, 0, 0);
                xacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fj, xacc, 0, 0, 0);
                if (ORDER == 4) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7\n\ts_nop 7"); __builtin_amdgcn_sched_barrier(0); }
            }
        } else {
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7\n\ts_nop 7");
            __builtin_amdgcn_sched_barrier(0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7\n\ts_nop 7");
            __builtin_amdgcn_sched_barrier(0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a, 0, 0, 0);
        }
        x = x * 1.0f + 0.0f * a[0];   // (keeps the loop-carried value alive without changing it: a[0] is finite)
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a[r] + b[r] + c[r] + d[r] + e[r] + f[r];
    out[blockIdx.x * 64 + l] = s;
}

// aggressors
__global__ __launch_bounds__(256) void agg_mfma_f32(float* out, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float u = 1.0f + threadIdx.x * 1e-9f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(u, u, acc[j], 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
__global__ __launch_bounds__(256) void agg_mfma_f16(float* out, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8 u;
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = (_Float16)(1.0f + threadIdx.x * 1e-4f);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u, u, acc[j], 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
__global__ __launch_bounds__(256) void agg_mem(const float4* __restrict__ in, float* out, size_t n4, int iters) {
    float s = 0.f;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const float4 v = in[(i + (size_t)it * 1048573u) % n4];
        s += v.x + v.y + v.z + v.w;
        // a short fp32 MFMA burst between loads, as a GEMM's k step would issue
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = s;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, v.y, acc, 0, 0, 0);
        s += acc[3] * 1e-30f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ORDER>
static long run_victim(hipStream_t s, const float* src, float* out, int wgs) {
    hipLaunchKernelGGL(victim<ORDER>, dim3(wgs), dim3(64), 0, s, src, out);
    return 0;
}

int main() {
    const int wgs = 256 * 4 * 4;   // four waves per SIMD if alone
    float *src, *out, *aout;
    float4* big;
    const size_t n4 = (size_t)64 << 20;   // 1 GiB
    CHECK(hipMalloc(&src, 64)); CHECK(hipMalloc(&out, (size_t)wgs * 64 * 4)); CHECK(hipMalloc(&aout, (size_t)4096 * 256 * 4)); CHECK(hipMalloc(&big, n4 * 16));
    const float ones[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    CHECK(hipMemcpy(src, ones, 32, hipMemcpyHostToDevice));
    CHECK(hipMemset(big, 0, n4 * 16));
    hipStream_t sv, sa;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    // expected per-lane sum: each step adds 16 * m per MFMA and element (m = (it & 3) + 1), three MFMAs per step, 16 registers summed
    double per_elem = 0;
    for (int it = 0; it < ITER; ++it) per_elem += 16.0 * ((it & 3) + 1) * 3;
    const float want3 = (float)(per_elem * 16);   // (exactly representable? 16 * 3 * 16 * 10240 = 7,864,320 < 2^24: yes)
    printf("expected lane sum %.1f (orders 4 / 5: three times that, summed per accumulator below 2^24, the lane total rounds identically in every wave)\n", want3);
    std::vector<float> h((size_t)wgs * 64);
    const char* agg_names[] = {"none", "fp32 MFMA loop", "fp16 MFMA loop", "memory stream + fp32 MFMA bursts"};
    const char* ord_names[] = {"one accumulator back to back", "two accumulators alternating", "three accumulators in rotation", "one accumulator, 16 idle slots between",
                               "per column tile: cvt, fence, x m x, fence", "per column tile: cvt, x m x (no fences)"};
    for (int agg = 0; agg < 4; ++agg)
        for (int ord = 0; ord < 6; ++ord) {
            const float want = ord >= 4 ? 3.0f * want3 : want3;
            long bad = 0, waves_bad = 0;
            double worst = 0;
            for (int rep = 0; rep < 6; ++rep) {
                if (agg == 1) hipLaunchKernelGGL(agg_mfma_f32, dim3(512), dim3(256), 0, sa, aout, 60000);
                if (agg == 2) hipLaunchKernelGGL(agg_mfma_f16, dim3(512), dim3(256), 0, sa, aout, 120000);
                if (agg == 3) hipLaunchKernelGGL(agg_mem, dim3(1024), dim3(256), 0, sa, big, aout, n4, 3000);
                if (ord == 0) run_victim<0>(sv, src, out, wgs);
                if (ord == 1) run_victim<1>(sv, src, out, wgs);
                if (ord == 2) run_victim<2>(sv, src, out, wgs);
                if (ord == 3) run_victim<3>(sv, src, out, wgs);
                if (ord == 4) run_victim<4>(sv, src, out, wgs);
                if (ord == 5) run_victim<5>(sv, src, out, wgs);
                CHECK(hipStreamSynchronize(sv));
                CHECK(hipStreamSynchronize(sa));
                CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
                for (int w = 0; w < wgs; ++w) {
                    bool wb = false;
                    for (int l = 0; l < 64; ++l) {
                        const float v = h[(size_t)w * 64 + l];
                        if (v != want) { ++bad; wb = true; const double d = (double)v - want; if ((d < 0 ? -d : d) > worst) worst = d < 0 ? -d : d; }
                    }
                    waves_bad += wb;
                }
            }
            printf("aggressor: %-34s order: %-40s wrong lanes %ld (waves %ld of %d), worst |diff| %.1f\n", agg_names[agg], ord_names[ord], bad, waves_bad, 6 * wgs, worst);
            fflush(stdout);
        }
    return 0;
}

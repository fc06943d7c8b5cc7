// mfma_f16_operand_hazard.hip — does v_mfma_f32_32x32x16_f16 read a 128-bit operand register that a VALU instruction wrote N wait states earlier (RAW), and is a
// VALU write M issue slots AFTER the MFMA too early (WAR)?  Fixed registers, hand-written issue order (round 4, DESIGN.md 8.4: the two adjacencies seen in the ISA of
// the first attention build that was not bit-stable).
//   A = v[32:35], B = v[36:39] (fp16 ones), accumulator v[48:63].  A correct MFMA adds 16 to every element; one that sees the poisoned value (2, 2) in v35 adds 20.
//   RAW<N>: v35 <- (2,2); idle; v35 <- (1,1) by v_cvt_pk_f16_f32; s_nop N-1 (N wait states; N = 0: none); MFMA
//   WAR<M>: MFMA; s_nop M-1; v35 <- (2,2); idle; v35 <- (1,1); idle
// Each beside: nothing / a memory-bound kernel on another stream (so that the wave issues back to back).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_operand_hazard scripts/ubench/mfma_f16_operand_hazard.hip && /tmp/mfma_f16_operand_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define STR2(x) #x
#define STR(x) STR2(x)
#define ITERS 2048

#define PROLOGUE                                                                                    \
    "v_mov_b32 v30, 1.0\n v_mov_b32 v31, 2.0\n"                                                     \
    "v_cvt_pk_f16_f32 v32, v30, v30\n v_cvt_pk_f16_f32 v33, v30, v30\n v_cvt_pk_f16_f32 v34, v30, v30\n v_cvt_pk_f16_f32 v35, v30, v30\n" \
    "v_cvt_pk_f16_f32 v36, v30, v30\n v_cvt_pk_f16_f32 v37, v30, v30\n v_cvt_pk_f16_f32 v38, v30, v30\n v_cvt_pk_f16_f32 v39, v30, v30\n" \
    "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n" \
    "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n" \
    "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_movk_i32 s20, " STR(ITERS) "\n"
#define IDLE "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n"
#define LOOP_END "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n v_mov_b32 %0, v48\n v_mov_b32 %1, v63\n"
#define CLOBBERS "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "s20", "scc"
#define MFMA "v_mfma_f32_32x32x16_f16 v[48:63], v[32:35], v[36:39], v[48:63]\n"

template <int N>
__global__ __launch_bounds__(64) void raw_kernel(float* out) {
    float r0, r1;
    if constexpr (N == 0)
        asm volatile(PROLOGUE "1:\n v_cvt_pk_f16_f32 v35, v31, v31\n" IDLE "v_cvt_pk_f16_f32 v35, v30, v30\n" MFMA IDLE LOOP_END : "=v"(r0), "=v"(r1) : : CLOBBERS);
    else if constexpr (N == 1)
        asm volatile(PROLOGUE "1:\n v_cvt_pk_f16_f32 v35, v31, v31\n" IDLE "v_cvt_pk_f16_f32 v35, v30, v30\n s_nop 0\n" MFMA IDLE LOOP_END : "=v"(r0), "=v"(r1) : : CLOBBERS);
    else if constexpr (N == 2)
        asm volatile(PROLOGUE "1:\n v_cvt_pk_f16_f32 v35, v31, v31\n" IDLE "v_cvt_pk_f16_f32 v35, v30, v30\n s_nop 1\n" MFMA IDLE LOOP_END : "=v"(r0), "=v"(r1) : : CLOBBERS);
    else
        asm volatile(PROLOGUE "1:\n v_cvt_pk_f16_f32 v35, v31, v31\n" IDLE "v_cvt_pk_f16_f32 v35, v30, v30\n s_nop 3\n" MFMA IDLE LOOP_END : "=v"(r0), "=v"(r1) : : CLOBBERS);
    out[(blockIdx.x * 64 + threadIdx.x) * 2] = r0;
    out[(blockIdx.x * 64 + threadIdx.x) * 2 + 1] = r1;
}
template <int M>
__global__ __launch_bounds__(64) void war_kernel(float* out) {
    float r0, r1;
    if constexpr (M == 0)
        asm volatile(PROLOGUE "1:\n" MFMA "v_cvt_pk_f16_f32 v35, v31, v31\n" IDLE "v_cvt_pk_f16_f32 v35, v30, v30\n" IDLE LOOP_END : "=v"(r0), "=v"(r1) : : CLOBBERS);
    else if constexpr (M == 1)
        asm volatile(PROLOGUE "1:\n" MFMA "s_nop 0\n v_cvt_pk_f16_f32 v35, v31, v31\n" IDLE "v_cvt_pk_f16_f32 v35, v30, v30\n" IDLE LOOP_END : "=v"(r0), "=v"(r1) : : CLOBBERS);
    else if constexpr (M == 2)
        asm volatile(PROLOGUE "1:\n" MFMA "s_nop 1\n v_cvt_pk_f16_f32 v35, v31, v31\n" IDLE "v_cvt_pk_f16_f32 v35, v30, v30\n" IDLE LOOP_END : "=v"(r0), "=v"(r1) : : CLOBBERS);
    else
        asm volatile(PROLOGUE "1:\n" MFMA "s_nop 7\n v_cvt_pk_f16_f32 v35, v31, v31\n" IDLE "v_cvt_pk_f16_f32 v35, v30, v30\n" IDLE LOOP_END : "=v"(r0), "=v"(r1) : : CLOBBERS);
    out[(blockIdx.x * 64 + threadIdx.x) * 2] = r0;
    out[(blockIdx.x * 64 + threadIdx.x) * 2 + 1] = r1;
}
__global__ __launch_bounds__(256) void agg_mem(const float4* __restrict__ in, float* out, size_t n4, int iters) {
    float s = 0.f;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const float4 v = in[(i + (size_t)it * 1048573u) % n4];
        s += v.x + v.y + v.z + v.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int wgs = 256 * 4 * 2;
    float *out, *aout;
    float4* big;
    const size_t n4 = (size_t)64 << 20;
    CHECK(hipMalloc(&out, (size_t)wgs * 64 * 2 * 4)); CHECK(hipMalloc(&aout, (size_t)2048 * 256 * 4)); CHECK(hipMalloc(&big, n4 * 16));
    CHECK(hipMemset(big, 0, n4 * 16));
    hipStream_t sv, sa;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    std::vector<float> h((size_t)wgs * 128);
    const float want = 16.0f * ITERS;
    printf("a correct run leaves %.0f in every accumulator element; every MFMA that saw the poisoned register adds 4 more\n", want);
    for (int agg = 0; agg < 2; ++agg)
        for (int t = 0; t < 8; ++t) {
            long bad = 0; double worst = 0;
            for (int rep = 0; rep < 4; ++rep) {
                if (agg) hipLaunchKernelGGL(agg_mem, dim3(2048), dim3(256), 0, sa, big, aout, n4, 2000);
                switch (t) {
                    case 0: hipLaunchKernelGGL(raw_kernel<0>, dim3(wgs), dim3(64), 0, sv, out); break;
                    case 1: hipLaunchKernelGGL(raw_kernel<1>, dim3(wgs), dim3(64), 0, sv, out); break;
                    case 2: hipLaunchKernelGGL(raw_kernel<2>, dim3(wgs), dim3(64), 0, sv, out); break;
                    case 3: hipLaunchKernelGGL(raw_kernel<4>, dim3(wgs), dim3(64), 0, sv, out); break;
                    case 4: hipLaunchKernelGGL(war_kernel<0>, dim3(wgs), dim3(64), 0, sv, out); break;
                    case 5: hipLaunchKernelGGL(war_kernel<1>, dim3(wgs), dim3(64), 0, sv, out); break;
                    case 6: hipLaunchKernelGGL(war_kernel<2>, dim3(wgs), dim3(64), 0, sv, out); break;
                    default: hipLaunchKernelGGL(war_kernel<8>, dim3(wgs), dim3(64), 0, sv, out); break;
                }
                CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sa));
                CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < h.size(); ++i)
                    if (h[i] != want) { ++bad; const double d = (double)h[i] - want; if (d > worst) worst = d; if (-d > worst) worst = -d; }
            }
            static const char* names[] = {"RAW, 0 wait states", "RAW, 1 wait state", "RAW, 2 wait states (the compiler's choice)", "RAW, 4 wait states",
                                          "WAR, write in the next slot", "WAR, 1 slot between", "WAR, 2 slots between", "WAR, 8 slots between"};
            printf("%-28s %-44s wrong values %ld of %zu, worst |diff| %.0f (= %.0f poisoned MFMAs of %d)\n", agg ? "beside a memory-bound kernel" : "alone", names[t], bad, 4 * h.size(), worst, worst / 4, ITERS);
            fflush(stdout);
        }
    return 0;
}

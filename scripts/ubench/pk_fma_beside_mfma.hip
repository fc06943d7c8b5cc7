// pk_fma_beside_mfma.hip — round 5: the mechanism behind round 4's "split-fp16 kernels are not bit-stable next to other streams".
//
// scripts/race_localise.py localised the deviations: NOT in the split-fp16 kernels, but in a DIFFERENT kernel running beside them on the same CUs — the fp32
// GEMM with the GroupNorm'ed residual (tgemm_dev.h, tg_epilogue phase 2b), whose epilogue hipcc compiles to
//     global_load_dwordx2 v[8:9], v[8:9], off ; s_waitcnt vmcnt(0)
//     v_pk_fma_f32 v[34:35], v[4:5], v[8:9], v[8:9] op_sel:[0,0,1] op_sel_hi:[1,0,1]      ; x.lo * a + b , x.hi * a + b
//     v_pk_fma_f32 v[8:9],  v[6:7], v[8:9], v[8:9] op_sel:[0,0,1] op_sel_hi:[1,0,1]
// and whose wrong words are exactly: the LOW half of one of the two packed FMAs, lanes 48-63, with the addend b read as 0 (offset == -b to the last bit).
// This micro-benchmark replays that: victim waves run packed / scalar fp32 FMAs on exact small-integer data and count wrong results per (instruction, half, lane
// quarter); aggressor waves on another stream (same CUs) issue MFMAs densely or with idle slots (s_nop) between them, as the split kernels' operand fences did.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_fma_beside_mfma scripts/ubench/pk_fma_beside_mfma.hip && /tmp/pk_fma_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NCH = 192;
// counters: [form][instr 0/1][half lo/hi][lane quarter][kind: 0 any mismatch, 1 == x*a (addend read as 0), 2 == x*a + a (low register instead of high)]
struct Counters { unsigned c[8][2][2][4][3]; };

// FORM 0: the observed sequence (load into the address registers, wait, two packed FMAs with op_sel picking the high register as the addend)
// FORM 1: the same packed FMAs, (a, b) produced by VALU moves instead of a load
// FORM 2: packed FMAs without op_sel tricks: (a, a) and (b, b) in their own register pairs
// FORM 3: four scalar v_fma_f32 (control)
// FORM 4: v_pk_mul_f32 then v_pk_add_f32, op_sel as in FORM 0
template <int FORM>
__global__ __launch_bounds__(64) void victim(const float* __restrict__ coef, Counters* cnt, int iters) {
    const int l = threadIdx.x;
    unsigned bad[2][2][3] = {};
    for (int it = 0; it < iters; ++it) {
        const int ch = (it * 7 + (l >> 3) + blockIdx.x) % NCH;
        const int ai = (ch % 5) + 1, bi = (ch % 11) - 5 + 16;                 // a in 1..5, b in 11..21 (never 0, never == a)
        const int x0 = (l & 7) + (it & 3), x1 = x0 + 3, x2 = x0 + 9, x3 = x0 + 17;
        const f32x2 x01 = {(float)x0, (float)x1}, x23 = {(float)x2, (float)x3};
        f32x2 r0, r1;
        if constexpr (FORM == 0) {
            unsigned long long c = (unsigned long long)(coef + 2 * ch);
            asm volatile("global_load_dwordx2 %[c], %[c], off\n\t"
                         "s_waitcnt vmcnt(0)\n\t"
                         "v_pk_fma_f32 %[r0], %[x01], %[c], %[c] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
                         "v_pk_fma_f32 %[r1], %[x23], %[c], %[c] op_sel:[0,0,1] op_sel_hi:[1,0,1]"
                         : [r0] "=&v"(r0), [r1] "=&v"(r1), [c] "+v"(c) : [x01] "v"(x01), [x23] "v"(x23) : "memory");
        } else if constexpr (FORM == 1) {
            f32x2 c = {(float)ai, (float)bi};
            asm volatile("v_pk_fma_f32 %[r0], %[x01], %[c], %[c] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
                         "v_pk_fma_f32 %[r1], %[x23], %[c], %[c] op_sel:[0,0,1] op_sel_hi:[1,0,1]"
                         : [r0] "=&v"(r0), [r1] "=&v"(r1) : [c] "v"(c), [x01] "v"(x01), [x23] "v"(x23));
        } else if constexpr (FORM == 2) {
            f32x2 ca = {(float)ai, (float)ai}, cb = {(float)bi, (float)bi};
            asm volatile("v_pk_fma_f32 %[r0], %[x01], %[ca], %[cb]\n\t"
                         "v_pk_fma_f32 %[r1], %[x23], %[ca], %[cb]"
                         : [r0] "=&v"(r0), [r1] "=&v"(r1) : [ca] "v"(ca), [cb] "v"(cb), [x01] "v"(x01), [x23] "v"(x23));
        } else if constexpr (FORM == 3) {
            const float a = (float)ai, b = (float)bi;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0[0]) : "v"(x01[0]), "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0[1]) : "v"(x01[1]), "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1[0]) : "v"(x23[0]), "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r1[1]) : "v"(x23[1]), "v"(a), "v"(b));
        } else {
            f32x2 c = {(float)ai, (float)bi}, t0, t1;
            asm volatile("v_pk_mul_f32 %[t0], %[x01], %[c] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
                         "v_pk_mul_f32 %[t1], %[x23], %[c] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
                         "v_pk_add_f32 %[r0], %[t0], %[c] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                         "v_pk_add_f32 %[r1], %[t1], %[c] op_sel:[0,1] op_sel_hi:[1,1]"
                         : [r0] "=&v"(r0), [r1] "=&v"(r1), [t0] "=&v"(t0), [t1] "=&v"(t1) : [c] "v"(c), [x01] "v"(x01), [x23] "v"(x23));
        }
        const int xs[4] = {x0, x1, x2, x3};
        const float rs[4] = {r0[0], r0[1], r1[0], r1[1]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float want = (float)(xs[k] * ai + bi);
            if (rs[k] != want) {
                ++bad[k >> 1][k & 1][0];
                if (rs[k] == (float)(xs[k] * ai)) ++bad[k >> 1][k & 1][1];
                if (rs[k] == (float)(xs[k] * ai + ai)) ++bad[k >> 1][k & 1][2];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (bad[i][h][k]) atomicAdd(&cnt->c[FORM][i][h][l >> 4][k], bad[i][h][k]);
}

// aggressors: PAT 0 none; 1 fp16 MFMAs back to back; 2 fp16 MFMA, 16 idle slots, ...; 3 fp32 MFMAs back to back; 4 fp32 MFMA + idle slots; 5 bf16 MFMA + idle slots;
// 6 idle slots only (no MFMA); 7 fp16 MFMA pair (cross, main) + idle + cross + idle: the order-2 attention build
template <int PAT>
__global__ __launch_bounds__(64) void aggressor(float* out, int iters) {
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8 u; bf16x8 ub;
#pragma unroll
    for (int i = 0; i < 8; ++i) { u[i] = (_Float16)(1.0f + threadIdx.x * 1e-3f); ub[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); }
    const float uf = 1.0f + threadIdx.x * 1e-9f;
#define IDLE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7\n\ts_nop 7"); __builtin_amdgcn_sched_barrier(0); } while (0)
    for (int it = 0; it < iters; ++it) {
        if (PAT == 1) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u, u, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u, u, acc[1], 0, 0, 0); }
        if (PAT == 2) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u, u, acc[0], 0, 0, 0); IDLE(); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u, u, acc[1], 0, 0, 0); IDLE(); }
        if (PAT == 3) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf, uf, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf, uf, acc[1], 0, 0, 0); }
        if (PAT == 4) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf, uf, acc[0], 0, 0, 0); IDLE(); acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf, uf, acc[1], 0, 0, 0); IDLE(); }
        if (PAT == 5) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ub, ub, acc[0], 0, 0, 0); IDLE(); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ub, ub, acc[1], 0, 0, 0); IDLE(); }
        if (PAT == 6) { IDLE(); IDLE(); }
        if (PAT == 7) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u, u, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u, u, acc[1], 0, 0, 0);
            IDLE();
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u, u, acc[0], 0, 0, 0);
            IDLE();
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc[0][0] + acc[1][1];
}

// gap sweep: MFMA of TYPE (0 fp16 32x32x16, 1 bf16 32x32x16, 2 fp32 32x32x2, 3 fp32 16x16x4 (8 passes), 4 fp16 16x16x32) / GAP idle slots (s_nop) or GAP plain VALU instructions (FILL) / MFMA ...
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int TYPE, int GAP, bool FILL>
__global__ __launch_bounds__(64) void aggressor_gap(float* out, int iters) {
    f32x16 acc[2];
    f32x4 acc4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[j][r] = 0.f;
    }
    f16x8 u; bf16x8 ub;
#pragma unroll
    for (int i = 0; i < 8; ++i) { u[i] = (_Float16)(1.0f + threadIdx.x * 1e-3f); ub[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); }
    const float uf = 1.0f + threadIdx.x * 1e-9f;
    float fill = uf;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (TYPE == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u, u, acc[j], 0, 0, 0);
            if (TYPE == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ub, ub, acc[j], 0, 0, 0);
            if (TYPE == 2) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf, uf, acc[j], 0, 0, 0);
            if (TYPE == 3) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf, uf, acc4[j], 0, 0, 0);
            if (TYPE == 4) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, u, acc4[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (FILL) {
#pragma unroll
                for (int g = 0; g < GAP; ++g) asm volatile("v_add_f32 %0, %0, %1" : "+v"(fill) : "v"(uf));
            } else {
#pragma unroll
                for (int g = 0; g < GAP; ++g) asm volatile("s_nop 0");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc4[0][0] + acc4[1][1] + fill;
}
template <int TYPE, int GAP, bool FILL> static void run_gap(hipStream_t s, float* out, int wgs, int iters) { hipLaunchKernelGGL((aggressor_gap<TYPE, GAP, FILL>), dim3(wgs), dim3(64), 0, s, out, iters); }

// more victim instruction classes beside the worst aggressor: which operand-select forms are affected?
// VF 0: pk_fma src2 lo<-hi, hi<-hi (observed)   1: src2 lo<-hi, hi<-lo (swap)   2: src2 lo<-lo, hi<-lo (broadcast low)   3: src0 lo<-hi   4: src1 lo<-hi
// VF 5: v_fma_f64 on exact integers   6: v_pk_mov_b32 lo<-hi   7: v_lshl_add_u64 / v_mad_u64_u32   8: v_pk_add_f32 src1 lo<-hi   9: v_pk_fma_f16 lo<-hi (packed halves of ONE register)
struct Counters2 { unsigned c[12][4][2]; };   // [form][lane quarter][lo / hi]
template <int VF>
__global__ __launch_bounds__(64) void victim2(Counters2* cnt, int iters) {
    const int l = threadIdx.x;
    unsigned bad[2] = {0, 0};
    for (int it = 0; it < iters; ++it) {
        const int ai = ((it + l) % 5) + 1, bi = ((it * 3 + l) % 11) + 11, x0 = (l & 7) + (it & 3) + 1, x1 = x0 + 3;
        f32x2 x = {(float)x0, (float)x1}, c = {(float)ai, (float)bi}, r = {0.f, 0.f};
        float w0 = 0.f, w1 = 0.f;
        if constexpr (VF == 0) { asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=&v"(r) : "v"(x), "v"(c)); w0 = x0 * ai + bi; w1 = x1 * ai + bi; }
        if constexpr (VF == 1) { asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,0]" : "=&v"(r) : "v"(x), "v"(c)); w0 = x0 * ai + bi; w1 = x1 * ai + ai; }
        if constexpr (VF == 2) { asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=&v"(r) : "v"(x), "v"(c)); w0 = x0 * bi + ai; w1 = x1 * bi + ai; }
        if constexpr (VF == 3) { asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(r) : "v"(x), "v"(c)); w0 = x1 * ai + ai; w1 = x1 * ai + ai; }
        if constexpr (VF == 4) { asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=&v"(r) : "v"(x), "v"(c)); w0 = x0 * bi + x0; w1 = x1 * bi + x1; }
        if constexpr (VF == 5) {
            double dx = (double)x0, da = (double)ai, db = (double)bi, dr;
            asm volatile("v_fma_f64 %0, %1, %2, %3" : "=&v"(dr) : "v"(dx), "v"(da), "v"(db));
            r[0] = (float)dr; r[1] = (float)(x1 * ai + bi); w0 = x0 * ai + bi; w1 = x1 * ai + bi;
        }
        if constexpr (VF == 6) { asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=&v"(r) : "v"(c)); w0 = bi; w1 = ai; }
        if constexpr (VF == 7) {
            unsigned long long a64 = ((unsigned long long)bi << 32) | (unsigned)ai, b64 = ((unsigned long long)x1 << 32) | (unsigned)x0, r64;
            asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=&v"(r64) : "v"(a64), "v"(b64));
            r[0] = (float)(unsigned)(r64 & 0xffffffffu); r[1] = (float)(unsigned)(r64 >> 32); w0 = ai + x0; w1 = bi + x1;
        }
        if constexpr (VF == 8) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=&v"(r) : "v"(x), "v"(c)); w0 = x0 + bi; w1 = x1 + bi; }
        if constexpr (VF == 9) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 hx = {(_Float16)x0, (_Float16)x1}, hc = {(_Float16)ai, (_Float16)bi}, hr;
            asm volatile("v_pk_fma_f16 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=&v"(hr) : "v"(hx), "v"(hc));
            r[0] = (float)hr[0]; r[1] = (float)hr[1]; w0 = x0 * ai + bi; w1 = x1 * ai + bi;
        }
        if (r[0] != w0) ++bad[0];
        if (r[1] != w1) ++bad[1];
    }
    if (bad[0]) atomicAdd(&cnt->c[VF][l >> 4][0], bad[0]);
    if (bad[1]) atomicAdd(&cnt->c[VF][l >> 4][1], bad[1]);
}
template <int VF> static void run_vic2(hipStream_t s, Counters2* c, int wgs, int iters) { hipLaunchKernelGGL(victim2<VF>, dim3(wgs), dim3(64), 0, s, c, iters); }

template <int PAT> static void run_agg(hipStream_t s, float* out, int wgs, int iters) { hipLaunchKernelGGL(aggressor<PAT>, dim3(wgs), dim3(64), 0, s, out, iters); }
template <int FORM> static void run_vic(hipStream_t s, const float* coef, Counters* c, int wgs, int iters) { hipLaunchKernelGGL(victim<FORM>, dim3(wgs), dim3(64), 0, s, coef, c, iters); }

int main() {
    const int wgs = 256 * 4 * 2;   // two victim waves per SIMD; as many aggressor waves
    float *coef, *aout;
    Counters* cnt;
    CHECK(hipMalloc(&coef, NCH * 2 * 4)); CHECK(hipMalloc(&aout, (size_t)wgs * 64 * 4)); CHECK(hipMalloc(&cnt, sizeof(Counters)));
    std::vector<float> hc(NCH * 2);
    for (int ch = 0; ch < NCH; ++ch) { hc[2 * ch] = (float)((ch % 5) + 1); hc[2 * ch + 1] = (float)((ch % 11) - 5 + 16); }
    CHECK(hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sa;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    const char* pat_names[] = {"none", "fp16 MFMAs back to back", "fp16 MFMA / 16 idle slots", "fp32 MFMAs back to back", "fp32 MFMA / 16 idle slots", "bf16 MFMA / 16 idle slots",
                               "idle slots only", "fp16 (x, m) / idle / x / idle"};
    const char* form_names[] = {"load + pk_fma op_sel (observed)", "pk_fma op_sel, no load", "pk_fma plain", "scalar v_fma_f32", "pk_mul + pk_add op_sel"};
    const int vit = 40000;
    for (int pat = 0; pat < 8; ++pat) {
        for (int form = 0; form < 5; ++form) {
            CHECK(hipMemset(cnt, 0, sizeof(Counters)));
            for (int rep = 0; rep < 4; ++rep) {
                const int ait = 400000;
                switch (pat) {
                    case 1: run_agg<1>(sa, aout, wgs, ait); break; case 2: run_agg<2>(sa, aout, wgs, ait / 3); break; case 3: run_agg<3>(sa, aout, wgs, ait / 2); break;
                    case 4: run_agg<4>(sa, aout, wgs, ait / 4); break; case 5: run_agg<5>(sa, aout, wgs, ait / 3); break; case 6: run_agg<6>(sa, aout, wgs, ait / 2); break;
                    case 7: run_agg<7>(sa, aout, wgs, ait / 4); break; default: break;
                }
                switch (form) {
                    case 0: run_vic<0>(sv, coef, cnt, wgs, vit); break; case 1: run_vic<1>(sv, coef, cnt, wgs, vit); break; case 2: run_vic<2>(sv, coef, cnt, wgs, vit); break;
                    case 3: run_vic<3>(sv, coef, cnt, wgs, vit); break; default: run_vic<4>(sv, coef, cnt, wgs, vit); break;
                }
                CHECK(hipStreamSynchronize(sv));
                CHECK(hipStreamSynchronize(sa));
            }
            Counters h;
            CHECK(hipMemcpy(&h, cnt, sizeof h, hipMemcpyDeviceToHost));
            unsigned long tot = 0;
            for (int i = 0; i < 2; ++i) for (int hf = 0; hf < 2; ++hf) for (int q = 0; q < 4; ++q) tot += h.c[form][i][hf][q][0];
            printf("aggressor %-30s victim %-34s wrong results %lu of %.2e", pat_names[pat], form_names[form], tot, 4.0 * wgs * 64.0 * vit * 4);
            if (tot) {
                printf("  [instr.half: quarter counts (==x*a / ==x*a+a)]");
                for (int i = 0; i < 2; ++i) for (int hf = 0; hf < 2; ++hf) {
                    unsigned long t2 = 0; for (int q = 0; q < 4; ++q) t2 += h.c[form][i][hf][q][0];
                    if (!t2) continue;
                    printf("  %d.%s:", i, hf ? "hi" : "lo");
                    for (int q = 0; q < 4; ++q) printf(" %u(%u/%u)", h.c[form][i][hf][q][0], h.c[form][i][hf][q][1], h.c[form][i][hf][q][2]);
                }
            }
            printf("\n");
            fflush(stdout);
        }
    }
    // ---- gap sweep beside the observed victim (FORM 1: pk_fma op_sel, no load)
    printf("\ngap sweep: aggressor = MFMA / GAP slots / MFMA ..., victim = pk_fma op_sel (no load); wrong results of %.2e\n", 2.0 * wgs * 64.0 * vit * 4);
    auto sweep = [&](const char* name, auto launch) {
        CHECK(hipMemset(cnt, 0, sizeof(Counters)));
        for (int rep = 0; rep < 2; ++rep) { launch(); run_vic<1>(sv, coef, cnt, wgs, vit); CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sa)); }
        Counters h; CHECK(hipMemcpy(&h, cnt, sizeof h, hipMemcpyDeviceToHost));
        unsigned long tot = 0, q3lo = 0;
        for (int i = 0; i < 2; ++i) for (int hf = 0; hf < 2; ++hf) for (int q = 0; q < 4; ++q) { tot += h.c[1][i][hf][q][0]; if (q == 3 && hf == 0) q3lo += h.c[1][i][hf][q][0]; }
        printf("  %-52s wrong %lu (low half, lanes 48-63: %lu)\n", name, tot, q3lo); fflush(stdout);
    };
    const int git = 150000;
#define SW(T, G, F, NAME) sweep(NAME, [&] { run_gap<T, G, F>(sa, aout, wgs, git); })
    SW(0, 0, false, "fp16 32x32x16, gap 0"); SW(0, 1, false, "fp16 32x32x16, 1 idle slot"); SW(0, 2, false, "fp16 32x32x16, 2 idle slots"); SW(0, 4, false, "fp16 32x32x16, 4 idle slots");
    SW(0, 8, false, "fp16 32x32x16, 8 idle slots"); SW(0, 16, false, "fp16 32x32x16, 16 idle slots"); SW(0, 32, false, "fp16 32x32x16, 32 idle slots"); SW(0, 64, false, "fp16 32x32x16, 64 idle slots");
    SW(0, 4, true, "fp16 32x32x16, 4 VALU instructions"); SW(0, 16, true, "fp16 32x32x16, 16 VALU instructions"); SW(0, 64, true, "fp16 32x32x16, 64 VALU instructions");
    SW(1, 0, false, "bf16 32x32x16, gap 0"); SW(1, 16, false, "bf16 32x32x16, 16 idle slots"); SW(1, 16, true, "bf16 32x32x16, 16 VALU instructions");
    SW(2, 16, false, "fp32 32x32x2, 16 idle slots"); SW(2, 64, false, "fp32 32x32x2, 64 idle slots"); SW(2, 16, true, "fp32 32x32x2, 16 VALU instructions");
    SW(3, 0, false, "fp32 16x16x4, gap 0"); SW(3, 16, false, "fp32 16x16x4, 16 idle slots"); SW(3, 16, true, "fp32 16x16x4, 16 VALU instructions");
    SW(4, 0, false, "fp16 16x16x32, gap 0"); SW(4, 16, false, "fp16 16x16x32, 16 idle slots"); SW(4, 16, true, "fp16 16x16x32, 16 VALU instructions");
    // ---- victim instruction classes beside "fp16 32x32x16, 16 idle slots"
    Counters2* cnt2; CHECK(hipMalloc(&cnt2, sizeof(Counters2))); CHECK(hipMemset(cnt2, 0, sizeof(Counters2)));
    const char* vf_names[] = {"pk_fma_f32 src2 lo<-hi hi<-hi", "pk_fma_f32 src2 lo<-hi hi<-lo", "pk_fma_f32 src1 hi<-hi lo<-hi? (both b) src2 lo", "pk_fma_f32 src0 lo<-hi", "pk_fma_f32 src1 lo<-hi",
                              "v_fma_f64", "v_pk_mov_b32 lo<-hi", "v_lshl_add_u64", "v_pk_add_f32 src1 lo<-hi", "v_pk_fma_f16 lo<-hi"};
    printf("\nvictim instruction classes beside fp16 32x32x16 MFMAs with 16 idle slots between them (wrong results per lane quarter, low / high half):\n");
    for (int vf = 0; vf < 10; ++vf) {
        for (int rep = 0; rep < 2; ++rep) {
            run_gap<0, 16, false>(sa, aout, wgs, git);
            switch (vf) { case 0: run_vic2<0>(sv, cnt2, wgs, vit); break; case 1: run_vic2<1>(sv, cnt2, wgs, vit); break; case 2: run_vic2<2>(sv, cnt2, wgs, vit); break; case 3: run_vic2<3>(sv, cnt2, wgs, vit); break;
                          case 4: run_vic2<4>(sv, cnt2, wgs, vit); break; case 5: run_vic2<5>(sv, cnt2, wgs, vit); break; case 6: run_vic2<6>(sv, cnt2, wgs, vit); break; case 7: run_vic2<7>(sv, cnt2, wgs, vit); break;
                          case 8: run_vic2<8>(sv, cnt2, wgs, vit); break; default: run_vic2<9>(sv, cnt2, wgs, vit); break; }
            CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sa));
        }
        Counters2 h2; CHECK(hipMemcpy(&h2, cnt2, sizeof h2, hipMemcpyDeviceToHost));
        printf("  %-50s", vf_names[vf]);
        for (int q = 0; q < 4; ++q) printf("  q%d %u / %u", q, h2.c[vf][q][0], h2.c[vf][q][1]);
        printf("\n"); fflush(stdout);
    }
    return 0;
}

// split_f16.h — fp32-accurate matrix products on the fp16 matrix pipe (round 4).
#pragma once
#include <hip/hip_runtime.h>

namespace said {

typedef float f32x4s __attribute__((ext_vector_type(4)));
// Split-fp16 operands (attn_kernel PM == 2, fgemm_kernel SP): x ~= h + 2^-11 l with h = RN16(x), l = RN16((x - h) * 2^11) — the remainder is stored at h's own
// magnitude, so both halves are normal fp16 numbers for 2^-14 <= |x| < 65504 and x is represented to 2^-22 relative (22 significand bits + the
// remainder's sign; below 2^-14 the pair still resolves 2^-36 absolute where the matrix pipe keeps fp16 denormals) — and
//   a . b ~= h_a . h_b + 2^-11 (h_a . l_b + l_a . h_b)
// on v_mfma_f32_32x32x16_f16: three 8-pass MFMAs per 16 contraction steps against eight 16-pass v_mfma_f32_32x32x2_f32 (5.3 x fewer
// matrix-pipe clocks).  Accumulation is fp32 as before; the cross terms have their own accumulator (merged with one fma per element).  The
// dropped term 2^-22 l_a . l_b is below the representation error.  Domain: |x| < 65504 (q, k, v are projections of LayerNorm'ed rows; an
// overflow shows as inf / NaN).
typedef _Float16 f16x8a __attribute__((ext_vector_type(8)));
struct SplitH { f16x8a h, l; };
static __device__ __forceinline__ SplitH split_f16x8(const f32x4s a, const f32x4s b) {
    SplitH r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 ha = (_Float16)a[i], hb = (_Float16)b[i];
        r.h[i] = ha;
        r.h[4 + i] = hb;
        r.l[i] = (_Float16)((a[i] - (float)ha) * 2048.f);      // (a - ha is exact in fp32)
        r.l[4 + i] = (_Float16)((b[i] - (float)hb) * 2048.f);
    }
    return r;
}
// TWO RULES for v_mfma_f32_32x32x16_f16 in this engine, both found the hard way (profiles/r04i_attn_split_hazard.txt: runs that were bit-stable alone and
// differed by up to 5e-2 in a few clips, silently, as soon as waves of this engine's OTHER kernels shared the SIMD):
//  1. ROTATION: two MFMAs that accumulate into the same registers have at least two other MFMAs of the wave between them and no idle slots — an accumulate
//     never needs the result of an MFMA still in flight.  (Hence three accumulators per product: main, l.h, h.l.)  Back to back, one apart, or with idle slots
//     between them: not bit-stable under concurrent clip groups; the compiler's wait states do not cover it.
//  2. OPERAND FENCE: all split operands of a tile are computed first, then the scheduler is fenced and the wave idles SAID_SP_FENCE_NOPS issue slots, then the
//     MFMAs go out together — the machine scheduler otherwise interleaves the next operand's conversions with the MFMAs (a v_cvt_pk_f16_f32 writing a register
//     of a 128-bit operand two wait states before the MFMA that reads it, or rewriting one two issue slots after it), which breaks rule 1's "no other work
//     between the MFMAs" and was the first thing seen failing.  Free: the MFMAs hide behind the other waves' VALU work anyway.
#ifndef SAID_SP_FENCE_NOPS
#define SAID_SP_FENCE_NOPS 16
#endif
static __device__ __forceinline__ void operand_fence() {
    __builtin_amdgcn_sched_barrier(0);
#if SAID_SP_FENCE_NOPS >= 16
    asm volatile("s_nop 7\n\ts_nop 7");
#elif SAID_SP_FENCE_NOPS >= 8
    asm volatile("s_nop 7");
#elif SAID_SP_FENCE_NOPS >= 4
    asm volatile("s_nop 3");
#endif
    __builtin_amdgcn_sched_barrier(0);
}

}  // namespace said

// sched_math.h — the DDIM update as device functions shared by the stand-alone scheduler kernels (misc.hip) and the
// fused output-conv + scheduler kernel (out_sched.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace said {

// ------------------------------------------------------------------------------------------
// Scheduler arithmetic.  Every operation is an explicitly rounded fp32 op in diffusers'
// DDIMScheduler.step order (no FMA contraction), so given the same eps the result is
// bit-identical to the CPU restatement (oracle/scheduler.py).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float cfg_combine(float e_c, float e_u, float s) {
    // diffusion.py:430-434: noise_pred_audio + guidance_scale * (noise_pred_audio - noise_pred_uncond)
    return __fadd_rn(e_c, __fmul_rn(s, __fsub_rn(e_c, e_u)));
}
__device__ __forceinline__ float ddim_prev(float model_out, float x, const float* cf, int pred) {
    const float sa = cf[0], sb = cf[1], sap = cf[2], dir = cf[3];
    float x0, e;
    if (pred == 0) {
        x0 = __fdiv_rn(__fsub_rn(x, __fmul_rn(sb, model_out)), sa);
        e = model_out;
    } else if (pred == 1) {
        x0 = model_out;
        e = __fdiv_rn(__fsub_rn(x, __fmul_rn(sa, x0)), sb);
    } else {
        x0 = __fsub_rn(__fmul_rn(sa, x), __fmul_rn(sb, model_out));
        e = __fadd_rn(__fmul_rn(sa, model_out), __fmul_rn(sb, x));
    }
    x0 = (x0 != x0) ? x0 : fminf(fmaxf(x0, -1.0f), 1.0f);  // clip_sample=True, range 1.0 (torch.clamp keeps a NaN; fminf / fmaxf alone would turn it into -1)
    return __fadd_rn(__fmul_rn(sap, x0), __fmul_rn(dir, e));
}
// The model output of a step is not finite: remember the first such step (said_numeric_status).  fp32 mode multiplies on split-fp16 operands
// (split_f16.h: |x| < 65504); an operand beyond that becomes inf / NaN in its product and reaches this point through every later layer.
__device__ __forceinline__ void note_nonfinite(float model_out, int* status, int step) {
    if (status && !(fabsf(model_out) <= 3.4028234663852886e38f)) atomicCAS(status, 0, step + 1);
}
__device__ __forceinline__ float mask_blend(float prev, float init, float enoise, float mask, const float* cf) {
    // diffusion.py:446-456: add_noise(init, noise, t_next) * mask + latents * (1 - mask)
    const float noisy = __fadd_rn(__fmul_rn(cf[5], init), __fmul_rn(cf[6], enoise));
    return __fadd_rn(__fmul_rn(noisy, mask), __fmul_rn(prev, __fsub_rn(1.0f, mask)));
}


// ------------------------------------------------------------------------------------------
// eta > 0 variance noise generated on the device (said_loop_params::use_step_noise == 2): the reference draws
// `randn(model_output.shape)` inside DDIMScheduler.step once per step (diffusion.py:441-443) — a stream no other
// implementation reproduces bit for bit — so the product path draws its own standard normals, counter-based:
// Philox4x32-10 (Salmon et al., SC'11) with key = the call's 64-bit seed and counter = (element index in (B, T, C)
// order, step, 0, 0); the first two output words make one Box-Muller normal.  Layout- and launch-shape-independent,
// nothing is stored: no (N, B, T, C) buffer exists.  said_philox_normal fills a tensor with exactly these values.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float philox_normal(unsigned seed_lo, unsigned seed_hi, unsigned step, unsigned elem) {
    unsigned r[4];
    philox4x32_10(elem, step, 0u, 0u, seed_lo, seed_hi, r);
    const float u1 = ((float)(r[0] >> 8) + 1.0f) * 5.9604644775390625e-8f;    // (0, 1], 24 bits
    const float u2 = (float)(r[1] >> 8) * 5.9604644775390625e-8f;             // [0, 1)
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

}  // namespace said

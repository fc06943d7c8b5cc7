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
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);  // clip_sample=True, range 1.0
    return __fadd_rn(__fmul_rn(sap, x0), __fmul_rn(dir, e));
}
__device__ __forceinline__ float mask_blend(float prev, float init, float enoise, float mask, const float* cf) {
    // diffusion.py:446-456: add_noise(init, noise, t_next) * mask + latents * (1 - mask)
    const float noisy = __fadd_rn(__fmul_rn(cf[5], init), __fmul_rn(cf[6], enoise));
    return __fadd_rn(__fmul_rn(noisy, mask), __fmul_rn(prev, __fsub_rn(1.0f, mask)));
}

}  // namespace said

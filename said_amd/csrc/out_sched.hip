// out_sched.hip — the last kernel of a denoise step: `out` of the UNet (GroupNorm -> SiLU -> Conv1d(192 -> 32, k=3),
// openaimodel.py:652-656, 709) fused with classifier-free guidance (diffusion.py:430-434), the DDIM update
// (DDIMScheduler.step as called at diffusion.py:441-443), the eta noise and the editing mask blend
// (diffusion.py:446-456).  The model output never reaches HBM: a workgroup computes the unconditional AND the
// conditional 32-channel x 32-token tile of one clip (two accumulators sharing every weight fragment), combines them
// and updates the latents in place.
//
// Structure = gemm_lds.hip's 3-tap GroupNorm variant: 8 waves split the 192 input channels, each stages its
// 24-channel x 32-token slice (per half) through a wave-private LDS tile after applying GroupNorm+SiLU once, the
// main loop is ds_read + v_mfma_f32_32x32x2_f32, the split-K partials are summed through LDS in a fixed order.
// The scheduler arithmetic is sched_math.h's explicitly rounded op sequence, so given the same model output the
// update is bit-identical to the stand-alone scheduler kernel (and to oracle/scheduler.py).
// guidance_rescale > 0 needs per-sample standard deviations of the whole model output (rescale_noise_cfg): that case
// keeps the unfused path (conv -> partial statistics -> sched_step_kernel).
#include "gemm_common.h"
#include "sched_math.h"

namespace said {

typedef float f32x4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ f32x4 bload4(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

constexpr int OS_KS = 8, OS_C = 192, OS_CW = OS_C / OS_KS, OS_XP = 40;
static_assert(OS_CW == 24, "one 24-channel block per wave");

// SP (round 5, the default): the convolution on split-fp16 operands (split_f16.h; gemm_lds.hip SP): the wave-private tile is token-major, two fp16 planes [34 rows][24 channels],
// the weights out.2 in engine.cpp's pack_rows_split layout (a.ws: [8 blocks][5 k16 steps][2 planes][64 lanes][8 halfs], GroupNorm affine behind) — 30 v_mfma_f32_32x32x16_f16
// of 8 passes per wave instead of 72 v_mfma_f32_32x32x2_f32 of 16 (two waves share a SIMD's matrix pipe: 4.6 of the kernel's 12.8 us were this loop).
// The leading parameters are what the first requests need (preloaded into SGPRs: build.py): with everything inside the by-value struct the kernel began with a scalar-memory round trip.
template <bool CFG, bool SP>
__global__ __launch_bounds__(64 * OS_KS) void out_sched_kernel(const float* hx, const float* hpart, const float* hw, const int* hstep, int hT_pitch, int hxbs, int hpbs, int hB_np,
                                                               const OutSchedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NH = CFG ? 2 : 1;
    constexpr int SPR = 34, SPP = 24;                       // SP tile: token rows (32 + 2 halo), halfs per row
    constexpr int TILE_F = SP ? (2 * SPR * SPP) / 2 : OS_CW * OS_XP;   // floats of one half's staging tile
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const int tid = threadIdx.x, l = tid & 63, lt = l & 31, lh = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * 32, b = blockIdx.y;
    const int T = hT_pitch & 0xffff, pitch = (int)((unsigned)hT_pitch >> 16), pitch4 = pitch * 4;
    const int nB = hB_np & 0xffff, nparts = (int)((unsigned)hB_np >> 16);
    const int sr = l >> 3, sq = l & 7;
    float* coefS = smem;                                   // [NH][2 * 192] GroupNorm (a, b) per channel
    float* gnS = coefS + NH * 2 * OS_C + w * GN_SCRATCH;   // per-wave GroupNorm scratch
    float* mainS = coefS + NH * 2 * OS_C + OS_KS * GN_SCRATCH;
    float* xt = mainS + w * (NH * TILE_F);                 // this wave's X tiles [NH][24][XP] (SP: [NH][2 planes][34][24] halfs)
    float* red = mainS;                                    // [KS][NH][16][64] after the MFMA loop

    // ---- requests: statistics partials first (head of the dependent chain), then operands, weights last ----
    constexpr unsigned W_DW = SP ? 8u * 5u * 2u * 256u : 3u * (OS_C / 8) * 256u;   // dwords of the packed weights; gamma[192], beta[192] follow
    const GnP gp = {OS_C / 32, nparts, T, 1e-5f, hw + W_DW, hw + W_DW + OS_C, OS_C};
    GnLoads gl[NH];
    rsrc_t rp[NH], rx[NH];
    f32x4 xv[NH][3];
    float halo[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int sb = b + h * nB;   // unconditional half first (diffusion.py:397-400)
        rp[h] = make_rsrc(hpart + (long long)sb * hpbs, (unsigned)OS_C * (unsigned)nparts * 8u);
        gn_issue(gp, rp[h], w * OS_CW, OS_CW, l, gl[h]);
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int sb = b + h * nB;
        rx[h] = make_rsrc(hx + (long long)sb * hxbs, (unsigned)OS_C * (unsigned)pitch * 4u);
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) xv[h][rr] = bload4(rx[h], sr * pitch4 + (t0 + 4 * sq) * 4, (w * OS_CW + rr * 8) * pitch4);
        const int row = l >> 1, tin = (l & 1) ? (t0 + 32) : (t0 - 1);
        const bool ok = (row < OS_CW) && ((unsigned)tin < (unsigned)T);
        halo[h] = bload(rx[h], ok ? (row * pitch4 + tin * 4) : (int)0x80000000, (w * OS_CW) * pitch4);
    }
    const rsrc_t rw = make_rsrc(hw, W_DW * 4u);
    constexpr int WD0 = SP ? 5 : 3, WD1 = SP ? 2 : 3;      // SP: [k16 step][plane], else [tap][8-channel round]
    f32x4 wv[WD0][WD1];
#pragma unroll
    for (int i0 = 0; i0 < WD0; ++i0)
#pragma unroll
        for (int i1 = 0; i1 < WD1; ++i1)
            wv[i0][i1] = SP ? bload4(rw, l * 16, ((w * 5 + i0) * 2 + i1) * 1024) : bload4(rw, l * 16, (i0 * (OS_C / 8) + 3 * w + i1) * 1024);
    // epilogue operands of this wave's two output rows (channels n0, n1 of token t0 + lt)
    const int step = *hstep;
    const float* cf = a.coef + step * 8;
    float cfv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cfv[i] = cf[i];
    const int t = t0 + lt;
    const bool tok = t < T;
    const rsrc_t rlat = make_rsrc(a.lat + (long long)b * a.lat_bstride, (unsigned)a.Cout * (unsigned)a.pitch * 4u);
    const rsrc_t rnz = make_rsrc(a.step_noise ? a.step_noise + ((long long)step * a.B + b) * a.lat_bstride : nullptr,
                                 a.step_noise ? (unsigned)a.Cout * (unsigned)a.pitch * 4u : 0u);
    const rsrc_t rin = make_rsrc(a.mask ? a.init + (long long)b * a.lat_bstride : nullptr, a.mask ? (unsigned)a.Cout * (unsigned)a.pitch * 4u : 0u);
    const rsrc_t ren = make_rsrc(a.mask ? a.edit_noise + (long long)b * a.lat_bstride : nullptr, a.mask ? (unsigned)a.Cout * (unsigned)a.pitch * 4u : 0u);
    const rsrc_t rmk = make_rsrc(a.mask ? a.mask + (long long)b * a.lat_bstride : nullptr, a.mask ? (unsigned)a.Cout * (unsigned)a.pitch * 4u : 0u);
    float e_x[2], e_nz[2], e_in[2], e_en[2], e_mk[2], e_bias[2];
    int e_n[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = w + j * OS_KS;
        const int n = (r & 3) + 8 * (r >> 2) + 4 * lh;
        e_n[j] = n;
        const int vo = (tok && n < a.Cout) ? (n * a.pitch + t) * 4 : (int)0x80000000;
        e_x[j] = bload(rlat, vo, 0);
        e_nz[j] = bload(rnz, vo, 0);
        e_in[j] = bload(rin, vo, 0);
        e_en[j] = bload(ren, vo, 0);
        e_mk[j] = bload(rmk, vo, 0);
        e_bias[j] = a.bias[n < a.Cout ? n : 0];
    }

    // ---- GroupNorm coefficients of the wave's own 24 channels, per half ----
#pragma unroll
    for (int h = 0; h < NH; ++h)   // (the second sample's scratch is the wave's own, not yet written, staging tile: two independent chains the scheduler can interleave)
        gn_finish(gp, rp[h], w * OS_CW, OS_CW, l, gl[h], h == 0 ? gnS : xt, coefS + h * 2 * OS_C);

    // ---- stage: GroupNorm + SiLU once per element, wave-private LDS tiles ----
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        float* xth = xt + h * TILE_F;
        _Float16* xhh = reinterpret_cast<_Float16*>(xth);
        const float2* cG = reinterpret_cast<const float2*>(coefS + h * 2 * OS_C);
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const float2 g = cG[w * OS_CW + rr * 8 + sr];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = silu_f(fmaf(xv[h][rr][e], g.x, g.y));
                o[e] = (t0 + 4 * sq + e < T) ? v : 0.f;
            }
            if constexpr (SP) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 hv = (_Float16)o[e];
                    asm volatile("" : "+v"(hv));   // (one conversion for the stored half and the remainder: split_f16.h)
                    xhh[(1 + 4 * sq + e) * SPP + rr * 8 + sr] = hv;
                    xhh[(SPR + 1 + 4 * sq + e) * SPP + rr * 8 + sr] = (_Float16)((o[e] - (float)hv) * 2048.f);
                }
            } else {
                *reinterpret_cast<f32x4*>(xth + (rr * 8 + sr) * OS_XP + 4 + 4 * sq) = o;
            }
        }
        const int row = l >> 1, tin = (l & 1) ? (t0 + 32) : (t0 - 1);
        if (row < OS_CW) {
            const float2 g = cG[w * OS_CW + row];
            const float v = silu_f(fmaf(halo[h], g.x, g.y));
            const float hv = ((unsigned)tin < (unsigned)T) ? v : 0.f;
            if constexpr (SP) {
                _Float16 hh = (_Float16)hv;
                asm volatile("" : "+v"(hh));
                xhh[((l & 1) ? 33 : 0) * SPP + row] = hh;
                xhh[(SPR + ((l & 1) ? 33 : 0)) * SPP + row] = (_Float16)((hv - (float)hh) * 2048.f);
            } else {
                xth[row * OS_XP + ((l & 1) ? 36 : 3)] = hv;
            }
        }
    }

    // ---- MFMA: both halves share every weight fragment ----
    f32x16 acc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
    if constexpr (SP) {
        f32x16 accx[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) accx[h][r] = 0.f;
        // row `token` of the tile is the im2col row of that token (three rows of 24 channels = nine K-groups of 8; the tenth re-reads the ninth against zero weights)
        const _Float16* xh = reinterpret_cast<const _Float16*>(xt) + lt * SPP;
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            const int g = (2 * st + lh < 9) ? 2 * st + lh : 8;
            const f16x8 wh = __builtin_bit_cast(f16x8, wv[st][0]), wl = __builtin_bit_cast(f16x8, wv[st][WD1 - 1]);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const f16x8 fh = *reinterpret_cast<const f16x8*>(xh + h * (2 * SPR * SPP) + 8 * g);
                const f16x8 fl = *reinterpret_cast<const f16x8*>(xh + h * (2 * SPR * SPP) + SPR * SPP + 8 * g);
                accx[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fh, accx[h], 0, 0, 0);
                acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fh, acc[h], 0, 0, 0);
                accx[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fl, accx[h], 0, 0, 0);
            }
        }
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][r] = fmaf(accx[h][r], 0x1p-11f, acc[h][r]);   // main + 2^-11 cross
    } else {
        const float* xrow = xt + lh * OS_XP + lt + 3;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const float xf = xrow[h * TILE_F + (rr * 8 + 2 * j) * OS_XP + tap];
                        acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[tap < WD0 ? tap : 0][rr < WD1 ? rr : 0][j], xf, acc[h], 0, 0, 0);
                    }
    }

    // ---- split-K reduction (fixed order) ----
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((w * NH + h) * 16 + r) * 64 + l] = acc[h][r];
    __syncthreads();

    // ---- epilogue: bias, guidance, DDIM update, noise, mask blend; latents updated in place ----
    float* latp = a.lat + (long long)b * a.lat_bstride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = w + j * OS_KS;
        float eh[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float s = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < OS_KS; ++w2) s += red[((w2 * NH + h) * 16 + r) * 64 + l];
            eh[h] = s + e_bias[j];
        }
        const float e = CFG ? cfg_combine(eh[NH - 1], eh[0], a.guidance_scale) : eh[0];
        const int n = e_n[j];
        if (!(tok && n < a.Cout)) continue;
        const float x = e_x[j];
        note_nonfinite(e, a.status, step);
        if (a.inter) a.inter[(((long long)step * a.B + b) * T + t) * a.Cout + n] = x / a.latent_scale;
        float prev = ddim_prev(e, x, cfv, a.prediction_type);
        if (a.step_noise) prev = __fadd_rn(prev, __fmul_rn(cfv[4], e_nz[j]));
        else if (a.noise_seed) prev = __fadd_rn(prev, __fmul_rn(cfv[4], philox_normal(a.noise_seed[0], a.noise_seed[1], (unsigned)step, a.noise_elem0 + (unsigned)((b * T + t) * a.Cout + n))));
        if (a.mask) prev = mask_blend(prev, e_in[j], e_en[j], e_mk[j], cfv);
        latp[(long long)n * a.pitch + t] = prev;
    }
}

template <bool CFG> __global__ __launch_bounds__(256) void out_sched_tm_kernel(const OutSchedArgs a);   // (below)
static int out_sched_smem(bool cfg) {   // (the split-fp16 tile is the smaller one: 2 x 34 x 24 halfs against 24 x 40 floats)
    const int nh = cfg ? 2 : 1;
    const int stage = OS_KS * nh * OS_CW * OS_XP, red = OS_KS * nh * 16 * 64;
    return (nh * 2 * OS_C + OS_KS * GN_SCRATCH + (stage > red ? stage : red)) * (int)sizeof(float);
}
void configure_out_sched_kernel() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&out_sched_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&out_sched_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&out_sched_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&out_sched_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&out_sched_tm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&out_sched_tm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
}
bool out_sched_supports(const OutSchedArgs& a) {
    return a.Cin == OS_C && a.Cout <= 32 && a.guidance_rescale <= 0.f && a.gn_nparts < 0x7fff && a.T <= 0xffff && a.pitch <= 0xffff && a.B <= 0xffff &&
           a.x_bstride <= 0x7fffffffLL && a.gn_part_bstride <= 0x7fffffffLL;
}
void launch_out_sched(const OutSchedArgs& a, hipStream_t s) {
    dim3 grid((a.T + 31) / 32, a.B);
    const int tp = a.T | (a.pitch << 16), bn = a.B | (a.gn_nparts << 16);
    const bool sp = a.ws != nullptr;   // split-fp16 products (engine.cpp: fp32 mode's default; said_debug_option "out_split")
#define OS_LAUNCH(CFG, SP) hipLaunchKernelGGL((out_sched_kernel<CFG, SP>), grid, dim3(64 * OS_KS), out_sched_smem(CFG), s, a.x, a.gn_part, (SP) ? a.ws : a.w4, a.step_ptr, tp, (int)a.x_bstride, \
                                              (int)a.gn_part_bstride, bn, a)
    if (a.cfg) { if (sp) OS_LAUNCH(true, true); else OS_LAUNCH(true, false); }
    else { if (sp) OS_LAUNCH(false, true); else OS_LAUNCH(false, false); }
#undef OS_LAUNCH
}


// ------------------------------------------------------------------------------------------------------------------
// out_sched_tm_kernel (round 4) — the same last kernel for the bf16 large-batch schedule, whose activations are token-major bf16
// [sample][seg rows][192] (rgemm.hip): until now the last block's folded proj_out had to leave that schedule (xgemm_kernel writing
// channel-major fp32, 60 us at 64 x 600 tokens) to feed out_sched_kernel (19 x 32 workgroups re-reading 2 x 14.7 MB of fp32: 41 us).
// Workgroup = one 32-token tile of one clip, four waves: GroupNorm tables of the clip's unconditional and conditional sample (wave w: the
// 48-channel slice w, all partial tiles in flight: gn20_*), GroupNorm + SiLU once per element into a bf16 LDS tile [sample][34 rows][192]
// (rgemm's 400-byte rows), the convolution as a TRANSPOSED product D[out channel][token] on v_mfma_f32_32x32x16_bf16 — lane == token, as
// out_sched_kernel's epilogue wants it — with the 36 k-steps split over the four waves (fixed-order LDS reduction), both samples sharing every
// weight fragment, then exactly out_sched_kernel's epilogue (sched_math.h: same explicitly rounded op sequence).
typedef __bf16 os_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int os_u32x4 __attribute__((ext_vector_type(4)));
constexpr int OT_AP = 200, OT_ROWS = 34, OT_NP = 4;   // tile row pitch (elements), rows, 16-byte pieces per thread and sample (34 x 24 = 816 <= 4 x 256)

template <bool CFG>
__global__ __launch_bounds__(256) void out_sched_tm_kernel(const OutSchedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NH = CFG ? 2 : 1;
    const int tid = threadIdx.x, l = tid & 63, lt = l & 31, lh = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * 32, b = blockIdx.y, T = a.T;
    float* const coefS = smem;                                   // [NH][192][2]
    float* const gnS = coefS + NH * 384 + w * GN_SCRATCH;
    unsigned short* const At = reinterpret_cast<unsigned short*>(coefS + NH * 384 + 4 * GN_SCRATCH);   // [NH][34][OT_AP] bf16
    float* const red = reinterpret_cast<float*>(At);             // [4 waves][NH][16][64] after the MFMAs
    const unsigned short* xb = reinterpret_cast<const unsigned short*>(a.x_tm);

    // ---- requests: source rows (raw), epilogue operands, weight fragments; then the statistics partials
    os_u32x4 raw[NH][OT_NP];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const long long sb = b + h * a.B;
#pragma unroll
        for (int i = 0; i < OT_NP; ++i) {
            const int idx = min(tid + 256 * i, OT_ROWS * 24 - 1);
            const int row = idx / 24, pc = idx - row * 24;
            const int tt = min(max(t0 + row - 1, 0), T - 1);
            raw[h][i] = *reinterpret_cast<const os_u32x4*>(xb + ((sb * a.seg + tt) * 192 + 8 * pc));
        }
    }
    const int step = *a.step_ptr;
    const float* cf = a.coef + step * 8;
    float cfv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cfv[i] = cf[i];
    const int t = t0 + lt;
    const bool tok = t < T;
    const rsrc_t rlat = make_rsrc(a.lat + (long long)b * a.lat_bstride, (unsigned)a.Cout * (unsigned)a.pitch * 4u);
    const rsrc_t rnz = make_rsrc(a.step_noise ? a.step_noise + ((long long)step * a.B + b) * a.lat_bstride : nullptr,
                                 a.step_noise ? (unsigned)a.Cout * (unsigned)a.pitch * 4u : 0u);
    const rsrc_t rin = make_rsrc(a.mask ? a.init + (long long)b * a.lat_bstride : nullptr, a.mask ? (unsigned)a.Cout * (unsigned)a.pitch * 4u : 0u);
    const rsrc_t ren = make_rsrc(a.mask ? a.edit_noise + (long long)b * a.lat_bstride : nullptr, a.mask ? (unsigned)a.Cout * (unsigned)a.pitch * 4u : 0u);
    const rsrc_t rmk = make_rsrc(a.mask ? a.mask + (long long)b * a.lat_bstride : nullptr, a.mask ? (unsigned)a.Cout * (unsigned)a.pitch * 4u : 0u);
    float e_x[4], e_nz[4], e_in[4], e_en[4], e_mk[4], e_bias[4];
    int e_n[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = w + 4 * j;
        const int n = (r & 3) + 8 * (r >> 2) + 4 * lh;
        e_n[j] = n;
        const int vo = (tok && n < a.Cout) ? (n * a.pitch + t) * 4 : (int)0x80000000;
        e_x[j] = bload(rlat, vo, 0);
        e_nz[j] = bload(rnz, vo, 0);
        e_in[j] = bload(rin, vo, 0);
        e_en[j] = bload(ren, vo, 0);
        e_mk[j] = bload(rmk, vo, 0);
        e_bias[j] = a.bias[n < a.Cout ? n : 0];
    }
    // this wave's nine k16 steps of the [32][576] weight matrix (row = output channel, k = tap * 192 + channel); rows past Cout: row 0 again (unused)
    os_bf16x8 wf[9];
    {
        const unsigned short* wb = reinterpret_cast<const unsigned short*>(a.wb) + (long long)(lt < a.Cout ? lt : 0) * 576 + 8 * lh;
#pragma unroll
        for (int i = 0; i < 9; ++i) wf[i] = __builtin_bit_cast(os_bf16x8, *reinterpret_cast<const os_u32x4*>(wb + 16 * (9 * w + i)));
    }
    // ---- GroupNorm tables: wave w = channels [48 w, 48 w + 48) of each sample
    const GnP gp = {OS_C / 32, a.gn_nparts, T, 1e-5f, a.gn_gamma, a.gn_beta, OS_C};
    {   // (both samples' partials in flight together: one memory round trip — at the start of a launch that is 8k clocks, rgemm.hip's stamps)
        rsrc_t rp[NH];
        GnL20 gl[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            rp[h] = make_rsrc(a.gn_part + (long long)(b + h * a.B) * a.gn_part_bstride, (unsigned)OS_C * (unsigned)a.gn_nparts * 8u);
            gn20_issue(gp, rp[h], 48 * w, l, gl[h]);
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            gn20_finish(gp, rp[h], 48 * w, l, gl[h], gnS, coefS + h * 384);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // ---- GroupNorm + SiLU once per element -> bf16 tile (rows outside [0, T): zeros = the convolution's padding)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const float2* cG = reinterpret_cast<const float2*>(coefS + h * 384);
#pragma unroll
        for (int i = 0; i < OT_NP; ++i) {
            const int idx = tid + 256 * i;
            if (idx < OT_ROWS * 24) {
                const int row = idx / 24, pc = idx - row * 24;
                const int tt = t0 + row - 1;
                const bool valid = tt >= 0 && tt < T;
                os_u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 g0 = cG[8 * pc + 2 * e], g1 = cG[8 * pc + 2 * e + 1];
                    const float x0 = silu_f(fmaf(__builtin_bit_cast(float, raw[h][i][e] << 16), g0.x, g0.y));
                    const float x1 = silu_f(fmaf(__builtin_bit_cast(float, raw[h][i][e] & 0xffff0000u), g1.x, g1.y));
                    typedef __bf16 os_bf16x2 __attribute__((ext_vector_type(2)));
                    const os_bf16x2 p2 = {(__bf16)x0, (__bf16)x1};
                    o[e] = valid ? __builtin_bit_cast(unsigned, p2) : 0u;
                }
                *reinterpret_cast<os_u32x4*>(At + (h * OT_ROWS + row) * OT_AP + 8 * pc) = o;
            }
        }
    }
    __syncthreads();
    // ---- D[out channel][token] += W[out channel][k] A[token + tap][k]: lane == token
    f32x16 acc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int sidx = 9 * w + i, tap = sidx / 12, c0 = (sidx - 12 * tap) * 16;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const os_bf16x8 fa = __builtin_bit_cast(os_bf16x8, *reinterpret_cast<const os_u32x4*>(At + (h * OT_ROWS + lt + tap) * OT_AP + c0 + 8 * lh));
            acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], fa, acc[h], 0, 0, 0);
        }
    }
    __syncthreads();   // (the tile is dead: its memory is the reduction buffer)
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((w * NH + h) * 16 + r) * 64 + l] = acc[h][r];
    __syncthreads();
    // ---- epilogue: bias, guidance, DDIM update, noise, mask blend; latents updated in place (out_sched_kernel's)
    float* latp = a.lat + (long long)b * a.lat_bstride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = w + 4 * j;
        float eh[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float s = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) s += red[((w2 * NH + h) * 16 + r) * 64 + l];
            eh[h] = s + e_bias[j];
        }
        const float e = CFG ? cfg_combine(eh[NH - 1], eh[0], a.guidance_scale) : eh[0];
        const int n = e_n[j];
        if (!(tok && n < a.Cout)) continue;
        const float x = e_x[j];
        note_nonfinite(e, a.status, step);
        if (a.inter) a.inter[(((long long)step * a.B + b) * T + t) * a.Cout + n] = x / a.latent_scale;
        float prev = ddim_prev(e, x, cfv, a.prediction_type);
        if (a.step_noise) prev = __fadd_rn(prev, __fmul_rn(cfv[4], e_nz[j]));
        else if (a.noise_seed) prev = __fadd_rn(prev, __fmul_rn(cfv[4], philox_normal(a.noise_seed[0], a.noise_seed[1], (unsigned)step, a.noise_elem0 + (unsigned)((b * T + t) * a.Cout + n))));
        if (a.mask) prev = mask_blend(prev, e_in[j], e_en[j], e_mk[j], cfv);
        latp[(long long)n * a.pitch + t] = prev;
    }
}
static int out_sched_tm_smem(bool cfg) {
    const int nh = cfg ? 2 : 1;
    const int tile = nh * OT_ROWS * OT_AP * 2, red = 4 * nh * 16 * 64 * 4;
    return (nh * 384 + 4 * GN_SCRATCH) * (int)sizeof(float) + (tile > red ? tile : red);
}
bool out_sched_tm_supports(const OutSchedArgs& a) {
    return out_sched_supports(a) && a.x_tm && a.wb && a.seg > 0 && (long long)(a.cfg ? 2 : 1) * a.B * a.seg * 192 < 0x7fffffffLL;
}
void launch_out_sched_tm(const OutSchedArgs& a, hipStream_t s) {
    dim3 grid((a.T + 31) / 32, a.B);
    if (a.cfg) hipLaunchKernelGGL(out_sched_tm_kernel<true>, grid, dim3(256), out_sched_tm_smem(true), s, a);
    else hipLaunchKernelGGL(out_sched_tm_kernel<false>, grid, dim3(256), out_sched_tm_smem(false), s, a);
}

}  // namespace said

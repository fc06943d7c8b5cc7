// gemm.hip — channel-major fp32 MFMA GEMM / implicit-GEMM Conv1d for gfx950.
//
//   Y[b][n][t] = epi( sum_seg sum_tap sum_c W[n][c][tap] * xform(X_seg[b][c][t*stride + tap - pad]) )
//
// Design (MI355X-first):
//  * v_mfma_f32_32x32x2_f32, exact fp32.  With channel-major activations the B operand of
//    lane l is X[c0 + (l>>5)][t0 + (l&31)] — one coalesced 128-B row segment per half-wave —
//    and the A operand is a host-packed weight fragment (256 contiguous bytes per wave), so
//    neither operand is staged through LDS.
//  * A workgroup owns NB 32-row output tiles x 32 tokens; its KS waves split K by input
//    channel, so each X element is fetched and transformed by exactly one wave and reused for
//    all NB tiles.  Partial accumulators are reduced through LDS in a fixed order
//    (deterministic), then the epilogue runs on the reduced tile.
//  * Producer-side elementwise work is fused into the operand load: GroupNorm+SiLU,
//    LayerNorm, GroupNorm->LayerNorm, SiLU.  GroupNorm statistics arrive as per-(channel,
//    32-token tile) Welford partials written by the producing kernel's epilogue and are
//    combined here (Chan) — no separate statistics kernels, no atomics.
//  * Epilogues: bias / activation / timestep-embedding add / residual (optionally
//    GroupNorm'ed) / GN partials out; QKV split with V written token-major for the attention
//    kernel; GEGLU; banded cross-attention (ldm/attention.py:170-191) behind the q projection.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace said {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float half_sum(float v) {  // within each 32-lane half
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Combine the Welford partials of one GroupNorm group and emit per-channel affine
// coefficients a_c = rstd*gamma_c, b_c = beta_c - mean*a_c for channels [c_begin, c_end).
// part: [C][nparts][2] for this batch; executed by one whole wave.
__device__ __forceinline__ void gn_group_coefs(const float* part, int nparts, int Tin, int cpg, int grp, float eps,
                                               const float* gamma, const float* beta, int c_begin, int c_end,
                                               float* outA, float* outB, int out_base, int lane) {
    const int entries = cpg * nparts;
    const float* p = part + (long long)grp * cpg * nparts * 2;
    float s = 0.f;
    for (int e = lane; e < entries; e += 64) {
        const int pi = e % nparts;
        const float cnt = (float)min(32, Tin - pi * 32);
        s += cnt * p[2 * e];
    }
    const float total = (float)cpg * (float)Tin;
    const float mean = wave_sum(s) / total;
    float q = 0.f;
    for (int e = lane; e < entries; e += 64) {
        const int pi = e % nparts;
        const float cnt = (float)min(32, Tin - pi * 32);
        const float d = p[2 * e] - mean;
        q += p[2 * e + 1] + cnt * d * d;
    }
    const float var = wave_sum(q) / total;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane < cpg) {
        const int c = grp * cpg + lane;
        if (c >= c_begin && c < c_end) {
            const float av = rstd * gamma[c];
            outA[c - out_base] = av;
            outB[c - out_base] = beta[c] - mean * av;
        }
    }
}

template <int XF>
__device__ __forceinline__ float xform_apply(float v, float2 gn, float mu, float rs, float2 ln) {
    if (XF == XF_NONE) return v;
    if (XF == XF_SILU) return silu_f(v);
    if (XF == XF_GN_SILU) return silu_f(fmaf(v, gn.x, gn.y));
    if (XF == XF_LN) return fmaf((v - mu) * rs, ln.x, ln.y);
    if (XF == XF_GN_LN) return fmaf((fmaf(v, gn.x, gn.y) - mu) * rs, ln.x, ln.y);
    return v;
}

// LDS carve (floats).  Must match gemm_smem_floats() on the host side below.
__host__ __device__ inline int seg_coef_floats(const Seg& s) {
    int f = 0;
    if (s.xform == XF_GN_SILU || s.xform == XF_GN_LN) f += 2 * s.C;
    if (s.xform == XF_LN || s.xform == XF_GN_LN) f += 2 * s.C;
    return f;
}
template <int NACC>
__host__ __device__ inline int epi_scratch_floats(int epi, int KS) {
    if (epi == EPI_BAND) return 32 * 32 + (KS * 2) * 8 * 32;  // q tile + score partials [groups][wmax<=8][32]
    return 2 * 32 * NACC;                                      // residual GN coefficients
}

template <int XF, int NACC, bool TRANS>
__device__ __forceinline__ void seg_accumulate(const Seg& sg, const float* __restrict__ xb, const float* __restrict__ wb,
                                               const long long tile_wstride, const int* tile_off, int t0, int lt, int lh,
                                               int l, int c_lo, int cw, const float2* coefGN, const float2* coefLN,
                                               float mu, float rs, f32x16 (&acc)[NACC]) {
    const int taps = sg.taps, pitch = sg.x_pitch, halfC = sg.C >> 1;
    for (int tap = 0; tap < taps; ++tap) {
        const int tin = (t0 + lt) * sg.stride + tap - sg.pad;
        const bool valid = (tin >= 0) && (tin < sg.Tin);
        const float* xp = xb + (long long)(c_lo + lh) * pitch + (valid ? tin : 0);
        const float* wp = wb + ((long long)tap * halfC + (c_lo >> 1)) * 64 + l;
#pragma unroll 4
        for (int cp = 0; cp < (cw >> 1); ++cp) {
            float xv = xp[(long long)(2 * cp) * pitch];
            const int c = c_lo + 2 * cp + lh;
            float2 gn = make_float2(1.f, 0.f), ln = make_float2(1.f, 0.f);
            if (XF == XF_GN_SILU || XF == XF_GN_LN) gn = coefGN[c];
            if (XF == XF_LN || XF == XF_GN_LN) ln = coefLN[c];
            xv = xform_apply<XF>(xv, gn, mu, rs, ln);
            xv = valid ? xv : 0.f;
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                const float wv = wp[(long long)tile_off[i] * tile_wstride + (long long)cp * 64];
                if (TRANS)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, wv, acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, xv, acc[i], 0, 0, 0);
            }
        }
    }
}

template <int NB, int KS, int EPI, bool TRANS>
__device__ __forceinline__ void cgemm_body(const GemmArgs& a, float* smem) {
    constexpr int NACC = (EPI == EPI_GEGLU) ? 2 * NB : NB;
    const int tid = threadIdx.x, l = tid & 63, lt = l & 31, lh = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * 32;
    const int b = blockIdx.z;
    const int tb_per_group = a.ntiles_per_group / NB;
    const int g = blockIdx.y / tb_per_group;
    const int tile0 = (blockIdx.y % tb_per_group) * NB;
    const int w_tiles_pg = (EPI == EPI_GEGLU) ? a.ntiles_per_group + a.geglu_gate_tiles : a.ntiles_per_group;
    (void)w_tiles_pg;

    int tile_off[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) tile_off[i] = (i < NB) ? (tile0 + i) : (tile0 + (i - NB) + a.geglu_gate_tiles);

    const int epi_sz = epi_scratch_floats<NACC>(EPI, KS);
    float* epiS = smem;
    float* mainS = smem + epi_sz;

    // ---- prologue A: GroupNorm coefficients (main-loop segments and residual) ----
    int coef_off[3];
    {
        int off = 0;
        for (int s = 0; s < a.nseg; ++s) {
            coef_off[s] = off;
            off += seg_coef_floats(a.seg[s]);
        }
    }
    for (int s = 0; s < a.nseg; ++s) {
        const Seg& sg = a.seg[s];
        if (sg.xform == XF_GN_SILU || sg.xform == XF_GN_LN) {
            const int sb = sg.b_mod > 0 ? b % sg.b_mod : b;
            const float* part = sg.gn_part + (long long)sb * sg.gn_part_bstride;
            float* cA = mainS + coef_off[s];  // interleaved (a, b) pairs
            const int G = sg.C / sg.gn_cpg;
            for (int grp = w; grp < G; grp += KS) {
                // write interleaved: use stride-2 views
                const int entries = sg.gn_cpg * sg.gn_nparts;
                const float* p = part + (long long)grp * entries * 2;
                float sacc = 0.f;
                for (int e = l; e < entries; e += 64) {
                    const int pi = e % sg.gn_nparts;
                    sacc += (float)min(32, sg.Tin - pi * 32) * p[2 * e];
                }
                const float total = (float)sg.gn_cpg * (float)sg.Tin;
                const float mean = wave_sum(sacc) / total;
                float q = 0.f;
                for (int e = l; e < entries; e += 64) {
                    const int pi = e % sg.gn_nparts;
                    const float d = p[2 * e] - mean;
                    q += p[2 * e + 1] + (float)min(32, sg.Tin - pi * 32) * d * d;
                }
                const float var = wave_sum(q) / total;
                const float rstd = 1.0f / sqrtf(var + sg.gn_eps);
                if (l < sg.gn_cpg) {
                    const int c = grp * sg.gn_cpg + l;
                    const float av = rstd * sg.gn_gamma[c];
                    cA[2 * c] = av;
                    cA[2 * c + 1] = sg.gn_beta[c] - mean * av;
                }
            }
        }
        if (sg.xform == XF_LN || sg.xform == XF_GN_LN) {
            float* cL = mainS + coef_off[s] + ((sg.xform == XF_GN_LN) ? 2 * sg.C : 0);
            for (int c = tid; c < sg.C; c += 64 * KS) {
                cL[2 * c] = sg.ln_gamma[c];
                cL[2 * c + 1] = sg.ln_beta[c];
            }
        }
    }
    if (EPI == EPI_STORE && a.res_kind == RES_GN) {
        const float* part = a.res_gn_part + (long long)b * a.res_gn_part_bstride;
        const int c_begin = tile0 * 32, c_end = min(a.N, (tile0 + NB) * 32);
        const int g_first = c_begin / a.res_gn_cpg, g_last = (c_end - 1) / a.res_gn_cpg;
        for (int grp = g_first + w; grp <= g_last; grp += KS)
            gn_group_coefs(part, a.res_gn_nparts, a.T, a.res_gn_cpg, grp, a.res_gn_eps, a.res_gn_gamma, a.res_gn_beta,
                           c_begin, c_end, epiS, epiS + 32 * NACC, c_begin, l);
    }
    __syncthreads();

    // ---- prologue B: LayerNorm statistics of this token tile (segment 0 only) ----
    float mu = 0.f, rs = 1.f;
    {
        const Seg& sg = a.seg[0];
        if (sg.xform == XF_LN || sg.xform == XF_GN_LN) {
            const float2* cGN = reinterpret_cast<const float2*>(mainS + coef_off[0]);
            float* lnred = mainS + coef_off[a.nseg - 1] + seg_coef_floats(a.seg[a.nseg - 1]);
            const float* xb = sg.x + (long long)(sg.b_mod > 0 ? b % sg.b_mod : b) * sg.x_bstride;
            const int t = min(t0 + lt, sg.Tin - 1);
            const int cw = sg.C / KS, c_lo = w * cw;
            float ref = xb[t];
            if (sg.xform == XF_GN_LN) ref = fmaf(ref, cGN[0].x, cGN[0].y);
            float s1 = 0.f, s2 = 0.f;
            for (int c = c_lo + lh; c < c_lo + cw; c += 2) {
                float v = xb[(long long)c * sg.x_pitch + t];
                if (sg.xform == XF_GN_LN) v = fmaf(v, cGN[c].x, cGN[c].y);
                const float d = v - ref;
                s1 += d;
                s2 = fmaf(d, d, s2);
            }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lh == 0) {
                lnred[(w * 32 + lt) * 2] = s1;
                lnred[(w * 32 + lt) * 2 + 1] = s2;
            }
            __syncthreads();
            float S1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < KS; ++w2) {
                S1 += lnred[(w2 * 32 + lt) * 2];
                S2 += lnred[(w2 * 32 + lt) * 2 + 1];
            }
            const float invC = 1.0f / (float)sg.C;
            const float md = S1 * invC;
            const float var = fmaxf(S2 * invC - md * md, 0.f);
            mu = ref + md;
            rs = 1.0f / sqrtf(var + sg.ln_eps);
        }
    }

    // ---- main loop ----
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    for (int s = 0; s < a.nseg; ++s) {
        const Seg& sg = a.seg[s];
        const int cw = sg.C / KS, c_lo = w * cw;
        const float* xb = sg.x + (long long)(sg.b_mod > 0 ? b % sg.b_mod : b) * sg.x_bstride +
                          (long long)g * sg.c_group_stride * sg.x_pitch;
        const long long tile_wstride = (long long)sg.taps * (sg.C >> 1) * 64;
        const float* wb = sg.w + (long long)g * w_tiles_pg * tile_wstride;
        const float2* cGN = reinterpret_cast<const float2*>(mainS + coef_off[s]);
        const float2* cLN = reinterpret_cast<const float2*>(mainS + coef_off[s] + ((sg.xform == XF_GN_LN) ? 2 * sg.C : 0));
        switch (sg.xform) {
            case XF_NONE: seg_accumulate<XF_NONE, NACC, TRANS>(sg, xb, wb, tile_wstride, tile_off, t0, lt, lh, l, c_lo, cw, cGN, cLN, mu, rs, acc); break;
            case XF_GN_SILU: seg_accumulate<XF_GN_SILU, NACC, TRANS>(sg, xb, wb, tile_wstride, tile_off, t0, lt, lh, l, c_lo, cw, cGN, cLN, mu, rs, acc); break;
            case XF_LN: seg_accumulate<XF_LN, NACC, TRANS>(sg, xb, wb, tile_wstride, tile_off, t0, lt, lh, l, c_lo, cw, cGN, cLN, mu, rs, acc); break;
            case XF_GN_LN: seg_accumulate<XF_GN_LN, NACC, TRANS>(sg, xb, wb, tile_wstride, tile_off, t0, lt, lh, l, c_lo, cw, cGN, cLN, mu, rs, acc); break;
            default: seg_accumulate<XF_SILU, NACC, TRANS>(sg, xb, wb, tile_wstride, tile_off, t0, lt, lh, l, c_lo, cw, cGN, cLN, mu, rs, acc); break;
        }
    }

    // ---- split-K reduction through LDS (fixed order => deterministic) ----
    __syncthreads();
    float* red = mainS;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((w * NACC + i) * 16 + r) * 64 + l] = acc[i][r];
    __syncthreads();

    constexpr int NV = NB * 16;  // value vectors (one per (tile, acc register))
    static_assert(NV % KS == 0, "NB*16 must be divisible by KS");
    constexpr int VPW = NV / KS;
    const int nparts_out = (a.T + 31) >> 5;

#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        const int v = w + j * KS;
        const int i = v >> 4, r = v & 15;
        float val = 0.f, gate = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < KS; ++w2) val += red[((w2 * NACC + i) * 16 + r) * 64 + l];
        if (EPI == EPI_GEGLU) {
#pragma unroll
            for (int w2 = 0; w2 < KS; ++w2) gate += red[((w2 * NACC + i + NB) * 16 + r) * 64 + l];
        }
        const int frow = (r & 3) + 8 * (r >> 2) + 4 * lh;  // fragment row
        const int tile = tile0 + i;

        if (EPI == EPI_QKV && TRANS) {
            // D[i=t][j=n]: lane column = channel, fragment rows = tokens
            const int nl = tile * 32 + lt;
            const int t = t0 + frow;
            if (a.bias) val += a.bias[nl];
            const int vn = (tile - a.vt_first_tile) * 32 + lt;
            const int h = vn / a.vt_dim, d = vn % a.vt_dim;
            if (t < a.T && nl < a.N)
                a.vt[(((long long)b * a.vt_heads + h) * a.vt_rows + t) * a.vt_dim + d] = val;
            continue;
        }

        const int nl = tile * 32 + frow;  // channel within group
        const int t = t0 + lt;
        const bool ok = (nl < a.N) && (t < a.T);
        const int ng = g * a.N + nl;

        if (EPI == EPI_GEGLU) {
            const int ngate = nl + a.geglu_gate_tiles * 32;
            const float xv = val + (a.bias ? a.bias[nl] : 0.f);
            const float gv = gate + (a.bias ? a.bias[ngate] : 0.f);
            if (ok) a.y[(long long)b * a.y_bstride + (long long)nl * a.y_pitch + t] = xv * gelu_f(gv);
            continue;
        }

        if (EPI == EPI_BAND) {
            epiS[frow * 32 + lt] = val;  // q tile [d][t]
            continue;
        }

        // EPI_STORE / EPI_QKV (normal orientation)
        if (nl < a.N) {
            if (a.bias) val += a.bias[ng];
            if (a.act == ACT_SILU) val = silu_f(val);
            else if (a.act == ACT_GELU) val = gelu_f(val);
            if (a.emb) {
                const int row = (a.step_ptr ? *a.step_ptr : 0) + b * a.emb_b_stride;
                val += a.emb[(long long)ng * a.emb_pitch + row];
            }
        }
        if (EPI == EPI_STORE) {
            if (ok && a.res_kind != RES_NONE) {
                float rv = a.res[(long long)b * a.res_bstride + (long long)ng * a.res_pitch + t];
                if (a.res_kind == RES_GN) rv = fmaf(rv, epiS[nl - tile0 * 32], epiS[32 * NACC + nl - tile0 * 32]);
                val += rv;
            }
        }
        if (ok) a.y[(long long)b * a.y_bstride + (long long)ng * a.y_pitch + t] = val;
        if (EPI == EPI_STORE && a.stats_out) {
            const float cnt = (float)min(32, a.T - t0);
            const float vv = (t < a.T) ? val : 0.f;
            const float mean = half_sum(vv) / cnt;
            const float d = (t < a.T) ? (val - mean) : 0.f;
            const float m2 = half_sum(d * d);
            if (lt == 0 && nl < a.N) {
                float* so = a.stats_out + (long long)b * a.stats_bstride + ((long long)ng * nparts_out + blockIdx.x) * 2;
                so[0] = mean;
                so[1] = m2;
            }
        }
    }

    if (EPI == EPI_BAND) {
        // banded cross-attention on the finished q tile (one head x 32 queries)
        static_assert(EPI != EPI_BAND || NB == 1, "EPI_BAND needs NB == 1");
        __syncthreads();
        constexpr int NG = KS * 2;        // 32-thread groups
        constexpr int DPG = 32 / NG;      // head-dim rows per group
        const float* qt = epiS;
        float* part = epiS + 32 * 32;
        const int gi = tid >> 5, tt = tid & 31;
        const int t = t0 + tt;
        const bool tv = t < a.T;
        const int lo = tv ? a.band.lo[t] : 0, hi = tv ? a.band.hi[t] : 0;
        const int head = tile0;
        const float* kb = a.band.k + (long long)b * a.band.kv_bstride + (long long)(head * 32) * a.band.kv_pitch;
        const float* vb = a.band.v + (long long)b * a.band.kv_bstride + (long long)(head * 32) * a.band.kv_pitch;
        const int wmax = a.band.wmax;
        for (int wi = 0; wi < wmax; ++wi) {
            const int s = lo + wi;
            float p = 0.f;
            if (s < hi) {
#pragma unroll
                for (int dd = 0; dd < DPG; ++dd) {
                    const int d = gi * DPG + dd;
                    p = fmaf(qt[d * 32 + tt], kb[(long long)d * a.band.kv_pitch + s], p);
                }
            }
            part[(gi * 8 + wi) * 32 + tt] = p;
        }
        __syncthreads();
        float sc[8];
        float mx = -3.0e38f;
#pragma unroll
        for (int wi = 0; wi < 8; ++wi) {
            float sum = 0.f;
            if (wi < wmax) {
#pragma unroll
                for (int g2 = 0; g2 < NG; ++g2) sum += part[(g2 * 8 + wi) * 32 + tt];
            }
            const bool vis = (wi < wmax) && (lo + wi < hi);
            sc[wi] = vis ? sum * a.band.scale : -3.0e38f;
            mx = fmaxf(mx, sc[wi]);
        }
        float den = 0.f;
#pragma unroll
        for (int wi = 0; wi < 8; ++wi) {
            const bool vis = (wi < wmax) && (lo + wi < hi);
            sc[wi] = vis ? __expf(sc[wi] - mx) : 0.f;
            den += sc[wi];
        }
        const float inv = 1.0f / den;
        if (tv) {
#pragma unroll
            for (int dd = 0; dd < DPG; ++dd) {
                const int d = gi * DPG + dd;
                float o = 0.f;
#pragma unroll
                for (int wi = 0; wi < 8; ++wi)
                    if ((wi < wmax) && (lo + wi < hi)) o = fmaf(sc[wi] * inv, vb[(long long)d * a.band.kv_pitch + lo + wi], o);
                a.y[(long long)b * a.y_bstride + (long long)(head * 32 + d) * a.y_pitch + t] = o;
            }
        }
    }
}

template <int NB, int KS, int EPI>
__global__ __launch_bounds__(64 * KS) void cgemm_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (EPI == EPI_QKV) {
        const int tb_per_group = a.ntiles_per_group / NB;
        const int tile0 = (blockIdx.y % tb_per_group) * NB;
        if (tile0 >= a.vt_first_tile) {
            cgemm_body<NB, KS, EPI, true>(a, smem);
            return;
        }
    }
    cgemm_body<NB, KS, EPI, false>(a, smem);
}

template <int NB, int EPI>
static int gemm_smem_floats(const GemmArgs& a, int KS) {
    constexpr int NACC = (EPI == EPI_GEGLU) ? 2 * NB : NB;
    int coef = 0;
    for (int s = 0; s < a.nseg; ++s) coef += seg_coef_floats(a.seg[s]);
    coef += KS * 32 * 2;  // LN reduction scratch
    const int red = KS * NACC * 16 * 64;
    return epi_scratch_floats<NACC>(EPI, KS) + (coef > red ? coef : red);
}

constexpr int kMaxLdsBytes = 160 * 1024;

template <int NB, int KS, int EPI>
static void launch_one(const GemmArgs& a, int batch, hipStream_t s) {
    const int smem = gemm_smem_floats<NB, EPI>(a, KS) * (int)sizeof(float);
    if (smem > kMaxLdsBytes) {
        fprintf(stderr, "said: gemm needs %d B of LDS (> %d)\n", smem, kMaxLdsBytes);
        abort();
    }
    dim3 grid((a.T + 31) / 32, a.groups * (a.ntiles_per_group / NB), batch);
    hipLaunchKernelGGL((cgemm_kernel<NB, KS, EPI>), grid, dim3(64 * KS), smem, s, a);
}
template <int NB, int KS, int EPI>
static void configure_one() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgemm_kernel<NB, KS, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes);
}

#define SAID_GEMM_CONFIGS(X)                                                                   \
    X(EPI_STORE, 1, 8) X(EPI_STORE, 2, 8) X(EPI_STORE, 3, 4) X(EPI_STORE, 6, 4) X(EPI_STORE, 4, 4) \
    X(EPI_QKV, 1, 8) X(EPI_QKV, 2, 8) X(EPI_QKV, 3, 4) X(EPI_QKV, 6, 4)                        \
    X(EPI_GEGLU, 1, 8) X(EPI_GEGLU, 3, 4)                                                      \
    X(EPI_BAND, 1, 8) X(EPI_BAND, 1, 4)

void configure_gemm_kernels() {
#define X(E, nb, ks) configure_one<nb, ks, E>();
    SAID_GEMM_CONFIGS(X)
#undef X
}

void launch_gemm(const GemmArgs& a, int epi, int batch, int NB, int KS, hipStream_t s) {
#define X(E, nb, ks) \
    if (epi == E && NB == nb && KS == ks) { launch_one<nb, ks, E>(a, batch, s); return; }
    SAID_GEMM_CONFIGS(X)
#undef X
    fprintf(stderr, "said: unsupported gemm config epi=%d NB=%d KS=%d\n", epi, NB, KS);
    abort();
}

}  // namespace said

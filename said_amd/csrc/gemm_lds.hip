// gemm_lds.hip — the UNet GEMM: channel-major fp32 MFMA GEMM / Conv1d(k=1|3, stride 1) with the X operand
// staged ONCE per wave through LDS.
//
// Why (measured on MI355X, scripts/ubench/load_issue.hip): a CU sustains only ~20 B/clk of dword
// buffer loads (~44 B/clk as dwordx4), so the first version of this kernel — two dword loads per MFMA,
// X re-fetched for each of the three conv taps — spent 3-7k clocks just issuing its 72 loads per wave.
// Here a wave
//   * fetches its 24-channel x 32-token slice of X with three dwordx4 loads (+ one dword load for the
//     two halo columns of a k=3 conv), applies the fused operand transform (GroupNorm+SiLU /
//     LayerNorm / GroupNorm->LayerNorm / SiLU) ONCE per element, and parks the result in a wave-private
//     LDS tile — no cross-wave synchronisation is needed for it;
//   * fetches its weight fragments as dwordx4 (host packing puts a lane's four consecutive k-pairs side
//     by side): 9 loads instead of 36 for a conv slice;
//   * runs a main loop that is nothing but ds_read_b32 + v_mfma_f32_32x32x2_f32.
// LayerNorm statistics come from the very registers that were loaded for staging; GroupNorm statistics
// are finalised per wave for its own channel slice from the producer's Welford partials (gemm_common.h).
// Work split, reduction and epilogues are those of gemm.hip: NB output tiles x 32 tokens per workgroup,
// KS waves splitting K by input channel, fixed-order LDS reduction, fused epilogues.
#include <cstdio>
#include <cstdlib>

#include "gemm_common.h"

namespace said {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bload4(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

constexpr int XP = 40;        // LDS tile pitch (floats): col = tin - t0 + 4  (halo at 3 and 36)
constexpr int NRMAX = 3;      // row rounds per block: CB = 8 * NR <= 24 channels

template <int XF>
__device__ __forceinline__ float xf1(float v, float2 gn, float mu, float rs, float2 ln) { return xform_apply<XF>(v, gn, mu, rs, ln); }

// 16 dwords of launch-invariant scalars passed as LEADING kernel parameters: with -amdgpu-kernarg-preload-count=16
// the command processor delivers them in SGPRs, so the first operand loads need no memory round trip at all.
struct FastHdr {
    const float* x;    // segment 0 source (batch 0)
    const float* w4;   // segment 0 packed weights
    int pack;          // C | taps << 16 | xform << 20 | nseg << 24
    int pitch, Tin, bstride;   // segment 0 pitch, valid length, batch stride (floats)
    int bmod_b0;       // b_mod | b0 << 16
    int T, N, ntiles, gate_tiles;
    int r0, r1, r2;
};
static_assert(sizeof(FastHdr) == 64, "FastHdr must be exactly 16 dwords");

struct UBlock {   // one (segment, channel block) of this wave
    rsrc_t rx, rw;
    int c0;        // first channel of the block (segment-relative)
    int nr;        // row rounds (CB / 8)
    int taps, Tin, pitch4, C8 /* C / 8 */;
    int xform;
    const float2* cGN;   // LDS coefficient tables, indexed by segment channel
    const float2* cLN;
};

template <int NB, int KS, int EPI, bool TRANS>
__device__ __forceinline__ void ugemm_body(const FastHdr& hd, float* smem, int bx, int by, int bz) {
    constexpr int NACC = (EPI == EPI_GEGLU) ? 2 * NB : NB;
    constexpr int NV = NB * 16;
    static_assert(NV % KS == 0, "NB*16 must be divisible by KS");
    constexpr int VPW = NV / KS;
    constexpr bool EPRE = (VPW <= 4) && (EPI == EPI_STORE || EPI == EPI_QKV) && !TRANS;
    constexpr int TMAX = (EPI == EPI_STORE) ? 3 : 1;   // only plain convolutions have 3 taps
    const int tid = threadIdx.x, l = tid & 63, lt = l & 31, lh = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = bz + (hd.bmod_b0 >> 16);
    const int t0 = bx * 32;
    const int tile0 = by * NB;
    const int nseg = (hd.pack >> 24) & 3, aT = hd.T, aN = hd.N;
    const int gate_tiles = hd.gate_tiles;
    const int w_tiles = (EPI == EPI_GEGLU) ? hd.ntiles + gate_tiles : hd.ntiles;
    const int sr = l >> 3, sq = l & 7;   // staging map: row-in-round, token quad
    const int C0 = hd.pack & 0xffff, taps0 = (hd.pack >> 16) & 15, xf0 = (hd.pack >> 20) & 15;

    const int epi_sz = epi_scratch_floats<NACC>(EPI, KS);
    float* epiS = smem;
    float* gnS = smem + epi_sz + w * GN_SCRATCH;
    float* mainS = smem + epi_sz + KS * GN_SCRATCH;

    int tile_wo[NACC];   // output tile index of each accumulator
#pragma unroll
    for (int i = 0; i < NACC; ++i) tile_wo[i] = (i < NB) ? (tile0 + i) : (tile0 + (i - NB) + gate_tiles);

    // block 0 of segment 0 entirely from the preloaded header: its operand loads are issued first thing
    UBlock u0;
    {
        const int cw = C0 / KS;
        const int cb = (cw % 24 == 0) ? 24 : cw;
        const int bmod = hd.bmod_b0 & 0xffff;
        const int sb = bmod > 0 ? b % bmod : b;
        u0.rx = make_rsrc(hd.x + (long long)sb * hd.bstride, (unsigned)C0 * (unsigned)hd.pitch * 4u);
        u0.rw = make_rsrc(hd.w4, (unsigned)w_tiles * (unsigned)taps0 * (unsigned)(C0 >> 3) * 1024u);
        u0.c0 = w * cw;
        u0.nr = cb >> 3;
        u0.taps = taps0; u0.Tin = hd.Tin; u0.pitch4 = hd.pitch * 4; u0.C8 = C0 >> 3; u0.xform = xf0;
        u0.cGN = reinterpret_cast<const float2*>(mainS);
        u0.cLN = reinterpret_cast<const float2*>(mainS + ((xf0 == XF_GN_LN) ? 2 * C0 : 0));
    }

    // raw X slice of a block -> registers: NR dwordx4 (row sr of each round, tokens t0+4*sq..+3) + halo dwords
    auto issue_x = [&](const UBlock& u, f32x4 (&xv)[NRMAX], float& halo) {
#pragma unroll
        for (int rr = 0; rr < NRMAX; ++rr) {
            const bool on = rr < u.nr;
            xv[rr] = bload4(u.rx, on ? (sr * u.pitch4 + (t0 + 4 * sq) * 4) : (int)0x80000000, (u.c0 + rr * 8) * u.pitch4);
        }
        halo = 0.f;
        if (u.taps == 3) {   // lane -> (row = l >> 1, side = l & 1): token t0-1 or t0+32
            const int row = l >> 1, tin = (l & 1) ? (t0 + 32) : (t0 - 1);
            const bool ok = (row < u.nr * 8) && ((unsigned)tin < (unsigned)u.Tin);
            halo = bload(u.rx, ok ? (row * u.pitch4 + tin * 4) : (int)0x80000000, u.c0 * u.pitch4);
        }
    };
    auto issue_w = [&](const UBlock& u, f32x4 (&wv)[TMAX][NRMAX][NACC]) {
#pragma unroll
        for (int tap = 0; tap < TMAX; ++tap)
#pragma unroll
            for (int rr = 0; rr < NRMAX; ++rr)
#pragma unroll
                for (int i = 0; i < NACC; ++i) {
                    const bool on = (tap < u.taps) && (rr < u.nr);
                    const int so = ((tile_wo[i] * u.taps + tap) * u.C8 + (u.c0 >> 3) + rr) * 1024;
                    wv[tap][rr][i] = bload4(u.rw, on ? l * 16 : (int)0x80000000, on ? so : 0);
                }
    };

    // ================= phase 0: requests =================
    f32x4 xv[NRMAX], wv[TMAX][NRMAX][NACC];
    float halo;
    // request order matters: vector loads return in order, so the argument block goes first (GroupNorm partial loads
    // wait on it), then the operands, which need only the preloaded header
    const ArgView V = arg_view(l, 16);   // the rest of the argument block: 4 coalesced loads, fields via v_readlane
    issue_x(u0, xv, halo);
    const bool has_ln = (xf0 == XF_LN || xf0 == XF_GN_LN);
    f32x4 lnref = {0.f, 0.f, 0.f, 0.f};
    if (has_ln) lnref = bload4(u0.rx, (t0 + 4 * sq) * 4, 0);   // raw channel 0 of this lane's 4 tokens: common shift
    // Weights are not needed before the MFMA loop.  If segment 0 is GroupNorm'ed, its statistics partials are the
    // head of the critical chain (partials -> coefficients -> staging), so they must be requested BEFORE the weights
    // (loads return in order) — but they need the argument block; otherwise the weights go out right away.
    const bool gn0 = (xf0 == XF_GN_SILU || xf0 == XF_GN_LN);
    if (!gn0) issue_w(u0, wv);
    long long* const clkp = AH(clk);
    clk_stamp_p(clkp, w, l, 0);
    auto SV = [&](int s) -> unsigned { return s == 0 ? V.s0 : (s == 1 ? V.s1 : V.s2); };
    int coef_off[3];
    int coef_total = 0;
    for (int s = 0; s < nseg; ++s) {
        coef_off[s] = coef_total;
        const int xf = AS(SV(s), xform), sc = AS(SV(s), C);
        coef_total += ((xf == XF_GN_SILU || xf == XF_GN_LN) ? 2 * sc : 0) + ((xf == XF_LN || xf == XF_GN_LN) ? 2 * sc : 0);
    }
    float* lnred = mainS + coef_total;                      // [KS][32][2]
    float* xt = lnred + KS * 64 + w * (8 * NRMAX * XP);    // this wave's X tile [CB][XP]
    auto make_block = [&](int s, int blk) {
        const unsigned sv = SV(s);
        UBlock u;
        const int sC = AS(sv, C), pitch = AS(sv, x_pitch), taps = AS(sv, taps), bmod = AS(sv, b_mod);
        const int cw = sC / KS;
        const int cb = (cw % 24 == 0) ? 24 : cw;   // host guarantees cw % 24 == 0 or cw in {8, 16}
        const int sb = bmod > 0 ? b % bmod : b;
        u.rx = make_rsrc(AS(sv, x) + (long long)sb * AS(sv, x_bstride), (unsigned)sC * (unsigned)pitch * 4u);
        u.rw = make_rsrc(AS(sv, w4), (unsigned)w_tiles * (unsigned)taps * (unsigned)(sC >> 3) * 1024u);
        u.c0 = w * cw + blk * cb;
        u.nr = cb >> 3;
        u.taps = taps; u.Tin = AS(sv, Tin); u.pitch4 = pitch * 4; u.C8 = sC >> 3; u.xform = AS(sv, xform);
        u.cGN = reinterpret_cast<const float2*>(mainS + coef_off[s]);
        u.cLN = reinterpret_cast<const float2*>(mainS + coef_off[s] + ((u.xform == XF_GN_LN) ? 2 * sC : 0));
        return u;
    };
    auto nblocks = [&](int s) {
        const int cw = AS(SV(s), C) / KS;
        return (cw % 24 == 0) ? cw / 24 : 1;
    };
    auto seg_gnp = [&](int s) {
        const unsigned sv = SV(s);
        GnP p = {AS(sv, gn_cpg), AS(sv, gn_nparts), AS(sv, Tin), AS(sv, gn_eps), AS(sv, gn_gamma), AS(sv, gn_beta)};
        return p;
    };
    GnLoads gl[2];
    rsrc_t grp_rsrc[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int xfs = (s < nseg) ? AS(SV(s), xform) : XF_NONE;
        if (xfs == XF_GN_SILU || xfs == XF_GN_LN) {
            const unsigned sv = SV(s);
            const int bmod = AS(sv, b_mod), sC = AS(sv, C);
            const int sb = bmod > 0 ? b % bmod : b;
            grp_rsrc[s] = make_rsrc(AS(sv, gn_part) + (long long)sb * AS(sv, gn_part_bstride), (unsigned)sC * (unsigned)AS(sv, gn_nparts) * 8u);
            gn_issue(seg_gnp(s), grp_rsrc[s], w * (sC / KS), sC / KS, l, gl[s]);
        }
    }
    if (gn0) issue_w(u0, wv);
    float e_bias[EPRE ? VPW : 1], e_emb[EPRE ? VPW : 1], e_res[EPRE ? VPW : 1];
    const float* const e_biasp = AH(bias);
    const int e_act = AH(act);
    const int res_kind = (EPI == EPI_STORE) ? AH(res_kind) : RES_NONE;
    if (EPRE) {
        const float* embp = AH(emb);
        int erow = 0;
        if (embp) { const int* sp = AH(step_ptr); erow = (sp ? *sp : 0) + b * AH(emb_b_stride); }
        const int emb_pitch = AH(emb_pitch);
        const float* resp = AH(res);
        const long long res_bs = AH(res_bstride);
        const int res_pitch = AH(res_pitch);
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            const int v = w + j * KS;
            const int i = v >> 4, r = v & 15;
            const int nl = (tile0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int t = t0 + lt;
            const bool nok = nl < aN;
            const int ng = nok ? nl : 0;
            e_bias[j] = (e_biasp && nok) ? e_biasp[ng] : 0.f;
            e_emb[j] = (embp && nok) ? embp[(long long)ng * emb_pitch + erow] : 0.f;
            e_res[j] = 0.f;
            if (EPI == EPI_STORE && res_kind != RES_NONE && nok && t < aT)
                e_res[j] = resp[(long long)b * res_bs + (long long)ng * res_pitch + t];
        }
    }
    int band_lo = 0, band_hi = 0;
    if (EPI == EPI_BAND) {
        const int t = t0 + (tid & 31);
        if (t < aT) { band_lo = AB(lo)[t]; band_hi = AB(hi)[t]; }
    }
    clk_stamp_p(clkp, w, l, 1);

    // ================= phase 1: GroupNorm coefficients of the wave's own slice; LN affine =================
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int xfs = (s < nseg) ? AS(SV(s), xform) : XF_NONE;
        if (xfs == XF_GN_SILU || xfs == XF_GN_LN) {
            const int sC = AS(SV(s), C);
            gn_finish(seg_gnp(s), grp_rsrc[s], w * (sC / KS), sC / KS, l, gl[s], gnS, mainS + coef_off[s]);
        }
    }
    if (has_ln) {
        float* cL = mainS + coef_off[0] + ((xf0 == XF_GN_LN) ? 2 * C0 : 0);
        const float* lg = AS(V.s0, ln_gamma);
        const float* lb = AS(V.s0, ln_beta);
        const int cw = C0 / KS;
        for (int c = w * cw + l; c < (w + 1) * cw; c += 64) {
            cL[2 * c] = lg[c];
            cL[2 * c + 1] = lb[c];
        }
    }
    if (EPI == EPI_STORE && res_kind == RES_GN) {
        const int rcpg = AH(res_gn_cpg), rnp = AH(res_gn_nparts);
        const float* part = AH(res_gn_part) + (long long)b * AH(res_gn_part_bstride);
        const rsrc_t rp = make_rsrc(part, (unsigned)aN * (unsigned)rnp * 8u);
        const int c_begin = tile0 * 32, c_end = min(aN, (tile0 + NB) * 32);
        const int g_first = c_begin / rcpg, g_last = (c_end - 1) / rcpg;
        const GnP fake = {rcpg, rnp, aT, AH(res_gn_eps), AH(res_gn_gamma), AH(res_gn_beta)};
        for (int gb = g_first; gb <= g_last; gb += KS) {
            const int grp = min(gb + w, g_last);
            GnLoads L;
            gn_issue(fake, rp, grp * rcpg, rcpg, l, L);
            float* tmp = gnS + 64 * 3 + 32;
            gn_finish(fake, rp, grp * rcpg, rcpg, l, L, gnS, tmp - 2 * grp * rcpg);
            if (gb + w <= g_last && l < rcpg) {
                const int c = grp * rcpg + l;
                if (c >= c_begin && c < c_end) {
                    epiS[c - c_begin] = tmp[2 * l];
                    epiS[32 * NACC + c - c_begin] = tmp[2 * l + 1];
                }
            }
        }
    }
    clk_stamp_p(clkp, w, l, 2);

    // ================= phase 2: LayerNorm statistics from the staged registers =================
    f32x4 mu4 = {0.f, 0.f, 0.f, 0.f}, rs4 = {1.f, 1.f, 1.f, 1.f};
    if (has_ln) {   // single block (host guarantees C/KS == CB), taps == 1
        const bool gnx = xf0 == XF_GN_LN;
        const float ln_eps = AS(V.s0, ln_eps);
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < NRMAX; ++rr) {
            if (rr < u0.nr) {
                const int c = u0.c0 + rr * 8 + sr;
                float2 cg = make_float2(1.f, 0.f);
                if (gnx) cg = u0.cGN[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float vv = gnx ? fmaf(xv[rr][e], cg.x, cg.y) : xv[rr][e];
                    const float d = vv - lnref[e];
                    s1[e] += d;
                    s2[e] = fmaf(d, d, s2[e]);
                }
            }
        }
        // sum over the 8 staging rows (lanes with equal token quad): xor 8 (DPP row_ror 8), 16, 32
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1[e] = dpp_add<0x128>(s1[e]); s2[e] = dpp_add<0x128>(s2[e]);
            s1[e] += __shfl_xor(s1[e], 16); s2[e] += __shfl_xor(s2[e], 16);
            s1[e] += __shfl_xor(s1[e], 32); s2[e] += __shfl_xor(s2[e], 32);
        }
        if (sr == 0) {   // lanes 0..7: tokens 4*sq..+3
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lnred[(w * 32 + 4 * sq + e) * 2] = s1[e];
                lnred[(w * 32 + 4 * sq + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        const float invC = 1.0f / (float)C0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float S1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < KS; ++w2) {
                S1 += lnred[(w2 * 32 + 4 * sq + e) * 2];
                S2 += lnred[(w2 * 32 + 4 * sq + e) * 2 + 1];
            }
            const float md = S1 * invC;
            const float var = fmaxf(S2 * invC - md * md, 0.f);
            mu4[e] = lnref[e] + md;
            rs4[e] = 1.0f / sqrtf(var + ln_eps);
        }
    }
    clk_stamp_p(clkp, w, l, 3);

    // banded cross-attention: this thread's K and V window values, requested before the main loop
    constexpr int BNG = KS * 2, BDPG = (32 / BNG) > 0 ? (32 / BNG) : 1;
    float kq[EPI == EPI_BAND ? BDPG : 1][8], vq[EPI == EPI_BAND ? BDPG : 1][8];
    if (EPI == EPI_BAND) {
        const int kvp = AB(kv_pitch), bwmax = AB(wmax);
        const long long kvo = (long long)b * AB(kv_bstride) + (long long)(tile0 * 32) * kvp;
        const rsrc_t rk = make_rsrc(AB(k) + kvo, 32u * (unsigned)kvp * 4u);
        const rsrc_t rv_ = make_rsrc(AB(v) + kvo, 32u * (unsigned)kvp * 4u);
        const int gi = tid >> 5;
#pragma unroll
        for (int dd = 0; dd < BDPG; ++dd)
#pragma unroll
            for (int wi = 0; wi < 8; ++wi) {
                const bool vis = (wi < bwmax) && (band_lo + wi < band_hi);
                const int vo = vis ? ((gi * BDPG + dd) * kvp + band_lo + wi) * 4 : (int)0x80000000;
                kq[dd][wi] = bload(rk, vo, 0);
                vq[dd][wi] = bload(rv_, vo, 0);
            }
    }

    // ================= phase 3: stage -> LDS, MFMA =================
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    clk_stamp_p(clkp, w, l, 4);

    // transform the staged registers once and write the wave-private LDS tile
    auto stage = [&](const UBlock& u, const f32x4 (&xs)[NRMAX], float hl) {
#pragma unroll
        for (int rr = 0; rr < NRMAX; ++rr) {
            if (rr < u.nr) {
                const int c = u.c0 + rr * 8 + sr;
                float2 gn = make_float2(1.f, 0.f), ln = make_float2(1.f, 0.f);
                if (u.xform == XF_GN_SILU || u.xform == XF_GN_LN) gn = u.cGN[c];
                if (u.xform == XF_LN || u.xform == XF_GN_LN) ln = u.cLN[c];
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = xs[rr][e];
                    switch (u.xform) {
                        case XF_NONE: break;
                        case XF_GN_SILU: v = xf1<XF_GN_SILU>(v, gn, 0.f, 1.f, ln); break;
                        case XF_LN: v = xf1<XF_LN>(v, gn, mu4[e], rs4[e], ln); break;
                        case XF_GN_LN: v = xf1<XF_GN_LN>(v, gn, mu4[e], rs4[e], ln); break;
                        default: v = xf1<XF_SILU>(v, gn, 0.f, 1.f, ln); break;
                    }
                    o[e] = (t0 + 4 * sq + e < u.Tin) ? v : 0.f;
                }
                *reinterpret_cast<f32x4*>(xt + (rr * 8 + sr) * XP + 4 + 4 * sq) = o;
            }
        }
        if (u.taps == 3) {
            const int row = l >> 1, tin = (l & 1) ? (t0 + 32) : (t0 - 1);
            if (row < u.nr * 8) {
                float v = hl;
                if (u.xform == XF_GN_SILU) v = xf1<XF_GN_SILU>(v, u.cGN[u.c0 + row], 0.f, 1.f, make_float2(1.f, 0.f));
                else if (u.xform == XF_SILU) v = xf1<XF_SILU>(v, make_float2(1.f, 0.f), 0.f, 1.f, make_float2(1.f, 0.f));
                xt[row * XP + ((l & 1) ? 36 : 3)] = ((unsigned)tin < (unsigned)u.Tin) ? v : 0.f;
            }
        }
    };
    // pure ds_read + MFMA loop over the block
    auto mma_block = [&](const UBlock& u, const f32x4 (&ws)[TMAX][NRMAX][NACC]) {
        const float* xrow = xt + lh * XP + lt + 3 + ((u.taps == 3) ? 0 : 1);   // col = lt + tap + 4 - pad
#pragma unroll
        for (int tap = 0; tap < TMAX; ++tap) {
            if (tap < u.taps) {
#pragma unroll
                for (int rr = 0; rr < NRMAX; ++rr) {
                    if (rr < u.nr) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float xf = xrow[(rr * 8 + 2 * j) * XP + tap];
#pragma unroll
                            for (int i = 0; i < NACC; ++i) {
                                if (TRANS)
                                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, ws[tap][rr][i][j], acc[i], 0, 0, 0);
                                else
                                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws[tap][rr][i][j], xf, acc[i], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    };

    {
        // flattened (segment, block) list; with one accumulator tile per wave there are registers to spare, so the
        // NEXT block's operands are requested before the current block is staged and multiplied (double buffer)
        int nblk_seg[3] = {0, 0, 0}, nblk_total = 0;
        for (int s = 0; s < nseg; ++s) { nblk_seg[s] = nblocks(s); nblk_total += nblk_seg[s]; }
        auto block_at = [&](int i) {
            int s = 0;
            while (s < 2 && i >= nblk_seg[s]) { i -= nblk_seg[s]; ++s; }
            return make_block(s, i);
        };
        if constexpr (NACC == 1) {
            f32x4 xv2[NRMAX], wv2[TMAX][NRMAX][NACC];
            float halo2 = 0.f;
            UBlock ua = u0, ub = u0;
            for (int i = 0; i < nblk_total; i += 2) {
                const bool has_b = i + 1 < nblk_total;
                if (has_b) { ub = block_at(i + 1); issue_x(ub, xv2, halo2); issue_w(ub, wv2); }
                stage(ua, xv, halo);
                mma_block(ua, wv);
                if (has_b) {
                    if (i + 2 < nblk_total) { ua = block_at(i + 2); issue_x(ua, xv, halo); issue_w(ua, wv); }
                    stage(ub, xv2, halo2);
                    mma_block(ub, wv2);
                }
            }
        } else {
            for (int i = 0; i < nblk_total; ++i) {
                const UBlock u = (i == 0) ? u0 : block_at(i);
                if (i > 0) { issue_x(u, xv, halo); issue_w(u, wv); }
                stage(u, xv, halo);
                mma_block(u, wv);
            }
        }
    }
    clk_stamp_p(clkp, w, l, 6);

    // ================= phase 4: split-K reduction through LDS (fixed order => deterministic) =================
    __syncthreads();
    clk_stamp_p(clkp, w, l, 7);
    float* red = mainS;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((w * NACC + i) * 16 + r) * 64 + l] = acc[i][r];
    __syncthreads();
    clk_stamp_p(clkp, w, l, 8);

    // ================= phase 5: epilogue (same as gemm.hip) =================
    const int nparts_out = (aT + 31) >> 5;
    float* const yp = AH(y);
    const long long y_bs = AH(y_bstride);
    const int y_pitch = AH(y_pitch);
    float* const statsp = (EPI == EPI_STORE) ? AH(stats_out) : nullptr;
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
        const int frow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int tile = tile0 + i;

        if (EPI == EPI_QKV && TRANS) {
            const int nl = tile * 32 + lt;
            const int t = t0 + frow;
            if (e_biasp) val += e_biasp[nl];
            const int vdim = AH(vt_dim);
            const int vn = (tile - AH(vt_first_tile)) * 32 + lt;
            const int h = vn / vdim, d = vn % vdim;
            if (t < aT && nl < aN)
                AH(vt)[(((long long)b * AH(vt_heads) + h) * AH(vt_rows) + t) * vdim + d] = val;
            continue;
        }
        const int nl = tile * 32 + frow;
        const int t = t0 + lt;
        const bool ok = (nl < aN) && (t < aT);
        const int ng = nl;
        if (EPI == EPI_GEGLU) {
            const int ngate = nl + gate_tiles * 32;
            const float xv_ = val + (e_biasp ? e_biasp[nl] : 0.f);
            const float gv = gate + (e_biasp ? e_biasp[ngate] : 0.f);
            if (ok) yp[(long long)b * y_bs + (long long)nl * y_pitch + t] = xv_ * gelu_f(gv);
            continue;
        }
        if (EPI == EPI_BAND) {
            epiS[frow * 32 + lt] = val;
            continue;
        }
        float rv = 0.f;
        if (EPRE) {
            val += e_bias[j];
            if (e_act == ACT_SILU) val = silu_f(val);
            else if (e_act == ACT_GELU) val = gelu_f(val);
            val += e_emb[j];
            rv = e_res[j];
        } else {
            if (nl < aN) {
                if (e_biasp) val += e_biasp[ng];
                if (e_act == ACT_SILU) val = silu_f(val);
                else if (e_act == ACT_GELU) val = gelu_f(val);
                const float* embp = AH(emb);
                if (embp) {
                    const int* sp = AH(step_ptr);
                    const int row = (sp ? *sp : 0) + b * AH(emb_b_stride);
                    val += embp[(long long)ng * AH(emb_pitch) + row];
                }
            }
            if (EPI == EPI_STORE && ok && res_kind != RES_NONE)
                rv = AH(res)[(long long)b * AH(res_bstride) + (long long)ng * AH(res_pitch) + t];
        }
        if (EPI == EPI_STORE) {
            if (ok && res_kind != RES_NONE) {
                if (res_kind == RES_GN) rv = fmaf(rv, epiS[nl - tile0 * 32], epiS[32 * NACC + nl - tile0 * 32]);
                val += rv;
            }
        }
        if (ok) yp[(long long)b * y_bs + (long long)ng * y_pitch + t] = val;
        if (EPI == EPI_STORE && statsp) {
            const float cnt = (float)min(32, aT - t0);
            const float vv = (t < aT) ? val : 0.f;
            const float mean = half32_sum(vv) / cnt;
            const float d = (t < aT) ? (val - mean) : 0.f;
            const float m2 = half32_sum(d * d);
            if (lt == 0 && nl < aN) {
                float* so = statsp + (long long)b * AH(stats_bstride) + ((long long)ng * nparts_out + bx) * 2;
                so[0] = mean;
                so[1] = m2;
            }
        }
    }
    clk_stamp_p(clkp, w, l, 9);

    if (EPI == EPI_BAND) {
        static_assert(EPI != EPI_BAND || NB == 1, "EPI_BAND needs NB == 1");
        constexpr int NG = KS * 2, DPG = 32 / NG;
        const float* qt = epiS;
        float* part = epiS + 32 * 32;
        const int gi = tid >> 5, tt = tid & 31;
        const int t = t0 + tt;
        const bool tv = t < aT;
        const int lo = band_lo, hi = band_hi;
        const int head = tile0;
        const int wmax = AB(wmax);
        const float bscale = AB(scale);
        __syncthreads();
#pragma unroll
        for (int wi = 0; wi < 8; ++wi) {
            if (wi < wmax) {
                float p = 0.f;
#pragma unroll
                for (int dd = 0; dd < DPG; ++dd) p = fmaf(qt[(gi * DPG + dd) * 32 + tt], kq[dd][wi], p);
                part[(gi * 8 + wi) * 32 + tt] = p;
            }
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
            sc[wi] = vis ? sum * bscale : -3.0e38f;
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
                for (int wi = 0; wi < 8; ++wi) o = fmaf(sc[wi] * inv, vq[dd][wi], o);
                yp[(long long)b * y_bs + (long long)(head * 32 + d) * y_pitch + t] = o;
            }
        }
    }
}

template <int NB, int KS, int EPI>
__global__ __launch_bounds__(64 * KS) void ugemm_kernel(const float* hx, const float* hw4, int hpack, int hpitch, int hTin, int hbstride,
                                                        int hbmod_b0, int hT, int hN, int hntiles, int hgate, int hvft, int r1, int r2,
                                                        const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // the 16 leading dwords are the preloaded FastHdr; `a` only reserves the kernarg layout for arg_view()
    const FastHdr hd = {hx, hw4, hpack, hpitch, hTin, hbstride, hbmod_b0, hT, hN, hntiles, hgate, hvft, r1, r2};
    // XCD-aware block order.  Hardware places block id on XCD id % 8; with the natural order every XCD's L2 ends up
    // fetching ALL weights and ALL activations of the launch (rocprofv3 FETCH_SIZE: 4x the algorithmic bytes).  Here
    // each XCD gets a contiguous run of the logical order (n-tile fastest, then batch, then t-tile), i.e. a few whole
    // token tiles: it still needs every weight tile but only its own slice of X.  Placement affects speed only.
    const int ny = hntiles / NB, nz = r1 /* batch count */, nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, q = nwg >> 3, r = nwg & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int by = L % ny, bz = (L / ny) % nz, bx = L / (ny * nz);
    if constexpr (EPI == EPI_QKV) {
        if (by * NB >= hvft) {
            ugemm_body<NB, KS, EPI, true>(hd, smem, bx, by, bz);
            return;
        }
    }
    ugemm_body<NB, KS, EPI, false>(hd, smem, bx, by, bz);
}

template <int NB, int EPI>
static int ugemm_smem_floats(const GemmArgs& a, int KS) {
    constexpr int NACC = (EPI == EPI_GEGLU) ? 2 * NB : NB;
    int coef = 0;
    for (int s = 0; s < a.nseg; ++s) coef += seg_coef_floats(a.seg[s]);
    const int stage = coef + KS * 64 + KS * 8 * NRMAX * XP;
    const int red = KS * NACC * 16 * 64;
    return epi_scratch_floats<NACC>(EPI, KS) + KS * GN_SCRATCH + (stage > red ? stage : red);
}

constexpr int kMaxLds = 160 * 1024;
template <int NB, int KS, int EPI>
static void ulaunch_one(const GemmArgs& a, int batch, hipStream_t s) {
    int smem = ugemm_smem_floats<NB, EPI>(a, KS) * (int)sizeof(float);
    static const int min_lds = getenv("SAID_MIN_LDS") ? atoi(getenv("SAID_MIN_LDS")) : 0;   // experiment: force one workgroup per CU
    if (smem < min_lds) smem = min_lds;
    if (smem > kMaxLds) { fprintf(stderr, "said: ugemm needs %d B of LDS\n", smem); abort(); }
    dim3 grid(((a.T + 31) / 32) * (a.ntiles_per_group / NB) * batch);   // 1-D: decoded XCD-aware in the kernel
    const Seg& s0 = a.seg[0];
    const int pack = s0.C | (s0.taps << 16) | (s0.xform << 20) | (a.nseg << 24);
    const int bmod_b0 = (s0.b_mod & 0xffff) | (a.b0 << 16);
    hipLaunchKernelGGL((ugemm_kernel<NB, KS, EPI>), grid, dim3(64 * KS), smem, s, s0.x, s0.w4, pack, s0.x_pitch, s0.Tin, (int)s0.x_bstride,
                       bmod_b0, a.T, a.N, a.ntiles_per_group, a.geglu_gate_tiles, a.vt_first_tile, batch, 0, a);
}
template <int NB, int KS, int EPI>
static void uconfigure_one() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ugemm_kernel<NB, KS, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
}

// small-batch tile shapes only (<= 2 accumulators per wave keep every weight fragment of a block in
// registers); large batches use the generic kernel's NB = 3..6 shapes
#define SAID_UGEMM_CONFIGS(X)               \
    X(EPI_STORE, 1, 8) X(EPI_STORE, 2, 8)   \
    X(EPI_QKV, 1, 8) X(EPI_QKV, 2, 8)       \
    X(EPI_GEGLU, 1, 8) X(EPI_GEGLU, 2, 8)   \
    X(EPI_BAND, 1, 8)

void configure_ugemm_kernels() {
#define X(E, nb, ks) uconfigure_one<nb, ks, E>();
    SAID_UGEMM_CONFIGS(X)
#undef X
}

// The LDS-staged kernel covers stride-1, k in {1,3}, ungrouped GEMMs whose per-wave channel slice is a
// multiple of 24 (or 8 / 16); everything else stays on the generic kernel.
bool ugemm_supports(const GemmArgs& a, int epi, int NB, int KS) {
    bool cfg = false;
#define X(E, nb, ks) cfg = cfg || (epi == E && NB == nb && KS == ks);
    SAID_UGEMM_CONFIGS(X)
#undef X
    if (!cfg || a.groups != 1 || a.ntiles_per_group % NB) return false;
    if (a.seg[0].x_bstride > 0x7fffffffLL || a.b0 > 0x7fff || a.seg[0].b_mod > 0xffff || a.seg[0].C > 0xffff) return false;
    for (int s = 0; s < a.nseg; ++s) {
        const Seg& sg = a.seg[s];
        if (!sg.w4 || sg.stride != 1 || !(sg.taps == 1 || sg.taps == 3) || sg.pad != (sg.taps - 1) / 2) return false;
        if (sg.taps == 3 && epi != EPI_STORE) return false;
        if (sg.C % KS) return false;
        const int cw = sg.C / KS;
        if (!(cw % 24 == 0 || cw == 8 || cw == 16)) return false;
        if ((sg.xform == XF_GN_SILU || sg.xform == XF_GN_LN) && (cw % sg.gn_cpg || cw > 64)) return false;
        if ((sg.xform == XF_LN || sg.xform == XF_GN_LN) && (sg.taps != 1 || cw > 24 || s != 0)) return false;
    }
    return true;
}

void launch_ugemm(const GemmArgs& a, int epi, int batch, int NB, int KS, hipStream_t s) {
#define X(E, nb, ks) \
    if (epi == E && NB == nb && KS == ks) { ulaunch_one<nb, ks, E>(a, batch, s); return; }
    SAID_UGEMM_CONFIGS(X)
#undef X
    fprintf(stderr, "said: unsupported ugemm config epi=%d NB=%d KS=%d\n", epi, NB, KS);
    abort();
}

}  // namespace said

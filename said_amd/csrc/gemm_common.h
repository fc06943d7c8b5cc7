// gemm_common.h — device helpers shared by the generic GEMM (gemm.hip) and the LDS-staged UNet GEMM (gemm_lds.hip).
#pragma once
#include <cstddef>

#include "kernels.h"

namespace said {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// erf by Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute (about two fp32 ulps at 1): ~12 VALU ops against ~40
// for the library erff, which was 0.7 % of a denoise step in the GEGLU / GELU epilogues.  The UNet's distance to the
// reference goldens is unchanged by it (8.9e-7 of the output range; tolerance 1e-4).
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f)); }
// value * gelu(gate) (GEGLU, attention.py:25-32) with the same A&S 7.1.26 polynomial, without the copysign / 1 + erf / 0.5 round trip:
// gelu(g) = g * Phi(g),  Phi = g < 0 ? q : 1 - q,  q = (p(t) / 2) t exp(-g^2 / 2),  t = 1 / (1 + (0.3275911 / sqrt 2) |g|)  — 14 vector instructions + 2 transcendentals
// per element against 20 + 2 for value * gelu_f(gate) (the fused tails are bound by vector-instruction issue: profiles/r06m_sq_lds_l2_counters.txt), and for g < 0 the
// tail q keeps its own bits instead of passing through 1 - (1 - 2 q).  |error of Phi| <= 0.75e-7 as before.
#ifndef SAID_GEGLU_FORM
#define SAID_GEGLU_FORM 1
#endif
__device__ __forceinline__ float geglu_f(float val, float gate) {
#if SAID_GEGLU_FORM == 0
    return val * gelu_f(gate);
#else
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(gate), 1.0f));
    float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f); p = fmaf(p, t, 0.5f * -0.284496736f); p = fmaf(p, t, 0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f((gate * gate) * (-0.5f * 1.4426950408889634f));
    const float q = (p * t) * e;
    return (val * gate) * (gate < 0.f ? q : 1.0f - q);
#endif
}

// value * gelu(gate) for two elements at a time on packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two floats per lane
// and issue slot).  The GEGLU launch is bound by VALU ISSUE in its MFMA waves' epilogue — 16 products per lane and tile, 19 VALU instructions
// + 2 transcendentals each in the scalar form against 24 MFMAs — so the erf (gemm_common.h: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7)
// is evaluated pairwise and without the copysign / 1 + erf round trip:  gelu(g) = g * (x < 0 ? q : 1 - q),  q = (p(t) / 2) t exp(-x^2),
// x = g / sqrt(2), t = 1 / (1 + 0.3275911 |x|)  (for x < 0 the scalar form computes 1 - (1 - 2q): this one keeps q's own bits).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 geglu2(f32x2 val, f32x2 gate) {
    const f32x2 x = gate * 0.70710678118654752440f;
    const f32x2 ax = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 d = __builtin_elementwise_fma(ax, (f32x2){0.3275911f, 0.3275911f}, one);
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 p = __builtin_elementwise_fma(t, (f32x2){0.5f * 1.061405429f, 0.5f * 1.061405429f}, (f32x2){0.5f * -1.453152027f, 0.5f * -1.453152027f});
    p = __builtin_elementwise_fma(p, t, (f32x2){0.5f * 1.421413741f, 0.5f * 1.421413741f});
    p = __builtin_elementwise_fma(p, t, (f32x2){0.5f * -0.284496736f, 0.5f * -0.284496736f});
    p = __builtin_elementwise_fma(p, t, (f32x2){0.5f * 0.254829592f, 0.5f * 0.254829592f});
    const f32x2 ea = (ax * ax) * -1.4426950408889634f;
    const f32x2 e = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
    const f32x2 q = (p * t) * e;
    const f32x2 omq = one - q;
    const f32x2 sel = {x[0] < 0.f ? q[0] : omq[0], x[1] < 0.f ? q[1] : omq[1]};
    return (val * gate) * sel;
}

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

// ---- argument block access without scalar-memory round trips ----
// GemmArgs is four 256-byte blocks [common][seg0][seg1][seg2].  Each wave fetches them with four coalesced
// vector loads (lane k holds dword k of every block) and reads fields with v_readlane.
struct ArgView { unsigned h, s0, s1, s2; };
__device__ __forceinline__ ArgView arg_view(int lane, int dword_offset = 0) {
    const unsigned* p = (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr() + dword_offset;
    ArgView v;
    v.h = p[lane]; v.s0 = p[64 + lane]; v.s1 = p[128 + lane]; v.s2 = p[192 + lane];
    return v;
}
// only the blocks a kernel variant reads: the common block, plus segments 1 and 2 for multi-segment launches (segment 0
// is fully described by the preloaded header).  Every vector load costs the CU's address path >= 13 clocks and all
// 8 waves fetch the same block, so the three spare loads were ~300 clocks of queueing per workgroup.
template <bool MULTI>
__device__ __forceinline__ ArgView arg_view_hs(int lane, int dword_offset) {
    const unsigned* p = (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr() + dword_offset;
    ArgView v;
    v.h = p[lane]; v.s0 = 0u; v.s1 = 0u; v.s2 = 0u;
    if constexpr (MULTI) { v.s1 = p[128 + lane]; v.s2 = p[192 + lane]; }
    return v;
}
template <typename T>
__device__ __forceinline__ T rl(unsigned v, int dw) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_readlane((int)v, dw));
    } else {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)v, dw), hi = (unsigned)__builtin_amdgcn_readlane((int)v, dw + 1);
        return __builtin_bit_cast(T, (unsigned long long)lo | ((unsigned long long)hi << 32));
    }
}
#define AH(f) rl<decltype(GemmCommon::f)>(V.h, (int)(offsetof(GemmCommon, f) / 4))
#define AB(f) rl<decltype(BandArgs::f)>(V.h, (int)((offsetof(GemmCommon, band) + offsetof(BandArgs, f)) / 4))
#define AS(sv, f) rl<decltype(SegFields::f)>(sv, (int)(offsetof(SegFields, f) / 4))

// optional phase timing (GemmArgs::clk != null): wave w of workgroup (1,0,0) stamps the shader clock
__device__ __forceinline__ void clk_stamp(const GemmArgs& a, int w, int lane, int slot) {
    if (a.clk && blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (lane == 0) a.clk[w * 16 + slot] = (long long)t;
    }
}

// Compiled in only with -DSAID_CLK_STAMPS (SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force, as
// tests/debug_clocks.py's header says): ten inlined stamp sites are ~1 KB of instructions and ten branches on the hot
// path of every wave, and a kernel's first pass through its code is instruction-cache cold.
__device__ __forceinline__ void clk_stamp_p(long long* clk, int w, int lane, int slot) {
#ifndef SAID_CLK_STAMPS
    (void)clk; (void)w; (void)lane; (void)slot;
    return;
#endif
    if (clk && blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 0) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (lane == 0) clk[w * 16 + slot] = (long long)t;
    }
}

// Loads through pointers rebuilt from v_readlane values have no known address space: the compiler emits flat_load,
// which counts on BOTH vmcnt and lgkmcnt and makes every later wait a full drain (s_waitcnt vmcnt(0) lgkmcnt(0)).
// cload: uniform address, read-only data -> constant address space -> s_load.  gload: per-lane address -> global_load.
typedef const float __attribute__((address_space(4))) * cfloat_p;
typedef const int __attribute__((address_space(4))) * cint_p;
typedef const float __attribute__((address_space(1))) * gfloat_p;
__device__ __forceinline__ float cload(const float* p, long long i) { return ((cfloat_p)(unsigned long long)p)[i]; }
__device__ __forceinline__ int cload(const int* p, long long i) { return ((cint_p)(unsigned long long)p)[i]; }
__device__ __forceinline__ float gload(const float* p, long long i) { return ((gfloat_p)(unsigned long long)p)[i]; }
__device__ __forceinline__ void gstore(float* p, long long i, float v) { ((float __attribute__((address_space(1)))*)(unsigned long long)p)[i] = v; }

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// 32-bit-offset load: byte address = base + voff (VGPR) + soff (SGPR) [+ folded immediate]; out-of-range
// offsets (e.g. the t = -1 halo of the first row) return 0 from the hardware bounds check.
__device__ __forceinline__ float bload(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float2 bload2(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

// ---- cross-lane sums on DPP (1 VALU each) instead of ds_bpermute chains ----
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, x);
}
// sum over each 16-lane row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ float half32_sum(float v) { v = row16_sum(v); return v + __shfl_xor(v, 16); }

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
constexpr int GN_SCRATCH = 64 * 3 + 32 + 32;  // per-wave GroupNorm scratch (floats)
template <int NACC>
__host__ __device__ inline int epi_scratch_floats(int epi, int KS) {
    if (epi == EPI_BAND) return 32 * 32 + (KS * 2) * 8 * 32;  // q tile + score partials [groups][wmax<=8][32]
    return 2 * 32 * NACC;                                      // residual GN coefficients
}

// ------------------------------------------------------------------------------------------------
// GroupNorm coefficients of the wave's own channel slice [c_lo, c_lo + cw), cw <= 64, cpg | cw.
// lane <-> (channel, phase): each lane accumulates its channel's Welford partials (relative to the
// group's first partial mean) over every NPH-th 32-token tile with all loads in flight; the 6/12
// channels of a group are then combined through a small per-wave LDS scratch (no shuffle chains).
// scratch: 64*3 + 16*2 floats per wave.
// ------------------------------------------------------------------------------------------------
struct GnLoads { float2 v[10]; float ref; float gamma, beta; };
// GroupNorm partial statistics are stored TILE-major: part[b][tile][channel][2] (mean, M2), `ct` channels per tile row.  A
// consuming wave reads them with lane <-> channel, so one load instruction touches the 1-4 cache lines that hold its 24-64
// consecutive channels of one tile; the channel-major layout of round 1 ([channel][tile]) put every lane on a cache line of
// its own — 10+ loads x 64 lines per wave in 17 kernels of every step.
struct GnP { int gn_cpg, gn_nparts, Tin; float gn_eps; const float* gn_gamma; const float* gn_beta; int ct; };
__device__ __forceinline__ GnP gnp_of(const Seg& sg) { return {sg.gn_cpg, sg.gn_nparts, sg.Tin, sg.gn_eps, sg.gn_gamma, sg.gn_beta, sg.C}; }

__device__ __forceinline__ void gn_issue(const GnP sg, rsrc_t rp, int c_lo, int cw, int lane, GnLoads& L) {
    const int nph = (cw <= 32) ? 2 : 1;
    const int ch = (nph == 2) ? (lane & 31) : lane;
    const int ph = (nph == 2) ? (lane >> 5) : 0;
    const bool chok = ch < cw;
    const int c = c_lo + (chok ? ch : 0);
    // c / cpg for small non-negative ints through one float multiply (exact: the +0.5 keeps the quotient clear of
    // rounding at multiples of cpg); an integer division is ~25 dependent instructions per lane
    const int gfirst = (int)(((float)c + 0.5f) * __builtin_amdgcn_rcpf((float)sg.gn_cpg)) * sg.gn_cpg;
    L.ref = bload(rp, gfirst * 8, 0);   // tile 0 of the group's first channel
    L.gamma = gload(sg.gn_gamma, c);
    L.beta = gload(sg.gn_beta, c);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const int pi = ph + nph * r;
        const bool ok = chok && (pi < sg.gn_nparts);
        L.v[r] = bload2(rp, ok ? (pi * sg.ct + c) * 8 : (int)0x80000000, 0);
    }
}

__device__ __forceinline__ void gn_finish(const GnP sg, rsrc_t rp, int c_lo, int cw, int lane, const GnLoads& L, float* scratch,
                                          float* cA /* interleaved (a,b), indexed by segment channel */) {
    const int nph = (cw <= 32) ? 2 : 1;
    const int ch = (nph == 2) ? (lane & 31) : lane;
    const int ph = (nph == 2) ? (lane >> 5) : 0;
    const bool chok = ch < cw;
    const int c = c_lo + (chok ? ch : 0);
    const int nparts = sg.gn_nparts;
    const int tail = sg.Tin - (nparts - 1) * 32;
    float s1 = 0.f, s2 = 0.f, sm = 0.f;
    // Tokens of partial tile pi = clamp(Tin - 32 pi, 0, 32): 32, the tail, or 0 past the end — as arithmetic on a per-lane base (one subtract + one
    // v_med3 per tile) instead of two compares and two selects; tiles past the end were requested out of range and read as (0, 0), so their
    // M2 adds an exact zero without a select, and cnt = 0 keeps s1 / s2 bit for bit (round 4: the selects were ~60 of this kernel family's 720
    // VALU instructions per wave, on the critical path behind the statistics loads).  Lanes with ch >= cw compute on and are never stored.
    const float cnt_base = (float)(sg.Tin - 32 * ph);
    (void)tail;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const float cnt = __builtin_amdgcn_fmed3f(cnt_base - (float)(32 * nph * r), 0.f, 32.f);
        const float d = L.v[r].x - L.ref;
        s1 = fmaf(cnt, d, s1);
        s2 = fmaf(cnt * d, d, s2);
        sm += L.v[r].y;
    }
    for (int r0 = 10; ph + nph * r0 < nparts; r0 += 10) {  // long sequences: further rounds of 10 tiles
        float2 v[10];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const int pi = ph + nph * (r0 + r);
            const bool ok = chok && (pi < nparts);
            v[r] = bload2(rp, ok ? (pi * sg.ct + c) * 8 : (int)0x80000000, 0);
        }
        const float cb = cnt_base - (float)(32 * nph * r0);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const float cnt = __builtin_amdgcn_fmed3f(cb - (float)(32 * nph * r), 0.f, 32.f);
            const float d = v[r].x - L.ref;
            s1 = fmaf(cnt, d, s1);
            s2 = fmaf(cnt * d, d, s2);
            sm += v[r].y;
        }
    }
    if (nph == 2) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        sm += __shfl_xor(sm, 32);
    }
    float* sc = scratch;            // [64][3]
    if (ph == 0 && chok) { sc[ch * 3] = s1; sc[ch * 3 + 1] = s2; sc[ch * 3 + 2] = sm; }
    const float rcp_cpg = __builtin_amdgcn_rcpf((float)sg.gn_cpg);
    // Round 6: EVERY lane sums its own group's channels (the lanes of a group read the same scratch words: LDS broadcasts) and derives (mean, rstd) itself — the first version
    // let one lane per group do that and hand the pair to the others through LDS: a second write -> read round trip and a serial 3 x cpg-read loop in 4 of 64 lanes on the
    // critical path of every GroupNorm'ed kernel.  Same summation order (q = 0 .. cpg - 1 onto 0), same formulas: bit-identical.
    if (ph == 0 && chok) {
        const int gi = (int)(((float)ch + 0.5f) * rcp_cpg);
        const float* g0 = sc + gi * sg.gn_cpg * 3;
        float S1 = 0.f, S2 = 0.f, SM = 0.f;
        if (sg.gn_cpg == 6) {
#pragma unroll
            for (int q = 0; q < 6; ++q) { S1 += g0[q * 3]; S2 += g0[q * 3 + 1]; SM += g0[q * 3 + 2]; }
        } else {
            for (int q = 0; q < sg.gn_cpg; ++q) { S1 += g0[q * 3]; S2 += g0[q * 3 + 1]; SM += g0[q * 3 + 2]; }
        }
        const float total = (float)sg.gn_cpg * (float)sg.Tin;
        const float inv_total = __builtin_amdgcn_rcpf(total);   // 1-ulp reciprocal / rsqrt: this chain is on every
        const float md = S1 * inv_total;                        // GroupNorm'ed kernel's critical path
        const float var = fmaxf((SM + S2 - total * md * md) * inv_total, 0.f);
        const float mean = L.ref + md;  // (md: mean relative to the group's ref)
        const float av = __builtin_amdgcn_rsqf(var + sg.gn_eps) * L.gamma;
        cA[2 * c] = av;
        cA[2 * c + 1] = L.beta - mean * av;
    }
}


// GroupNorm coefficients of a 48-channel slice (lane <-> channel, 48 lanes active) with up to TWENTY partial tiles in flight per lane
// (T <= 640: one memory round trip; gemm_common.h's gn_issue / gn_finish take ten per round).  Same combination order as there.
struct GnL20 { float2 v[20]; float ref, gamma, beta; };
__device__ __forceinline__ void gn20_issue(const GnP sg, rsrc_t rp, int c_lo, int lane, GnL20& L) {
    const bool chok = lane < 48;
    const int c = c_lo + (chok ? lane : 0);
    const int gfirst = (int)(((float)c + 0.5f) * __builtin_amdgcn_rcpf((float)sg.gn_cpg)) * sg.gn_cpg;
    L.ref = bload(rp, gfirst * 8, 0);
    L.gamma = gload(sg.gn_gamma, c);
    L.beta = gload(sg.gn_beta, c);
#pragma unroll
    for (int r = 0; r < 20; ++r) {
        const bool ok = chok && (r < sg.gn_nparts);
        L.v[r] = bload2(rp, ok ? (r * sg.ct + c) * 8 : (int)0x80000000, 0);
    }
}
__device__ __forceinline__ void gn20_finish(const GnP sg, rsrc_t rp, int c_lo, int lane, const GnL20& L, float* scratch, float* cA) {
    const bool chok = lane < 48;
    const int c = c_lo + (chok ? lane : 0);
    const int nparts = sg.gn_nparts;
    const int tail = sg.Tin - (nparts - 1) * 32;
    float s1 = 0.f, s2 = 0.f, sm = 0.f;
    const float cnt_base = (float)sg.Tin;   // (counts by arithmetic, zeros from the out-of-range loads: see gn_finish)
    (void)tail;
#pragma unroll
    for (int r = 0; r < 20; ++r) {
        const float cnt = __builtin_amdgcn_fmed3f(cnt_base - (float)(32 * r), 0.f, 32.f);
        const float d = L.v[r].x - L.ref;
        s1 = fmaf(cnt, d, s1);
        s2 = fmaf(cnt * d, d, s2);
        sm += L.v[r].y;
    }
    for (int r0 = 20; r0 < nparts; r0 += 10) {   // long sequences: further rounds of 10 tiles
        float2 v[10];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const bool ok = chok && (r0 + r < nparts);
            v[r] = bload2(rp, ok ? ((r0 + r) * sg.ct + c) * 8 : (int)0x80000000, 0);
        }
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const float cnt = __builtin_amdgcn_fmed3f(cnt_base - (float)(32 * (r0 + r)), 0.f, 32.f);
            const float d = v[r].x - L.ref;
            s1 = fmaf(cnt, d, s1);
            s2 = fmaf(cnt * d, d, s2);
            sm += v[r].y;
        }
    }
    float* sc = scratch;            // [64][3]
    if (chok) { sc[lane * 3] = s1; sc[lane * 3 + 1] = s2; sc[lane * 3 + 2] = sm; }
    const float rcp_cpg = __builtin_amdgcn_rcpf((float)sg.gn_cpg);
    if (chok) {   // (every lane sums its own group: gn_finish)
        const int gi = (int)(((float)lane + 0.5f) * rcp_cpg);
        const float* g0 = sc + gi * sg.gn_cpg * 3;
        float S1 = 0.f, S2 = 0.f, SM = 0.f;
        if (sg.gn_cpg == 6) {
#pragma unroll
            for (int q = 0; q < 6; ++q) { S1 += g0[q * 3]; S2 += g0[q * 3 + 1]; SM += g0[q * 3 + 2]; }
        } else {
            for (int q = 0; q < sg.gn_cpg; ++q) { S1 += g0[q * 3]; S2 += g0[q * 3 + 1]; SM += g0[q * 3 + 2]; }
        }
        const float total = (float)sg.gn_cpg * (float)sg.Tin;
        const float inv_total = __builtin_amdgcn_rcpf(total);
        const float md = S1 * inv_total;
        const float var = fmaxf((SM + S2 - total * md * md) * inv_total, 0.f);
        const float mean = L.ref + md;
        const float av = __builtin_amdgcn_rsqf(var + sg.gn_eps) * L.gamma;
        cA[2 * c] = av;
        cA[2 * c + 1] = L.beta - mean * av;
    }
}


}  // namespace said

// stchain.hip — round 5: the per-token tail of a SpatialTransformer block as ONE launch (small batches, fp32 mode).
//
// Everything behind self-attention is per-token work (ldm/attention.py:131-193, 196-234): attn1.to_out + GroupNorm'ed residual, LayerNorm ->
// to_q -> banded cross-attention over the step-invariant audio K / V -> to_out + residual, LayerNorm -> GEGLU -> (proj_out o ff.net.2) + x_in.
// Rounds 1-4 ran it as five launches of 114-228 small workgroups — 5.0 + 6.8 + 5.8 + 9.4 + 10.0 us in situ at Be = 2, T = 600, of which the matrix
// work is a tenth: each launch pays its dispatch, a first memory round trip (~1.2 us: the producer's tile comes from another XCD), wave-start
// skew, a split-K reduction and an epilogue.  Here one 8-wave workgroup owns a 32-token tile of one sample from the attention output to the
// block's output:
//   * activations never leave the CU: every GEMM's B operand is a token-major split-fp16 tile in LDS (two planes h, l of [32 tokens][K]: split_f16.h),
//     written by the previous GEMM's epilogue straight from the accumulator registers (a lane holds 16 channels of ONE token);
//   * weights are pre-split, pre-ordered by the host into one stream per wave in exactly the order the wave consumes them (engine.cpp:
//     pack_chain_stream): 1 KB fragments of v_mfma_f32_32x32x16_f16 A operands, fetched with a 12-deep register ring that runs across GEMM
//     boundaries — the launch is bound by that stream (2.36 MB per workgroup through one CU's 64 B/clk L2 port), nothing else is on the critical path;
//   * waves 0-5 own output columns [32 j, 32 j + 32) of the 192-wide GEMMs (= head j of the cross-attention), all eight share GEGLU's 24 (value, gate)
//     tile pairs; LayerNorm statistics are merged across the six column owners through 1.5 KB of LDS (Chan's update, fixed order); LayerNorm affines are
//     folded into the following GEMM's weights and bias on the host;
//   * the cross-attention window of the tile's tokens (<= 56 keys) is copied once from the key-major copy of K / V (made once per loop) into LDS.
// Products: split-fp16 (x = h + 2^-11 l, three MFMAs per 16 k, fp32 accumulation, cross terms in their own accumulator) — the fp32 mode's arithmetic since
// round 5 (gemm_lds.hip SP).  Unconditional samples of a guided batch skip the cross-attention (x2 = x1 + c2: engine.cpp, run_transformer).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gemm_common.h"
#include "split_f16.h"
#include "stchain.h"
#ifndef SAID_GEGLU_BIAS_INIT
#define SAID_GEGLU_BIAS_INIT 1
#endif

namespace said {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1))) * gf4_p;

// ---- LDS carve (bytes) ----
constexpr int CH_AP = 200;                         // halfs per token row of a 192-wide activation plane (400 B: conflict-free 16-byte reads)
constexpr int CH_HP = 776;                         // halfs per token row of the 768-wide GEGLU product plane (1552 B)
constexpr int CH_KP = 196;                         // floats per key row of the K / V window tiles
constexpr int CH_APL = 32 * CH_AP * 2;             // bytes of one activation plane (12,800)
constexpr int CH_HPL = 32 * CH_HP * 2;             // bytes of one GEGLU product plane (49,664)
constexpr int CH_R0 = 0;                           // GroupNorm scratch -> K / V window tiles [CHAIN_KW][CH_KP] fp32 -> GEGLU product planes (h, l)
constexpr int CH_R1 = CH_R0 + 2 * CH_HPL;          // attention output -> cross-attention output -> LayerNorm3(x2)
constexpr int CH_R2 = CH_R1 + 2 * CH_APL;          // block input tile (fp32) -> LayerNorm2(x1) -> x2
constexpr int CH_VEC = CH_R2 + 2 * CH_APL;         // b1, bq, bo2 | c2, bffp (192 each), bff (1536)
constexpr int CH_GNC = CH_VEC + CHAIN_VEC_FLOATS_LDS * 4;   // GroupNorm (a, b) per channel of x_in
constexpr int CH_LNP = CH_GNC + 384 * 4;           // LayerNorm partials [32 tokens][6 column owners][2]
constexpr int CH_LDS = CH_LNP + 32 * 6 * 2 * 4;
static_assert(CH_LDS <= 160 * 1024, "chain kernel LDS budget");
static_assert(CHAIN_KW * CH_KP * 4 * 2 <= 2 * CH_HPL, "K / V window tiles fit the GEGLU product region");
// The bf16 variant's carve: <= 80 KB and <= 128 VGPRs, so that TWO workgroups share a CU (large batches: 1216 workgroups at 32 clips) and one's un-overlapped phases
// (prologue, LayerNorm exchanges, the band) run under the other's weight stream.  One plane per region; the K / V window tiles in bf16 (43,904 B) leave 5,760 B of the
// GEGLU product region (49,664 B) free until GEGLU starts: the GroupNorm coefficients, the LayerNorm partials and b1 / bq / bo2 | c2 — all dead by then — live there;
// bffp comes from memory (requested before the last barrier), bff keeps its own 6 KB.
template <bool BF> struct Carve {   // fp32 (split-fp16 planes): the constants above
    static constexpr int R0 = CH_R0, R1 = CH_R1, R2 = CH_R2, VEC = CH_VEC, BFF = CH_VEC + 4 * 192 * 4, GNC = CH_GNC, LNP = CH_LNP, LDS = CH_LDS;
};
template <> struct Carve<true> {
    static constexpr int KV = CHAIN_KW * CH_KP * 2 * 2;
    static constexpr int R0 = 0, GNC = KV, LNP = GNC + 384 * 4, VEC = LNP + 32 * 6 * 2 * 4;
    static constexpr int R1 = CH_HPL, R2 = R1 + CH_APL, BFF = R2 + CH_APL, LDS = BFF + 1536 * 4;
    static_assert(VEC + 3 * 192 * 4 <= CH_HPL, "the early tables fit behind the K / V tiles");
    static_assert(LDS <= 80 * 1024, "two workgroups per CU");
};
constexpr int CH_NR_OWNER = 8;                     // ring depths (units of one k16 step: h + l fragments = 2 KB per wave): what is in flight is what bounds the stream's
constexpr int CH_NR_HELPER = 9;                    // rate (latency x bandwidth ~ 160 KB per CU): 6 x 16 + 2 x 18 = 132 KB
constexpr int CH_NR_BF = 6;                        // bf16 variant: 1 KB units; 6 KB per wave in flight x 16 waves of the CU's two workgroups (128 VGPRs)

// units of a wave's stream: [to_out1 12][to_q 12][to_out2 12][GEGLU 3 pairs x 12 steps x (value, gate)][ffproj 60 | 30]; waves 6, 7: [GEGLU 72][ffproj 30].
// The folded proj_out (60 k16 steps per column tile) is the one phase where six column owners on four SIMDs are unbalanced (two SIMDs with two owners: 11.5k clocks of
// MFMA against 5.8k): column tiles 4 and 5 are split over K — waves 4, 5 take steps 0 .. 29, the helper waves 6, 7 steps 30 .. 59 and hand their partial sums over through LDS.
// Slices (round 6, small launches): NS = 3 or 2 workgroups per token tile, each with 1 / NS of the GEGLU / folded-proj_out work — slice c owns GEGLU pairs
// (24 / NS) c .. + 24 / NS - 1 and the matching part of the folded proj_out's K (48 / NS k16 steps over its GEGLU columns + 12 / NS over x2); everything in front of GEGLU is
// computed by every slice.  The stream a workgroup pulls through its CU's L2 port shrinks from 1152 units (2.36 MB) to 684 (NS = 2) / 528 (NS = 3); the partial sums
// meet in memory (see the epilogue).  NS = 3: one pair per wave.  NS = 2: twelve pairs on eight waves — waves 0-3 take two (local pairs w, w + 4), waves 4-7 one
// (w + 4): three pairs per SIMD; the two wave classes have different stream layouts behind GEGLU, hence WC.
// SC = 2 NS + WC (wave class: 0 = waves 0-3 — and every wave when NS != 2 —, 1 = waves 4-7 of a two-slice workgroup).
template <int SC> struct CU {
    static constexpr int NS = SC >> 1, WC = SC & 1;
    static_assert(NS >= 1 && NS <= 3 && (WC == 0 || NS == 2), "slice configuration");
    static constexpr int G1 = 0, G2 = 12, G3 = 24, GE = 36;
    static constexpr int NP0 = NS == 1 ? 3 : (NS == 2 ? 2 : 1), NP1 = NS == 1 ? 3 : 1;   // GEGLU (value, gate) pairs of waves 0-3 / 4-7
    static constexpr int NPAIR = WC ? NP1 : NP0;               // ... of this wave class
    static constexpr int LSTEP = NS == 1 ? 8 : 4, L0 = (NS == 2 && WC) ? 4 : 0;          // local pair of (wave w, pair pi) = w + LSTEP pi + L0
    static constexpr int FF = GE + 24 * NPAIR;                 // first unit of the folded proj_out in this wave class's streams
    static constexpr int FFS = 60 / NS;                        // its k16 steps per column tile in this slice ...
    static constexpr int FFH = 48 / NS;                        // ... of which over the GEGLU product (the rest, 12 / NS, over x2)
    static constexpr int HALF = FFS / 2;                       // steps of waves 4-7 (column tiles 4, 5 are split over K between an owner and a helper wave)
    static constexpr int END = GE + 24 * NP0 + FFS;            // units of waves 0-3 (168 | 114 | 80)
    static constexpr int W45 = GE + 24 * NP1 + HALF, W67 = 24 * NP1 + HALF;   // units of waves 4, 5 (138 | 75 | 70) and 6, 7 (102 | 39 | 34)
    static constexpr int SLICE = 4 * END + 2 * W45 + 2 * W67;  // units of one workgroup's stream (1152 | 684 | 528)
};
static_assert(CU<2>::SLICE == (int)CHAIN_STREAM_UNITS && CU<4>::SLICE == (int)CHAIN2_SLICE_UNITS && CU<6>::SLICE == (int)CHAIN3_SLICE_UNITS, "stchain.h");
constexpr int U_G1 = 0, U_G2 = 12, U_G3 = 24, U_GE = 36;
// MODE 0: column owner, conditional sample; 1: column owner, unconditional (skips to_q / to_out2); 2: helper wave.  A request past the end of the wave's stream (the
// shorter streams of waves 4-7, the ring running ahead at the end) is out of the buffer's range: no memory access, zeros.
template <int MODE> __device__ __forceinline__ constexpr int unit_of(int q) { return MODE == 0 ? q : (MODE == 1 ? (q < U_G2 ? q : q + (U_GE - U_G2)) : q); }
template <int MODE, int SC> __device__ __forceinline__ constexpr int n_units() { return MODE == 0 ? CU<SC>::FF + CU<SC>::FFS : (MODE == 1 ? CU<SC>::FF + CU<SC>::FFS - (U_GE - U_G2) : CU<SC>::W67); }

template <int B, int E, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        sfor<B + 1, E>(f);
    }
}

// (buffer loads: the unit's offset rides in an SGPR / the immediate — with flat pointers the compiler kept a 64-bit VGPR address per in-flight load and spilled the ring)
template <int NR> struct Ring { f32x4 h[NR], l[NR]; };
struct WStream { rsrc_t r; int vo; };   // the wave's stream, lane * 16
// BF (bf16 mode, large batches): ONE plane everywhere — weights and activations rounded to bf16 (RNE), v_mfma_f32_32x32x16_bf16, a unit is 1 KB
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4c __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2c __attribute__((ext_vector_type(2)));
template <int MODE, bool BF, int SC, int NR, int Q>
__device__ __forceinline__ void ring_issue(Ring<NR>& R, const WStream& wp) {
    if constexpr (Q < n_units<MODE, SC>()) {
        constexpr int u = unit_of<MODE>(Q);
        R.h[Q % NR] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wp.r, wp.vo, u * (BF ? 1024 : 2048), 0));
        if constexpr (!BF) R.l[Q % NR] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wp.r, wp.vo + 1024, u * 2048, 0));
    }
}

__device__ __forceinline__ void clk_stamp_c(long long* clk, int w, int lane, int slot) {
#ifndef SAID_CLK_STAMPS
    (void)clk; (void)w; (void)lane; (void)slot;
    return;
#endif
    if (clk && blockIdx.x == 8 && blockIdx.y + 1 == gridDim.y && blockIdx.z == 0) {   // (three-slice launches: slice 0)
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        if (lane == 0) clk[w * 16 + slot] = (long long)t;
    }
}

// one 192-deep (NS = 12) or longer run of k16 steps: B fragments from the token-major planes at `bh` (this lane's row + k-group offset; low plane `pl` bytes behind)
// (the B fragments are double-buffered by hand, one step ahead, and a scheduling fence closes every step: left alone the compiler hoists all 2 NS fragment
//  reads above the MFMAs — 96 registers at NS = 12, which the ring and the accumulators need)
template <int MODE, bool BF, int SC, int NR, int Q0, int NS>
__device__ __forceinline__ void gemm_run(Ring<NR>& R, const WStream& wp, const char* bh, int pl, f32x16& acc, f32x16& accx) {
    f16x8 xh = *reinterpret_cast<const f16x8*>(bh), xl = xh;
    if constexpr (!BF) xl = *reinterpret_cast<const f16x8*>(bh + pl);
    sfor<0, NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        f16x8 nh = xh, nl = xl;
        if constexpr (s + 1 < NS) {
            nh = *reinterpret_cast<const f16x8*>(bh + 32 * (s + 1));
            if constexpr (!BF) nl = *reinterpret_cast<const f16x8*>(bh + pl + 32 * (s + 1));
        }
        if constexpr (BF) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8c, R.h[(Q0 + s) % NR]), __builtin_bit_cast(bf16x8c, xh), acc, 0, 0, 0);
        } else {
            const f16x8 wh = __builtin_bit_cast(f16x8, R.h[(Q0 + s) % NR]), wl = __builtin_bit_cast(f16x8, R.l[(Q0 + s) % NR]);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, accx, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, accx, 0, 0, 0);
        }
        ring_issue<MODE, BF, SC, NR, Q0 + s + NR>(R, wp);
        __builtin_amdgcn_sched_barrier(0);
        xh = nh; xl = nl;
    });
}
// GEGLU: a (value, gate) tile pair shares every B fragment; units alternate value, gate.  `between(s)` runs behind step s's MFMAs (the previous pair's epilogue in
// four pieces: its erf / split VALU work rides under this pair's matrix work, and the weight stream never pauses for an epilogue)
template <int MODE, bool BF, int SC, int NR, int Q0, typename F>
__device__ __forceinline__ void geglu_run(Ring<NR>& R, const WStream& wp, const char* bh, int pl, f32x16& av, f32x16& avx, f32x16& ag, f32x16& agx, F&& between) {
    f16x8 xh = *reinterpret_cast<const f16x8*>(bh), xl = xh;
    if constexpr (!BF) xl = *reinterpret_cast<const f16x8*>(bh + pl);
    sfor<0, 12>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        f16x8 nh = xh, nl = xl;
        if constexpr (s + 1 < 12) {
            nh = *reinterpret_cast<const f16x8*>(bh + 32 * (s + 1));
            if constexpr (!BF) nl = *reinterpret_cast<const f16x8*>(bh + pl + 32 * (s + 1));
        }
        constexpr int qv = Q0 + 2 * s, qg = Q0 + 2 * s + 1;
        if constexpr (BF) {
            av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8c, R.h[qv % NR]), __builtin_bit_cast(bf16x8c, xh), av, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8c, R.h[qg % NR]), __builtin_bit_cast(bf16x8c, xh), ag, 0, 0, 0);
        } else {
            const f16x8 vh = __builtin_bit_cast(f16x8, R.h[qv % NR]), vl = __builtin_bit_cast(f16x8, R.l[qv % NR]);
            const f16x8 gh = __builtin_bit_cast(f16x8, R.h[qg % NR]), gl = __builtin_bit_cast(f16x8, R.l[qg % NR]);
            avx = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, xh, avx, 0, 0, 0);
            agx = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, xh, agx, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, xh, av, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xh, ag, 0, 0, 0);
            avx = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, xl, avx, 0, 0, 0);
            agx = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xl, agx, 0, 0, 0);
        }
        ring_issue<MODE, BF, SC, NR, qv + NR>(R, wp);
        ring_issue<MODE, BF, SC, NR, qg + NR>(R, wp);
        between(sc);
        __builtin_amdgcn_sched_barrier(0);
        xh = nh; xl = nl;
    });
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// accumulator pair -> values: main + 2^-11 cross
__device__ __forceinline__ void merge16(float (&v)[16], const f32x16& acc, const f32x16& accx) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaf(accx[r], 0x1p-11f, acc[r]);
}
// a lane's 16 values (token row `row`, channels col0 + (r & 3) + 8 (r >> 2) [col0 includes 4 lh]) -> split planes: four 8-byte pieces per plane
template <bool BF>
__device__ __forceinline__ void put_split(char* plane_h, int pl, int row_bytes, int col0, const float (&v)[16]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        char* p = plane_h + row_bytes + (col0 + 8 * m) * 2;
        if constexpr (BF) {
            bf16x4c h;
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = (__bf16)v[4 * m + i];
            *reinterpret_cast<bf16x4c*>(p) = h;
        } else {
            f16x4 h, lo;
            split_f16x4(v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3], h, lo);
            *reinterpret_cast<f16x4*>(p) = h;
            *reinterpret_cast<f16x4*>(p + pl) = lo;
        }
    }
}
// a 192-vector in LDS -> the lane's 16 channels (col0 = 32 j + 4 lh)
__device__ __forceinline__ void get_vec(const float* tab, int col0, float (&o)[16]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tab + col0 + 8 * m);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[4 * m + i] = t[i];
    }
}

// LayerNorm statistics of the token's 192 channels, spread over six column owners x two lane halves x 16 registers: per-lane (mean, M2) of 16, merged with
// the other half (Chan), one (mean, M2) of 32 per owner through LDS, merged in owner order by every lane.  Returns (mean, rstd); contains one barrier.
__device__ __forceinline__ float2 ln_stats(const float (&v)[16], float* lnp, int j, int lt, int lh) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[r];
    const float m16 = s * 0.0625f;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float d = v[r] - m16; q = fmaf(d, d, q); }
    const float mo = __shfl_xor(m16, 32), qo = __shfl_xor(q, 32);
    const float dl = mo - m16;
    const float m32 = 0.5f * (m16 + mo), q32 = q + qo + dl * dl * 8.f;
    if (lh == 0) { lnp[(lt * 6 + j) * 2] = m32; lnp[(lt * 6 + j) * 2 + 1] = q32; }
    __syncthreads();
    float mean = lnp[(lt * 6) * 2], M2 = lnp[(lt * 6) * 2 + 1];
#pragma unroll
    for (int k = 1; k < 6; ++k) {
        const float mk = lnp[(lt * 6 + k) * 2], qk = lnp[(lt * 6 + k) * 2 + 1];
        const float d = mk - mean;
        const float n = 32.f * (float)k, nn = n + 32.f;
        mean = fmaf(d, 32.f / nn, mean);
        M2 += qk + d * d * (n * 32.f / nn);
    }
    return make_float2(mean, __builtin_amdgcn_rsqf(M2 * (1.0f / 192.f) + 1e-5f));
}

struct ChainHdr {          // the leading kernel parameters (preloaded into SGPRs): what the first requests need
    const float* wstream;
    const float* o;
    const float* xin;
    const int* lo;         // (the window tile's first key row is lo[first token]: one scalar load from the header instead of two round trips)
    int T, pitch;
    int o_bs, x_bs;        // floats between samples
    int in_mod, n_uncond, wmax;
};

template <int MODE, bool BF, int SC>
__device__ __forceinline__ void chain_body(const ChainHdr& hd, const ChainArgs& a, char* smem, int w, int l, int s_idx, int in_idx, int t0, int slice) {
    using U = CU<SC>;
    constexpr int NSL = U::NS;
    constexpr bool S3 = NSL > 1;   // sliced launch
    constexpr bool COEF = !BF;     // fp32 variants read the block input's finalised GroupNorm coefficients (ChainArgs::gn_coef); the bf16 variant finalises them from the partials
    static_assert(!(BF && S3), "the sliced variants exist for the fp32 mode's small launches only");
    constexpr int U_FF = U::FF, U_HALF = U::HALF, U_END = U::END, U_W45 = U::W45, U_W67 = U::W67;
    constexpr int NR = BF ? CH_NR_BF : ((MODE == 2) ? CH_NR_HELPER : CH_NR_OWNER);
    const int tid = threadIdx.x;
    const int lt = l & 31, lh = l >> 5;
    const int j = w;                       // column owner index (MODE 0 / 1)
    const int col0 = 32 * j + 4 * lh;      // this lane's first channel of the owner's 32
    const int t = t0 + lt;
    const bool tv = t < hd.T;
    using CV = Carve<BF>;
    char* const r1h = smem + CV::R1;
    char* const r2h = smem + CV::R2;
    char* const hh = smem + CV::R0;
    const float* vec = reinterpret_cast<const float*>(smem + CV::VEC);   // b1, bq, bo2 | c2 (fp32 variant: bffp and bff behind)
    const float* gnc = reinterpret_cast<const float*>(smem + CV::GNC);
    float* lnp = reinterpret_cast<float*>(smem + CV::LNP);
    long long* const clk = a.clk;
    const int browA = lt * (CH_AP * 2) + 16 * lh;   // this lane's B-fragment row in a 192-wide plane (bytes)
    const int browH = lt * (CH_HP * 2) + 16 * lh;
    const int wrowA = lt * (CH_AP * 2), wrowH = lt * (CH_HP * 2);
    // this wave's weight stream: column owners 168 units of 2 KB each, helper waves 72
    const int w_units = w < 4 ? U_END : (w < 6 ? U_W45 : U_W67);
    const int w_first = w < 4 ? w * U_END : (w < 6 ? 4 * U_END + (w - 4) * U_W45 : 4 * U_END + 2 * U_W45 + (w - 6) * U_W67);
    constexpr int UB = BF ? 1024 : 2048;   // bytes of a unit
    const WStream wp = {make_rsrc(reinterpret_cast<const char*>(hd.wstream) + ((long long)slice * U::SLICE + w_first) * UB, (unsigned)w_units * (unsigned)UB), l * 16};
    Ring<NR> R;
    clk_stamp_c(clk, w, l, 0);

    int kmin = 0;
    if constexpr (MODE == 0) kmin = cload(hd.lo, t0);
    if constexpr (MODE == 2) {
        // ---- helper waves: the cross-attention window tile of this token tile (conditional samples), then their share of GEGLU ----
        // The key rows lo[t0] .. hi[last token] - 1 (<= CHAIN_KW: engine.cpp set_band) of K and V from the key-major copy: 26 loads per lane requested first thing (one round for
        // the 34 rows of S == T), parked in LDS behind the first barrier — the column owners' operand staging does not wait for them; first read behind the third barrier.
        const bool uncond = s_idx < hd.n_uncond;
        if constexpr (BF) {
            constexpr int KVL = 13;
            f32x4 kvv[KVL];
            int kv_n = 0;
            const int l2 = (w - 6) * 64 + l;
            // bf16 key-major copy (engine.cpp run_kv, bf16 mode): a key row of this block is 384 consecutive bf16 (K then V) = 48 pieces of 16 bytes
            const rsrc_t rkv = make_rsrc(reinterpret_cast<const char*>(a.kvt) + (long long)s_idx * a.kvt_bs * 2, (unsigned)a.S * 1536u * 2u);
            auto kv_walk = [&](int i0, auto&& f) {
                int idx0 = l2 + 128 * i0;
                int key = (int)(((unsigned)idx0 * 43691u) >> 21), f8 = idx0 - key * 48;   // idx0 / 48 (exact for idx0 < 2^17)
                const int lastk = (kv_n - 1) / 48, lastf = (kv_n - 1) - lastk * 48;
    #pragma unroll
                for (int i = 0; i < KVL; ++i) {
                    const bool in = (key * 48 + f8) < kv_n;
                    const int k = in ? key : lastk, p8 = in ? f8 : lastf;
                    f(i, k, p8);
                    f8 += 32; key += 2;
                    if (f8 >= 48) { f8 -= 48; key += 1; }
                }
            };
            auto kv_issue = [&](int i0) {
                kv_walk(i0, [&](int i, int k, int p8) {
                    kvv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rkv, ((kmin + k) * 1536 + a.koff + 8 * p8) * 2, 0, 0));
                });
            };
            auto kv_park = [&](int i0) {
                __bf16* kt = reinterpret_cast<__bf16*>(smem + CH_R0);
                kv_walk(i0, [&](int i, int k, int p8) {
                    const int e = (p8 < 24 ? 0 : CHAIN_KW * CH_KP) + k * CH_KP + 8 * (p8 < 24 ? p8 : p8 - 24);   // (392-byte rows: two 8-byte stores)
                    typedef unsigned int u32x2k __attribute__((ext_vector_type(2)));
                    typedef unsigned int u32x4k __attribute__((ext_vector_type(4)));
                    const u32x4k v = __builtin_bit_cast(u32x4k, kvv[i]);
                    const u32x2k v0 = {v[0], v[1]}, v1 = {v[2], v[3]};
                    *reinterpret_cast<u32x2k*>(kt + e) = v0;
                    *reinterpret_cast<u32x2k*>(kt + e + 4) = v1;
                });
            };
            if (!uncond) {
                kmin = cload(hd.lo, min(t0, hd.T - 1));
                kv_n = (min(cload(hd.lo, min(t0 + 31, hd.T - 1)) + hd.wmax - kmin, CHAIN_KW)) * 48;
                kv_issue(0);
            }
            __syncthreads();   // operands staged
            if (!uncond) {
                kv_park(0);
                for (int i0 = KVL; i0 * 128 < kv_n; i0 += KVL) { kv_issue(i0); kv_park(i0); }
            }
        } else {
            constexpr int KVL = 26;
            f32x4 kvv[KVL];
            int kv_n = 0;
            const int l2 = (w - 6) * 64 + l;
            const rsrc_t rkv = make_rsrc(a.kvt + (long long)s_idx * a.kvt_bs, (unsigned)a.S * 1536u * 4u);
            // flat piece index idx = l2 + 128 i -> (key row, piece of the row's 96): advanced incrementally (a division per piece costs ~100 clocks of VALU beside the
            // owners' MFMAs); pieces past the end repeat the last one (same source, same destination: harmless) instead of being predicated
            auto kv_walk = [&](int i0, auto&& f) {
                int idx0 = l2 + 128 * i0;
                int key = (int)(((unsigned)idx0 * 43691u) >> 22), f4 = idx0 - key * 96;   // idx0 / 96 (exact for idx0 < 2^15)
                const int lastk = (kv_n - 1) / 96, lastf = (kv_n - 1) - lastk * 96;
    #pragma unroll
                for (int i = 0; i < KVL; ++i) {
                    const bool in = (key * 96 + f4) < kv_n;
                    const int k = in ? key : lastk, p4 = in ? f4 : lastf;
                    f(i, k, p4);
                    f4 += 32; key += 1;
                    if (f4 >= 96) { f4 -= 96; key += 1; }
                }
            };
            auto kv_issue = [&](int i0) {
                kv_walk(i0, [&](int i, int k, int p4) {
                    kvv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rkv, ((kmin + k) * 1536 + a.koff + 4 * p4) * 4, 0, 0));
                });
            };
            auto kv_park = [&](int i0) {
                float* kt = reinterpret_cast<float*>(smem + CH_R0);
                kv_walk(i0, [&](int i, int k, int p4) {
                    const int e = (p4 < 48 ? 0 : CHAIN_KW * CH_KP) + k * CH_KP + 4 * (p4 < 48 ? p4 : p4 - 48);
                    if constexpr (BF) {   // bf16 tiles (RNE)
                        bf16x4c b;
    #pragma unroll
                        for (int q = 0; q < 4; ++q) b[q] = (__bf16)kvv[i][q];
                        *reinterpret_cast<bf16x4c*>(reinterpret_cast<__bf16*>(kt) + e) = b;
                    } else {
                        *reinterpret_cast<f32x4*>(kt + e) = kvv[i];
                    }
                });
            };
            if (!uncond) {
                kmin = cload(hd.lo, t0);
                kv_n = (min(cload(hd.lo, min(t0 + 31, hd.T - 1)) + hd.wmax - kmin, CHAIN_KW)) * 96;   // float4 pieces: 48 of K and 48 of V per key row
                kv_issue(0);
            }
            __syncthreads();   // operands staged
            clk_stamp_c(clk, w, l, 1);
            if (!uncond) {
                kv_park(0);
                for (int i0 = KVL; i0 * 128 < kv_n; i0 += KVL) { kv_issue(i0); kv_park(i0); }
            }
        }
        clk_stamp_c(clk, w, l, 2);
        sfor<0, NR>([&](auto qc) { ring_issue<MODE, BF, SC, NR, decltype(qc)::value>(R, wp); });   // (not needed before GEGLU; behind the window tile, whose registers it reuses)
        clk_stamp_c(clk, w, l, 3);
        if (!uncond) {   // LayerNorm2 partials, LayerNorm2(x1) planes, cross-attention output planes
            __syncthreads();
            clk_stamp_c(clk, w, l, 4);
            __syncthreads();
            clk_stamp_c(clk, w, l, 5);
            __syncthreads();
            clk_stamp_c(clk, w, l, 6);
        }
        __syncthreads();   // LayerNorm3 partials
        __syncthreads();   // LayerNorm3(x2) / x2 planes
    } else {
        // ---- requests: statistics, operands, then the weights (loads return in order; the ring is not needed before the operands are staged) ----
        // (first what the preloaded header alone addresses — the rest of the arguments is a scalar-memory round trip away)
        // attention output tile: thread <-> (channel-in-round tid / 8 (48 per round), token quad tid & 7), four rounds
        // (BF: the operands are token-major bf16 already — rows t0 .. t0 + 31 of [sample][row][192] are copied as they are: 768 16-byte pieces per tile, two per thread;
        //  rows past T are out of the buffer's range and read as zeros)
        f32x4 ov[4];
        const int cr = tid >> 3, tq = tid & 7;
        const int prow0 = (tid * 2731) >> 16, ppc0 = tid - 24 * prow0;                      // BF: piece tid -> (row tid / 24, 16-byte piece of the row)
        const int prow1 = ((tid + 384) * 2731) >> 16, ppc1 = (tid + 384) - 24 * prow1;
        if constexpr (BF) {
            const rsrc_t ro = make_rsrc(reinterpret_cast<const char*>(hd.o) + (long long)in_idx * hd.o_bs * 2, (unsigned)hd.T * 384u);
            ov[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ro, (t0 + prow0) * 384 + ppc0 * 16, 0, 0));
            ov[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ro, (t0 + prow1) * 384 + ppc1 * 16, 0, 0));
        } else {
            const rsrc_t ro = make_rsrc(hd.o + (long long)in_idx * hd.o_bs, 192u * (unsigned)hd.pitch * 4u);
#pragma unroll
            for (int rnd = 0; rnd < 4; ++rnd) ov[rnd] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ro, ((48 * rnd + cr) * hd.pitch + t0 + 4 * tq) * 4, 0, 0));
        }
        // the block input's tile (the GroupNorm'ed residual of to_out1) the same way: four 16-byte loads per thread and an fp32 tile in LDS instead of sixteen
        // 4-byte loads per lane in accumulator layout (the CU's address path is the prologue's bottleneck: every load instruction costs it >= 13 clocks)
        f32x4 xv4[4];
        if constexpr (BF) {
            const rsrc_t rxin = make_rsrc(reinterpret_cast<const char*>(hd.xin) + (long long)in_idx * hd.x_bs * 2, (unsigned)hd.T * 384u);
            xv4[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxin, (t0 + prow0) * 384 + ppc0 * 16, 0, 0));
            xv4[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxin, (t0 + prow1) * 384 + ppc1 * 16, 0, 0));
        } else {
            const rsrc_t rxin = make_rsrc(hd.xin + (long long)in_idx * hd.x_bs, 192u * (unsigned)hd.pitch * 4u);
#pragma unroll
            for (int rnd = 0; rnd < 4; ++rnd) xv4[rnd] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxin, ((48 * rnd + cr) * hd.pitch + t0 + 4 * tq) * 4, 0, 0));
        }
        GnL20 gl;
        const GnP gp = {6, a.np, hd.T, 1e-6f, a.gn_gamma, a.gn_beta, 192};
        const rsrc_t rpart = make_rsrc(a.xin_part + (long long)in_idx * a.part_bs, 192u * (unsigned)a.np * 8u);
        float2 cf2 = make_float2(0.f, 0.f);
        if constexpr (COEF) { // fp32 mode: the coefficients were finalised by the block's q/k/v GEMM or its operand preparation (ugemm_kernel EPI_QKV: GemmCommon::gn_coef_out;
                              // prep_kernel: PrepArgs::coef_out; engine.cpp runs gn_coef_kernel otherwise): waves 0-3 fetch 48 channels' (a, b) each — 1 load instead of 23, nothing to finalise
            const rsrc_t rco = make_rsrc(a.gn_coef + (long long)in_idx * a.coef_bs, 192u * 8u);
            if (w < 4) cf2 = bload2(rco, (l < 48) ? (48 * w + l) * 8 : (int)0x80000000, 0);
        } else {
            if (w < 4) gn20_issue(gp, rpart, 48 * w, l, gl);   // GroupNorm partials of x_in: waves 0-3 finalise 48 channels each (eps 1e-6, six channels per group)
        }
        int blo = 0, bhi = 0;
        if constexpr (MODE == 0) {
            const rsrc_t rlo = make_rsrc(hd.lo, (unsigned)hd.T * 4u), rhi = make_rsrc(a.hi, (unsigned)hd.T * 4u);
            blo = __builtin_bit_cast(int, bload(rlo, tv ? t * 4 : (int)0x80000000, 0));   // (tokens past T: an empty window)
            bhi = __builtin_bit_cast(int, bload(rhi, tv ? t * 4 : (int)0x80000000, 0));
        }
        // the vectors b1, bq, bo2 | c2, bffp, bff: requested with everything else, parked in LDS behind the operand tiles (a load-then-store loop here made the whole
        // request phase wait for its own first loads: one memory round trip before the weight ring was even requested)
        f32x4 vecv[2];
        {
            const rsrc_t rvec = make_rsrc(a.vec, (unsigned)CHAIN_VEC_FLOATS * 4u);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = tid + 384 * k;
                // slot 2 of the table is bo2 for conditional samples, c2 for unconditional ones (global layout: b1, bq, bo2, c2, bffp, bff)
                const int gi = (i < 96) ? i : ((i < 144) ? (MODE == 1 ? i + 48 : i) : i + 48);
                vecv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rvec, (i < CHAIN_VEC_FLOATS_LDS / 4) ? gi * 16 : (int)0x80000000, 0, 0));
            }
        }
        clk_stamp_c(clk, w, l, 12);
        if constexpr (BF) {   // both tiles as they are: bf16 rows of 384 bytes at the planes' 400-byte pitch
            *reinterpret_cast<f32x4*>(smem + CV::R1 + prow0 * (CH_AP * 2) + ppc0 * 16) = ov[0];
            *reinterpret_cast<f32x4*>(smem + CV::R1 + prow1 * (CH_AP * 2) + ppc1 * 16) = ov[1];
            *reinterpret_cast<f32x4*>(smem + CV::R2 + prow0 * (CH_AP * 2) + ppc0 * 16) = xv4[0];
            *reinterpret_cast<f32x4*>(smem + CV::R2 + prow1 * (CH_AP * 2) + ppc1 * 16) = xv4[1];
        } else {
            // attention output -> split planes (token-major)
            {
                _Float16* ph = reinterpret_cast<_Float16*>(smem + CH_R1);
#pragma unroll
                for (int rnd = 0; rnd < 4; ++rnd)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = (t0 + 4 * tq + e < hd.T) ? ov[rnd][e] : 0.f;
                        _Float16 hv = (_Float16)x;
                        asm volatile("" : "+v"(hv));   // (one conversion: split_f16.h)
                        ph[(4 * tq + e) * CH_AP + 48 * rnd + cr] = hv;
                        ph[32 * CH_AP + (4 * tq + e) * CH_AP + 48 * rnd + cr] = (_Float16)((x - (float)hv) * 2048.f);
                    }
            }
        }
        clk_stamp_c(clk, w, l, 14);
        // the weight ring is primed HERE, behind the attention tile's staging, not at the end of the request phase: the requests of a CU are a queue (13-23 clocks each), and the
        // ~100 ring loads of the six owners in front of the other waves' operand loads held up the first barrier; the first units still arrive by the time it falls (round 6:
        // bit-identical, headline +0.3 %)
        __builtin_amdgcn_sched_barrier(0);
        sfor<0, NR>([&](auto qc) { ring_issue<MODE, BF, SC, NR, decltype(qc)::value>(R, wp); });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!BF) {   // x_in tile [192][32] fp32 (tokens past T: 0)
            float* xl = reinterpret_cast<float*>(smem + CH_R2);
#pragma unroll
            for (int rnd = 0; rnd < 4; ++rnd) {
                f32x4 v = xv4[rnd];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (t0 + 4 * tq + e < hd.T) ? v[e] : 0.f;
                *reinterpret_cast<f32x4*>(xl + (48 * rnd + cr) * 32 + 4 * tq) = v;
            }
        }
        {   // float4 index i of the LDS order: [0, 144) b1, bq, bo2 | c2; [144, 192) bffp; [192, 576) bff
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = tid + 384 * k;
                if constexpr (BF) {
                    if (i < 144) reinterpret_cast<f32x4*>(smem + CV::VEC)[i] = vecv[k];
                    else if (i >= 192 && i < CHAIN_VEC_FLOATS_LDS / 4) reinterpret_cast<f32x4*>(smem + CV::BFF)[i - 192] = vecv[k];
                } else {
                    if (i < CHAIN_VEC_FLOATS_LDS / 4) reinterpret_cast<f32x4*>(smem + CV::VEC)[i] = vecv[k];
                }
            }
        }
        if constexpr (COEF) {
            if (w < 4 && l < 48) reinterpret_cast<float2*>(smem + CV::GNC)[48 * w + l] = cf2;
        } else {
            if (w < 4) gn20_finish(gp, rpart, 48 * w, l, gl, reinterpret_cast<float*>(smem + CH_R0) + w * GN_SCRATCH, reinterpret_cast<float*>(smem + CV::GNC));
        }
        __syncthreads();
        clk_stamp_c(clk, w, l, 1);
        // ---- to_out1 + GroupNorm'ed residual (attention.py:127, 168, 226-227) ----
        f32x16 acc, accx;
        zero16(acc); zero16(accx);
        gemm_run<MODE, BF, SC, NR, U_G1, 12>(R, wp, r1h + browA, CH_APL, acc, accx);
        clk_stamp_c(clk, w, l, 2);
        float x1[16];
        merge16(x1, acc, accx);
        {
            float b1[16];
            get_vec(vec, col0, b1);
            float xr[16];
            if constexpr (BF) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const bf16x4c xb = *reinterpret_cast<const bf16x4c*>(r2h + wrowA + (col0 + 8 * m) * 2);
#pragma unroll
                    for (int i = 0; i < 4; ++i) xr[4 * m + i] = (float)xb[i];
                }
            } else {
                const float* xl = reinterpret_cast<const float*>(smem + CH_R2) + col0 * 32 + lt;
#pragma unroll
                for (int r = 0; r < 16; ++r) xr[r] = xl[((r & 3) + 8 * (r >> 2)) * 32];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(gnc + 2 * (col0 + 8 * m)), c1 = *reinterpret_cast<const f32x4*>(gnc + 2 * (col0 + 8 * m) + 4);
                x1[4 * m + 0] += b1[4 * m + 0] + fmaf(xr[4 * m + 0], c0[0], c0[1]);
                x1[4 * m + 1] += b1[4 * m + 1] + fmaf(xr[4 * m + 1], c0[2], c0[3]);
                x1[4 * m + 2] += b1[4 * m + 2] + fmaf(xr[4 * m + 2], c1[0], c1[1]);
                x1[4 * m + 3] += b1[4 * m + 3] + fmaf(xr[4 * m + 3], c1[2], c1[3]);
            }
        }
        float x2[16];
        if constexpr (MODE == 1) {   // unconditional half: the cross-attention output is the per-channel constant c2 (engine.cpp)
            float c2[16];
            get_vec(vec + 2 * 192, col0, c2);
#pragma unroll
            for (int r = 0; r < 16; ++r) x2[r] = x1[r] + c2[r];
        } else {
            // ---- LayerNorm2 -> to_q (affine folded into the weights / bias) ----
            const float2 st = ln_stats(x1, lnp, j, lt, lh);
            {
                float y[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = (x1[r] - st.x) * st.y;
                put_split<BF>(r2h, CH_APL, wrowA, col0, y);
            }
            __syncthreads();
            clk_stamp_c(clk, w, l, 3);
            zero16(acc); zero16(accx);
            gemm_run<MODE, BF, SC, NR, U_G2, 12>(R, wp, r2h + browA, CH_APL, acc, accx);
            float q[16];
            merge16(q, acc, accx);
            {
                float bq[16];
                get_vec(vec + 192, col0, bq);
#pragma unroll
                for (int r = 0; r < 16; ++r) q[r] += bq[r];
            }
            clk_stamp_c(clk, w, l, 4);
            // ---- banded cross-attention of head j over the window tile (attention.py:170-191) ----
            float o2[16];
            {
                const int lo = blo, hi = bhi;
                const float* kt = reinterpret_cast<const float*>(smem + CH_R0);
                auto kv4 = [&](int e) -> f32x4 {   // four consecutive elements of the window tiles (element index from the K tile's start; BF: bf16 tiles)
                    if constexpr (BF) {
                        const bf16x4c b = *reinterpret_cast<const bf16x4c*>(reinterpret_cast<const __bf16*>(kt) + e);
                        f32x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = (float)b[q];
                        return o;
                    } else {
                        return *reinterpret_cast<const f32x4*>(kt + e);
                    }
                };
                float sc[8];
                float mx = -3.0e38f;
#pragma unroll
                for (int wi = 0; wi < 8; ++wi) {
                    float p = 0.f;
                    if (wi < hd.wmax) {
                        const int row = min(max(lo - kmin + wi, 0), CHAIN_KW - 1);
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            const f32x4 kv = kv4(row * CH_KP + col0 + 8 * m);
#pragma unroll
                            for (int i = 0; i < 4; ++i) p = fmaf(q[4 * m + i], kv[i], p);
                        }
                    }
                    p += __shfl_xor(p, 32);
                    const bool vis = (wi < hd.wmax) && (lo + wi < hi);
                    sc[wi] = vis ? p * a.scale : -3.0e38f;
                    mx = fmaxf(mx, sc[wi]);
                }
                float den = 0.f;
#pragma unroll
                for (int wi = 0; wi < 8; ++wi) {
                    const bool vis = (wi < hd.wmax) && (lo + wi < hi);
                    sc[wi] = vis ? __expf(sc[wi] - mx) : 0.f;
                    den += sc[wi];
                }
                const float inv = den > 0.f ? 1.0f / den : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) o2[r] = 0.f;
#pragma unroll
                for (int wi = 0; wi < 8; ++wi) {
                    if (wi < hd.wmax) {
                        const int row = min(max(lo - kmin + wi, 0), CHAIN_KW - 1);
                        const float pw = sc[wi] * inv;
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            const f32x4 vv = kv4(CHAIN_KW * CH_KP + row * CH_KP + col0 + 8 * m);
#pragma unroll
                            for (int i = 0; i < 4; ++i) o2[4 * m + i] = fmaf(pw, vv[i], o2[4 * m + i]);
                        }
                    }
                }
            }
            put_split<BF>(r1h, CH_APL, wrowA, col0, o2);
            __syncthreads();
            clk_stamp_c(clk, w, l, 5);
            // ---- to_out2 + x1 ----
            zero16(acc); zero16(accx);
            gemm_run<MODE, BF, SC, NR, U_G3, 12>(R, wp, r1h + browA, CH_APL, acc, accx);
            merge16(x2, acc, accx);
            float bo[16];
            get_vec(vec + 2 * 192, col0, bo);
#pragma unroll
            for (int r = 0; r < 16; ++r) x2[r] += bo[r] + x1[r];
        }
        clk_stamp_c(clk, w, l, 6);
        if (a.dbg_x1) {   // bring-up taps (said_debug_option "st_chain_dbg")
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long o = (long long)s_idx * 192 * hd.pitch + (long long)(col0 + (r & 3) + 8 * (r >> 2)) * hd.pitch + t;
                if (tv) { gstore(a.dbg_x1, o, x1[r]); gstore(a.dbg_x2, o, x2[r]); }
            }
        }
        // ---- LayerNorm3 (folded into GEGLU's weights) and the raw x2 (second K segment of the folded proj_out) as B operands ----
        const float2 st3 = ln_stats(x2, lnp, j, lt, lh);
        {
            float y[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = (x2[r] - st3.x) * st3.y;
            put_split<BF>(r1h, CH_APL, wrowA, col0, y);
            put_split<BF>(r2h, CH_APL, wrowA, col0, x2);
        }
        __syncthreads();
    }
    clk_stamp_c(clk, w, l, 7);
    // ---- GEGLU (attention.py:25-32): pairs w, w + 8, w + 16 ----
    // A pair is 72 MFMAs (2.3k clocks of the SIMD's matrix pipe) and an erf / split epilogue of about as many VALU clocks; run one after the other (25k clocks measured for
    // the phase) the weight stream pauses during every epilogue.  The epilogue of pair pi is therefore executed in four pieces inside pair pi + 1's MFMA loop.
    {
        constexpr int QG = (MODE == 0) ? U_GE : (MODE == 1 ? U_G2 : 0);   // logical position of the wave's first GEGLU unit
        const float* bff = reinterpret_cast<const float*>(smem + CV::BFF);
        float pv[16], pg[16];   // the previous pair's value and gate sums
        const int gp0 = (24 / NSL) * slice;   // sliced: this workgroup's pairs are the hidden tiles gp0 .. gp0 + 24 / NS - 1; its product plane holds them from column 0
        auto lpair = [&](int pi) { return w + U::LSTEP * pi + U::L0; };   // local pair (= column tile of the product plane) of this wave's pair pi
        auto epi_piece = [&](int p, int m) {   // channels 32 p + 4 lh + 8 m .. + 3 of the token: bias, gelu, product, split, 8 bytes per plane
            float hv[4];
#if SAID_GEGLU_BIAS_INIT   // (the pair's accumulators started from the bias: below)
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = geglu_f(pv[4 * m + i], pg[4 * m + i]);
#else
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bff + 32 * (gp0 + p) + 4 * lh + 8 * m), bg = *reinterpret_cast<const f32x4*>(bff + 768 + 32 * (gp0 + p) + 4 * lh + 8 * m);
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = geglu_f(pv[4 * m + i] + bv[i], pg[4 * m + i] + bg[i]);
#endif
            (void)p;
            char* pp = hh + wrowH + (32 * p + 4 * lh + 8 * m) * 2;
            if constexpr (BF) {
                bf16x4c h;
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = (__bf16)hv[i];
                *reinterpret_cast<bf16x4c*>(pp) = h;
            } else {
                f16x4 h, lo;
                split_f16x4(hv[0], hv[1], hv[2], hv[3], h, lo);
                *reinterpret_cast<f16x4*>(pp) = h;
                *reinterpret_cast<f16x4*>(pp + CH_HPL) = lo;
            }
        };
        sfor<0, U::NPAIR>([&](auto pc) {
            constexpr int pi = decltype(pc)::value;
            f32x16 av, avx, ag, agx;
            zero16(avx); zero16(agx);
#if SAID_GEGLU_BIAS_INIT
            // the value and gate sums start from ff.net.0's bias (a lane's 16 channels of the pair's tiles: four 16-byte LDS reads each) instead of from zero: two additions per
            // element leave the epilogue, which is what the fused tails' waves are short of (vector-instruction issue: profiles/r06m_sq_lds_l2_counters.txt)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bff + 32 * (gp0 + lpair(pi)) + 4 * lh + 8 * m), bg = *reinterpret_cast<const f32x4*>(bff + 768 + 32 * (gp0 + lpair(pi)) + 4 * lh + 8 * m);
#pragma unroll
                for (int i = 0; i < 4; ++i) { av[4 * m + i] = bv[i]; ag[4 * m + i] = bg[i]; }
            }
#else
            zero16(av); zero16(ag);
#endif
            geglu_run<MODE, BF, SC, NR, QG + 24 * pi>(R, wp, r1h + browA, CH_APL, av, avx, ag, agx, [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (!BF && pi > 0 && s % 3 == 1) epi_piece(lpair(pi - 1), s / 3);
            });
            clk_stamp_c(clk, w, l, 11 + 2 * pi);
#pragma unroll
            for (int r = 0; r < 16; ++r) { pv[r] = fmaf(avx[r], 0x1p-11f, av[r]); pg[r] = fmaf(agx[r], 0x1p-11f, ag[r]); }
            if constexpr (BF) {   // two workgroups share the CU: the other one's stream runs under this epilogue, and the 32 registers of a carried pair buy ring depth instead
#pragma unroll
                for (int m = 0; m < 4; ++m) epi_piece(lpair(pi), m);
            }
        });
        if constexpr (!BF) {
#pragma unroll
            for (int m = 0; m < 4; ++m) epi_piece(lpair(U::NPAIR - 1), m);
        }
    }
    f32x4 bpv[4];   // BF: bffp of this lane's 16 channels, from memory (global layout of ChainArgs::vec: b1, bq, bo2, c2, bffp, bff)
    if constexpr (BF && MODE != 2) {
        const rsrc_t rvec = make_rsrc(a.vec, (unsigned)CHAIN_VEC_FLOATS * 4u);
#pragma unroll
        for (int m = 0; m < 4; ++m) bpv[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rvec, (4 * 192 + col0 + 8 * m) * 4, 0, 0));
    }
    __syncthreads();
    clk_stamp_c(clk, w, l, 8);
    if constexpr (MODE == 2) {   // steps 30 .. 59 of column tile w - 2: 18 over the GEGLU product, 12 over x2; the partial sums go to the tile's owner through LDS
        constexpr int QH = U_FF - U_GE;
        f32x16 acc, accx;
        zero16(acc); zero16(accx);
        gemm_run<MODE, BF, SC, NR, QH, U::FFH - U_HALF>(R, wp, hh + browH + 32 * U_HALF, CH_HPL, acc, accx);
        gemm_run<MODE, BF, SC, NR, QH + U::FFH - U_HALF, U::FFS - U::FFH>(R, wp, r2h + browA + 32 * (12 / NSL) * slice, CH_APL, acc, accx);   // (sliced: x2's k16 steps (12 / NS) slice ..)
        float* const fpart = reinterpret_cast<float*>(r1h);
#pragma unroll
        for (int r = 0; r < 16; ++r) fpart[((w - 6) * 16 + r) * 64 + l] = fmaf(accx[r], 0x1p-11f, acc[r]);
        __syncthreads();
    }
    if constexpr (MODE != 2) {
        // ---- (proj_out o ff.net.2) over [h ; x2] + x_in, result channel-major + GroupNorm partials of the tile ----
        constexpr int QF = (MODE == 0) ? U_FF : U_FF - (U_GE - U_G2);
        float xr[16];
        const rsrc_t rxin = BF ? make_rsrc(reinterpret_cast<const char*>(hd.xin) + (long long)in_idx * hd.x_bs * 2, (unsigned)hd.T * 384u)
                               : make_rsrc(hd.xin + (long long)in_idx * hd.x_bs, 192u * (unsigned)hd.pitch * 4u);
        const int xvo = BF ? (t * 384 + col0 * 2) : (tv ? (col0 * hd.pitch + t) * 4 : (int)0x80000000);   // (BF: rows past T are out of the buffer's range)
        if constexpr (BF) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bf16x4c xb = __builtin_bit_cast(bf16x4c, __builtin_amdgcn_raw_buffer_load_b64(rxin, xvo + 16 * m, 0, 0));
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[4 * m + i] = (float)xb[i];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) xr[r] = bload(rxin, xvo, ((r & 3) + 8 * (r >> 2)) * hd.pitch * 4);
        }
        f32x16 acc, accx;
        zero16(acc); zero16(accx);
        float y[16], bp[16];
        float* const fpart = reinterpret_cast<float*>(r1h);   // [2 helper waves][16][64]: their partial sums of column tiles 4, 5 (the LayerNorm3 planes are dead)
        if (w < 4) {
            gemm_run<MODE, BF, SC, NR, QF, U::FFH>(R, wp, hh + browH, CH_HPL, acc, accx);
            gemm_run<MODE, BF, SC, NR, QF + U::FFH, U::FFS - U::FFH>(R, wp, r2h + browA + 32 * (12 / NSL) * slice, CH_APL, acc, accx);
            merge16(y, acc, accx);
            __syncthreads();
        } else {
            gemm_run<MODE, BF, SC, NR, QF, U_HALF>(R, wp, hh + browH, CH_HPL, acc, accx);
            merge16(y, acc, accx);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] += fpart[((w - 4) * 16 + r) * 64 + l];
        }
        clk_stamp_c(clk, w, l, 9);
        if constexpr (S3) {
            // The slices' partial sums of this (token tile, column tile) meet in memory, per WAVE: each stores its 32 x 32 partial (agent-scope stores: written
            // through the XCD's L2 — the workgroups of a tile run on different XCDs), waits for the stores, and takes a ticket; the wave that draws the last one reads all
            // of them back and sums them in slice order (bit-reproducible whoever arrives last), then finishes the tile as the one-workgroup kernel does.  No barrier, no
            // fence: the partial buffer and the ticket are only ever touched by agent-scope accesses.
            float* const P = a.part + (((long long)s_idx * a.np + (t0 >> 5)) * NSL) * (6 * 16 * 64);
            int* const tk = a.ticket + ((long long)s_idx * a.np + (t0 >> 5)) * 6 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) __hip_atomic_store(P + ((slice * 6 + j) * 16 + r) * 64 + l, y[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int old = 0;
            if (l == 0) old = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != NSL - 1) return;
            if (l == 0) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (kernel boundaries order it)
            asm volatile("" ::: "memory");
            float ps[NSL][16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c = 0; c < NSL; ++c) ps[c][r] = __hip_atomic_load(P + ((c * 6 + j) * 16 + r) * 64 + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float acc3 = ps[0][r];
#pragma unroll
                for (int c = 1; c < NSL; ++c) acc3 += ps[c][r];
                y[r] = acc3;
            }
        }
        if constexpr (BF) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) bp[4 * m + i] = bpv[m][i];
        } else {
            get_vec(vec + 3 * 192, col0, bp);
        }
        const rsrc_t ryo = BF ? make_rsrc(reinterpret_cast<char*>(a.y) + (long long)s_idx * a.y_bs * 2, (unsigned)hd.T * 384u)
                              : make_rsrc(a.y + (long long)s_idx * a.y_bs, 192u * (unsigned)hd.pitch * 4u);
        float* const so = a.stats_out ? a.stats_out + (long long)s_idx * a.stats_bs + ((long long)(t0 >> 5) * 192) * 2 : nullptr;
        const float cnt = (float)min(32, hd.T - t0);
        const float rcnt = __builtin_amdgcn_rcpf(cnt);
        float mean[16], m2[16];
        if constexpr (BF) {   // token-major bf16 result [sample][row][192]; the statistics are those of the STORED (rounded) values
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                bf16x4c yb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    yb[i] = (__bf16)(y[4 * m + i] + bp[4 * m + i] + xr[4 * m + i]);
                    y[4 * m + i] = (float)yb[i];
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2c, yb), ryo, xvo + 16 * m, 0, 0);   // (rows past T: out of range, dropped)
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y[r] += bp[r] + xr[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, y[r]), ryo, xvo, ((r & 3) + 8 * (r >> 2)) * hd.pitch * 4, 0);   // (xvo is out of range past T: dropped)
            }
        }
        if (so) {   // (all sixteen means, then all sixteen M2s: independent reductions the hardware can overlap)
#pragma unroll
            for (int r = 0; r < 16; ++r) mean[r] = half32_sum(tv ? y[r] : 0.f) * rcnt;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = tv ? (y[r] - mean[r]) : 0.f; m2[r] = half32_sum(d * d); }
            if (lt == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = col0 + (r & 3) + 8 * (r >> 2);
                    gstore(so, 2 * c, mean[r]); gstore(so, 2 * c + 1, m2[r]);
                }
            }
        }
    }
    clk_stamp_c(clk, w, l, 10);
}

// NS: slices (workgroups per token tile): 1, or 2 / 3 for small launches of fp32 mode (CU<> above)
template <bool BF, int NS>
__global__ __launch_bounds__(512, BF ? 4 : 1) void stchain_kernel(const float* h_w, const float* h_o, const float* h_x, const int* h_lo, int h_T, int h_pitch, int h_obs, int h_xbs, int h_inmod,
                                                         int h_nunc_wmax, const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char csmem[];
#ifdef SAID_AB_FLOOR
    if (h_T > 0) return;
#endif
    const ChainHdr hd = {h_w, h_o, h_x, h_lo, h_T, h_pitch, h_obs, h_xbs, h_inmod, h_nunc_wmax & 0xffffff, (int)((unsigned)h_nunc_wmax >> 24)};
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (Confining small launches to fewer XCDs — a padded 1-D grid whose workgroups leave at once on the unwanted XCDs, so that one L2's copy of the weight stream serves more
    //  workgroups — was measured slower at every setting: profiles/r05i_stchain_xcds_ab.txt; and the sample count it read from the kernel arguments in
    //  memory cost every launch a scalar-memory round trip before its first request: 29.9 -> 31.4 us.  The grid is (token tiles, samples[, slices]).)
    const int s_idx = blockIdx.y, t0 = blockIdx.x * 32;
    const int slice = NS > 1 ? (int)blockIdx.z : 0;
    const bool uncond = s_idx < hd.n_uncond;
    const int in_idx = hd.in_mod > 0 ? s_idx % hd.in_mod : s_idx;
    // self-contained roles (each with its own prologue: nothing but scalars is live across this branch, so each gets its own register allocation); two slices:
    // waves 0-3 and 4-7 also differ in their streams' layout (CU<>: WC)
    constexpr int SC0 = 2 * NS, SC1 = NS == 2 ? 2 * NS + 1 : 2 * NS;
    if (w >= 6) chain_body<2, BF, SC1>(hd, a, csmem, w, l, s_idx, in_idx, t0, slice);
    else if (NS == 2 && w >= 4) {
        if (uncond) chain_body<1, BF, SC1>(hd, a, csmem, w, l, s_idx, in_idx, t0, slice);
        else chain_body<0, BF, SC1>(hd, a, csmem, w, l, s_idx, in_idx, t0, slice);
    } else if (uncond) chain_body<1, BF, SC0>(hd, a, csmem, w, l, s_idx, in_idx, t0, slice);
    else chain_body<0, BF, SC0>(hd, a, csmem, w, l, s_idx, in_idx, t0, slice);
}

bool stchain_supports(const ChainArgs& a, int T, int pitch, long long o_bs, long long x_bs) {
    if (a.wmax < 1 || a.wmax > 8) return false;
    if (o_bs > 0x7fffffffLL || x_bs > 0x7fffffffLL || pitch > 0xffff) return false;
    if ((long long)a.S * 1536 * 4 > 0x7ffffff0LL || 192LL * pitch * 4 > 0x7ffffff0LL) return false;
    if (T < 1 || a.np != (T + 31) / 32) return false;
    return true;
}

void launch_stchain(const ChainArgs& a, const float* o, const float* xin, int T, int pitch, long long o_bs, long long x_bs, int in_mod, int n_uncond, int nsamp, hipStream_t s, bool bf16) {
    if (!stchain_supports(a, T, pitch, o_bs, x_bs)) { launch_fault("stchain: unsupported arguments (T %d, window %d)", T, a.wmax); return; }
    dim3 grid((T + 31) / 32, nsamp);
    if (!bf16 && !a.gn_coef) { launch_fault("stchain: the fp32 kernels read the block input's GroupNorm coefficients (ChainArgs::gn_coef)"); return; }
    if (bf16) hipLaunchKernelGGL((stchain_kernel<true, 1>), grid, dim3(512), Carve<true>::LDS, s, a.wstream, o, xin, a.lo, T, pitch, (int)o_bs, (int)x_bs, in_mod, n_uncond | (a.wmax << 24), a);
    else if (a.slices == 3 || a.slices == 2) {
        if (!a.part || !a.ticket) { launch_fault("stchain: a sliced launch needs the partial-sum buffer and the tickets"); return; }
        if (a.slices == 3) hipLaunchKernelGGL((stchain_kernel<false, 3>), dim3(grid.x, grid.y, 3), dim3(512), CH_LDS, s, a.wstream, o, xin, a.lo, T, pitch, (int)o_bs, (int)x_bs, in_mod, n_uncond | (a.wmax << 24), a);
        else hipLaunchKernelGGL((stchain_kernel<false, 2>), dim3(grid.x, grid.y, 2), dim3(512), CH_LDS, s, a.wstream, o, xin, a.lo, T, pitch, (int)o_bs, (int)x_bs, in_mod, n_uncond | (a.wmax << 24), a);
    } else hipLaunchKernelGGL((stchain_kernel<false, 1>), grid, dim3(512), CH_LDS, s, a.wstream, o, xin, a.lo, T, pitch, (int)o_bs, (int)x_bs, in_mod, n_uncond | (a.wmax << 24), a);
}
void configure_stchain_kernel() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stchain_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stchain_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stchain_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stchain_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, Carve<true>::LDS);
}

}  // namespace said

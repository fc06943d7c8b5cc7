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
#include <type_traits>

#include "gemm_common.h"
#include "split_f16.h"

namespace said {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bload4(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int XP = 40;        // LDS tile pitch (floats): col = tin - t0 + 4  (halo at 3 and 36)
// bf16 MFMA mode (BF): the tile is kept token-major in bf16, X[col = tin - t0 + 1][channel], pitch PB halfs, so that the
// B operand of v_mfma_f32_32x32x8_bf16_1k — 4 consecutive channels of one token — is a single ds_read_b64
constexpr int PB = 28;
constexpr int NRMAX = 3;      // row rounds per block: CB = 8 * NR <= 24 channels

template <int XF>
__device__ __forceinline__ float xf1(float v, float2 gn, float mu, float rs, float2 ln) { return xform_apply<XF>(v, gn, mu, rs, ln); }

// 16 dwords of launch-invariant scalars passed as LEADING kernel parameters: with -amdgpu-kernarg-preload-count=16
// the command processor delivers them in SGPRs, so the first operand loads need no memory round trip at all.
struct FastHdr {       // unpacked view of the preloaded kernel parameters
    const float* x;    // segment 0 source (batch 0)
    const float* w4;   // segment 0 packed weights (+ GroupNorm / LayerNorm affine tails)
    int pack;          // C | taps << 16 | xform << 20 | nseg << 24 | (gn_eps == 1e-6) << 26 | tiles per workgroup << 28
    int pitch, T, bstride;   // segment 0 pitch, length (stride-1 conv: Tin == T), batch stride (floats)
    int bmod_b0;       // ny_magic (17 bits) | b0 << 17  (ny_magic = floor(65536 / ny) + 1: block decode without a division)
    int N;
    int gate_vft;      // EPI_GEGLU: gate tile offset; EPI_QKV: number of leading token-major tiles
    // GroupNorm statistics of segment 0: with these the partial loads — the head of the longest dependent chain of
    // a GroupNorm'ed GEMM — go out at kernel entry instead of one memory round trip later
    const float* gn_part;
    int gn_bstride;    // floats between batches of the partials
    int gn_cfg;        // gn_cpg | gn_nparts << 16
};
// The header travels as the 14 leading dwords of the kernel parameters: that is how many the hardware preloads into
// SGPRs (2 of the 16 user SGPRs hold the kernarg pointer).  Anything beyond — and implicit arguments such as gridDim —
// costs a scalar-memory round trip before the first request can be issued, so T|N and pitch|gate_vft share dwords
// and the grid width is passed explicitly.
constexpr int kHdrDwords = 14;

// dwords of one output tile's split-fp16 weights (Seg::ws): per-block layout [C / 24][5 | 2 steps][2 planes][64 lanes][4 dwords]; the flat layout of the
// one-tile-per-wave GEGLU shape, [C / 16][2][64][4], has the same size as the fp32 packing
__host__ __device__ constexpr unsigned sp_tile_dwords(int C, int taps, bool flat = false) {
    return flat ? (unsigned)(C / 16) * 2u * 256u : (unsigned)(C / 24) * (taps == 3 ? 5u : 2u) * 2u * 256u;
}
struct UBlock {   // one (segment, channel block) of this wave
    rsrc_t rx, rw;
    int c0;        // first channel of the block (segment-relative)
    int taps, Tin, pitch4, C8 /* C / 8 */;
    int xform;
    const float2* cGN;   // LDS coefficient tables, indexed by segment channel
    const float2* cLN;
};

// VAR (compile-time launch variant): which conditional load groups exist.  Every vector-memory request of the request
// phase is then UNCONDITIONAL code (lanes/groups that do not apply use an out-of-range offset and get 0 back), so the
// compiler can count outstanding loads exactly.  With loads inside run-time branches it falls back to
// s_waitcnt vmcnt(0) and e.g. the GroupNorm coefficients wait for the epilogue operands requested after them: one
// extra memory round trip on the critical path (measured 21.7k -> see profiles/ for the phase clocks).
enum UVar : int { UV_T3 = 1 /* segment 0 is a 3-tap conv */, UV_GN0 = 2 /* GroupNorm'ed segment 0 */,
                  UV_GN1 = 4 /* GroupNorm'ed segment 1 (concatenated skip) */, UV_RGN = 8 /* GroupNorm'ed residual */,
                  UV_MULTI = 16 /* more than one K segment: the argument blocks of segments 1, 2 are fetched */,
                  UV_DEEP = 32 /* single segment whose per-wave K slice spans several 24-channel blocks (FF out) */,
                  UV_DUP = 64 /* the result is stored twice: y and y2 (+ a per-channel constant), kernels.h GemmCommon::y2 */ };

// SP (round 5, fp32 mode's default at small batch): the products run on SPLIT-fp16 operands (split_f16.h: x = h + 2^-11 l) — three v_mfma_f32_32x32x16_f16 (8 passes
// each, 16 k) per eight v_mfma_f32_32x32x2_f32 (16 passes each, 2 k): 5.3 x fewer matrix-pipe clocks, fp32 accumulation (cross terms in their own accumulators),
// as close to a float64 evaluation as the fp32 MFMAs.  The 16-deep MFMA wants 8 CONSECUTIVE k per lane, so the wave-private tile is token-major: two fp16 planes
// (h, l) [34 token rows][24 channels] (48-byte rows: conflict-free 16-byte fragment reads), written once while staging — the split costs 4 VALU instructions per
// ELEMENT, not per use: the three taps of a convolution are three row offsets into the same tile, and K-group g = 3 tap + r (8 channels) of a row is simply
// halfs 8 g .. 8 g + 7 from the start of row `token` (the tile IS the im2col row).  A k16 step takes groups (2 s, 2 s + 1) for the two lane halves; a block's
// 9 (3 taps) or 3 (1 tap) groups are padded to 5 / 2 steps with zero weights (Seg::ws, engine.cpp: pack_rows_split).  Reduction and epilogues are untouched.
template <int NB, int KS, int EPI, int VAR, bool TRANS, bool BF, bool MT, bool SP = false>
__device__ __forceinline__ void ugemm_body(const FastHdr& hd, float* smem, int bx, int by, int bz) {
    // MT (multi-tile, large batches): the workgroup walks over up to `tt_run` consecutive 32-token tiles of one sample.
    // Weights, arguments and GroupNorm coefficients are fetched / finalised ONCE and stay in registers / LDS; per tile
    // only X (requested one tile ahead), the per-tile epilogue operands and the LayerNorm statistics are new.  At large
    // batch the per-tile weight re-fetch (147 of 171 KB for a two-tile conv workgroup) is what saturates a CU's load
    // path.  Single-block launches only (the host checks).
    // BF: operands are rounded to bf16 (weights on the host, activations after the fused transform) and multiplied with
    // v_mfma_f32_32x32x8_bf16_1k; everything else — statistics, transforms, accumulation, epilogues — stays fp32.
    static_assert(!(SP && (BF || MT)), "split-fp16 products: fp32 mode, single-tile workgroups");
    using WT = std::conditional_t<BF, float2, f32x4>;   // one lane's weight fragment of an 8-channel round (SP: of one (k16 step, plane))
    constexpr int WB = BF ? 512 : 1024;                  // bytes per (tile, tap, 8-channel round) in the packed weights (SP: per (step, plane))
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    constexpr int SPR = 34, SPP = 24;                    // SP tile: token rows (32 + 2 halo), halfs per row
    constexpr int NACC = (EPI == EPI_GEGLU) ? 2 * NB : NB;
    constexpr int NV = NB * 16;
    static_assert(NV % KS == 0, "NB*16 must be divisible by KS");
    constexpr int VPW = NV / KS;
    constexpr bool EPRE = (VPW <= 4) && (EPI == EPI_STORE || EPI == EPI_QKV) && !TRANS;
    constexpr bool T3 = (VAR & UV_T3) != 0, GN0 = (VAR & UV_GN0) != 0, GN1 = (VAR & UV_GN1) != 0, RGN = (VAR & UV_RGN) != 0;
    constexpr bool MULTI = (VAR & UV_MULTI) != 0, DEEP = (VAR & UV_DEEP) != 0, DUP = (VAR & UV_DUP) != 0;
    constexpr bool ONE_BLOCK = MT || (!MULTI && !DEEP);   // exactly one (segment, block) per wave: no block loop at all
    static_assert(!GN1 || MULTI, "a GroupNorm'ed segment 1 implies several segments");
    constexpr bool HAS_LN = (EPI != EPI_STORE);   // q/k/v, GEGLU and band projections read LayerNorm'ed input
    static_assert(EPI == EPI_STORE || (VAR & ~UV_GN0) == 0, "variants other than GN0 exist only for EPI_STORE");
    static_assert(!(MULTI && DEEP), "DEEP describes single-segment launches");
    static_assert(EPI != EPI_QKV || GN0, "q/k/v reads GroupNorm -> LayerNorm input");
    constexpr int TMAX = T3 ? 3 : 1;
    constexpr int NSX = T3 ? 5 : 2;   // SP: k16 steps of a 24-channel block (9 / 3 K-groups of 8, padded)
    constexpr bool SPFLAT = SP && EPI == EPI_GEGLU && NB == 4 && !MT && KS == 8 && !(VAR & (UV_MULTI | UV_DEEP));   // == NSPL below: one tile per wave over the whole K
    const int tid = threadIdx.x, l = tid & 63, lt = l & 31, lh = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = bz + (hd.bmod_b0 >> 17);
    const int tt_run = MT ? ((hd.pack >> 28) & 15) : 1;   // tiles per workgroup
    const int ntt_all = (hd.T + 31) >> 5;
    const int ntr = MT ? min(tt_run, ntt_all - bx * tt_run) : 1;
    int t0 = bx * tt_run * 32;   // first token of the current tile
    const int tile0 = by * NB;
    const int nseg = MULTI ? ((hd.pack >> 24) & 3) : 1, aT = hd.T, aN = hd.N;
    const int gate_tiles = (EPI == EPI_GEGLU) ? hd.gate_vft : 0;
    const int ntiles = (aN + 31) >> 5;
    const int w_tiles = (EPI == EPI_GEGLU) ? ntiles + gate_tiles : ntiles;
    const int sr = l >> 3, sq = l & 7;   // staging map: row-in-round, token quad
    const int C0 = hd.pack & 0xffff, taps0 = T3 ? 3 : 1, xf0 = (hd.pack >> 20) & 15;

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
        const int sb = b;   // b_mod (all samples reading sample b % b_mod) exists only on the generic kernel: the host checks
        u0.rx = make_rsrc(hd.x + (long long)sb * hd.bstride, (unsigned)C0 * (unsigned)hd.pitch * 4u);
        // (SP: [tile][C0 / 24 blocks][NSX steps][2 planes] x 1024 bytes — or, flat (GEGLU): [tile][C0 / 16 steps][2 planes])
        u0.rw = SP ? make_rsrc(hd.w4, (unsigned)w_tiles * sp_tile_dwords(C0, taps0, SPFLAT) * 4u)
                   : make_rsrc(hd.w4, (unsigned)w_tiles * (unsigned)taps0 * (unsigned)(C0 >> 3) * (unsigned)WB);
        u0.c0 = w * cw;
        u0.taps = taps0; u0.Tin = hd.T; u0.pitch4 = hd.pitch * 4; u0.C8 = C0 >> 3; u0.xform = xf0;
        u0.cGN = reinterpret_cast<const float2*>(mainS);
        u0.cLN = reinterpret_cast<const float2*>(mainS + (GN0 ? 2 * C0 : 0));
    }

    // raw X slice of a block -> registers: NR dwordx4 (row sr of each round, tokens t0+4*sq..+3) + halo dwords.
    // `valid` = false turns every request into an out-of-range one (software-pipeline tail).
    auto issue_x_at = [&](const UBlock& u, f32x4 (&xv)[NRMAX], float& halo, bool valid, int tb) {
#pragma unroll
        for (int rr = 0; rr < NRMAX; ++rr) {
            const int oor = valid ? 0 : (int)0x80000000;   // scalar select, no branch
            xv[rr] = bload4(u.rx, (sr * u.pitch4 + (tb + 4 * sq) * 4) | oor, (u.c0 + rr * 8) * u.pitch4);
        }
        halo = 0.f;
        if constexpr (T3) {   // lane -> (row = l >> 1, side = l & 1): token tb-1 or tb+32
            const int row = l >> 1, tin = (l & 1) ? (tb + 32) : (tb - 1);
            const bool ok = valid && (u.taps == 3) && (row < NRMAX * 8) && ((unsigned)tin < (unsigned)u.Tin);
            halo = bload(u.rx, ok ? (row * u.pitch4 + tin * 4) : (int)0x80000000, u.c0 * u.pitch4);
        }
    };
    auto issue_x = [&](const UBlock& u, f32x4 (&xv)[NRMAX], float& halo, bool valid) { issue_x_at(u, xv, halo, valid, t0); };
    // With 8 accumulator tiles the weight fragments of a whole block (3 rounds x 8 tiles x 4 registers) no longer fit
    // beside the accumulators: ROLL keeps two rounds in flight and requests round 2 into round 0's registers once
    // round 0 has been multiplied (it lands during round 1's 32 MFMAs).
    // NSPL (GEGLU, 4 value + 4 gate tiles, fp32): instead of splitting K over the 8 waves — which needs a 2 x 128 KB
    // reduction through LDS for the 8 tiles — every wave owns ONE output tile over the whole K = 192.  The X tile is
    // staged cooperatively (the wave-private staging tiles are contiguous, i.e. already one [192][XP] array), each wave
    // keeps its tile's 24 weight fragments in registers, and the only exchange is the gate tiles handed to the value
    // waves (16 KB).  Same MFMA count per wave; 8x the ds_read traffic in the main loop, 1/16 of the reduction traffic.
#ifdef SAID_NO_NSPLIT
    constexpr bool NSPL = false;
#else
    constexpr bool NSPL = (EPI == EPI_GEGLU && NB == 4 && !MT && KS == 8 && !(VAR & (UV_MULTI | UV_DEEP)));
#endif
    static_assert(!SP || NSPL == SPFLAT, "the flat split-fp16 weight layout is the one-tile-per-wave shape's");
    constexpr bool ROLL = (NACC >= 8) && !BF && !NSPL;   // bf16 fragments are half the size: all three rounds fit
    static_assert(!(SP && ROLL), "split-fp16 products: no rolling weight rounds (such shapes stay on the fp32 MFMAs)");
    static_assert(!(MT && ROLL), "multi-tile mode keeps every weight fragment in registers");
    static_assert(!ROLL || TMAX == 1, "rolling weight rounds are for 1-tap GEMMs");
    constexpr int WR = ROLL ? 2 : NRMAX;
    constexpr int WD0 = SP ? NSX : TMAX, WD1 = SP ? 2 : WR;   // weight fragments of a block: [tap][round] or, SP, [step][plane]
    auto wload = [&](rsrc_t r, int oor, int so) -> WT {
        if constexpr (BF) return bload2(r, (l * 8) | oor, so);
        else return bload4(r, (l * 16) | oor, so);   // the range check covers voffset only: soffset is don't-care when masked
    };
    auto issue_w_round = [&](const UBlock& u, int rr, WT (&wr)[NACC], bool valid) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const int oor = valid ? 0 : (int)0x80000000;
            wr[i] = wload(u.rw, oor, (tile_wo[i] * u.C8 + (u.c0 >> 3) + rr) * WB);
        }
    };
    auto issue_w = [&](const UBlock& u, WT (&wv)[WD0][WD1][NACC], bool valid) {
        if constexpr (SP) {
            const int ns_u = (u.taps == 3) ? 5 : 2, nblk = (u.C8 * 0x5556) >> 16 /* C / 24 */, bi = (u.c0 * 0xAAB) >> 16 /* c0 / 24 */;
#pragma unroll
            for (int st = 0; st < NSX; ++st)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int i = 0; i < NACC; ++i) {
                        const int oor = (valid && st < ns_u) ? 0 : (int)0x80000000;
                        wv[st][pl][i] = wload(u.rw, oor, (((tile_wo[i] * nblk + bi) * ns_u + st) * 2 + pl) * 1024);
                    }
        } else if constexpr (ROLL) {
            issue_w_round(u, 0, wv[0][0], valid);
            issue_w_round(u, 1, wv[0][1], valid);
        } else {
#pragma unroll
            for (int tap = 0; tap < TMAX; ++tap)
#pragma unroll
                for (int rr = 0; rr < NRMAX; ++rr)
#pragma unroll
                    for (int i = 0; i < NACC; ++i) {
                        const int oor = (valid && (tap < u.taps)) ? 0 : (int)0x80000000;
                        wv[tap][rr][i] = wload(u.rw, oor, ((tile_wo[i] * u.taps + tap) * u.C8 + (u.c0 >> 3) + rr) * WB);
                    }
        }
    };

    // ================= phase 0: requests =================
    f32x4 xv[NRMAX];
    WT wv[WD0][WD1][NACC];
    float halo;
    // request order matters (vector loads return in order): the statistics partials of a GroupNorm'ed segment 0 head the
    // longest chain (partials -> coefficients -> staging) and need only the header; then the rest of the argument
    // block, the operands, and last the weights, which are not needed before the MFMA loop
    GnLoads gl0, gl1, glr;
    rsrc_t grp_rsrc0 = u0.rx, grp_rsrc1 = u0.rx, grp_rsrcr = u0.rx;
    GnP gp0 = {1, 1, hd.T, 1e-5f, nullptr, nullptr, 0};
    if constexpr (GN0) {
        const float* gb = hd.w4 + (SP ? (long long)w_tiles * sp_tile_dwords(C0, taps0, SPFLAT) : (long long)w_tiles * taps0 * (C0 >> 3) * (WB / 4));   // gamma[C0], beta[C0] behind the weights
        gp0 = {hd.gn_cfg & 0xffff, hd.gn_cfg >> 16, hd.T, ((hd.pack >> 26) & 1) ? 1e-6f : 1e-5f, gb, gb + C0, C0};
        const int sb = b;
        grp_rsrc0 = make_rsrc(hd.gn_part + (long long)sb * hd.gn_bstride, (unsigned)C0 * (unsigned)gp0.gn_nparts * 8u);
        gn_issue(gp0, grp_rsrc0, w * (C0 / KS), C0 / KS, l, gl0);
    }
    const ArgView V = arg_view_hs<MULTI>(l, kHdrDwords);   // common block (+ segments 1, 2): coalesced loads, fields via v_readlane
    issue_x(u0, xv, halo, true);
    WT wn[NSPL ? 24 : 1];   // NSPL: this wave's tile, all 24 eight-channel rounds (SP: 12 k16 steps x 2 planes, flat layout)
    if constexpr (NSPL) {
        const int tw = (w < NB) ? (tile0 + w) : (tile0 + (w - NB) + gate_tiles);
#pragma unroll
        for (int rr = 0; rr < 24; ++rr) wn[rr] = wload(u0.rw, 0, (tw * u0.C8 + rr) * WB);   // (SP: entry rr = 2 * step + plane of the same 24 KB per tile)
    } else {
        if constexpr (!(HAS_LN && ONE_BLOCK && !MT)) issue_w(u0, wv, true);   // (LayerNorm'ed single-tile launches: behind the statistics barrier, see phase 2)
    }
    long long* const clkp = AH(clk);
    clk_stamp_p(clkp, w, l, 0);
    auto SV = [&](int s) -> unsigned { return s == 1 ? V.s1 : V.s2; };   // segment 0 lives in the header
    auto seg_coefs = [&](int s) {   // segments 1, 2 only (segment 0 comes from the header)
        const int xf = AS(SV(s), xform), sc = AS(SV(s), C);
        return ((xf == XF_GN_SILU || xf == XF_GN_LN) ? 2 * sc : 0) + ((xf == XF_LN || xf == XF_GN_LN) ? 2 * sc : 0);
    };
    const int coef_off0 = 0;
    const int coef_off1 = (GN0 ? 2 * C0 : 0) + (HAS_LN ? 2 * C0 : 0);
    const int coef_off2 = coef_off1 + (nseg > 1 ? seg_coefs(1) : 0);
    const int coef_total = coef_off2 + (nseg > 2 ? seg_coefs(2) : 0);
    auto coef_off = [&](int s) { return s == 0 ? coef_off0 : (s == 1 ? coef_off1 : coef_off2); };
    float* lnred = mainS + coef_total;                      // [KS][32][2]
    float* xt = lnred + KS * 64 + w * (8 * NRMAX * XP);    // this wave's X tile [CB][XP]
    auto make_block = [&](int s, int blk) {
        if (s == 0) {   // further channel blocks of segment 0 (K slices wider than 24 channels)
            UBlock u = u0;
            u.c0 = w * (C0 / KS) + blk * 24;
            return u;
        }
        const unsigned sv = SV(s);
        UBlock u;
        const int sC = AS(sv, C), pitch = AS(sv, x_pitch), taps = AS(sv, taps);
        const int cw = sC / KS;
        constexpr int cb = 24;   // every block is three 8-channel rounds (the host guarantees cw % 24 == 0)
        const int sb = b;
        u.rx = make_rsrc(AS(sv, x) + (long long)sb * AS(sv, x_bstride), (unsigned)sC * (unsigned)pitch * 4u);
        u.rw = SP ? make_rsrc(AS(sv, ws), (unsigned)w_tiles * sp_tile_dwords(sC, taps) * 4u)
                  : make_rsrc(BF ? AS(sv, w2) : AS(sv, w4), (unsigned)w_tiles * (unsigned)taps * (unsigned)(sC >> 3) * (unsigned)WB);
        u.c0 = w * cw + blk * cb;
        u.taps = taps; u.Tin = AS(sv, Tin); u.pitch4 = pitch * 4; u.C8 = sC >> 3; u.xform = AS(sv, xform);
        u.cGN = reinterpret_cast<const float2*>(mainS + coef_off(s));
        u.cLN = reinterpret_cast<const float2*>(mainS + coef_off(s) + ((u.xform == XF_GN_LN) ? 2 * sC : 0));
        return u;
    };
    auto nblocks = [&](int s) {
        const int cw = (s == 0 ? C0 : AS(SV(s), C)) / KS;
        return cw / 24;
    };
    auto seg_gnp = [&](int s) {
        const unsigned sv = SV(s);
        GnP p = {AS(sv, gn_cpg), AS(sv, gn_nparts), AS(sv, Tin), AS(sv, gn_eps), AS(sv, gn_gamma), AS(sv, gn_beta), AS(sv, C)};
        return p;
    };
    GnP gp1 = gp0, gpr = gp0;
    int sC1 = 0;
    if constexpr (GN1) {   // segment 1 (concatenated skip input of a ResBlock): GroupNorm parameters from the argument block
        const unsigned sv = SV(1);
        sC1 = AS(sv, C);
        const int sb = b;
        gp1 = seg_gnp(1);
        grp_rsrc1 = make_rsrc(AS(sv, gn_part) + (long long)sb * AS(sv, gn_part_bstride), (unsigned)sC1 * (unsigned)gp1.gn_nparts * 8u);
        gn_issue(gp1, grp_rsrc1, w * (sC1 / KS), sC1 / KS, l, gl1);
    }
    // GroupNorm'ed residual (SpatialTransformer: x_in of attention.py:226 is the un-normalised input, the residual
    // of attn1 inside the block is norm(x)): wave w finalises group g_first + w of the workgroup's channel range
    int rg_first = 0, rg_last = 0, rg_grp = 0;
    if constexpr (RGN) {
        const int rcpg = AH(res_gn_cpg), rnp = AH(res_gn_nparts);
        const float* part = AH(res_gn_part) + (long long)b * AH(res_gn_part_bstride);
        grp_rsrcr = make_rsrc(part, (unsigned)aN * (unsigned)rnp * 8u);
        const int c_begin = tile0 * 32, c_end = min(aN, (tile0 + NB) * 32);
        const float rcp_rcpg = __builtin_amdgcn_rcpf((float)rcpg);   // small-int division by one float multiply
        rg_first = (int)(((float)c_begin + 0.5f) * rcp_rcpg);
        rg_last = (int)(((float)(c_end - 1) + 0.5f) * rcp_rcpg);   // host guarantees rg_last - rg_first < KS
        rg_grp = min(rg_first + w, rg_last);
        gpr = {rcpg, rnp, aT, AH(res_gn_eps), AH(res_gn_gamma), AH(res_gn_beta), aN};
        gn_issue(gpr, grp_rsrcr, rg_grp * rcpg, rcpg, l, glr);
    }
    float ln_g = 0.f, ln_b = 0.f;
    if constexpr (HAS_LN) {   // LayerNorm affine of the wave's channel slice (<= 24 channels): lane c
        const int cw = C0 / KS;
        // gamma[C0], beta[C0] sit behind the weights (after the GroupNorm pair, if any)
        const float* lnp = hd.w4 + (SP ? (long long)w_tiles * sp_tile_dwords(C0, taps0, SPFLAT) : (long long)w_tiles * taps0 * (C0 >> 3) * (WB / 4)) + (GN0 ? 2 * C0 : 0);
        const rsrc_t rg = make_rsrc(lnp, (unsigned)C0 * 8u);
        const int vo = (l < cw) ? (w * cw + l) * 4 : (int)0x80000000;
        ln_g = bload(rg, vo, 0);
        ln_b = bload(rg, vo, C0 * 4);
    }
    float e_bias[EPRE ? VPW : 1], e_emb[EPRE ? VPW : 1], e_res[EPRE ? VPW : 1], e_add2[(EPRE && DUP) ? VPW : 1];
    const float* const e_biasp = AH(bias);
    const int e_act = AH(act);
    const int res_kind = (EPI == EPI_STORE) ? AH(res_kind) : RES_NONE;
    // null-safe descriptors: a null pointer becomes an empty range, every load from it returns 0
    const int nbias = (EPI == EPI_GEGLU) ? (gate_tiles * 32 + aN) : aN;
    const rsrc_t r_bias = make_rsrc(e_biasp, e_biasp ? (unsigned)nbias * 4u : 0u);
    const bool has_res = (EPI == EPI_STORE) && res_kind != RES_NONE;
    const int res_pitch = AH(res_pitch);
    const rsrc_t r_res = make_rsrc(AH(res) + (has_res ? (long long)b * AH(res_bstride) : 0LL), has_res ? (unsigned)aN * (unsigned)res_pitch * 4u : 0u);
    int band_lo = 0, band_hi = 0;
    // operands that change from tile to tile: residual rows and the alignment window of the tile's queries
    auto issue_tile_operands = [&](int tb) {
        if (EPRE) {
#pragma unroll
            for (int j = 0; j < VPW; ++j) {
                const int v = w + j * KS;
                const int i = v >> 4, r = v & 15;
                const int nl = (tile0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int t = tb + lt;
                e_res[j] = bload(r_res, (nl < aN && t < aT) ? (nl * res_pitch + t) * 4 : (int)0x80000000, 0);
            }
        }
        if constexpr (EPI == EPI_BAND) {
            const int t = tb + (tid & 31);
            const rsrc_t rlo = make_rsrc(AB(lo), (unsigned)aT * 4u), rhi = make_rsrc(AB(hi), (unsigned)aT * 4u);
            band_lo = __builtin_bit_cast(int, bload(rlo, t * 4, 0));
            band_hi = __builtin_bit_cast(int, bload(rhi, t * 4, 0));
        }
    };
    if (EPRE) {
        // Unconditional scalar loads: an absent bias / embedding table / step counter is read from the start of the
        // weight block instead (always mapped, finite) and multiplied by 0 — a pointer select costs two scalar ops, a
        // null test per load costs a branch and 64-bit compares, and this code runs in front of every wave's MFMAs.
        const float* embp0 = AH(emb);
        const int* sp0 = AH(step_ptr);
        const float* bp = e_biasp ? e_biasp : hd.w4;
        const float* mp = embp0 ? embp0 : hd.w4;
        const float bsc = e_biasp ? 1.f : 0.f, msc = embp0 ? 1.f : 0.f;
        const int mhas = embp0 ? 1 : 0;
        const int step_now = cload(sp0 ? sp0 : reinterpret_cast<const int*>(hd.w4), 0) * (sp0 ? 1 : 0);
        const int erow = (step_now + b * AH(emb_b_stride)) * mhas;
        const int emb_pitch = AH(emb_pitch) * mhas;
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            const int v = w + j * KS;
            const int i = v >> 4, r = v & 15;
            // per-row constants: two uniform addresses per row (lane halves) -> scalar loads, which do not occupy the
            // CU's vector address path; rows past N read row N-1 (never stored)
            const int na = min((tile0 + i) * 32 + (r & 3) + 8 * (r >> 2), aN - 1), nb = min(na + 4, aN - 1);
            const float b_a = cload(bp, na) * bsc, b_b = cload(bp, nb) * bsc;
            const float m_a = cload(mp, (long long)na * emb_pitch + erow) * msc, m_b = cload(mp, (long long)nb * emb_pitch + erow) * msc;
            e_bias[j] = lh ? b_b : b_a;
            e_emb[j] = lh ? m_b : m_a;
            if constexpr (DUP) {   // per-channel constant of the second copy (absent: read the weight block, times 0)
                const float* a2 = AH(y2_add);
                const float* ap = a2 ? a2 : hd.w4;
                const float asc = a2 ? 1.f : 0.f;
                const float a_a = cload(ap, na) * asc, a_b = cload(ap, nb) * asc;
                e_add2[j] = lh ? a_b : a_a;
            }
        }
    }
    issue_tile_operands(t0);
    // GEGLU: the value and gate biases of the workgroup's tiles go through LDS (one load per thread, requested here)
    float geglu_bias = 0.f;
    if constexpr (EPI == EPI_GEGLU) {
        const int half = tid / (32 * NB), k = tid % (32 * NB);
        const int nl = tile0 * 32 + k;
        geglu_bias = bload(r_bias, (tid < 64 * NB && nl < aN) ? (nl + half * gate_tiles * 32) * 4 : (int)0x80000000, 0);
    }
    clk_stamp_p(clkp, w, l, 1);

    // ================= phase 1: GroupNorm coefficients of the wave's own slice; LN affine =================
    if constexpr (GN0) gn_finish(gp0, grp_rsrc0, w * (C0 / KS), C0 / KS, l, gl0, gnS, mainS + coef_off(0));
    if constexpr (GN0 && EPI == EPI_QKV) {   // the sample's coefficients for stchain_kernel (one workgroup per sample; every wave its own channel slice)
        float* const co = AH(gn_coef_out);
        if (co && bx == 0 && by == 0 && l < 2 * (C0 / KS)) gstore(co, (long long)b * AH(gn_coef_bs) + 2 * w * (C0 / KS) + l, (mainS + coef_off(0))[2 * w * (C0 / KS) + l]);
    }
    if constexpr (GN1) gn_finish(gp1, grp_rsrc1, w * (sC1 / KS), sC1 / KS, l, gl1, gnS, mainS + coef_off(1));
    if constexpr (HAS_LN) {
        float* cL = mainS + coef_off(0) + (GN0 ? 2 * C0 : 0);
        const int cw = C0 / KS;
        if (l < cw) {
            cL[2 * (w * cw + l)] = ln_g;
            cL[2 * (w * cw + l) + 1] = ln_b;
        }
    }
    if constexpr (RGN) {
        const int rcpg = gpr.gn_cpg;
        const int c_begin = tile0 * 32, c_end = min(aN, (tile0 + NB) * 32);
        float* tmp = gnS + 64 * 3 + 32;
        gn_finish(gpr, grp_rsrcr, rg_grp * rcpg, rcpg, l, glr, gnS, tmp - 2 * rg_grp * rcpg);
        if (rg_first + w <= rg_last && l < rcpg) {
            const int c = rg_grp * rcpg + l;
            if (c >= c_begin && c < c_end) {
                epiS[c - c_begin] = tmp[2 * l];
                epiS[32 * NACC + c - c_begin] = tmp[2 * l + 1];
            }
        }
    }
    clk_stamp_p(clkp, w, l, 2);

    // next tile's X (MT): requested before the current tile is staged
    f32x4 xvn[MT ? NRMAX : 1];
    float halon = 0.f;
    for (int ti = 0; ti < ntr; ++ti) {
    if constexpr (MT) {
        const bool more = ti + 1 < ntr;
        issue_x_at(u0, xvn, halon, more, t0 + 32);
    }
    // ================= phase 2: LayerNorm statistics from the staged registers =================
    f32x4 mu4 = {0.f, 0.f, 0.f, 0.f}, rs4 = {1.f, 1.f, 1.f, 1.f};
    if constexpr (HAS_LN) {   // single block (host guarantees C/KS == CB), taps == 1
        constexpr bool gnx = GN0;
        const float ln_eps = 1e-5f;   // host guarantees (ugemm_supports)
        // Round 6: (mean, M2) pairs merged with Chan's update all the way — per lane over its NRMAX values of a token (two passes over registers), over the eight
        // lanes that hold the token's other channels of this wave's block, over the KS waves through LDS.  (Rounds 1-5 summed d = v - ref and d^2 with ref = the RAW
        // channel 0 of the token and took E[d^2] - E[d]^2: behind a GroupNorm the values are O(1) while the raw residual stream is O(100) on trained-like weights, and
        // the subtraction cancelled five digits — tests/test_gpu_round6.py, the trained-like fill: 2e-3 of the output range instead of 1e-5.)
        f32x4 m4, q4;
        {
            float vg[NRMAX][4];
#pragma unroll
            for (int rr = 0; rr < NRMAX; ++rr) {
                const int c = u0.c0 + rr * 8 + sr;
                float2 cg = make_float2(1.f, 0.f);
                if (gnx) cg = u0.cGN[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) vg[rr][e] = gnx ? fmaf(xv[rr][e], cg.x, cg.y) : xv[rr][e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sm = 0.f;
#pragma unroll
                for (int rr = 0; rr < NRMAX; ++rr) sm += vg[rr][e];
                m4[e] = sm * (1.0f / (float)NRMAX);
                float qq = 0.f;
#pragma unroll
                for (int rr = 0; rr < NRMAX; ++rr) { const float d = vg[rr][e] - m4[e]; qq = fmaf(d, d, qq); }
                q4[e] = qq;
            }
        }
        // the 8 staging rows (lanes with equal token quad): xor 8, 16, 32 — equal counts n on both sides: mean' = (ma + mb) / 2, M2' = M2a + M2b + (mb - ma)^2 n / 2
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const float half_n = 0.5f * (float)(NRMAX << st);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float mo = __shfl_xor(m4[e], 8 << st), qo = __shfl_xor(q4[e], 8 << st);
                const float d = mo - m4[e];
                q4[e] = q4[e] + qo + d * d * half_n;
                m4[e] = 0.5f * (m4[e] + mo);
            }
        }
        if (sr == 0) {   // lanes 0..7: tokens 4*sq..+3
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lnred[(w * 32 + 4 * sq + e) * 2] = m4[e];
                lnred[(w * 32 + 4 * sq + e) * 2 + 1] = q4[e];
            }
        }
        __syncthreads();
        const float invC = 1.0f / (float)C0;
        const float nw = (float)(8 * NRMAX);   // values per wave and token
        // Round 6: a lane merges the KS partials of ONE token (lane & 31) and the wave shares the 32 (mean, rstd) pairs through its own scratch — every lane used to merge all
        // four tokens of its quad, the same four as the seven other staging rows of the wave: 4 x 7 merges (~170 VALU instructions) per lane on the critical path of every
        // LayerNorm'ed GEMM, 3/4 of them redundant.  Same merge order per token: bit-identical.
        {
            const int tk = l & 31;
            float mean = lnred[tk * 2], M2 = lnred[tk * 2 + 1];
#pragma unroll
            for (int w2 = 1; w2 < KS; ++w2) {   // fixed order
                const float mk = lnred[(w2 * 32 + tk) * 2], qk = lnred[(w2 * 32 + tk) * 2 + 1];
                const float d = mk - mean;
                const float n = nw * (float)w2, nn = n + nw;
                mean = fmaf(d, nw / nn, mean);
                M2 += qk + d * d * (n * nw / nn);
            }
            const float rstd = __builtin_amdgcn_rsqf(M2 * invC + ln_eps);   // v_rsq_f32 (1 ulp): the IEEE 1/sqrt sequence is ~40 VALU ops
            float* const lst = gnS;   // (the wave's GroupNorm scratch is free by now; same-wave LDS accesses are ordered: no barrier)
            if (l < 32) { lst[2 * tk] = mean; lst[2 * tk + 1] = rstd; }
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(lst + 8 * sq), p1 = *reinterpret_cast<const f32x4*>(lst + 8 * sq + 4);
            mu4[0] = p0[0]; rs4[0] = p0[1]; mu4[1] = p0[2]; rs4[1] = p0[3];
            mu4[2] = p1[0]; rs4[2] = p1[1]; mu4[3] = p1[2]; rs4[3] = p1[3];
        }
        // Round 6: the weights of a LayerNorm'ed launch are requested HERE, behind the statistics barrier.  The request phase is a queue (a CU takes 13-23 clocks per
        // vector-memory instruction: the second wave of every SIMD trails the first by the length of that queue), and this barrier comes early: with the 12 weight requests per
        // wave out of its way the late waves reach it sooner; the weights still have the staging (2-3k clocks) to arrive.  Same loads: bit-identical; headline +0.5 %
        // (the convolutions have no early barrier: there the order is neutral, profiles/r06g_kconv_ab.txt).
        if constexpr (!NSPL && ONE_BLOCK && !MT) {
            __builtin_amdgcn_sched_barrier(0);
            issue_w(u0, wv, true);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    clk_stamp_p(clkp, w, l, 3);

    // banded cross-attention: this thread's K and V window values, requested before the main loop
    constexpr int BNG = KS * 2, BDPG = (32 / BNG) > 0 ? (32 / BNG) : 1;
    f32x4 kraw[EPI == EPI_BAND ? BDPG : 1][2], vraw[EPI == EPI_BAND ? BDPG : 1][2];
    if (EPI == EPI_BAND) {
        // the window of query t is keys lo .. lo+wmax-1, contiguous in a K/V row: two dwordx4 per row instead of eight
        // dword loads (the second one is masked out of range when the band is at most 4 wide); the registers are only
        // touched again in the band epilogue
        const int kvp = AB(kv_pitch), bwmax = AB(wmax);
        const long long kvo = (long long)b * AB(kv_bstride) + (long long)(tile0 * 32) * kvp;
        const rsrc_t rk = make_rsrc(AB(k) + kvo, 32u * (unsigned)kvp * 4u);
        const rsrc_t rv_ = make_rsrc(AB(v) + kvo, 32u * (unsigned)kvp * 4u);
        const int gi = tid >> 5;
        const int oor_hi = (bwmax > 4) ? 0 : (int)0x80000000;
#pragma unroll
        for (int dd = 0; dd < BDPG; ++dd) {
            const int vo = ((gi * BDPG + dd) * kvp + band_lo) * 4;
            kraw[dd][0] = bload4(rk, vo, 0); kraw[dd][1] = bload4(rk, (vo + 16) | oor_hi, 0);
            vraw[dd][0] = bload4(rv_, vo, 0); vraw[dd][1] = bload4(rv_, (vo + 16) | oor_hi, 0);
        }
    }

    // ================= phase 3: stage -> LDS, MFMA =================
    f32x16 acc[NACC], accx[SP ? NACC : 1];   // accx: the split mode's cross terms (h.l + l.h, x 2^11)
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[i][r] = 0.f;
            if (SP) accx[SP ? i : 0][r] = 0.f;
        }
    clk_stamp_p(clkp, w, l, 4);

    // transform the staged registers once and write the wave-private LDS tile
    // (one copy per transform: a run-time switch per element compiled to a jump table per value)
    auto stage_t = [&](auto xfc, const UBlock& u, const f32x4 (&xs)[NRMAX], float hl) {
        constexpr int XF = decltype(xfc)::value;
        constexpr bool GNX = (XF == XF_GN_SILU || XF == XF_GN_LN), LNX = (XF == XF_LN || XF == XF_GN_LN);
#pragma unroll
        for (int rr = 0; rr < NRMAX; ++rr) {
            {
                const int c = u.c0 + rr * 8 + sr;
                float2 gn = make_float2(1.f, 0.f), ln = make_float2(1.f, 0.f);
                if constexpr (GNX) gn = u.cGN[c];
                if constexpr (LNX) ln = u.cLN[c];
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = xf1<XF>(xs[rr][e], gn, mu4[e], rs4[e], ln);
                    o[e] = (t0 + 4 * sq + e < u.Tin) ? v : 0.f;
                }
                if constexpr (SP) {
                    _Float16* xh = reinterpret_cast<_Float16*>(xt);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 hv = (_Float16)o[e];
                        asm volatile("" : "+v"(hv));   // (the stored half and the one the remainder is taken from must be ONE conversion: split_f16.h)
                        xh[(1 + 4 * sq + e) * SPP + rr * 8 + sr] = hv;
                        xh[(SPR + 1 + 4 * sq + e) * SPP + rr * 8 + sr] = (_Float16)((o[e] - (float)hv) * 2048.f);
                    }
                } else if constexpr (BF) {
                    __bf16* xb = reinterpret_cast<__bf16*>(xt);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xb[(1 + 4 * sq + e) * PB + rr * 8 + sr] = (__bf16)o[e];
                } else {
                    *reinterpret_cast<f32x4*>(xt + (rr * 8 + sr) * XP + 4 + 4 * sq) = o;
                }
            }
        }
        if constexpr (T3) {
            if (u.taps == 3) {
                const int row = l >> 1, tin = (l & 1) ? (t0 + 32) : (t0 - 1);
                if (row < NRMAX * 8) {
                    float2 gn = make_float2(1.f, 0.f);
                    if constexpr (GNX) gn = u.cGN[u.c0 + row];
                    const float v = xf1<XF>(hl, gn, 0.f, 1.f, make_float2(1.f, 0.f));
                    const float hv = ((unsigned)tin < (unsigned)u.Tin) ? v : 0.f;
                    if constexpr (SP) {
                        _Float16* xh = reinterpret_cast<_Float16*>(xt);
                        _Float16 hh = (_Float16)hv;
                        asm volatile("" : "+v"(hh));
                        xh[((l & 1) ? 33 : 0) * SPP + row] = hh;
                        xh[(SPR + ((l & 1) ? 33 : 0)) * SPP + row] = (_Float16)((hv - (float)hh) * 2048.f);
                    } else if constexpr (BF) reinterpret_cast<__bf16*>(xt)[((l & 1) ? 33 : 0) * PB + row] = (__bf16)hv;
                    else xt[row * XP + ((l & 1) ? 36 : 3)] = hv;
                }
            }
        }
    };
    auto stage = [&](const UBlock& u, const f32x4 (&xs)[NRMAX], float hl) {
        if constexpr (EPI == EPI_QKV) {
            stage_t(std::integral_constant<int, XF_GN_LN>{}, u, xs, hl);
        } else if constexpr (HAS_LN) {
            stage_t(std::integral_constant<int, XF_LN>{}, u, xs, hl);
        } else {
            if constexpr (GN0 && !MULTI) {   // the only segment is the GroupNorm'ed one
                stage_t(std::integral_constant<int, XF_GN_SILU>{}, u, xs, hl);
            } else {
                if ((GN0 || GN1) && u.xform == XF_GN_SILU) stage_t(std::integral_constant<int, XF_GN_SILU>{}, u, xs, hl);
                else if (u.xform == XF_SILU) stage_t(std::integral_constant<int, XF_SILU>{}, u, xs, hl);
                else stage_t(std::integral_constant<int, XF_NONE>{}, u, xs, hl);
            }
        }
    };
    // pure ds_read + MFMA loop over the block
    auto mma_block = [&](const UBlock& u, WT (&ws)[WD0][WD1][NACC]) {
        if constexpr (SP) {
            // row `token` of the tile is the im2col row of that token: K-group g at halfs 8 g; 1-tap blocks start one row down (no left halo)
            const _Float16* xh = reinterpret_cast<const _Float16*>(xt) + (lt + ((u.taps == 3) ? 0 : 1)) * SPP;
            const int ns_u = (u.taps == 3) ? 5 : 2, ng = (u.taps == 3) ? 9 : 3;
#pragma unroll
            for (int st = 0; st < NSX; ++st) {
                if (st < ns_u) {
                    const int g = min(2 * st + lh, ng - 1);   // (the padding group re-reads the last one: finite data against zero weights)
                    const f16x8 fh = *reinterpret_cast<const f16x8*>(xh + 8 * g);
                    const f16x8 fl = *reinterpret_cast<const f16x8*>(xh + SPR * SPP + 8 * g);
#pragma unroll
                    for (int i = 0; i < NACC; ++i) {
                        const f16x8 wh = __builtin_bit_cast(f16x8, ws[st][0][i]), wl = __builtin_bit_cast(f16x8, ws[st][1][i]);
                        if (TRANS) {
                            accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, wl, accx[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, wh, acc[i], 0, 0, 0);
                            accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, wh, accx[i], 0, 0, 0);
                        } else {
                            accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fh, accx[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fh, acc[i], 0, 0, 0);
                            accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fl, accx[i], 0, 0, 0);
                        }
                    }
                }
            }
            return;
        } else if constexpr (BF) {
            // token row = lt + tap + 1 - pad; this lane's 4 channels of an 8-channel round start at 4 * lh
            const __bf16* xb = reinterpret_cast<const __bf16*>(xt) + (lt + ((u.taps == 3) ? 0 : 1)) * PB + 4 * lh;
            if constexpr (ROLL) {
#pragma unroll
                for (int rr = 0; rr < NRMAX; ++rr) {
                    {
                        const s16x4 xf = *reinterpret_cast<const s16x4*>(xb + rr * 8);
#pragma unroll
                        for (int i = 0; i < NACC; ++i)
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, ws[0][rr & 1][i]), xf, acc[i], 0, 0, 0);
                        if (rr == 0) issue_w_round(u, 2, ws[0][0], true);
                    }
                }
                return;
            } else {
#pragma unroll
                for (int tap = 0; tap < TMAX; ++tap) {
                    if (tap < u.taps) {
#pragma unroll
                        for (int rr = 0; rr < NRMAX; ++rr) {
                            {
                                const s16x4 xf = *reinterpret_cast<const s16x4*>(xb + tap * PB + rr * 8);
#pragma unroll
                                for (int i = 0; i < NACC; ++i) {
                                    const s16x4 wf = __builtin_bit_cast(s16x4, ws[tap][rr][i]);
                                    acc[i] = TRANS ? __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(xf, wf, acc[i], 0, 0, 0)
                                                   : __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(wf, xf, acc[i], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
                return;
            }
        } else {
        const float* xrow = xt + lh * XP + lt + 3 + ((u.taps == 3) ? 0 : 1);   // col = lt + tap + 4 - pad
        if constexpr (ROLL) {
#pragma unroll
            for (int rr = 0; rr < NRMAX; ++rr) {
                {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xf = xrow[(rr * 8 + 2 * j) * XP];
#pragma unroll
                        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws[0][rr & 1][i][j], xf, acc[i], 0, 0, 0);
                    }
                    if (rr == 0) issue_w_round(u, 2, ws[0][0], true);
                }
            }
            return;
        }
#pragma unroll
        for (int tap = 0; tap < TMAX; ++tap) {
            if (tap < u.taps) {
#pragma unroll
                for (int rr = 0; rr < NRMAX; ++rr) {
                    {
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
        }
    };

    {
        // flattened (segment, block) list, software-pipelined D blocks deep: the operands of block i + D - 1 are requested
        // before block i is staged and multiplied.  Requests past the end are issued out of range (they return 0 and
        // cost an issue slot each) so that the loop body contains no conditional loads.
        const int nb0 = nblocks(0), nb1 = nseg > 1 ? nblocks(1) : 0, nb2 = nseg > 2 ? nblocks(2) : 0;
        const int nblk_total = nb0 + nb1 + nb2;
        auto block_at = [&](int i) {
            if (i < nb0) return make_block(0, i);
            if (i < nb0 + nb1) return make_block(1, i - nb0);
            return make_block(2, i - nb0 - nb1);
        };
        if constexpr (NSPL) {
            stage(u0, xv, halo);
            __syncthreads();   // the eight staging tiles together are the [192][XP] X tile every wave multiplies
            clk_stamp_p(clkp, w, l, 5);
            if constexpr (SP) {   // eight split-fp16 tiles, one per staging wave: K-group G = 2 S + lh of the flat K = 192 lives in tile G / 3 at halfs 8 (G % 3)
                const _Float16* xb = reinterpret_cast<const _Float16*>(lnred + KS * 64) + (lt + 1) * SPP;
#pragma unroll
                for (int S = 0; S < 12; ++S) {
                    constexpr int TH = 2 * 8 * NRMAX * XP;   // halfs between the waves' tiles
                    const int o0 = ((2 * S) / 3) * TH + 8 * ((2 * S) % 3), o1 = ((2 * S + 1) / 3) * TH + 8 * ((2 * S + 1) % 3);
                    const _Float16* pg = xb + (lh ? o1 : o0);
                    const f16x8 fh = *reinterpret_cast<const f16x8*>(pg), fl = *reinterpret_cast<const f16x8*>(pg + SPR * SPP);
                    const f16x8 wh = __builtin_bit_cast(f16x8, wn[2 * S]), wl = __builtin_bit_cast(f16x8, wn[2 * S + 1]);
                    accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fh, accx[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fh, acc[0], 0, 0, 0);
                    accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fl, accx[0], 0, 0, 0);
                }
            } else if constexpr (BF) {   // eight token-major bf16 tiles [token][24 channels], one per staging wave
                const __bf16* xb = reinterpret_cast<const __bf16*>(lnred + KS * 64) + (lt + 1) * PB + 4 * lh;
#pragma unroll
                for (int rr = 0; rr < 24; ++rr) {
                    const s16x4 xf = *reinterpret_cast<const s16x4*>(xb + (rr / 3) * (2 * 8 * NRMAX * XP) + (rr % 3) * 8);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, wn[rr]), xf, acc[0], 0, 0, 0);
                }
            } else {
                const float* xrow = lnred + KS * 64 + lh * XP + lt + 4;
#pragma unroll
                for (int rr = 0; rr < 24; ++rr)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wn[rr][j], xrow[(rr * 8 + 2 * j) * XP], acc[0], 0, 0, 0);
            }
        } else if constexpr (ONE_BLOCK) {   // no block pipeline (and none of its code or registers)
            stage(u0, xv, halo);
            clk_stamp_p(clkp, w, l, 5);
            mma_block(u0, wv);
        } else if (ROLL || nblk_total == 1) {   // ROLL: the host guarantees a single block
            stage(u0, xv, halo);
            clk_stamp_p(clkp, w, l, 5);
            mma_block(u0, wv);
        } else if constexpr (!((NACC == 1) || (TMAX == 1 && NACC <= 3))) {
            for (int i = 0; i < nblk_total; ++i) {
                const UBlock u = (i == 0) ? u0 : block_at(i);
                if (i > 0) { issue_x(u, xv, halo, true); issue_w(u, wv, true); }
                stage(u, xv, halo);
                if (i == 0) clk_stamp_p(clkp, w, l, 5);
                mma_block(u, wv);
            }
        } else {
#ifndef SAID_PIPE_DEPTH_1TAP
#define SAID_PIPE_DEPTH_1TAP 3
#endif
            constexpr int D = (TMAX == 1 && NACC == 1) ? SAID_PIPE_DEPTH_1TAP : 2;
            f32x4 xq[D - 1][NRMAX];
            WT wq[D - 1][WD0][WD1][NACC];
            float hq[D - 1];
            UBlock uq[D - 1];
            const int last = nblk_total - 1;
#pragma unroll
            for (int d = 0; d < D - 2; ++d) {   // prologue (D == 3): block 1
                uq[d] = block_at(min(d + 1, last));
                issue_x(uq[d], xq[d], hq[d], d + 1 <= last);
                issue_w(uq[d], wq[d], d + 1 <= last);
            }
            UBlock uc = u0;
            for (int i = 0; i < nblk_total; ++i) {
                const int nb = i + D - 1;
                uq[D - 2] = block_at(min(nb, last));
                issue_x(uq[D - 2], xq[D - 2], hq[D - 2], nb <= last);
                issue_w(uq[D - 2], wq[D - 2], nb <= last);
                stage(uc, xv, halo);
                if (i == 0) clk_stamp_p(clkp, w, l, 5);
                mma_block(uc, wv);
                // rotate the queue (register renaming after unrolling)
                uc = uq[0];
#pragma unroll
                for (int rr = 0; rr < NRMAX; ++rr) xv[rr] = xq[0][rr];
                halo = hq[0];
#pragma unroll
                for (int tap = 0; tap < WD0; ++tap)
#pragma unroll
                    for (int rr = 0; rr < WD1; ++rr)
#pragma unroll
                        for (int ii = 0; ii < NACC; ++ii) wv[tap][rr][ii] = wq[0][tap][rr][ii];
#pragma unroll
                for (int d = 0; d + 1 < D - 1; ++d) {
                    uq[d] = uq[d + 1];
                    hq[d] = hq[d + 1];
#pragma unroll
                    for (int rr = 0; rr < NRMAX; ++rr) xq[d][rr] = xq[d + 1][rr];
#pragma unroll
                    for (int tap = 0; tap < WD0; ++tap)
#pragma unroll
                        for (int rr = 0; rr < WD1; ++rr)
#pragma unroll
                            for (int ii = 0; ii < NACC; ++ii) wq[d][tap][rr][ii] = wq[d + 1][tap][rr][ii];
                }
            }
        }
    }
    clk_stamp_p(clkp, w, l, 6);

    if constexpr (SP) {   // main + 2^-11 cross: one fma per element, before anything leaves the registers
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = fmaf(accx[i][r], 0x1p-11f, acc[i][r]);
    }
    // ================= phase 4: split-K reduction through LDS (fixed order => deterministic) =================
    if (EPI == EPI_GEGLU && tid < 64 * NB) epiS[tid] = geglu_bias;
    __syncthreads();
    clk_stamp_p(clkp, w, l, 7);
    if constexpr (NSPL) {   // no reduction: gate waves hand their tile to the value wave of the same index
        float* gx = mainS;
        if (w >= NB) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gx[((w - NB) * 16 + r) * 64 + l] = acc[0][r];
        }
        __syncthreads();
        clk_stamp_p(clkp, w, l, 8);
        if (w < NB) {
            float* const yo = AH(y) + (long long)b * AH(y_bstride);
            const int yp_ = AH(y_pitch);
            const int t = t0 + lt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int frow = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int nl = (tile0 + w) * 32 + frow;
                const float xv_ = acc[0][r] + epiS[w * 32 + frow];
                const float gv = gx[(w * 16 + r) * 64 + l] + epiS[(NB + w) * 32 + frow];
                if (nl < aN && t < aT) gstore(yo, (long long)nl * yp_ + t, geglu_f(xv_, gv));
            }
        }
        clk_stamp_p(clkp, w, l, 9);
        return;
    }
    float* red = MT ? mainS + coef_total : mainS;   // MT: the coefficient tables live on for the next tile
    // RP reduction passes: GEGLU with 4 value + 4 gate tiles would need 256 KB for one pass, so the value tiles and the
    // gate tiles go through the same buffer one after the other (the summation order per element is unchanged)
    constexpr int RP = (KS * NACC * 16 * 64 * 4 > 128 * 1024) ? 2 : 1;
    constexpr int NPP = NACC / RP;   // accumulator tiles per pass
    static_assert(RP == 1 || (EPI == EPI_GEGLU && NPP == NB), "two-pass reduction is the GEGLU value/gate split");
    float vsum[VPW], gsum[EPI == EPI_GEGLU ? VPW : 1];
#pragma unroll
    for (int ps = 0; ps < RP; ++ps) {
        if (ps > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < NPP; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((w * NPP + i) * 16 + r) * 64 + l] = acc[ps * NPP + i][r];
        __syncthreads();
        if (ps == 0) clk_stamp_p(clkp, w, l, 8);
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            const int v = w + j * KS;
            const int i = v >> 4, r = v & 15;
            if (RP == 1) {
                float a0 = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < KS; ++w2) a0 += red[((w2 * NPP + i) * 16 + r) * 64 + l];
                vsum[j] = a0;
                if (EPI == EPI_GEGLU) {
                    float g0 = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < KS; ++w2) g0 += red[((w2 * NPP + i + NB) * 16 + r) * 64 + l];
                    gsum[j] = g0;
                }
            } else {
                float a0 = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < KS; ++w2) a0 += red[((w2 * NPP + i) * 16 + r) * 64 + l];
                if (ps == 0) vsum[j] = a0; else gsum[EPI == EPI_GEGLU ? j : 0] = a0;
            }
        }
    }

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
        float val = vsum[j];
        const float gate = (EPI == EPI_GEGLU) ? gsum[EPI == EPI_GEGLU ? j : 0] : 0.f;
        const int frow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int tile = tile0 + i;

        if (EPI == EPI_QKV && TRANS) {
            const int nl = tile * 32 + lt;
            const int t = t0 + frow;
            if (e_biasp) val += gload(e_biasp, nl);
            const int vdim = AH(vt_dim);
            const int vn = tile * 32 + lt;
            const int h = vn >> __builtin_ctz(vdim), d = vn & (vdim - 1);   // head_dim is a power of two (host checks)
            if (AH(kv_split) && 2 * h >= AH(vt_heads)) val = pack_split_f16(val);   // (k heads: the second half of vt's head axis)
            if (t < aT && nl < aN)
                gstore(AH(vt), (((long long)b * AH(vt_heads) + h) * AH(vt_rows) + t) * vdim + d, val);
            continue;
        }
        const int nl = tile * 32 + frow;
        const int t = t0 + lt;
        const bool ok = (nl < aN) && (t < aT);
        const int ng = nl;
        if (EPI == EPI_GEGLU) {
            const int ngate = nl + gate_tiles * 32;
            (void)ngate;
            const float xv_ = val + epiS[i * 32 + frow];
            const float gv = gate + epiS[(NB + i) * 32 + frow];
            if (ok) gstore(yp, (long long)b * y_bs + (long long)nl * y_pitch + t, geglu_f(xv_, gv));
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
                if (e_biasp) val += gload(e_biasp, ng);
                if (e_act == ACT_SILU) val = silu_f(val);
                else if (e_act == ACT_GELU) val = gelu_f(val);
                const float* embp = AH(emb);
                if (embp) {
                    const int* sp = AH(step_ptr);
                    const int row = (sp ? cload(sp, 0) : 0) + b * AH(emb_b_stride);
                    val += gload(embp, (long long)ng * AH(emb_pitch) + row);
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
        if (EPI == EPI_QKV && AH(kv_split)) val = pack_split_f16(val);   // (v tiles)
        if (ok) gstore(yp, (long long)b * y_bs + (long long)ng * y_pitch + t, val);
        if constexpr (DUP) {
            float add2 = 0.f;
            if (EPRE) add2 = e_add2[EPRE ? j : 0];
            else { const float* a2 = AH(y2_add); if (a2 && nl < aN) add2 = gload(a2, ng); }
            if (ok) gstore(AH(y2), (long long)b * AH(y2_bstride) + (long long)ng * y_pitch + t, val + add2);
        }
        if (EPI == EPI_STORE && statsp) {
            const float cnt = (float)min(32, aT - t0);
            const float vv = (t < aT) ? val : 0.f;
            const float mean = half32_sum(vv) * __builtin_amdgcn_rcpf(cnt);
            const float d = (t < aT) ? (val - mean) : 0.f;
            const float m2 = half32_sum(d * d);
            if (lt == 0 && nl < aN) {
                float* so = statsp + (long long)b * AH(stats_bstride) + ((long long)(t0 >> 5) * aN + ng) * 2;   // [tile][channel][2]
                gstore(so, 0, mean);
                gstore(so, 1, m2);
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
                for (int dd = 0; dd < DPG; ++dd) p = fmaf(qt[(gi * DPG + dd) * 32 + tt], kraw[dd][wi >> 2][wi & 3], p);
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
                for (int wi = 0; wi < 8; ++wi) {
                    const bool vis = (wi < wmax) && (lo + wi < hi);   // invisible slots may hold another row's data
                    o = fmaf(sc[wi] * inv, vis ? vraw[dd][wi >> 2][wi & 3] : 0.f, o);
                }
                gstore(yp, (long long)b * y_bs + (long long)(head * 32 + d) * y_pitch + t, o);
            }
        }
    }
    if constexpr (MT) {
        if (ti + 1 < ntr) {
            __syncthreads();   // everybody is done with this tile's LDS (reduction buffer, band scratch)
#pragma unroll
            for (int rr = 0; rr < NRMAX; ++rr) xv[rr] = xvn[rr];
            halo = halon;
            t0 += 32;
            issue_tile_operands(t0);
        }
    }
    }   // tile loop
}


// ------------------------------------------------------------------------------------------------
// kconv_body (round 6): the K-long ResBlock convolutions of the UNet's up path on split-fp16 products, NB = 1, KS = 8 —
//   GN1  (in_layers over the concatenated [h ; skip]):   two 3-tap GroupNorm + SiLU segments of 8 x 24 channels;
//   !GN1 (out_layers + the 1x1 skip convolution):        one 3-tap GroupNorm + SiLU segment and one or two plain 1-tap segments.
// ugemm_body walks a wave's (segment, block) list in a run-time loop: the arguments of segments 1 / 2 are decoded per block, block i + 1 is requested only
// when block i is about to be staged (behind both GroupNorm finalisations), the transform is a run-time switch, and in the GN1 instantiation the compiler
// serialised the eight scalar epilogue loads (one s_waitcnt each) for lack of SGPRs: 25.4k / 21.9k clocks per launch against 12.5k for the single-segment
// convolution (profiles/r06g_kconv_ab.txt).  Here the wave's two or three blocks are straight-line code: every operand of every block is requested in
// the request phase (segment 0 from the preloaded header at entry, segments 1 / 2 as soon as the argument block has arrived), epilogue constants come
// through the vector path, the tiles of the blocks are separate (a block is staged while the previous one's MFMAs drain) and the reduction buffer no longer
// aliases them (one barrier).  Same weights (Seg::ws), same (block, step, product) order per accumulator, same reduction order: bit-identical results
// (tests/test_gpu_round6.py).  The host checks the shapes (ugemm_supports) and sizes the LDS (kconv_smem_floats).
// LDS (floats): [8 waves][GN_SCRATCH] | coefficient tables 2 C0 (+ 2 C1) | tiles [8 waves][NBLK][8 * NRMAX * XP] | reduction [8][16][64]
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int kconv_smem_floats(int C0, int C1gn, int nblk) {
    return 8 * GN_SCRATCH + 2 * C0 + 2 * C1gn + 8 * nblk * (8 * NRMAX * XP) + 8 * 16 * 64;
}
template <bool GN1>
__device__ __forceinline__ void kconv_body(const FastHdr& hd, float* smem, int bx, int by, int bz) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    constexpr int KS = 8, SPR = 34, SPP = 24, NBLK = GN1 ? 2 : 3, TILE = 8 * NRMAX * XP;
    constexpr int NS1 = GN1 ? 5 : 2;          // k16 steps of a block of segments 1 / 2 (3 taps: 9 K-groups of 8 -> 5; 1 tap: 3 -> 2)
    const int tid = threadIdx.x, l = tid & 63, lt = l & 31, lh = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = bz + (hd.bmod_b0 >> 17);
    const int t0 = bx * 32, tile0 = by;
    const int aT = hd.T, aN = hd.N;
    const int ntiles = (aN + 31) >> 5;
    const int sr = l >> 3, sq = l & 7;
    const int C0 = hd.pack & 0xffff;
    const int c0 = w * 24;                    // the wave's channel slice of every segment (the host guarantees C / 8 == 24)

    float* gnS = smem + w * GN_SCRATCH;
    float* coef0 = smem + KS * GN_SCRATCH;
    float* coef1 = coef0 + 2 * C0;

    // ================= requests: segment 0 from the preloaded header =================
    const rsrc_t rx0 = make_rsrc(hd.x + (long long)b * hd.bstride, (unsigned)C0 * (unsigned)hd.pitch * 4u);
    const unsigned wbytes0 = (unsigned)ntiles * sp_tile_dwords(C0, 3) * 4u;
    const rsrc_t rw0 = make_rsrc(hd.w4, wbytes0);
    const float* gb0 = hd.w4 + (long long)ntiles * sp_tile_dwords(C0, 3);   // gamma[C0], beta[C0] behind the weights
    const GnP gp0 = {hd.gn_cfg & 0xffff, hd.gn_cfg >> 16, hd.T, ((hd.pack >> 26) & 1) ? 1e-6f : 1e-5f, gb0, gb0 + C0, C0};
    const rsrc_t rp0 = make_rsrc(hd.gn_part + (long long)b * hd.gn_bstride, (unsigned)C0 * (unsigned)gp0.gn_nparts * 8u);
    GnLoads gl0;
    gn_issue(gp0, rp0, c0, 24, l, gl0);
    const ArgView V = arg_view_hs<true>(l, kHdrDwords);
    auto issue_x = [&](rsrc_t rx, int pitch4, int Tin, bool three, f32x4 (&xv)[NRMAX], float& halo, bool valid) {
        const int oor = valid ? 0 : (int)0x80000000;
#pragma unroll
        for (int rr = 0; rr < NRMAX; ++rr) xv[rr] = bload4(rx, (sr * pitch4 + (t0 + 4 * sq) * 4) | oor, (c0 + rr * 8) * pitch4);
        halo = 0.f;
        if (three) {   // lane -> (row = l >> 1, side = l & 1): token t0 - 1 or t0 + 32
            const int row = l >> 1, tin = (l & 1) ? (t0 + 32) : (t0 - 1);
            const bool ok = valid && (row < NRMAX * 8) && ((unsigned)tin < (unsigned)Tin);
            halo = bload(rx, ok ? (row * pitch4 + tin * 4) : (int)0x80000000, c0 * pitch4);
        }
    };
    // weights of (tile0, block w) of a segment: [tile][8 blocks][ns steps][2 planes] x 1024 bytes
    auto issue_w5 = [&](rsrc_t rw, f32x4 (&wv)[5][2], bool valid) {
        const int oor = valid ? 0 : (int)0x80000000;
#pragma unroll
        for (int st = 0; st < 5; ++st)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wv[st][pl] = bload4(rw, (l * 16) | oor, (((tile0 * 8 + w) * 5 + st) * 2 + pl) * 1024);
    };
    auto issue_w2 = [&](rsrc_t rw, f32x4 (&wv)[2][2], bool valid) {
        const int oor = valid ? 0 : (int)0x80000000;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wv[st][pl] = bload4(rw, (l * 16) | oor, (((tile0 * 8 + w) * 2 + st) * 2 + pl) * 1024);
    };
    f32x4 x0[NRMAX], w0[5][2];
    float halo0;
    issue_x(rx0, hd.pitch * 4, hd.T, true, x0, halo0, true);
#ifdef SAID_KCONV_W_FIRST
    issue_w5(rw0, w0, true);
#endif
    // (the machine scheduler otherwise hoists the argument-block loads and their readlanes above these requests and sinks the requests behind that round trip)
    __builtin_amdgcn_sched_barrier(0);
    long long* const clkp = AH(clk);
    clk_stamp_p(clkp, w, l, 0);

    // ================= requests: segments 1 (and 2) once the argument block is here =================
    const int nseg = (hd.pack >> 24) & 3;
    const unsigned sv1 = V.s1, sv2 = V.s2;
    const int* const stepp = AH(step_ptr);
    const float* const embp = AH(emb);
    // (the step counter heads a dependent chain — counter -> embedding row -> load: its scalar round trip runs under the requests below)
    const int step_raw = cload(stepp ? stepp : reinterpret_cast<const int*>(hd.w4), 0);
    const int C1 = AS(sv1, C), pitch1 = AS(sv1, x_pitch), Tin1 = AS(sv1, Tin);
    const rsrc_t rx1 = make_rsrc(AS(sv1, x) + (long long)b * AS(sv1, x_bstride), (unsigned)C1 * (unsigned)pitch1 * 4u);
    const rsrc_t rw1 = make_rsrc(AS(sv1, ws), (unsigned)ntiles * sp_tile_dwords(C1, GN1 ? 3 : 1) * 4u);
    GnLoads gl1;
    GnP gp1 = gp0;
    rsrc_t rp1 = rp0;
    if constexpr (GN1) {
        gp1 = {AS(sv1, gn_cpg), AS(sv1, gn_nparts), Tin1, AS(sv1, gn_eps), AS(sv1, gn_gamma), AS(sv1, gn_beta), C1};
        rp1 = make_rsrc(AS(sv1, gn_part) + (long long)b * AS(sv1, gn_part_bstride), (unsigned)C1 * (unsigned)gp1.gn_nparts * 8u);
        gn_issue(gp1, rp1, c0, 24, l, gl1);
    }
    f32x4 x1[NRMAX], w1[NS1][2];
    float halo1;
    issue_x(rx1, pitch1 * 4, Tin1, GN1, x1, halo1, true);
    f32x4 x2[GN1 ? 1 : NRMAX], w2[GN1 ? 1 : 2][2];
    int Tin2 = 0;
    rsrc_t rw2_ = rw1;
    const bool has2 = !GN1 && nseg > 2;
    if constexpr (!GN1) {   // (absent third segment: every request out of range — zeros against zero weights, never multiplied)
        const int C2 = AS(sv2, C), pitch2 = AS(sv2, x_pitch);
        Tin2 = AS(sv2, Tin);
        const rsrc_t rx2 = make_rsrc(AS(sv2, x) + (long long)b * AS(sv2, x_bstride), has2 ? (unsigned)C2 * (unsigned)pitch2 * 4u : 0u);
        const rsrc_t rw2 = make_rsrc(AS(sv2, ws), has2 ? (unsigned)ntiles * sp_tile_dwords(C2, 1) * 4u : 0u);
        float h2;
        issue_x(rx2, pitch2 * 4, Tin2, false, x2, h2, has2);
        rw2_ = rw2;
    }
    // epilogue constants of the wave's two accumulator rows (VPW = 2: r = w, w + 8) through the vector path: row by lane half
    const float* const biasp = AH(bias);
    const int e_act = AH(act);
    const rsrc_t r_bias = make_rsrc(biasp, biasp ? (unsigned)aN * 4u : 0u);
    const int emb_pitch = AH(emb_pitch);
    const rsrc_t r_emb = make_rsrc(embp, embp ? (unsigned)aN * (unsigned)emb_pitch * 4u : 0u);
    const int erow = (stepp ? step_raw : 0) + b * AH(emb_b_stride);
    float e_bias[2], e_emb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = w + j * KS;
        const int n = min(tile0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, aN - 1);   // rows past N read row N - 1 (never stored)
        e_bias[j] = bload(r_bias, n * 4, 0);
        e_emb[j] = bload(r_emb, (n * emb_pitch + erow) * 4, 0);
    }
    // The weights go out LAST: a CU's vector-memory path takes ~16 clocks per 1 KB wave-load, so the eight waves' requests queue behind each other (the second wave of
    // each SIMD finishes issuing 2-4k clocks after the first), and 10 KB of weights per block in front of the statistics and operands of the waves behind would hold up
    // the head of every wave's dependent chain (partials -> coefficients -> staging) for data the MFMAs need last.
    __builtin_amdgcn_sched_barrier(0);
#ifndef SAID_KCONV_W_FIRST
    issue_w5(rw0, w0, true);
#endif
    if constexpr (GN1) issue_w5(rw1, w1, true); else issue_w2(rw1, w1, true);
    if constexpr (!GN1) issue_w2(rw2_, w2, has2);
    __builtin_amdgcn_sched_barrier(0);
    clk_stamp_p(clkp, w, l, 1);

    // ================= GroupNorm coefficients of the wave's slice of segment 0 =================
    gn_finish(gp0, rp0, c0, 24, l, gl0, gnS, coef0);
    clk_stamp_p(clkp, w, l, 2);
    clk_stamp_p(clkp, w, l, 3);

    f32x16 acc, accx;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accx[r] = 0.f; }
    float* const tiles = coef1 + (GN1 ? 2 * C1 : 0) + w * (NBLK * TILE);
    clk_stamp_p(clkp, w, l, 4);

    // one conversion per stored half (split_f16.h): the remainder is taken from the very bits that are stored
    auto put = [&](_Float16* xh, int idx, float o) {
        _Float16 hv = (_Float16)o;
        asm volatile("" : "+v"(hv));
        xh[idx] = hv;
        xh[SPR * SPP + idx] = (_Float16)((o - (float)hv) * 2048.f);
    };
    auto stage_gn3 = [&](float* xt, const float2* cGN, const f32x4 (&xs)[NRMAX], float hl, int Tin) {   // GroupNorm + SiLU, three taps (halo rows 0 and 33)
        _Float16* xh = reinterpret_cast<_Float16*>(xt);
#pragma unroll
        for (int rr = 0; rr < NRMAX; ++rr) {
            const float2 gn = cGN[c0 + rr * 8 + sr];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = xf1<XF_GN_SILU>(xs[rr][e], gn, 0.f, 1.f, make_float2(1.f, 0.f));
                put(xh, (1 + 4 * sq + e) * SPP + rr * 8 + sr, (t0 + 4 * sq + e < Tin) ? v : 0.f);
            }
        }
        const int row = l >> 1, tin = (l & 1) ? (t0 + 32) : (t0 - 1);
        if (row < NRMAX * 8) {
            const float v = xf1<XF_GN_SILU>(hl, cGN[c0 + row], 0.f, 1.f, make_float2(1.f, 0.f));
            put(xh, ((l & 1) ? 33 : 0) * SPP + row, ((unsigned)tin < (unsigned)Tin) ? v : 0.f);
        }
    };
    auto stage_raw1 = [&](float* xt, const f32x4 (&xs)[NRMAX], int Tin) {   // plain operand, one tap
        _Float16* xh = reinterpret_cast<_Float16*>(xt);
#pragma unroll
        for (int rr = 0; rr < NRMAX; ++rr)
#pragma unroll
            for (int e = 0; e < 4; ++e) put(xh, (1 + 4 * sq + e) * SPP + rr * 8 + sr, (t0 + 4 * sq + e < Tin) ? xs[rr][e] : 0.f);
    };
    // row `token` of the tile is the im2col row of that token: K-group g at halfs 8 g; a 1-tap block starts one row down (no left halo)
    auto mma5 = [&](const float* xt, const f32x4 (&ws)[5][2]) {
        const _Float16* xh = reinterpret_cast<const _Float16*>(xt) + lt * SPP;
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            const int g = min(2 * st + lh, 8);   // (the padding group re-reads the last one: finite data against zero weights)
            const f16x8 fh = *reinterpret_cast<const f16x8*>(xh + 8 * g), fl = *reinterpret_cast<const f16x8*>(xh + SPR * SPP + 8 * g);
            const f16x8 wh = __builtin_bit_cast(f16x8, ws[st][0]), wl = __builtin_bit_cast(f16x8, ws[st][1]);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fh, accx, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fh, acc, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fl, accx, 0, 0, 0);
        }
    };
    auto mma2 = [&](const float* xt, const f32x4 (&ws)[2][2]) {
        const _Float16* xh = reinterpret_cast<const _Float16*>(xt) + (lt + 1) * SPP;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int g = min(2 * st + lh, 2);
            const f16x8 fh = *reinterpret_cast<const f16x8*>(xh + 8 * g), fl = *reinterpret_cast<const f16x8*>(xh + SPR * SPP + 8 * g);
            const f16x8 wh = __builtin_bit_cast(f16x8, ws[st][0]), wl = __builtin_bit_cast(f16x8, ws[st][1]);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, fh, accx, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fh, acc, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, fl, accx, 0, 0, 0);
        }
    };

    // ================= stage + multiply, block by block =================
    stage_gn3(tiles, reinterpret_cast<const float2*>(coef0), x0, halo0, hd.T);
    clk_stamp_p(clkp, w, l, 5);
    mma5(tiles, w0);
    if constexpr (GN1) {
        gn_finish(gp1, rp1, c0, 24, l, gl1, gnS, coef1);
        stage_gn3(tiles + TILE, reinterpret_cast<const float2*>(coef1), x1, halo1, Tin1);
        mma5(tiles + TILE, w1);
    } else {
        stage_raw1(tiles + TILE, x1, Tin1);
        mma2(tiles + TILE, w1);
        if (has2) {
            stage_raw1(tiles + 2 * TILE, x2, Tin2);
            mma2(tiles + 2 * TILE, w2);
        }
    }
    clk_stamp_p(clkp, w, l, 6);

    // ================= split-K reduction through LDS (fixed order => deterministic), epilogue =================
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fmaf(accx[r], 0x1p-11f, acc[r]);   // main + 2^-11 cross
    float* red = coef1 + (GN1 ? 2 * C1 : 0) + KS * (NBLK * TILE);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + l] = acc[r];
    clk_stamp_p(clkp, w, l, 7);
    __syncthreads();
    clk_stamp_p(clkp, w, l, 8);
    float* const yp = AH(y);
    const long long y_bs = AH(y_bstride);
    const int y_pitch = AH(y_pitch);
    float* const statsp = AH(stats_out);
    const long long st_bs = AH(stats_bstride);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = w + j * KS;
        float val = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < KS; ++w2) val += red[(w2 * 16 + r) * 64 + l];
        const int nl = tile0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int t = t0 + lt;
        const bool ok = (nl < aN) && (t < aT);
        val += e_bias[j];
        if (e_act == ACT_SILU) val = silu_f(val);
        else if (e_act == ACT_GELU) val = gelu_f(val);
        val += e_emb[j];
        if (ok) gstore(yp, (long long)b * y_bs + (long long)nl * y_pitch + t, val);
        if (statsp) {   // GroupNorm partials of the stored values: [tile][channel][2] (mean, M2)
            const float cnt = (float)min(32, aT - t0);
            const float vv = (t < aT) ? val : 0.f;
            const float mean = half32_sum(vv) * __builtin_amdgcn_rcpf(cnt);
            const float d = (t < aT) ? (val - mean) : 0.f;
            const float m2 = half32_sum(d * d);
            if (lt == 0 && nl < aN) {
                float* so = statsp + (long long)b * st_bs + ((long long)(t0 >> 5) * aN + nl) * 2;
                gstore(so, 0, mean);
                gstore(so, 1, m2);
            }
        }
    }
    clk_stamp_p(clkp, w, l, 9);
}

template <int NB, int KS, int EPI, int VAR, bool BF, bool MT, bool SP = false>
// multi-tile single-n-tile store kernels are compiled for <= 128 VGPRs (4 waves per SIMD): two workgroups share a CU, so one's
// staging / reduction / epilogue phases overlap the other's MFMA stream (large batches, SAID_BIG_NB=1)
__global__ __launch_bounds__(64 * KS, (MT && NB == 1 && EPI == EPI_STORE) ? 4 : 1) void ugemm_kernel(const float* hx, const float* hw4, int hpack, int hTN, int hpitch_gv, int hbstride,
                                                        int hbmod_b0, int hgx, const float* hgn_part, int hgn_bstride, int hgn_cfg,
                                                        const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef SAID_AB_FLOOR   // development: the launch's floor (every workgroup leaves at once; scripts/gpu_r5_floor.sh)
    if (hTN != 0x7fffffff) return;
#endif
    // the leading dwords are the preloaded header; `a` only reserves the kernarg layout for arg_view_hs()
    const int hN = (int)((unsigned)hTN >> 16), hgate_vft = (int)((unsigned)hpitch_gv >> 16);
    const FastHdr hd = {hx, hw4, hpack, hpitch_gv & 0xffff, hTN & 0xffff, hbstride, hbmod_b0, hN, hgate_vft, hgn_part, hgn_bstride, hgn_cfg};
    // XCD-aware block order.  Hardware places block id on XCD id % 8; with the natural order every XCD's L2 ends up
    // fetching ALL weights and ALL activations of the launch (rocprofv3 FETCH_SIZE: 4x the algorithmic bytes).  Here
    // each XCD gets a contiguous run of the logical order (n-tile fastest, then t-tile), i.e. a few whole token tiles:
    // it still needs every weight tile but only its own slice of X.  Placement affects speed only.
    // The grid is (token-tile runs x n-tile groups, samples): blocks of one sample with equal blockIdx.x & 7 share an
    // XCD whatever the row's phase, so the decode needs no division by the batch count; the one division left, by the
    // number of n-tile groups, is by one of a few small constants (a run-time integer division is ~25 dependent
    // instructions, and this sits in front of the first load request of every wave).
    const unsigned ny = (unsigned)(((hN + 31) >> 5) / NB), gx = (unsigned)hgx;
    const unsigned cls = blockIdx.x & 7u, slot = blockIdx.x >> 3, q = gx >> 3, r = gx & 7u;
    const unsigned L = (cls < r ? cls * (q + 1) : r * (q + 1) + (cls - r) * q) + slot;
    // L / ny as a multiply-shift with the host's magic number (exact for every L < grid width: the host checks)
    const unsigned ubx = (L * ((unsigned)hbmod_b0 & 0x1ffffu)) >> 16;
    const int bx = (int)ubx, by = (int)(L - ubx * ny), bz = (int)blockIdx.y;
    if constexpr (SP && !MT && NB == 1 && KS == 8 && EPI == EPI_STORE && (VAR == (UV_T3 | UV_GN0 | UV_MULTI) || VAR == (UV_T3 | UV_GN0 | UV_GN1 | UV_MULTI))) {
        if (!(hpack & (1 << 27))) {   // (bit 27: said_debug_option "kconv" = 0 — the block loop of ugemm_body, the A/B reference)
            kconv_body<(VAR & UV_GN1) != 0>(hd, smem, bx, by, bz);
            return;
        }
    }
    if constexpr (EPI == EPI_QKV) {
        if (by * NB < hgate_vft) {
            ugemm_body<NB, KS, EPI, VAR, true, BF, MT, SP>(hd, smem, bx, by, bz);
            return;
        }
    }
    ugemm_body<NB, KS, EPI, VAR, false, BF, MT, SP>(hd, smem, bx, by, bz);
}

template <int NB, int EPI>
static int ugemm_smem_floats(const GemmArgs& a, int KS, bool mt = false) {
    constexpr int NACC = (EPI == EPI_GEGLU) ? 2 * NB : NB;
    int coef = 0;
    for (int s = 0; s < a.nseg; ++s) coef += seg_coef_floats(a.seg[s]);
    const int stage = coef + KS * 64 + KS * 8 * NRMAX * XP;
    const int red = (KS * NACC * 16 * 64 * 4 > 128 * 1024) ? KS * (NACC / 2) * 16 * 64 : KS * NACC * 16 * 64;
    const int red_end = mt ? coef + red : red;   // multi-tile: the reduction buffer sits behind the coefficient tables
    return epi_scratch_floats<NACC>(EPI, KS) + KS * GN_SCRATCH + (stage > red_end ? stage : red_end);
}

constexpr int kMaxLds = 160 * 1024;
template <int NB, int KS, int EPI, int VAR, bool BF, bool MT, bool SP = false>
static void ulaunch_one(const GemmArgs& a, int batch, hipStream_t s, int tt) {
    int smem = ugemm_smem_floats<NB, EPI>(a, KS, MT) * (int)sizeof(float);
    constexpr bool KCONV = SP && !MT && NB == 1 && KS == 8 && EPI == EPI_STORE && (VAR == (UV_T3 | UV_GN0 | UV_MULTI) || VAR == (UV_T3 | UV_GN0 | UV_GN1 | UV_MULTI));
    if (KCONV && !a.kconv_off) {   // kconv_body's own carve (separate tiles per block, reduction buffer beside them)
        const int ks = kconv_smem_floats(a.seg[0].C, (VAR & UV_GN1) ? a.seg[1].C : 0, (VAR & UV_GN1) ? 2 : 3) * (int)sizeof(float);
        if (ks > smem) smem = ks;
    }
    static const int min_lds = dev_env("SAID_MIN_LDS") ? atoi(dev_env("SAID_MIN_LDS")) : 0;   // experiment: force one workgroup per CU
    if (smem < min_lds) smem = min_lds;
    if (smem > kMaxLds) { launch_fault("ugemm needs %d B of LDS", smem); return; }
    const int ntt = (a.T + 31) / 32;
    dim3 grid(((MT ? (ntt + tt - 1) / tt : ntt)) * (a.ntiles_per_group / NB), batch);   // x: decoded XCD-aware in the kernel
    const Seg& s0 = a.seg[0];
    const bool gn0 = s0.xform == XF_GN_SILU || s0.xform == XF_GN_LN;
    const int pack = s0.C | (s0.taps << 16) | (s0.xform << 20) | (a.nseg << 24) | ((gn0 && s0.gn_eps == 1e-6f) ? (1 << 26) : 0) |
                     ((MT ? tt : 0) << 28) | ((KCONV && a.kconv_off) ? (1 << 27) : 0);
    const unsigned ny_host = (unsigned)(a.ntiles_per_group / NB), magic = 65536u / ny_host + 1u;
    for (unsigned L = 0; L < grid.x; ++L)   // exactness of the multiply-shift over this launch's range (a few thousand at most)
        if (((L * magic) >> 16) != L / ny_host) { launch_fault("block decode magic inexact (grid %u, ny %u)", grid.x, ny_host); return; }
    const int bmod_b0 = (int)(magic & 0x1ffffu) | (a.b0 << 17);
    const int gate_vft = (EPI == EPI_GEGLU) ? a.geglu_gate_tiles : (EPI == EPI_QKV ? a.tm_tiles : 0);   // tm_tiles shares a union
    hipLaunchKernelGGL((ugemm_kernel<NB, KS, EPI, VAR, BF, MT, SP>), grid, dim3(64 * KS), smem, s, s0.x, SP ? s0.ws : (BF ? s0.w2 : s0.w4), pack,
                       a.T | (a.N << 16), s0.x_pitch | (gate_vft << 16), (int)s0.x_bstride, bmod_b0, (int)grid.x,
                       gn0 ? s0.gn_part : nullptr, (int)s0.gn_part_bstride, s0.gn_cpg | (s0.gn_nparts << 16), a);
}
template <int NB, int KS, int EPI, int VAR, bool BF, bool MT, bool SP = false>
static void uconfigure_one() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ugemm_kernel<NB, KS, EPI, VAR, BF, MT, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
}

// (epilogue, NB, KS, variant): small-batch tile shapes; large batches use the generic kernel's NB = 3..6 shapes
#ifdef SAID_DEV_ONE_CONFIG   // development aid: compile a single instantiation (resource-usage experiments)
#define SAID_UGEMM_EXPAND(X, ...) X(__VA_ARGS__)
#define SAID_UGEMM_CONFIGS(X) SAID_UGEMM_EXPAND(X, SAID_DEV_ONE_CONFIG)
#else
#define SAID_UGEMM_CONFIGS(X)                                                                                    \
    X(EPI_STORE, 1, 8, 0) X(EPI_STORE, 1, 8, UV_DEEP) X(EPI_STORE, 1, 8, UV_T3 | UV_GN0)                         \
    X(EPI_STORE, 1, 8, UV_MULTI) X(EPI_STORE, 2, 8, UV_MULTI)                                                    \
    X(EPI_STORE, 1, 8, UV_RGN | UV_DUP) X(EPI_STORE, 1, 8, UV_T3 | UV_GN0 | UV_DUP) X(EPI_STORE, 2, 8, UV_T3 | UV_GN0 | UV_DUP) \
    X(EPI_STORE, 1, 8, UV_T3 | UV_GN0 | UV_MULTI)                                                                \
    X(EPI_STORE, 1, 8, UV_T3 | UV_GN0 | UV_GN1 | UV_MULTI) X(EPI_STORE, 1, 8, UV_RGN)                            \
    X(EPI_STORE, 2, 8, 0) X(EPI_STORE, 2, 8, UV_DEEP) X(EPI_STORE, 2, 8, UV_T3 | UV_GN0)                         \
    X(EPI_STORE, 2, 8, UV_T3 | UV_GN0 | UV_MULTI)                                                                \
    X(EPI_STORE, 2, 8, UV_T3 | UV_GN0 | UV_GN1 | UV_MULTI)                                                       \
    X(EPI_STORE, 3, 8, 0) X(EPI_STORE, 3, 8, UV_DEEP) X(EPI_STORE, 3, 8, UV_T3 | UV_GN0) X(EPI_STORE, 3, 8, UV_MULTI) \
    X(EPI_STORE, 3, 8, UV_T3 | UV_GN0 | UV_DUP)                                                                  \
    X(EPI_QKV, 1, 8, UV_GN0) X(EPI_QKV, 2, 8, UV_GN0) X(EPI_QKV, 3, 8, UV_GN0)                                   \
    X(EPI_GEGLU, 1, 8, 0) X(EPI_GEGLU, 2, 8, 0) X(EPI_GEGLU, 4, 8, 0)                                            \
    X(EPI_BAND, 1, 8, 0)
#endif

// split-fp16 product shapes (SP; fp32 mode, single-tile workgroups): the small-batch step's launches.  Not built: the shapes that spill with the second
// accumulator set (NB = 2 with two or three K segments of 3-tap blocks, NB = 3 DEEP / MULTI: 148-180 bytes of scratch per lane) and GEGLU other than the one-tile-per-wave
// NB = 4 shape (its weights use the flat step layout) — those launches stay on the fp32 MFMAs.
#ifdef SAID_DEV_ONE_CONFIG
#define SAID_UGEMM_SP_CONFIGS(X) SAID_UGEMM_EXPAND(X, SAID_DEV_ONE_CONFIG)
#else
#define SAID_UGEMM_SP_CONFIGS(X)                                                                                 \
    X(EPI_STORE, 1, 8, 0) X(EPI_STORE, 1, 8, UV_DEEP) X(EPI_STORE, 1, 8, UV_T3 | UV_GN0)                         \
    X(EPI_STORE, 1, 8, UV_MULTI) X(EPI_STORE, 2, 8, UV_MULTI)                                                    \
    X(EPI_STORE, 1, 8, UV_RGN | UV_DUP) X(EPI_STORE, 1, 8, UV_T3 | UV_GN0 | UV_DUP) X(EPI_STORE, 2, 8, UV_T3 | UV_GN0 | UV_DUP) \
    X(EPI_STORE, 1, 8, UV_T3 | UV_GN0 | UV_MULTI)                                                                \
    X(EPI_STORE, 1, 8, UV_T3 | UV_GN0 | UV_GN1 | UV_MULTI) X(EPI_STORE, 1, 8, UV_RGN)                            \
    X(EPI_STORE, 2, 8, 0) X(EPI_STORE, 2, 8, UV_DEEP) X(EPI_STORE, 2, 8, UV_T3 | UV_GN0)                         \
    X(EPI_STORE, 3, 8, 0) X(EPI_STORE, 3, 8, UV_T3 | UV_GN0)                                                     \
    X(EPI_STORE, 3, 8, UV_T3 | UV_GN0 | UV_DUP)                                                                  \
    X(EPI_QKV, 1, 8, UV_GN0) X(EPI_QKV, 2, 8, UV_GN0) X(EPI_QKV, 3, 8, UV_GN0)                                   \
    X(EPI_GEGLU, 4, 8, 0)                                                                                        \
    X(EPI_BAND, 1, 8, 0)
#endif

// multi-tile shapes (large batches; single-block launches).  (Round 2's single-n-tile store shapes at two workgroups per CU — an
// experiment that measured slower, two of whose instantiations spilled 6 VGPRs — are gone.)  fp32 GEGLU uses NB = 2: with 8 accumulator tiles the
// weight fragments only fit by rolling them through the registers, which a multi-tile workgroup cannot do.
#ifdef SAID_DEV_ONE_CONFIG
#define SAID_UGEMM_MT_CONFIGS(X)
#else
#define SAID_UGEMM_MT_CONFIGS(X)                                                                   \
    X(EPI_STORE, 2, 8, 0, 0) X(EPI_STORE, 2, 8, UV_T3 | UV_GN0, 0) X(EPI_STORE, 1, 8, UV_RGN, 0)   \
    X(EPI_STORE, 2, 8, 0, 1) X(EPI_STORE, 2, 8, UV_T3 | UV_GN0, 1) X(EPI_STORE, 1, 8, UV_RGN, 1)   \
    X(EPI_STORE, 2, 8, UV_T3 | UV_GN0 | UV_DUP, 0) X(EPI_STORE, 1, 8, UV_RGN | UV_DUP, 0)          \
    X(EPI_STORE, 2, 8, UV_T3 | UV_GN0 | UV_DUP, 1) X(EPI_STORE, 1, 8, UV_RGN | UV_DUP, 1)          \
    X(EPI_QKV, 3, 8, UV_GN0, 0) X(EPI_QKV, 3, 8, UV_GN0, 1)                                        \
    X(EPI_GEGLU, 2, 8, 0, 0) X(EPI_GEGLU, 2, 8, 0, 1)                                              \
    X(EPI_BAND, 1, 8, 0, 0) X(EPI_BAND, 1, 8, 0, 1)
#endif

void configure_ugemm_kernels() {
#define X(E, nb, ks, var) uconfigure_one<nb, ks, E, var, false, false>(); uconfigure_one<nb, ks, E, var, true, false>();
    SAID_UGEMM_CONFIGS(X)
#undef X
#define X(E, nb, ks, var) uconfigure_one<nb, ks, E, var, false, false, true>();
    SAID_UGEMM_SP_CONFIGS(X)
#undef X
#define X(E, nb, ks, var, bf) uconfigure_one<nb, ks, E, var, bf != 0, true>();
    SAID_UGEMM_MT_CONFIGS(X)
#undef X
}

static inline bool is_gn(int xf) { return xf == XF_GN_SILU || xf == XF_GN_LN; }
static int uvar_of(const GemmArgs& a, int epi) {
    int v = 0;
    if (a.seg[0].taps == 3) v |= UV_T3;
    if (is_gn(a.seg[0].xform)) v |= UV_GN0;
    if (a.nseg > 1 && is_gn(a.seg[1].xform)) v |= UV_GN1;
    if (epi == EPI_STORE && a.res_kind == RES_GN) v |= UV_RGN;
    if (epi == EPI_STORE && a.y2) v |= UV_DUP;
    if (a.nseg > 1) v |= UV_MULTI;
    else if (a.seg[0].C > 24 * 8) v |= UV_DEEP;   // KS = 8 everywhere: more than one 24-channel block per wave
    return v;
}

// The LDS-staged kernel covers stride-1, k in {1,3}, ungrouped GEMMs whose per-wave channel slice is a
// multiple of 24 (or 8 / 16); everything else stays on the generic kernel.
bool ugemm_supports(const GemmArgs& a, int epi, int NB, int KS, int pm, int tt) {
    const bool bf16 = pm == 1, sp = pm == 2;
    bool cfg = false;
    const int var = uvar_of(a, epi);
    if (tt > 1) {   // multi-tile: own shape list, one K block per wave, at most 15 tiles per workgroup
        bool mt = false;
#define X(E, nb, ks, v, bf) mt = mt || (epi == E && NB == nb && KS == ks && var == (v) && bf16 == (bf != 0));
        SAID_UGEMM_MT_CONFIGS(X)
#undef X
        const int cw = a.seg[0].C / KS;
        if (!mt || tt > 15 || a.nseg != 1 || cw != 24) return false;
    }
#define X(E, nb, ks, v) cfg = cfg || (epi == E && NB == nb && KS == ks && var == (v));
    SAID_UGEMM_CONFIGS(X)
#undef X
    if (!cfg || a.groups != 1 || a.ntiles_per_group % NB) return false;
    if (sp) {   // split-fp16 products: single-tile workgroups, every segment packed for them, the flat layout exactly where the kernel expects it
        bool spc = false;
#define X(E, nb, ks, v) spc = spc || (epi == E && NB == nb && KS == ks && var == (v));
        SAID_UGEMM_SP_CONFIGS(X)
#undef X
        if (tt > 1 || !spc) return false;
        const bool flat_shape = epi == EPI_GEGLU && NB == 4 && KS == 8 && !(var & (UV_MULTI | UV_DEEP));
        for (int s = 0; s < a.nseg; ++s) if (!a.seg[s].ws || (a.seg[s].ws_flat != 0) != flat_shape) return false;
        if (epi == EPI_STORE && NB == 1 && KS == 8 && (var == (UV_T3 | UV_GN0 | UV_MULTI) || var == (UV_T3 | UV_GN0 | UV_GN1 | UV_MULTI)) && !a.kconv_off) {
            // kconv_body's shapes: every segment 8 x 24 channels; [3-tap GN + SiLU] x 2, or 3-tap GN + SiLU followed by one or two plain 1-tap segments
            const bool gn1 = (var & UV_GN1) != 0;
            if (a.res_kind != RES_NONE || a.y2 || (gn1 ? a.nseg != 2 : (a.nseg < 2 || a.nseg > 3))) return false;
            for (int s = 0; s < a.nseg; ++s) {
                const Seg& sg = a.seg[s];
                const bool gn3 = (s == 0) || gn1;
                if (sg.C != 24 * 8 || sg.taps != (gn3 ? 3 : 1) || sg.xform != (gn3 ? XF_GN_SILU : XF_NONE) || sg.Tin != a.T) return false;
                if (sg.x_bstride > 0x7fffffffLL || sg.gn_part_bstride > 0x7fffffffLL) return false;
            }
            if (a.emb && (long long)a.N * a.emb_pitch * 4 > 0x7fffffffLL) return false;
        }
    }
    {   // what the compile-time variant assumes about the arguments
        const int xf0 = a.seg[0].xform;
        if (epi == EPI_QKV && (xf0 != XF_GN_LN || a.vt_dim <= 0 || (a.vt_dim & (a.vt_dim - 1)))) return false;
        if ((epi == EPI_GEGLU || epi == EPI_BAND) && xf0 != XF_LN) return false;
        if (epi == EPI_STORE && !(xf0 == XF_NONE || xf0 == XF_SILU || xf0 == XF_GN_SILU)) return false;
        if (a.nseg > 2 && is_gn(a.seg[2].xform)) return false;
        if ((xf0 == XF_LN || xf0 == XF_GN_LN) && !(a.seg[0].w4_ln_tail && a.seg[0].ln_eps == 1e-5f)) return false;
        const int nacc = (epi == EPI_GEGLU) ? 2 * NB : NB;
        if (nacc >= 8 && !(a.nseg == 1 && a.seg[0].C == 24 * KS)) return false;   // rolling weight rounds: one block only
        if (epi != EPI_STORE && a.res_kind != RES_NONE) return false;
        if (!(var & UV_T3))
            for (int s = 0; s < a.nseg; ++s) if (a.seg[s].taps != 1) return false;
        if (var & UV_RGN) {   // one residual GroupNorm group per wave
            const int c_span = NB * 32;
            if (a.res_gn_cpg <= 0 || (c_span + a.res_gn_cpg - 1) / a.res_gn_cpg + 1 > KS) return false;
        }
    }
    if (a.seg[0].x_bstride > 0x7fffffffLL || a.b0 > 0x3fff || a.seg[0].C > 0xffff) return false;
    for (int s = 0; s < a.nseg; ++s) if (a.seg[s].b_mod != 0) return false;   // sample aliasing (b % b_mod) stays on the generic kernel
    if (a.seg[0].Tin != a.T || a.ntiles_per_group != (a.N + 31) / 32) return false;
    if (a.T > 0xffff || a.N > 0xffff || a.seg[0].x_pitch > 0xffff || a.geglu_gate_tiles > 0xffff || (epi == EPI_QKV && a.tm_tiles > 0xffff)) return false;   // packed header fields
    {   // header-only GroupNorm path of segment 0: parameters behind the weights, eps one of two known values
        const Seg& s0 = a.seg[0];
        if (s0.xform == XF_GN_SILU || s0.xform == XF_GN_LN) {
            if (!s0.w4_gn_tail || !(s0.gn_eps == 1e-5f || s0.gn_eps == 1e-6f)) return false;
            if (s0.gn_part_bstride > 0x7fffffffLL || s0.gn_cpg > 0xffff || s0.gn_nparts > 0x7fff) return false;
        }
    }
    for (int s = 0; s < a.nseg; ++s) {
        const Seg& sg = a.seg[s];
        if (bf16 && !sg.w2) return false;
        if (!sg.w4 || sg.stride != 1 || !(sg.taps == 1 || sg.taps == 3) || sg.pad != (sg.taps - 1) / 2) return false;
        if (sg.taps == 3 && epi != EPI_STORE) return false;
        if (sg.C % KS) return false;
        const int cw = sg.C / KS;
        if (cw % 24 != 0) return false;   // blocks of three 8-channel rounds; narrower slices stay on the generic kernel
        if ((sg.xform == XF_GN_SILU || sg.xform == XF_GN_LN) && (cw % sg.gn_cpg || cw > 64)) return false;
        if ((sg.xform == XF_LN || sg.xform == XF_GN_LN) && (sg.taps != 1 || cw > 24 || s != 0)) return false;
    }
    return true;
}

void launch_ugemm(const GemmArgs& a, int epi, int batch, int NB, int KS, hipStream_t s, int pm, int tt) {
    const int var = uvar_of(a, epi);
    const bool bf16 = pm == 1, sp = pm == 2;
    if (tt > 1) {
#define X(E, nb, ks, v, bf) \
        if (epi == E && NB == nb && KS == ks && var == (v) && bf16 == (bf != 0)) { ulaunch_one<nb, ks, E, v, bf != 0, true>(a, batch, s, tt); return; }
        SAID_UGEMM_MT_CONFIGS(X)
#undef X
    }
    if (sp) {
#define X(E, nb, ks, v) \
        if (epi == E && NB == nb && KS == ks && var == (v)) { ulaunch_one<nb, ks, E, v, false, false, true>(a, batch, s, 1); return; }
        SAID_UGEMM_SP_CONFIGS(X)
#undef X
        launch_fault("unsupported split-fp16 ugemm config epi=%d NB=%d KS=%d", epi, NB, KS);
        return;
    }
#define X(E, nb, ks, v) \
    if (epi == E && NB == nb && KS == ks && var == (v)) {                   \
        if (bf16) ulaunch_one<nb, ks, E, v, true, false>(a, batch, s, 1);   \
        else ulaunch_one<nb, ks, E, v, false, false>(a, batch, s, 1);       \
        return;                                                             \
    }
    SAID_UGEMM_CONFIGS(X)
#undef X
    launch_fault("unsupported ugemm config epi=%d NB=%d KS=%d", epi, NB, KS);
}

}  // namespace said

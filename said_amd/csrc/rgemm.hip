// rgemm.hip — REGISTER-STATIONARY weights, WAVE-SPECIALISED persistent GEMMs on token-major activations (round 4; bf16 mode,
// large batches): the 192-wide GEMMs with K <= 576 — the ResBlock convolutions 192 -> 192 (k = 3), attn1.to_out, attn2.to_out,
// the banded cross-attention's q projection — and q/k/v (N = 576, K = 192).
//
// What pgemm.hip (weights resident in LDS, one 4-wave workgroup per CU walking over row tiles) taught, by its shader-clock stamps
// (profiles/r04a_pgemm_clocks.txt): with ONE wave per SIMD every phase of a row tile is a latency chain — 5.0k clocks of MFMA phase
// against 1.7k of MFMA, 5.7k for the GroupNorm + SiLU transform of a 66 x 192 tile, 7.7k for the epilogue — and the 112 KB weight slice of a
// 3-tap convolution leaves no LDS for a second set of waves.  So the weights move to where a CU has most room, the register file:
//
//   * waves 0-5 are MFMA waves: wave j owns output columns [32 j, 32 j + 32) (q/k/v: head j of q, k and v) and keeps THEIR weight
//     fragments — 32 rows x K bf16 = 144 VGPRs at K = 576 — in registers for the whole launch (fetched once in full 128-byte lines
//     through a wave-private LDS staging area: fragment-shaped loads straight from memory took 25k clocks).  Per 32-token row tile it reads
//     the A fragments from an LDS tile (conflict-free 16-byte reads, 400-byte rows; the three taps of a convolution are three row
//     offsets; six fragments in flight, two accumulators), issues K / 16 MFMAs and finishes ITS 32 x 32 output tile itself: bias +
//     timestep-embedding term in the MFMA layout (lane == column), a transposition through a wave-private 4.6 KB scratch, residual (16-byte
//     loads issued before the MFMAs; optionally GroupNorm'ed), rounding, 16-byte stores, duplicate store, GroupNorm partials of the stored
//     values (read back lane == column: in-lane sums) — or direct stores where the MFMA layout is already coalesced (q / k: 128-byte rows
//     of a head; v: transposed product, 128 bytes of consecutive tokens per channel) — or the banded cross-attention of head j;
//   * waves 6-7 are helper waves (lane = (row slot, 6 consecutive channels = 12 bytes)): they load the source rows two tiles ahead, apply
//     the operand transform once per element (GroupNorm coefficients of the sample in registers; LayerNorm sums over the 32 lanes of a row
//     by DPP) and park the tile in the other LDS buffer, as STRAIGHT-LINE code (compile-time transform, clamped addresses instead of
//     branches: with loads inside run-time branches hipcc's wait counts fall back to vmcnt(0) and every use of a prefetched register
//     also waits for the loads issued after it);
//   * ONE workgroup barrier per tile (LDS data only: s_waitcnt lgkmcnt(0) + s_barrier, global loads stay in flight across it);
//   * prologue (later in round 4, from the stamps in profiles/r04g_rgemm_clocks_before.txt / r04h_rgemm_clocks.txt): loads requested in a launch's
//     first clocks come back 8-16k clocks later whatever their number, so MFMA waves 0-3 request the first samples' GroupNorm partials FIRST
//     (one 48-channel slice each: one round trip), then four rounds of weight rows; the first barrier (tables published) sits BEFORE the bulk of
//     the weights and the helpers park the first tile while the weights stream; a new sample's tables inside the loop stay with the helpers;
//   * the 16-byte result stores are agent-scope (sc1: written through L2): the launch does not end with 15-59 MB dirty (-2 % per step).
//
// One workgroup of 8 waves per CU (<= 256 VGPRs each), 64 KB of LDS, <= 256 workgroups for the whole batch, each walking over its
// contiguous share of the 32-token tiles.  All 192 columns of a row tile are produced by ONE workgroup: the operand transform runs once
// per tile.  Reference semantics: /root/reference/said/model/ldm/openaimodel.py:116-227 (ResBlock: in_layers, emb_layers, out_layers,
// skip), ldm/attention.py:86-128, 131-193 (to_q/k/v, to_out, norm1-3 and the residuals around them).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gemm_common.h"
#include "tgemm.h"
#include "tgemm_dev.h"

namespace said {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Cache policy of the result stores (buffer_store aux bits: 1 = sc0, 2 = nt, 16 = sc1).  sc1 = agent scope: the 16-byte result stores are
// written THROUGH the XCD's L2 while the launch runs instead of leaving 15 - 59 MB dirty for the write-back at its end, behind which the next
// launch's first loads queue: 1.243 -> 1.227 / 1.263 -> 1.237 ms per step on two boxes (plain / sc1; nt: 1.266; profiles/r04g_rgemm_ab.txt).
#ifndef SAID_RG_ST_AUX
#define SAID_RG_ST_AUX 16
#endif
constexpr int RG_AP = 200;                        // A tile row pitch in elements (192 + 8: 400 bytes)
constexpr int RG_AROWS = 34;                      // 32 tokens + 2 halo
constexpr int RG_A_ELEMS = RG_AROWS * RG_AP;
constexpr int RG_WS_FLOATS = 32 * 36;             // wave-private scratch: [32 rows][32 + 4] floats = 4608 bytes (= [32 rows][144 bytes] of weight staging)
constexpr int RG_COEF_FLOATS = 2 * 4 * 384;        // GroupNorm (a, b) tables [kind: source / residual][slot = sample & 3][192][2]
constexpr int RG_LDS_BYTES = 2 * RG_A_ELEMS * 2 + 6 * RG_WS_FLOATS * 4 + RG_COEF_FLOATS * 4 + 2 * GN_SCRATCH * 4;

// Workgroup barrier for data exchanged through LDS only: the LDS counter is drained, global loads and stores stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// (geglu2 — value * gelu(gate) for two elements at a time on packed fp32 instructions — lives in gemm_common.h since round 6: stchain.hip's A/B of the same form uses it)
// EK: 0 token-major activation out, 1 q/k/v split (direct stores), 4 banded cross-attention (transposed product)
// MODE: operand transform (0 raw, 1 silu(GroupNorm), 2 LayerNorm, 3 LayerNorm(GroupNorm)); RES: 0 none, 1 plain, 2 GroupNorm'ed residual;
// DUP: second copy of the result (+ per-channel constant); STATS: GroupNorm partials of the stored values
// NCH: 192-channel chunks along K (raw sources sa[0 .. NCH - 1], one tile buffer and one barrier per chunk; NTAP == 1)
template <int NTAP, int EK, int MODE, int RES, bool DUP, bool STATS, int NCH = 1>
__global__ __launch_bounds__(512, 2) void rgemm_kernel(const TGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    typedef unsigned short elt_t;
    constexpr int NCT = EK == 1 ? 3 : (EK == 2 ? 2 : 1);   // column tiles per MFMA wave (q/k/v: head j of each; GEGLU: a (value, gate) pair)
    constexpr int NACC = (NCT == 1 && NCH <= 2) ? 2 : 1;   // accumulators per column tile (one tile: two interleaved chains, registers permitting)
    static_assert(NCH == 1 || (NTAP == 1 && MODE == 0), "chunked K: raw sources, one tap");
    constexpr int NTS = NTAP * 12;                // k16 steps per chunk
    constexpr int NSEG = NTAP * NCH;              // 192-wide K segments of the weight rows: taps or chunks
    constexpr int HALO = NTAP == 3 ? 1 : 0;
    constexpr int NR = 32 + 2 * HALO;             // rows of an A tile
    constexpr int NPASS = (NR + 7) / 8;           // helper passes: four rows per wave and pass (34 rows: the fifth pass repeats row 33)
    constexpr int AP = RG_AP;
    elt_t* const Abuf = lds;
    float* const wscr = reinterpret_cast<float*>(lds + 2 * RG_A_ELEMS);   // [6 MFMA waves][RG_WS_FLOATS]
    float* const coefS = wscr + 6 * RG_WS_FLOATS; // [source / residual][sample & 3][192][2]
    float* const gnscr = coefS + RG_COEF_FLOATS;  // [helper wave][GN_SCRATCH]
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int ntv = (a.M + 31) >> 5;              // valid 32-token tiles per sample
    const int total = a.batch * ntv;
    // workgroup -> (column group, row-tile range); the groups of one range sit on one XCD (they read the same source rows)
    const unsigned Lb = blockIdx.x, xcd = Lb & 7u, slot_ = Lb >> 3;
    const int cg = (int)(slot_ % (unsigned)a.pg_s);
    const int tb = ((int)(slot_ / (unsigned)a.pg_s) * 8 + (int)xcd) * a.pg_per;
    const int n = min(total, tb + a.pg_per) - tb;
    if (n <= 0) return;
    const int t_last = tb + n - 1;

    if (w < 6) {
        // =================================================================== MFMA waves
        const int j = w, fr = l & 31, lh = l >> 5;
        float* const ws = wscr + j * RG_WS_FLOATS;
        const elt_t* W = reinterpret_cast<const elt_t*>(a.w);
        bf16x8 wf[NCT][NCH * NTS];
        const int w_ld = a.w_ld > 0 ? a.w_ld : a.K, w_seg = a.w_seg > 0 ? a.w_seg : 192;   // row pitch, distance between the segments
        {   // ---- weights: rounds of four k16 steps = 128 bytes of each of the wave's 32 rows: four 16-byte loads per lane that cover eight
            // full rows each -> the staging area [32][144 bytes] -> this lane's four fragments.  The loads run TWO rounds ahead of the
            // staging (three register sets): one round per memory round trip made this prologue 13k clocks of every launch.
            const int srow = l >> 3, spc = l & 7;           // staging: lane -> (row within 8, 16-byte piece)
            u32x4* const st = reinterpret_cast<u32x4*>(ws);
            constexpr int RPC = NSEG * 3, NRND = NCT * RPC; // rounds per column tile, rounds in all
            // RING register sets of raw rows: RING - 1 rounds are requested in the launch's first clocks — whatever is requested then comes back
            // together, 8k clocks later (stamps), and every later round is another memory round trip behind it
            constexpr int RING = NRND < 5 ? (NRND < 2 ? 2 : NRND) : 5;   // (7: the K = 576 kernels spill)
            u32x4 v[RING][4];
            auto wload = [&](int rnd, u32x4* dst) __attribute__((always_inline)) {
                const int ct = rnd / RPC, r4 = rnd - ct * RPC;
                const elt_t* wbase = W + (long long)(EK == 2 ? cg * 384 + 64 * j + 32 * ct : ct * 192 + 32 * j) * w_ld + a.w_k0 + (r4 / 3) * w_seg + 64 * (r4 % 3) + 8 * spc;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const u32x4*>(wbase + (long long)(8 * q + srow) * w_ld);
            };
            // The first samples' GroupNorm (a, b) tables are made HERE, by MFMA waves 0-3 (one 48-channel slice each, all partial tiles of the slice
            // in flight, requested before anything else of the launch), and published by the first barrier, which every wave reaches BEFORE the
            // bulk of the weights: the helper waves park the first tile while the weights stream.  (Until round 4's stamps the two helper waves
            // made the tables — four 48-channel slices, one memory round trip each, 11-16k clocks — then parked the first tile, 5k more, with the
            // six MFMA waves waiting at the barrier from 8-10k on: profiles/r04g_rgemm_clocks_before.txt.)
            constexpr bool GN_SRC_M = MODE == 1 || MODE == 3, RES_GN_M = EK == 0 && RES == 2;
            static_assert(!(GN_SRC_M && RES_GN_M), "one table kind per launch");
            bool w01 = false;
            if constexpr (GN_SRC_M || RES_GN_M) {
                if (j < 4) {
                    const int b0 = tb / ntv, b1 = min(tb + 1, t_last) / ntv;
#pragma unroll 1
                    for (int bi = 0; bi < 2; ++bi) {
                        const int b = bi ? b1 : b0;
                        if (bi && b1 == b0) break;
                        const GnP gp = GN_SRC_M ? GnP{a.gn_cpg, a.gn_nparts, a.M, a.gn_eps, a.gn_gamma, a.gn_beta, 192}
                                                : GnP{a.gn_cpg, a.gn_nparts, a.M, a.res_eps, a.res_gamma, a.res_beta, 192};
                        const rsrc_t rp = make_rsrc((GN_SRC_M ? a.gn_part[0] : a.res_part) + (long long)b * a.gn_part_bs, 192u * (unsigned)a.gn_nparts * 8u);
                        GnL20 g0;
                        gn20_issue(gp, rp, 48 * j, l, g0);
                        if (!bi) {   // (behind the partials in the queue, in flight during the finish)
#pragma unroll
                            for (int r0 = 0; r0 < RING - 1 && r0 < NRND; ++r0) wload(r0, v[r0]);
                            w01 = true;
                        }
                        gn20_finish(gp, rp, 48 * j, l, g0, ws, coefS + ((GN_SRC_M ? 0 : 4) + (b & 3)) * 384);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if (!w01) {
#pragma unroll
                for (int r0 = 0; r0 < RING - 1 && r0 < NRND; ++r0) wload(r0, v[r0]);
            }
            lds_barrier();                            // the tables are published; the helpers park the first tile from here on
            clk_stamp_p(a.clk, w, l, 12);
#pragma unroll
            for (int rnd = 0; rnd < NRND; ++rnd) {
                if (rnd + RING - 1 < NRND) wload(rnd + RING - 1, v[(rnd + RING - 1) % RING]);
#pragma unroll
                for (int q = 0; q < 4; ++q) st[(8 * q + srow) * 9 + spc] = v[rnd % RING][q];
                __builtin_amdgcn_wave_barrier();
                const int ct = rnd / RPC, r4 = rnd - ct * RPC;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) wf[ct][4 * r4 + s4] = __builtin_bit_cast(bf16x8, st[fr * 9 + 2 * s4 + lh]);
                __builtin_amdgcn_wave_barrier();
            }
        }
        clk_stamp_p(a.clk, w, l, 11);              // (the weight fragments are in registers)
        float bias_n[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) bias_n[ct] = a.bias ? a.bias[(EK == 2 ? cg * 384 + 64 * j + 32 * ct : ct * 192 + 32 * j) + fr] : 0.f;
        // epilogue mapping (EK 0): lane -> (row 16 q + (l >> 2), columns 8 (l & 3) .. + 7 of the wave's 32)
        const int er = l >> 2, ec = l & 3;
        float rca[8], rcb[8], add2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { rca[e] = 1.f; rcb[e] = 0.f; add2[e] = (EK == 0 && DUP && a.y2_add) ? a.y2_add[32 * j + 8 * ec + e] : 0.f; }
        int cur_b = -1;
        const rsrc_t rsrc_res = make_rsrc((EK == 0 && RES != 0) ? a.res_tm : a.w, 0x7ffffff0u);
        const rsrc_t rsrc_y = make_rsrc(EK == 0 ? a.y_tm : (EK == 2 ? a.yb : (void*)a.w), 0x7ffffff0u);
        const rsrc_t rsrc_y2 = make_rsrc((EK == 0 && DUP) ? a.y2_tm : (void*)a.w, 0x7ffffff0u);
        clk_stamp_p(a.clk, w, l, 0);
        lds_barrier();                            // the first tile is parked
        clk_stamp_p(a.clk, w, l, 1);
        for (int k = 0; k < n; ++k) {
            const int ti = tb + k;
            const int b = ti / ntv, t0 = (ti - b * ntv) * 32;
            const int nrows = min(32, a.M - t0);
            const int R0 = b * a.seg_rows + t0;
            float add = bias_n[0];
            u32x4 rres[2];
            if constexpr (EK == 0) {
                if (a.emb) add += a.emb[(long long)(32 * j + fr) * a.emb_pitch + (a.step_ptr ? *a.step_ptr : 0) + b * a.emb_b_stride];
                if constexpr (RES != 0) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int row = min(16 * q + er, nrows - 1);
                        rres[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_res, ((R0 + row) * a.ldr_tm + 32 * j + 8 * ec) * 2, 0, 0));
                    }
                }
                if constexpr (RES == 2) {
                    if (b != cur_b) {   // (the helpers published this sample's table at least one barrier ago)
#pragma unroll
                        for (int e = 0; e < 8; ++e) { rca[e] = coefS[(4 + (b & 3)) * 384 + 2 * (32 * j + 8 * ec + e)]; rcb[e] = coefS[(4 + (b & 3)) * 384 + 2 * (32 * j + 8 * ec + e) + 1]; }
                        cur_b = b;
                    }
                }
            }
            // ---- K / 16 MFMAs: A fragments six at a time, one group ahead
            f32x16 acc[NCT][NACC];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int x = 0; x < NACC; ++x)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ct][x][r] = 0.f;
            constexpr int GS = (NCT == 3 || NCH > 2) ? 3 : 6;   // fragments per group (q/k/v: three MFMAs per fragment; q/k/v and four-chunk K: ~190 registers of weights)
            constexpr int NG = NTS / GS;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const elt_t* pa = Abuf + ((k * NCH + ch) & 1) * RG_A_ELEMS + fr * AP + 8 * lh;
                bf16x8 fa[2][GS];
#pragma unroll
                for (int s = 0; s < GS; ++s) fa[0][s] = *reinterpret_cast<const bf16x8*>(pa + 16 * s);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 1 < NG) {
                        const int ts0 = (g + 1) * GS, tap = ts0 / 12, s0 = ts0 - 12 * tap;
#pragma unroll
                        for (int s = 0; s < GS; ++s) fa[(g + 1) & 1][s] = *reinterpret_cast<const bf16x8*>(pa + tap * AP + 16 * (s0 + s));
                    }
#pragma unroll
                    for (int s = 0; s < GS; ++s)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) {
                            const bool tr = EK == 4 || (EK == 1 && ct == 2);
                            f32x16& ac = acc[ct][NACC == 2 ? (s & 1) : 0];
                            if (tr) ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ct][ch * NTS + GS * g + s], fa[g & 1][s], ac, 0, 0, 0);
                            else ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g & 1][s], wf[ct][ch * NTS + GS * g + s], ac, 0, 0, 0);
                        }
                }
                if (ch + 1 < NCH) lds_barrier();   // the next chunk is parked
            }
            if constexpr (NACC == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] += acc[0][1][r];
            }
            if (k < 3) clk_stamp_p(a.clk, w, l, 2 + 3 * k);
            if constexpr (EK == 1) {
                // q, k: D[token][d] (lane == d); v: the transposed product D[channel][token] (lane == token).  Straight from the MFMA layout a
                // store instruction moves 4 bytes per lane — 48 of them per tile and wave, and the stamps showed the store issue, not the
                // MFMAs, setting the period (6.5k clocks).  Through the wave's scratch every lane stores 16 bytes: rows of [tokens][32 d]
                // (q, k: 128 contiguous bytes per token and head) / [channels][32 tokens] (v: 128 bytes per channel).
                const rsrc_t rq = make_rsrc(a.qk, 0x7ffffff0u), rv = make_rsrc(a.vt, 0x7ffffff0u);
                const int prow = l >> 3, pc4 = (l & 7) * 4;
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        float bv = bias_n[ct];
                        if (ct == 2) bv = a.bias ? a.bias[384 + 32 * j + row] : 0.f;
                        ws[row * 36 + fr] = acc[ct][0][r] + bv;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (a.qkv_bf16) {
                        // bf16 operands of battn_kernel (attn.hip): same indexing, 64-byte rows: lane -> (row 16 p + (l >> 2), eight elements)
#pragma unroll
                        for (int p2 = 0; p2 < 2; ++p2) {
                            const int row = 16 * p2 + er;
                            // v (rows = channels, columns = tokens): the stored token order inside a block of 16 is [0-3, 8-11, 4-7, 12-15], so the
                            // eight stored positions 8 ec .. 8 ec + 7 are tokens x0 .. x0 + 3 and x0 + 8 .. x0 + 11
                            const int x0 = ct == 2 ? 16 * (ec >> 1) + 4 * (ec & 1) : 8 * ec, x1 = ct == 2 ? x0 + 8 : x0 + 4;
                            f32x4t v0 = *reinterpret_cast<const f32x4t*>(ws + row * 36 + x0);
                            f32x4t v1 = *reinterpret_cast<const f32x4t*>(ws + row * 36 + x1);
                            int off;
                            if (ct < 2) {
                                off = row < nrows ? (((b * a.heads2 + ct * 6 + j) * a.rows + t0 + row) * 32 + 8 * ec) * 2 : (int)0x80000000;
                            } else {   // (padding tokens of the tile: zeros — attention multiplies them by p = 0)
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v0[e] = x0 + e < nrows ? v0[e] : 0.f; v1[e] = x1 + e < nrows ? v1[e] : 0.f; }
                                off = (int)(((long long)b * a.v_bs + (long long)(32 * j + row) * a.v_pitch + t0 + 8 * ec) * 2);
                            }
                            if (ct == 0) { v0 *= a.q_scale; v1 *= a.q_scale; }
                            const u32x4 ov = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
                            __builtin_amdgcn_raw_buffer_store_b128(ov, ct < 2 ? rq : rv, off, 0, SAID_RG_ST_AUX);
                        }
                    } else {
#pragma unroll
                    for (int p4 = 0; p4 < 4; ++p4) {
                        const int row = 8 * p4 + prow;
                        f32x4t v = *reinterpret_cast<const f32x4t*>(ws + row * 36 + pc4);
                        int off;
                        if (ct < 2) {
                            off = row < nrows ? (((b * a.heads2 + ct * 6 + j) * a.rows + t0 + row) * 32 + pc4) * 4 : (int)0x80000000;
                        } else {   // (padding tokens of the tile: zeros — attention multiplies them by p = 0)
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = pc4 + e < nrows ? v[e] : 0.f;
                            off = (int)(((long long)b * a.v_bs + (long long)(32 * j + row) * a.v_pitch + t0 + pc4) * 4);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ct < 2 ? rq : rv, off, 0, SAID_RG_ST_AUX);
                    }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else if constexpr (EK == 2) {
                // GEGLU (ldm/attention.py:25-32): value * gelu(gate) of this wave's 32 channels, lane == channel; rows through the scratch
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 pr = geglu2((f32x2){acc[0][0][r], acc[0][0][r + 1]} + bias_n[0], (f32x2){acc[1][0][r], acc[1][0][r + 1]} + bias_n[1]);
                    ws[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + fr] = pr[0];
                    ws[(((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * lh) * 36 + fr] = pr[1];
                }
                __builtin_amdgcn_wave_barrier();
                const int c0 = a.geglu_c0(cg * 384 + 64 * j);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = 16 * q + er;
                    const f32x4t v0 = *reinterpret_cast<const f32x4t*>(ws + row * 36 + 8 * ec);
                    const f32x4t v1 = *reinterpret_cast<const f32x4t*>(ws + row * 36 + 8 * ec + 4);
                    const u32x4 ov = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_y, row < nrows ? ((R0 + row) * a.ldy + c0 + 8 * ec) * 2 : (int)0x80000000, 0, SAID_RG_ST_AUX);
                }
                __builtin_amdgcn_wave_barrier();
            } else if constexpr (EK == 4) {
                const int t = t0 + fr;
                const bool tv = t < a.M;
                const int tc = min(t, a.M - 1);
                band_head<true>(a, acc[0][0], b, t, tv, a.band_lo[tc], a.band_hi[tc], j, l);
            } else {
                // ---- this wave's 32 x 32 tile: MFMA layout (lane == column) -> scratch -> rows (lane == 8 consecutive columns of a row)
#pragma unroll
                for (int r = 0; r < 16; ++r) ws[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + fr] = acc[0][0][r] + add;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = 16 * q + er;
                    const bool on = row < nrows;
                    const f32x4t v0 = *reinterpret_cast<const f32x4t*>(ws + row * 36 + 8 * ec);
                    const f32x4t v1 = *reinterpret_cast<const f32x4t*>(ws + row * 36 + 8 * ec + 4);
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    if constexpr (RES != 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float r0 = bf_lo(rres[q][e]), r1 = bf_hi(rres[q][e]);
                            v[2 * e] += RES == 2 ? fmaf(r0, rca[2 * e], rcb[2 * e]) : r0;
                            v[2 * e + 1] += RES == 2 ? fmaf(r1, rca[2 * e + 1], rcb[2 * e + 1]) : r1;
                        }
                    }
                    const u32x4 ov = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
                    const int off = on ? ((R0 + row) * a.ldy + 32 * j + 8 * ec) * 2 : (int)0x80000000;   // (out of range: the hardware drops the store)
                    __builtin_amdgcn_raw_buffer_store_b128(ov, rsrc_y, off, 0, SAID_RG_ST_AUX);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(ov[e]); v[2 * e + 1] = bf_hi(ov[e]); }   // the stored values
                    if constexpr (DUP) {
                        const u32x4 o2 = {pack_bf16(v[0] + add2[0], v[1] + add2[1]), pack_bf16(v[2] + add2[2], v[3] + add2[3]),
                                          pack_bf16(v[4] + add2[4], v[5] + add2[5]), pack_bf16(v[6] + add2[6], v[7] + add2[7])};
                        __builtin_amdgcn_raw_buffer_store_b128(o2, rsrc_y2, on ? off + (int)(a.y2_row_off * a.ldy * 2) : (int)0x80000000, 0, SAID_RG_ST_AUX);
                    }
                    if constexpr (STATS) {   // back into the scratch for the column sums (lane == column again)
                        const f32x4t w0 = {v[0], v[1], v[2], v[3]}, w1 = {v[4], v[5], v[6], v[7]};
                        *reinterpret_cast<f32x4t*>(ws + row * 36 + 8 * ec) = w0;
                        *reinterpret_cast<f32x4t*>(ws + row * 36 + 8 * ec + 4) = w1;
                    }
                }
                if constexpr (STATS) {
                    __builtin_amdgcn_wave_barrier();
                    const float ref = ws[fr];                       // row 0 of the column (always a valid row)
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float xv = ws[row * 36 + fr];         // (unconditional read, then a select: a guarded read is a branch per register)
                        const float d = row < nrows ? xv - ref : 0.f;
                        s1 += d; s2 = fmaf(d, d, s2);
                    }
                    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                    if (lh == 0) {
                        float* so = a.stats + (long long)b * a.stats_bs + ((long long)(t0 >> 5) * a.N + 32 * j + fr) * 2;   // [tile][channel][2]
                        const float cnt = (float)nrows, md = s1 / cnt;
                        so[0] = ref + md;
                        so[1] = fmaxf(s2 - cnt * md * md, 0.f);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (k < 3) clk_stamp_p(a.clk, w, l, 3 + 3 * k);
            lds_barrier();
            if (k < 3) clk_stamp_p(a.clk, w, l, 4 + 3 * k);
        }
        clk_stamp_p(a.clk, w, l, 14);
        return;
    }

    // ======================================================================= helper waves
    // lane = (row slot l >> 4, channels 12 (l & 15) .. + 11 = 24 bytes): a row is the 16 lanes of one DPP row, so the LayerNorm sums are four
    // DPP adds; pass p covers rows 8 p + 4 hw + slot of the tile (clamped to its last row: the fifth pass of a 34-row tile rewrites row 33)
    const int hw = w - 6, slot = l >> 4, c12 = l & 15;
    constexpr bool GN_SRC = MODE == 1 || MODE == 3;
    const rsrc_t rsrc_src = make_rsrc(MODE != 0 ? a.ra[0] : a.sa[0], 0x7ffffff0u);
    // chunked K: NCH == 2: chunk c is the 192-wide tensor sa[c]; NCH == 4: the chunks are the four 192-column blocks of sa[0] (row pitch sld[0])
    const rsrc_t rsrc_src1 = make_rsrc((NCH == 2 && a.sa[1]) ? a.sa[1] : a.w, 0x7ffffff0u);
    const int src_ld = MODE != 0 ? 192 : a.sld[0];
    float ga[12], gb[12], lg[12], lb[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        ga[e] = 1.f; gb[e] = 0.f;
        lg[e] = MODE >= 2 ? a.ln_gamma[12 * c12 + e] : 1.f;
        lb[e] = MODE >= 2 ? a.ln_beta[12 * c12 + e] : 0.f;
    }
    // GroupNorm (a, b) tables of sample b -> coefS[kind][b & 3]: helper wave hw finalises channels [96 hw, 96 hw + 96) — two 48-channel
    // slices, all partial tiles of a slice in flight together — for the source (GN_SRC) and for the GroupNorm'ed residual (RES 2).  Tables are
    // published by the NEXT workgroup barrier: they are computed one period before the first tile of the sample is parked (prologue: before
    // an extra barrier).  Four slots: a slot is rewritten three samples later, when nothing of its sample is in flight any more.
    auto sample_tables = [&](int b) __attribute__((always_inline)) {
        float* const gs = gnscr + hw * GN_SCRATCH;
        if constexpr (GN_SRC) {
            const GnP gp = {a.gn_cpg, a.gn_nparts, a.M, a.gn_eps, a.gn_gamma, a.gn_beta, 192};
            const rsrc_t rp = make_rsrc(a.gn_part[0] + (long long)b * a.gn_part_bs, 192u * (unsigned)a.gn_nparts * 8u);
#pragma unroll 1
            for (int q = 0; q < 2; ++q) {   // (one slice at a time: 43 registers of loads; both at once spilled in the K = 576 kernels, and all tiles at once through LDS-DMA — no registers, one round trip — measured 4 % SLOWER per step: profiles/r04g_rgemm_ab.txt)
                GnL20 g0;
                gn20_issue(gp, rp, 96 * hw + 48 * q, l, g0);
                gn20_finish(gp, rp, 96 * hw + 48 * q, l, g0, gs, coefS + (b & 3) * 384);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if constexpr (EK == 0 && RES == 2) {
            const GnP gp = {a.gn_cpg, a.gn_nparts, a.M, a.res_eps, a.res_gamma, a.res_beta, 192};
            const rsrc_t rp = make_rsrc(a.res_part + (long long)b * a.gn_part_bs, 192u * (unsigned)a.gn_nparts * 8u);
#pragma unroll 1
            for (int q = 0; q < 2; ++q) {   // (one slice at a time: 43 registers of loads; both at once spilled in the K = 576 kernels, and all tiles at once through LDS-DMA — no registers, one round trip — measured 4 % SLOWER per step: profiles/r04g_rgemm_ab.txt)
                GnL20 g0;
                gn20_issue(gp, rp, 96 * hw + 48 * q, l, g0);
                gn20_finish(gp, rp, 96 * hw + 48 * q, l, g0, gs, coefS + (4 + (b & 3)) * 384);
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    constexpr bool TABLES = GN_SRC || (EK == 0 && RES == 2);
    int cur_b_src = -1;

    u32x4 rawa[NPASS];
    u32x2 rawb[NPASS];
    // unit u = tile * NCH + chunk
    auto issue_tile = [&](int u_) __attribute__((always_inline)) {
        const int u = min(u_, t_last * NCH + NCH - 1);   // (units past the workgroup's range: clamped, their work is done and discarded)
        const int ti = u / NCH;
        const rsrc_t rs = (NCH == 2 && (u % NCH) == 1) ? rsrc_src1 : rsrc_src;
        const int col = NCH == 4 ? 192 * (u % NCH) : 0;
        const int b = ti / ntv, t0 = (ti - b * ntv) * 32;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = min(8 * p + 4 * hw + slot, NR - 1);
            const int tt = t0 + r - HALO;
            const int off = ((b * a.seg_rows + min(max(tt, 0), a.M - 1)) * src_ld + col + 12 * c12) * 2;
            rawa[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            rawb[p] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off + 16, 0, 0));
        }
    };
    auto park_tile = [&](int u_, int buf) __attribute__((always_inline)) {
        const int ti = min(u_, t_last * NCH + NCH - 1) / NCH;
        const int b = ti / ntv, t0 = (ti - b * ntv) * 32;
        if constexpr (GN_SRC) {
            if (b != cur_b_src) {   // (published at least one barrier ago)
                const float* const tab = coefS + (b & 3) * 384;
#pragma unroll
                for (int e = 0; e < 12; ++e) { ga[e] = tab[2 * (12 * c12 + e)]; gb[e] = tab[2 * (12 * c12 + e) + 1]; }
                cur_b_src = b;
            }
        }
        u32x2* const A64 = reinterpret_cast<u32x2*>(Abuf + buf * RG_A_ELEMS);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = min(8 * p + 4 * hw + slot, NR - 1);
            const int tt = t0 + r - HALO;
            const bool valid = tt >= 0 && tt < a.M;
            unsigned o[6] = {rawa[p][0], rawa[p][1], rawa[p][2], rawa[p][3], rawb[p][0], rawb[p][1]};
            if constexpr (MODE != 0) {
                float x[12];
#pragma unroll
                for (int e = 0; e < 6; ++e) { x[2 * e] = bf_lo(o[e]); x[2 * e + 1] = bf_hi(o[e]); }
                if constexpr (GN_SRC) {
#pragma unroll
                    for (int e = 0; e < 12; ++e) x[e] = fmaf(x[e], ga[e], gb[e]);
                }
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int e = 0; e < 12; ++e) x[e] = silu_f(x[e]);
                }
                if constexpr (MODE >= 2) {   // LayerNorm over the row's 192 channels = the 16 lanes of this DPP row (two-pass: mean, then squares)
                    float s1 = 0.f;
#pragma unroll
                    for (int e = 0; e < 12; ++e) s1 += x[e];
                    const float mu = row16_sum(s1) * (1.0f / 192.0f);
                    float s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 12; ++e) { const float d = x[e] - mu; s2 = fmaf(d, d, s2); }
                    const float rs = __builtin_amdgcn_rsqf(row16_sum(s2) * (1.0f / 192.0f) + 1e-5f);   // v_rsq_f32 (1 ulp); the IEEE 1 / sqrt sequence is ~15 VALU instructions per row
#pragma unroll
                    for (int e = 0; e < 12; ++e) x[e] = fmaf((x[e] - mu) * rs, lg[e], lb[e]);
                }
#pragma unroll
                for (int e = 0; e < 6; ++e) o[e] = pack_bf16(x[2 * e], x[2 * e + 1]);
            }
#pragma unroll
            for (int e = 0; e < 6; ++e) o[e] = valid ? o[e] : 0u;
            u32x2* d = A64 + r * (AP / 4) + 3 * c12;
            d[0] = u32x2{o[0], o[1]}; d[1] = u32x2{o[2], o[3]}; d[2] = u32x2{o[4], o[5]};
        }
    };

    // ---- prologue
    clk_stamp_p(a.clk, w, l, 0);
    const int u0 = tb * NCH;
    issue_tile(u0);
    clk_stamp_p(a.clk, w, l, 11);
    lds_barrier();                                 // (the tables of the first two tiles' samples: MFMA waves 0-3)
    clk_stamp_p(a.clk, w, l, 12);
    park_tile(u0, 0);
    issue_tile(u0 + 1);
    lds_barrier();
    clk_stamp_p(a.clk, w, l, 1);
    for (int q = 0; q < n * NCH; ++q) {
        park_tile(u0 + q + 1, (q + 1) & 1);
        if (q < 3) clk_stamp_p(a.clk, w, l, 2 + 3 * q);
        issue_tile(u0 + q + 2);
        if constexpr (TABLES) {   // a new sample two tiles ahead: its tables now, published by this period's barrier (rare: a self-contained block)
            const int b1 = min(tb + q + 1, t_last) / ntv, b2 = min(tb + q + 2, t_last) / ntv;
            if (b2 != b1) sample_tables(b2);
        }
        if (q < 3) clk_stamp_p(a.clk, w, l, 3 + 3 * q);
        lds_barrier();
        if (q < 3) clk_stamp_p(a.clk, w, l, 4 + 3 * q);
    }
    clk_stamp_p(a.clk, w, l, 14);
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct RgPlan { int ntap, ek, mode, res, dup, stats, nch, per, grid, groups; };
static bool rg_plan(const TGemmArgs& a, RgPlan& p) {
    if (a.f32 || a.seg_rows <= 0 || a.seg_rows % 32 || a.M < 1 || a.M > a.seg_rows) return false;
    if (a.ra[1] || a.sk[2] || a.y_cm) return false;
    p.nch = 1;
    if (a.ra[0]) {
        if (a.sk[0] || a.sk[1] || (a.rtaps != 1 && a.rtaps != 3) || a.K != a.rtaps * 192) return false;
        if (a.rmode < 1 || a.rmode > 3) return false;
        if ((a.rmode == 1 || a.rmode == 3) && (!a.gn_part[0] || !a.gn_gamma || !a.gn_beta || a.res_gn)) return false;
        if (a.rmode >= 2 && (!a.ln_gamma || !a.ln_beta || a.rtaps != 1)) return false;
        p.ntap = a.rtaps; p.mode = a.rmode;
    } else {
        if (!a.sa[0]) return false;
        if (a.sk[0] == 768) { if (a.sld[0] != 768 || a.sk[1] || a.K != 768) return false; p.nch = 4; }
        else if (a.sk[0] != 192 || a.sld[0] != 192) return false;
        else if (a.sk[1]) { if (a.sk[1] != 192 || a.sld[1] != 192 || !a.sa[1] || a.K != 384) return false; p.nch = 2; }
        else if (a.K != 192) return false;
        p.ntap = 1; p.mode = 0;
    }
    p.res = 0; p.dup = 0; p.stats = 0; p.groups = 1;
    if (a.geglu) {
        if (a.N % 384 || a.N / 384 > 32 || !a.yb || a.ldy % 8 || !a.ra[0] || p.ntap != 1 || a.band_k || a.qk) return false;
        if ((long long)a.batch * a.seg_rows * a.ldy * 2 > 0x7ffffff0LL) return false;
        p.ek = 2; p.groups = a.N / 384;
    } else if (a.band_k) {
        if (a.N != 192 || !a.ra[0] || !a.y_tm || !a.band_lo || !a.band_hi || a.band_wmax < 1 || a.band_wmax > 8 || p.ntap != 1) return false;
        p.ek = 4;
    } else if (a.qk) {
        if (a.N != 576 || a.qk_n != 384 || a.head_dim != 32 || a.heads2 != 12 || !a.vt || p.ntap != 1) return false;
        p.ek = 1;
    } else if (a.y_tm) {
        if (a.N != 192 || a.ldy != 192 || (a.res_tm && a.ldr_tm != 192)) return false;
        if (a.res_gn && (!a.res_tm || !a.res_part || !a.res_gamma || !a.res_beta)) return false;
        p.ek = 0;
        p.res = a.res_tm ? (a.res_gn ? 2 : 1) : 0;
        p.dup = a.y2_tm ? 1 : 0;
        p.stats = a.stats ? 1 : 0;
    } else return false;
    if ((long long)a.batch * a.seg_rows * (p.nch == 4 ? 768 : 192) * 2 > 0x7ffffff0LL) return false;   // 32-bit byte offsets into the activation tensors
    if (a.qk) {   // ... and into the q / k / v outputs: the fp32 q / k layout is 1536 B per token, four times the bf16 activations' 384
        const long long el = a.qkv_bf16 ? 2 : 4;
        if ((long long)a.batch * a.heads2 * a.rows * 32 * el > 0x7ffffff0LL) return false;
        if ((long long)a.batch * a.v_bs * el > 0x7ffffff0LL) return false;
    }
    const int total = a.batch * ((a.M + 31) / 32);
    const int wpg = std::max(8, 256 / p.groups / 8 * 8);          // workgroups per column group: one workgroup per CU in all
    p.per = (total + wpg - 1) / wpg;
    const int ranges = (total + p.per - 1) / p.per;
    p.grid = 8 * p.groups * ((ranges + 7) / 8);
    return true;
}
// the instantiations the UNet schedule needs (engine.cpp: run_resblock_tm, run_transformer_tm)
#define RG_VARIANTS(X)                                                                                     \
    X(3, 0, 1, 0, false, true, 1)   /* conv1: silu(GN(x)) -> conv3 + emb, statistics                       */ \
    X(3, 0, 1, 1, false, true, 1)   /* conv2 + identity skip                                              */ \
    X(3, 0, 1, 1, true, true, 1)    /* ... written to both guidance halves                                 */ \
    X(1, 0, 0, 2, false, false, 1)  /* attn1.to_out + GroupNorm(x_in)                                      */ \
    X(1, 0, 0, 2, true, false, 1)   /* ... + the unconditional half's x2 = x1 + c2                         */ \
    X(1, 0, 0, 1, false, false, 1)  /* attn2.to_out + x1                                                   */ \
    X(1, 1, 3, 0, false, false, 1)  /* q/k/v of LayerNorm(GroupNorm(x))                                    */ \
    X(1, 4, 2, 0, false, false, 1)  /* q of LayerNorm(x1) + banded cross-attention                         */ \
    X(1, 2, 2, 0, false, false, 1)  /* GEGLU of LayerNorm(x2): four column groups of six (value, gate) pairs */ \
    X(3, 0, 1, 0, false, false, 1)  /* one source of a concatenated-input convolution -> partial sum        */ \
    X(1, 0, 0, 0, false, false, 2)  /* 1x1 skip convolution over the concatenated raw input (two chunks)    */ \
    X(1, 0, 0, 1, false, false, 4)  /* folded proj_out o ff.net.2, the GEGLU product's 768 columns + x_in -> partial sum */ \
    X(1, 0, 0, 1, false, true, 1)   /* ... x2's 192 columns + bias + the partial sum, statistics            */
static int rg_variant(const RgPlan& p) {
    int i = 0;
#define X(NT_, EK_, MO_, RE_, DU_, ST_, NC_) if (p.ntap == NT_ && p.ek == EK_ && p.mode == MO_ && p.res == RE_ && (p.dup != 0) == DU_ && (p.stats != 0) == ST_ && p.nch == NC_) return i; ++i;
    RG_VARIANTS(X)
#undef X
    return -1;
}
bool rgemm_supports(const TGemmArgs& a_in, int batch) {
    TGemmArgs a = a_in; a.batch = batch;
    RgPlan p;
    return rg_plan(a, p) && rg_variant(p) >= 0;
}
void configure_rgemm_kernels() {
#define X(nt, ek, mo, re, du, st, nc) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rgemm_kernel<nt, ek, mo, re, du, st, nc>), hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS_BYTES);
    RG_VARIANTS(X)
#undef X
}
bool launch_rgemm(const TGemmArgs& a_in, int batch, hipStream_t s) {
    TGemmArgs a = a_in;
    a.batch = batch;
    RgPlan p;
    if (!rg_plan(a, p)) return false;
    const int v = rg_variant(p);
    if (v < 0) return false;
    a.pg_per = p.per; a.pg_s = p.groups;
    int i = 0;
#define X(nt, ek, mo, re, du, st, nc) if (v == i) hipLaunchKernelGGL((rgemm_kernel<nt, ek, mo, re, du, st, nc>), dim3((unsigned)p.grid), dim3(512), RG_LDS_BYTES, s, a); ++i;
    RG_VARIANTS(X)
#undef X
    return true;
}

}  // namespace said

// tgemm_dev.h — device code shared by the token-major GEMM kernels (tgemm.hip: tgemm / fgemm / xgemm; rgemm.hip: the persistent register-stationary GEMMs of round 4): the epilogues of one 32-row MFMA tile and the banded cross-attention of one head.
#pragma once
#include "gemm_common.h"
#include "tgemm.h"
#include "split_f16.h"

namespace said {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------------
// Epilogues shared by the two tile shapes, one 32-row tile of a wave at a time.
//
// The MFMA result layout is D[row][n]: lane -> output column n (l & 31 within a 32-column tile), register r -> row
// (r & 3) + 8 (r >> 2) + 4 (l >> 5).  Storing straight from that layout makes every store instruction touch up to 64
// different cache lines (channel-major results: one 16-byte piece per lane; token-major bf16: 2-byte elements), and the
// knock-out experiment of round 2 (SAID_TG_DBG, scripts/gpu_r2_j.sh) measured the epilogue at 53-57 % of the kernels' time.
// So the tile goes through a wave-private LDS scratch [32 rows][32 NJ + 4] first (the operand buffers are free after the K
// loop) — bias / timestep-embedding term / activation / GEGLU product are applied on the way in, where lane == column —
// and is read back in the layout the destination wants:
//   * token-major (audio encoder, GEGLU, q / k heads): 16 bytes of consecutive columns per lane, a row = 32 NJ / 4 lanes;
//   * channel-major (UNet results, v rows): 16 bytes of consecutive TOKENS per lane, 8 lanes per channel row — full 128-byte
//     lines; residual loads use the same mapping, GroupNorm partials reduce over the 8 lanes of a row.
// A row tile starts at global row rt; with seg_rows > 0 (UNet: all samples form one row axis, per-sample pitch seg_rows, a
// multiple of 32) the sample is rt / seg_rows and the first token rt % seg_rows, else sample b_grid and token rt.
// n0w = first output column of this wave's 32 NJ columns.  No workgroup barrier inside (waves may return early): LDS
// operations of one wave execute in order, so its own writes are visible to its later reads.
// ------------------------------------------------------------------------------------------------------------------
// (J0, NJE): the epilogue covers column tiles J0 .. J0 + NJE - 1 of the wave's NJ accumulator tiles, n0w = first column of tile J0
// (fgemm_kernel splits a tile's epilogue between the two K-half waves).
// round 3 (xgemm_kernel): `res_tm` adds a token-major residual in phase 1 (EK == 3 only; the token-major activation destination has
// its own all-wave epilogue inside xgemm_kernel).
// EK (xgemm_kernel): the epilogue kind is a compile-time constant there, so the paths a launch cannot take cost it no registers —
// -1: any (the round-2 kernels), 1: q/k/v split, 2: GEGLU, 3: channel-major fp32 result (+ token-major residual).
// PH (fgemm_kernel, round 3): 0 = both phases by the calling wave; 1 = phase 1 only; 2 = phase 2 only, on HALF `half` of the work (phase 2b:
// the lower / upper half of the channels, phases 2a / 2a': rows 0-15 / 16-31) — the two K-half waves of a row half then share the
// memory-facing phase instead of one of them idling (the scratch is the row half's, a workgroup barrier separates the phases).
template <int NJ, int J0 = 0, int NJE = NJ, int EK = -1, int PH = 0>
__device__ __forceinline__ void tg_epilogue(const TGemmArgs& a, f32x16 (&acc)[NJ], int b_grid, int rt, int n0w, int l, float* sc,
                                            const float* coefR = nullptr, int half = 0, int n_lim = 0) {
    constexpr bool ANY = EK < 0, P_CM = ANY || EK == 1 || EK == 3, P_BF = ANY || EK == 2, P_GEN = ANY || EK == 1 || EK == 2,
                   P_RES = EK == 3, P_GEGLU = ANY || EK == 2;
    constexpr int CW = 32 * NJE, CP = CW + 4;   // columns of this call, scratch row pitch (floats)
    const int lh = l >> 5, lc = l & 31;
    if (a.seg_rows > 0 && rt >= a.batch * a.seg_rows) return;   // row tile past the last sample (wave-uniform)
    const int b = a.seg_rows > 0 ? rt / a.seg_rows : b_grid;
    const int mt = a.seg_rows > 0 ? rt - b * a.seg_rows : rt;
    if (mt >= a.M) return;
    if (a.dbg & 1) {   // timing experiment (SAID_TG_DBG=1): no epilogue memory traffic — results are WRONG, never used in tests
        if (acc[0][0] == 12345.678f && a.yf) a.yf[0] = acc[0][1];
        return;
    }
    const int nrows = min(32, a.M - mt);
    const int n_store = n_lim ? n_lim : a.n_store;   // first column NOT stored (grouped launches: the group's own limit)
    const long long R0 = (long long)b * a.seg_rows + mt;   // global row of the tile's first token (token-major activation tensors)
    // ---- phase 1: registers -> scratch, elementwise work where lane == column
    const bool geglu = P_GEGLU && a.geglu != 0;
    constexpr int NJO = NJE;   // (GEGLU writes NJ / 2 column tiles; the scratch keeps the full pitch)
    if constexpr (PH != 2) {
#pragma unroll
    for (int j = 0; j < NJE; ++j) {
        if (geglu && (j & 1)) continue;   // gate tiles are consumed with their value tile
        const int n = n0w + j * 32 + lc;
        float add = a.bias ? a.bias[n] : 0.f;
        float gadd = 0.f;
        if (geglu) gadd = a.bias ? a.bias[n + 32] : 0.f;
        if (a.emb) add += a.emb[(long long)n * a.emb_pitch + (a.step_ptr ? *a.step_ptr : 0) + b * a.emb_b_stride];
        const int col = geglu ? (j >> 1) * 32 + lc : j * 32 + lc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = acc[J0 + j][r] + add;
            if constexpr (P_RES) {
                if (a.res_tm) {   // token-major residual (the one launch per step that feeds the channel-major `out` convolution)
                    const long long ro = (R0 + min(row, nrows - 1)) * a.ldr_tm + n;
                    v += a.f32 ? reinterpret_cast<const float*>(a.res_tm)[ro] : (float)reinterpret_cast<const __bf16*>(a.res_tm)[ro];
                }
            }
            if (geglu) {
                if constexpr (NJE % 2 == 0 && J0 % 2 == 0) v = geglu_f(v, acc[J0 + (j + 1) % NJE][r] + gadd);
            } else if (a.act == 1) {
                v = gelu_f(v);
            }
            sc[row * CP + col] = v;
        }
    }
    }
    (void)NJO;
    if constexpr (PH == 1) return;
    __builtin_amdgcn_wave_barrier();
    const int cw = geglu ? CW / 2 : CW;                 // live columns in the scratch
    const int c_lo = PH == 2 ? half * (cw / 2) : 0, c_hi = PH == 2 ? c_lo + cw / 2 : cw;      // phase 2b's channel range
    const int r_lo = PH == 2 ? 16 * half : 0, r_hi = PH == 2 ? min(nrows, r_lo + 16) : nrows;   // phase 2a's row range
    const int n_first = geglu ? a.geglu_c0(n0w) : n0w;   // first destination column
    const bool v_rows = a.qk && n0w >= a.qk_n;          // this wave holds v columns (channel-major) of a q/k/v projection
    if (P_CM && (a.y_cm || v_rows)) {
        // ---- phase 2b: channel-major destination: lane -> (channel l >> 3 of the pass, token quad l & 7)
        const int tq = l & 7;
        const int pitch = a.y_cm ? a.cm_pitch : a.v_pitch;
        float* const ybase = a.y_cm ? a.y_cm + (long long)b * a.cm_bs : a.vt + (long long)b * a.v_bs - (long long)a.qk_n * a.v_pitch;
        const int nparts = (a.M + 31) >> 5;
        for (int c0 = c_lo; c0 < c_hi; c0 += 8) {
            const int c = c0 + (l >> 3);
            const int n = n_first + c;
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = sc[(4 * tq + e) * CP + c];
            const int m = mt + 4 * tq;
            if (a.res_cm && m < a.M) {   // pitch >= roundup(M, 32): the whole quad is in bounds
                const float4 rv = *reinterpret_cast<const float4*>(a.res_cm + (long long)b * a.res_cm_bs + (long long)n * pitch + m);
                if (a.res_cm_coef) {   // GroupNorm'ed residual (attn1.to_out: + norm(x_in), attention.py:168)
                    const float2 cf = *reinterpret_cast<const float2*>(a.res_cm_coef + (long long)b * a.res_cm_coef_bs + 2 * n);
                    v4[0] += fmaf(rv.x, cf.x, cf.y); v4[1] += fmaf(rv.y, cf.x, cf.y); v4[2] += fmaf(rv.z, cf.x, cf.y); v4[3] += fmaf(rv.w, cf.x, cf.y);
                } else {
                    v4[0] += rv.x; v4[1] += rv.y; v4[2] += rv.z; v4[3] += rv.w;
                }
            }
            if (m < a.M) {   // tokens in [M, roundup(M, 4)) land in the row's padding: written as ZEROS (they come from operand rows
                             // nobody prepared; attention multiplies V's padding columns by p = 0, and 0 x NaN is NaN)
#pragma unroll
                for (int e = 1; e < 4; ++e) v4[e] = (m + e < a.M) ? v4[e] : 0.f;
                if (v_rows && a.kv_pack) {   // v for attn_kernel<PM = 3>: packed split pairs (0 stays all-zero bits)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = pack_split_f16(v4[e]);
                }
                const float4 o = make_float4(v4[0], v4[1], v4[2], v4[3]);
                *reinterpret_cast<float4*>(ybase + (long long)n * pitch + m) = o;
                if (a.y2_cm) {
                    const float ad = a.y2_add_cm ? a.y2_add_cm[n] : 0.f;
                    *reinterpret_cast<float4*>(a.y2_cm + (long long)b * a.y2_bs + (long long)n * pitch + m) =
                        make_float4(v4[0] + ad, (m + 1 < a.M) ? v4[1] + ad : 0.f, (m + 2 < a.M) ? v4[2] + ad : 0.f, (m + 3 < a.M) ? v4[3] + ad : 0.f);
                }
            }
            if (a.stats) {   // Welford partial of channel n over this 32-token tile: reduce over the row's 8 lanes
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += (m + e < a.M) ? v4[e] : 0.f;
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
                const float mean = sum / (float)nrows;
                float m2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = (m + e < a.M) ? v4[e] - mean : 0.f; m2 = fmaf(d, d, m2); }
                m2 += __shfl_xor(m2, 1); m2 += __shfl_xor(m2, 2); m2 += __shfl_xor(m2, 4);
                if (tq == 0) {
                    float* so = a.stats + (long long)b * a.stats_bs + ((long long)(mt >> 5) * a.N + n) * 2;   // [tile][channel][2]
                    so[0] = mean;
                    so[1] = m2;
                }
            }
        }
    } else if (P_BF && a.yb && !a.yf && !a.qk && !a.res && !n_store) {
        // ---- phase 2a', bf16-only token-major destination (GEGLU product, the audio encoder's conv / FFN activations): lane -> 8
        // consecutive columns = one 16-byte store (8-byte stores run at 0.54-0.70x the 16-byte rate)
        typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
        const int lanes_per_row = cw / 8;                // 16, 12, 8 or 4
        const int rows_pp = 64 / lanes_per_row;          // 4, 5 (60 lanes active), 8 or 16
        const int rr = l / lanes_per_row, cq = l - rr * lanes_per_row;
        const bool lane_on = rr < rows_pp;
        for (int r0 = r_lo; r0 < r_hi; r0 += rows_pp) {
            const int row = r0 + rr;
            if (!lane_on || row >= r_hi) continue;
            const f32x4t v0 = *reinterpret_cast<const f32x4t*>(sc + row * CP + 8 * cq);
            const f32x4t v1 = *reinterpret_cast<const f32x4t*>(sc + row * CP + 8 * cq + 4);
            const bf16x8s o = {(__bf16)v0[0], (__bf16)v0[1], (__bf16)v0[2], (__bf16)v0[3], (__bf16)v1[0], (__bf16)v1[1], (__bf16)v1[2], (__bf16)v1[3]};
            *reinterpret_cast<bf16x8s*>(reinterpret_cast<__bf16*>(a.yb) + (long long)b * a.y_bs + (long long)(mt + row) * a.ldy + n_first + 8 * cq) = o;
        }
    } else if (P_GEN) {
        // ---- phase 2a: token-major destination: lane -> 4 consecutive columns of a row; rows_pp rows per pass
        const int lanes_per_row = cw / 4;                // 32, 24, 16 or 8
        const int rows_pp = 64 / lanes_per_row;          // 2, 2 (48 lanes active), 4 or 8
        const int rr = l / lanes_per_row, cq = l - rr * lanes_per_row;
        const bool lane_on = rr < rows_pp;
        for (int r0 = r_lo; r0 < r_hi; r0 += rows_pp) {
            const int row = r0 + rr;
            if (!lane_on || row >= r_hi) continue;
            const int m = mt + row;
            const int n = n_first + 4 * cq;
            if (n_store && n >= n_store) continue;
            f32x4t v = *reinterpret_cast<const f32x4t*>(sc + row * CP + 4 * cq);
            if (a.res) {
                const f32x4t rv = *reinterpret_cast<const f32x4t*>(a.res + (long long)b * a.res_bs + (long long)m * a.ldr + n);
                v += rv;
            }
            if (a.qk) {   // q / k heads token-major [b][2 heads][rows][head_dim]; head_dim % 4 == 0
                const int h = n / a.head_dim, d = n - h * a.head_dim;
                if (a.kv_pack && 2 * h >= a.heads2) {   // k heads (the second half of the head axis) as packed split pairs
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pack_split_f16(v[e]);
                }
                *reinterpret_cast<f32x4t*>(a.qk + (((long long)b * a.heads2 + h) * a.rows + m) * a.head_dim + d) = v;
            } else {
                if (a.yf) *reinterpret_cast<f32x4t*>(a.yf + (long long)b * a.y_bs + (long long)m * a.ldy + n) = v;
                if (a.yb) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.yb) + (long long)b * a.y_bs + (long long)m * a.ldy + n) = o;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();   // the scratch is reused for the wave's next row tile
}

static __device__ __forceinline__ f32x4t xbload4(rsrc_t r, int voff) {
    return __builtin_bit_cast(f32x4t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

// banded cross-attention on the transposed q tile of one head: acc[r] = q[d = (r & 3) + 8 (r >> 2) + 4 lh][token lt]
template <bool BF>
__device__ __forceinline__ void band_head(const TGemmArgs& a, const f32x16& q, int b, int t, bool tv, int lo, int hi, int head, int l) {
    const int lh = l >> 5;
    const int kvp = a.band_kv_pitch;
    const long long kvo = (long long)b * a.band_kv_bs + (long long)(head * 32) * kvp;
    const rsrc_t rk = make_rsrc(a.band_k + kvo, 32u * (unsigned)kvp * 4u);
    const rsrc_t rv = make_rsrc(a.band_v + kvo, 32u * (unsigned)kvp * 4u);
    const int nh = a.band_wmax > 4 ? 2 : 1;
    float sc[8];
#pragma unroll
    for (int wi = 0; wi < 8; ++wi) sc[wi] = 0.f;
    for (int hf = 0; hf < nh; ++hf) {
        f32x4t kq[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) kq[r] = xbload4(rk, (((r & 3) + 8 * (r >> 2) + 4 * lh) * kvp + lo + 4 * hf) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (hf == 0) sc[e] = fmaf(q[r], kq[r][e], sc[e]); else sc[4 + e] = fmaf(q[r], kq[r][e], sc[4 + e]);
            }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int wi = 0; wi < 8; ++wi) {
        sc[wi] += __shfl_xor(sc[wi], 32);
        const bool vis = (wi < a.band_wmax) && (lo + wi < hi);
        sc[wi] = vis ? sc[wi] * a.band_scale : -3.0e38f;   // scale after QK^T (ldm/attention.py:101), masked keys at -max
        mx = fmaxf(mx, sc[wi]);
    }
    float den = 0.f;
#pragma unroll
    for (int wi = 0; wi < 8; ++wi) {
        const bool vis = (wi < a.band_wmax) && (lo + wi < hi);
        sc[wi] = vis ? __expf(sc[wi] - mx) : 0.f;
        den += sc[wi];
    }
    const float inv = 1.0f / den;
    float o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    for (int hf = 0; hf < nh; ++hf) {
        f32x4t vq[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) vq[r] = xbload4(rv, (((r & 3) + 8 * (r >> 2) + 4 * lh) * kvp + lo + 4 * hf) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int wi = 4 * hf + e;
                const bool vis = (wi < a.band_wmax) && (lo + wi < hi);   // invisible slots may hold another row's data
                o[r] = fmaf((hf == 0 ? sc[e] : sc[4 + e]) * inv, vis ? vq[r][e] : 0.f, o[r]);
            }
    }
    if (!tv) return;
    // registers 4 q .. 4 q + 3 are channels 8 q + 4 lh + (0 .. 3) of the head: four consecutive elements of the token's row
    const long long row = (long long)b * a.seg_rows + t;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const long long off = row * a.ldy + head * 32 + 8 * qd + 4 * lh;
        if constexpr (BF) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            const bf16x4 ov = {(__bf16)o[4 * qd], (__bf16)o[4 * qd + 1], (__bf16)o[4 * qd + 2], (__bf16)o[4 * qd + 3]};
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.y_tm) + off) = ov;
        } else {
            const f32x4t ov = {o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]};
            *reinterpret_cast<f32x4t*>(reinterpret_cast<float*>(a.y_tm) + off) = ov;
        }
    }
}

}  // namespace said

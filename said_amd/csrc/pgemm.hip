// pgemm.hip — PERSISTENT, WEIGHT-STATIONARY GEMMs on token-major activations (round 4; bf16 mode, large batches).
//
//     Y[row][n] = epi( sum_k xform(A)[row][k] * W[n][k] )       row = sample * seg_rows + token, activations [row][192] bf16
//
// Round 3's xgemm_kernel gave every 64-token row tile its own workgroup, which kept its SOURCE tile in LDS and streamed the
// weights of its column tiles through a single-stage pipeline: 1.5k clocks per k-step against 192 clocks of MFMA (the weight
// tile's L2 latency sat on every k-step), a 18k-clock prologue per 64 tokens, and launches of 640-1280 workgroups that load,
// multiply and store in lockstep.  This kernel turns the roles around:
//
//   * a workgroup owns ONE column slice of the weights — 32 NJ output columns x the whole K — for the whole launch: loaded
//     once into LDS ([n][K + 8] bf16: rows 16 bytes longer than K, so the 16-byte MFMA fragment reads of 32 consecutive rows
//     fall on distinct banks), 38 KB (K = 192, 96 columns) ... 124 KB (K = 960, 64 columns);
//   * it then walks over ITS contiguous share of the launch's row tiles.  A tile's A operand arrives as a sequence of
//     192-channel CHUNKS (one per source and K segment: the GroupNorm'ed / LayerNorm'ed input, the raw attention output, the
//     four 192-wide blocks of the GEGLU product, ...), each loaded into registers one step ahead — while the previous chunk
//     multiplies — transformed once per element (silu(GroupNorm) / LayerNorm / LayerNorm(GroupNorm) / raw) and parked in a
//     66-row LDS tile; the three taps of a convolution are three row offsets into it.  The k loop touches only LDS;
//   * the grid is 256 (one per CU, LDS > 80 KB) or 512 (two per CU) workgroups for the WHOLE batch, XCD-aware: the slices of
//     one row group sit on one XCD (they read the same source rows), so `clip groups` (round 3) are no longer needed to
//     fill the chip, and the GroupNorm coefficients are finalised once per sample a workgroup touches, not once per row tile.
//
// Wave roles: four waves = 2 row halves x 2 K halves (each wave 32 rows x 32 NJ columns over six of a chunk's twelve k16 steps;
// the halves meet through LDS once per row tile), except the GEGLU shape: 2 row halves x 2 (value, gate) column pairs, whole K.
// Epilogues: the token-major activation epilogue on all four waves (bias, timestep-embedding term, residual — optionally
// GroupNorm'ed —, rounding, duplicate store, GroupNorm partials of the stored values), or tgemm_dev.h's (q/k/v split, channel-major
// fp32), the GEGLU product, the banded cross-attention (transposed product).  Reference semantics:
// /root/reference/said/model/ldm/openaimodel.py:116-227 (ResBlock), ldm/attention.py:131-234 (transformer block).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gemm_common.h"
#include "tgemm.h"
#include "tgemm_dev.h"

namespace said {

constexpr int PG_AP = 200;                       // A tile row pitch in elements: 192 + 8 (400 bytes)
constexpr int PG_COEF_BYTES = 2 * 2 * 192 * 4;   // GroupNorm (a, b) of two 192-channel sources (slot 0 also: the GroupNorm'ed residual)
template <int NJ> __host__ __device__ constexpr int pg_statx_bytes() { return 2 * 2 * 32 * NJ * 4; }
__host__ __device__ constexpr int pg_a_bytes(int arows) { return arows * PG_AP * 2; }
template <int NJ>
__host__ __device__ constexpr int pg_lds_bytes(int K, int arows) {
    return 32 * NJ * (K + 8) * 2 + pg_a_bytes(arows) + PG_COEF_BYTES + pg_statx_bytes<NJ>();
}

// EK: 0 token-major activation out, 1 q/k/v split, 2 GEGLU product, 3 channel-major fp32 out, 4 banded cross-attention (TR)
// T3: chunk 0 may be a 3-tap convolution source (a third, two-row load pass for the halo)
// (the banded cross-attention epilogue holds 16 K / V fragments of 16 bytes per lane beside the accumulators: budgeted for one workgroup per CU)
template <int NJ, int EK, bool T3>
__global__ __launch_bounds__(256, (EK == 4 ? 1 : 2)) void pgemm_kernel(const TGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    typedef unsigned short elt_t;
    constexpr bool TR = EK == 4;
    constexpr bool KSPLIT = EK != 2;
    constexpr int NA = KSPLIT ? NJ : NJ / 2;      // accumulator tiles per wave
    constexpr int BN = 32 * NJ, AP = PG_AP;
    constexpr int NP = T3 ? 3 : 2;                // load passes per chunk: 2 x 32 rows (+ the two halo rows)
    const int K = a.K, WP = K + 8;
    elt_t* const Wl = lds;
    elt_t* const Al = lds + BN * WP;
    const int arows = T3 ? 66 : 64;
    float* const Af = reinterpret_cast<float*>(Al);
    float* const coefS = reinterpret_cast<float*>(Al + arows * AP);
    float* const statx = coefS + PG_COEF_BYTES / 4;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wr = w & 1, kh = w >> 1;            // GEGLU: kh = column pair
    // ---- which slice, which row tiles
    const int S = a.pg_s;
    const unsigned L = blockIdx.x, xcd = L & 7u, slot = L >> 3;
    const int slice = (int)(slot % (unsigned)S);
    const int rg = (int)(slot / (unsigned)S) * 8 + (int)xcd;
    const int tps = a.seg_rows >> 6;              // row tiles per sample
    const int MT = a.batch * tps;
    const int tb = rg * a.pg_per, te = min(MT, tb + a.pg_per);
    if (tb >= te) return;
    const int n0 = slice * BN;
    const elt_t* W = reinterpret_cast<const elt_t*>(a.w);

    // ---- chunks of one row tile
    const int nres = a.ra[0] ? (a.ra[1] ? 2 : 1) : 0;
    const int c0n = a.sk[0] / 192, c1n = a.sk[1] / 192, c2n = a.sk[2] / 192;
    const int nch = nres + c0n + c1n + c2n;
    const int kres = nres * a.rtaps * 192;
    struct Chunk { const elt_t* base; int ld, col, mode, taps, wk, wkt; };
    auto chunk_of = [&](int c) -> Chunk {
        Chunk ch;
        if (c < nres) {
            ch.base = reinterpret_cast<const elt_t*>(a.ra[c]); ch.ld = 192; ch.col = 0; ch.mode = a.rmode; ch.taps = a.rtaps;
            ch.wk = c * 192; ch.wkt = nres * 192;
        } else {
            int j = c - nres;
            ch.mode = 0; ch.taps = 1; ch.wk = kres + j * 192; ch.wkt = 0;
            if (j < c0n) { ch.base = reinterpret_cast<const elt_t*>(a.sa[0]); ch.ld = a.sld[0]; ch.col = j * 192; }
            else if (j < c0n + c1n) { ch.base = reinterpret_cast<const elt_t*>(a.sa[1]); ch.ld = a.sld[1]; ch.col = (j - c0n) * 192; }
            else { ch.base = reinterpret_cast<const elt_t*>(a.sa[2]); ch.ld = a.sld[2]; ch.col = (j - c0n - c1n) * 192; }
        }
        return ch;
    };

    // ---- GroupNorm coefficients of sample b -> coefS[slot]: 4 waves x 48 channels (scratch: the A tile, idle at every call site)
    auto gn_coefs = [&](const float* part, float eps, const float* gamma, const float* beta, int b, int cslot) {
        int lo_ = l, wo_ = w;
        asm volatile("" : "+v"(lo_), "+v"(wo_));   // (nothing of this is to stay alive in registers through the k loop)
        const GnP gp = {a.gn_cpg, a.gn_nparts, a.M, eps, gamma, beta, 192};
        const rsrc_t rp = make_rsrc(part + (long long)b * a.gn_part_bs, 192u * (unsigned)a.gn_nparts * 8u);
        GnLoads gl;
        float* const gsc = Af + wo_ * GN_SCRATCH;
        gn_issue(gp, rp, wo_ * 48, 48, lo_, gl);
        gn_finish(gp, rp, wo_ * 48, 48, lo_, gl, gsc, coefS + cslot * 384);
    };
    const bool gn_src = nres > 0 && (a.rmode == 1 || a.rmode == 3);
    auto sample_coefs = [&](int b) {   // (between two barriers: the previous tile's epilogue has read its coefficients)
        if (gn_src) {
            gn_coefs(a.gn_part[0], a.gn_eps, a.gn_gamma, a.gn_beta, b, 0);
            if (nres == 2) { __syncthreads(); gn_coefs(a.gn_part[1], a.gn_eps, a.gn_gamma + 192, a.gn_beta + 192, b, 1); }
        } else if (a.res_gn) {
            gn_coefs(a.res_part, a.res_eps, a.res_gamma, a.res_beta, b, 0);
        }
        __syncthreads();
    };

    // ---- chunk loads: 8 threads per row (24 channels = three 16-byte pieces each), 32 rows per pass; halo rows 64, 65 by 16 threads
    u32x4 raw[NP][3];
    auto issue_chunk = [&](const Chunk& ch, int b, int t0) {
        const int halo = ch.taps == 3 ? 1 : 0;
        const int q8 = tid & 7;
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            if (pass == 2 && (halo == 0 || tid >= 16)) break;
            const int r = pass * 32 + (tid >> 3);
            const int tt = t0 + r - halo;
            const elt_t* p = ch.base + ((long long)b * a.seg_rows + min(max(tt, 0), a.M - 1)) * ch.ld + ch.col + 24 * q8;
#pragma unroll
            for (int i = 0; i < 3; ++i) raw[pass][i] = *reinterpret_cast<const u32x4*>(p + 8 * i);
        }
    };
    auto park_chunk = [&](const Chunk& ch, int t0, int cslot) {
        const int halo = ch.taps == 3 ? 1 : 0;
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
        const int q8 = tid_ & 7;
        const float* cf = coefS + cslot * 384 + 48 * q8;
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            if (pass == 2 && (halo == 0 || tid_ >= 16)) break;
            const int r = pass * 32 + (tid_ >> 3);
            const int tt = t0 + r - halo;
            const bool valid = tt >= 0 && tt < a.M;
            elt_t* d = Al + r * AP + 24 * q8;
            if (ch.mode == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4*>(d + 8 * i) = valid ? raw[pass][i] : z;
                }
                continue;
            }
            float x[24];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[8 * i + 2 * e] = __builtin_bit_cast(float, raw[pass][i][e] << 16);
                    x[8 * i + 2 * e + 1] = __builtin_bit_cast(float, raw[pass][i][e] & 0xffff0000u);
                }
            if (ch.mode == 1 || ch.mode == 3) {
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    x[i] = fmaf(x[i], cf[2 * i], cf[2 * i + 1]);
                    if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (ch.mode == 1) {
#pragma unroll
                for (int i = 0; i < 24; ++i) x[i] = silu_f(x[i]);
            }
            if (ch.mode >= 2) {   // LayerNorm over the row's 192 channels: this thread's 24, then the row's eight threads
                const float ref = __shfl(x[0], (tid_ & 63) & ~7);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 24; ++i) { const float dd = x[i] - ref; s1 += dd; s2 = fmaf(dd, dd, s2); }
                s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
                s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
                s1 += __shfl_xor(s1, 4); s2 += __shfl_xor(s2, 4);
                const float md = s1 * (1.0f / 192.0f);
                const float var = fmaxf(s2 * (1.0f / 192.0f) - md * md, 0.f);
                const float mu = ref + md, rs = 1.0f / sqrtf(var + 1e-5f);
                const float* lg = a.ln_gamma + 24 * q8;
                const float* lb = a.ln_beta + 24 * q8;
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    x[i] = fmaf((x[i] - mu) * rs, lg[i], lb[i]);
                    if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!valid) {
#pragma unroll
                for (int i = 0; i < 24; ++i) x[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const bf16x8 o = {(__bf16)x[8 * i], (__bf16)x[8 * i + 1], (__bf16)x[8 * i + 2], (__bf16)x[8 * i + 3],
                                  (__bf16)x[8 * i + 4], (__bf16)x[8 * i + 5], (__bf16)x[8 * i + 6], (__bf16)x[8 * i + 7]};
                *reinterpret_cast<bf16x8*>(d + 8 * i) = o;
            }
        }
    };

    clk_stamp_p(a.clk, w, l, 0);
    // ---- the workgroup's weight slice -> LDS, once
    {
        const int kp8 = K >> 3;                     // 16-byte pieces per row
        const int npieces = BN * kp8;
        int row = tid / kp8, kp = tid - row * kp8;  // (one division per thread)
        const int drow = 256 / kp8, dkp = 256 - drow * kp8;
        for (int p0 = tid; p0 < npieces; p0 += 256 * 4) {
            u32x4 v[4]; int ro[4], ko[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ro[u] = row; ko[u] = kp;
                const bool ok = p0 + 256 * u < npieces;
                v[u] = *reinterpret_cast<const u32x4*>(W + ((long long)(n0 + (ok ? row : 0)) * K + (ok ? kp : 0) * 8));
                row += drow; kp += dkp;
                if (kp >= kp8) { kp -= kp8; ++row; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (p0 + 256 * u < npieces) *reinterpret_cast<u32x4*>(Wl + ro[u] * WP + ko[u] * 8) = v[u];
        }
    }

    // ---- accumulators, fragment pointers
    f32x16 acc[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int fr = l & 31, lh = l >> 5;
    const elt_t* const pa0 = Al + (wr * 32 + fr) * AP + 8 * lh;
    const elt_t* const pw0 = Wl + ((KSPLIT ? 0 : kh * 64) + fr) * WP + 8 * lh;
    auto mma_chunk = [&](const Chunk& ch) {
        for (int tap = 0; tap < ch.taps; ++tap) {
            const elt_t* pa = pa0 + tap * AP;
            const elt_t* pw = pw0 + ch.wk + tap * ch.wkt;
            constexpr int NS = KSPLIT ? 6 : 12;
            const int s0 = KSPLIT ? 6 * kh : 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(pa + 16 * (s0 + s));
                bf16x8 fb[NA];
#pragma unroll
                for (int j = 0; j < NA; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(pw + j * 32 * WP + 16 * (s0 + s));
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    if constexpr (TR) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa, acc[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[j], acc[j], 0, 0, 0);
                }
            }
        }
    };

    // ---- prologue: coefficients of the first sample, first chunk parked
    int b = tb / tps, t0 = (tb - b * tps) * 64;
    Chunk cur = chunk_of(0);
    issue_chunk(cur, b, t0);
    clk_stamp_p(a.clk, w, l, 1);
    if (gn_src || a.res_gn) sample_coefs(b); else __syncthreads();
    park_chunk(cur, t0, 0);
    __syncthreads();
    clk_stamp_p(a.clk, w, l, 2);

    constexpr int CW = 32 * NJ, CP = CW + 4;
    float* const xr = Af + wr * (32 * CP);        // K-half exchange, then transposition scratch, of this row half
    u32x4 rres[4];                                 // the epilogue's residual rows, requested before the tile's last chunk multiplies

    for (int ti = tb; ti < te; ++ti) {
        for (int c = 0; c < nch; ++c) {
            // -- next step's chunk: request now, park after this chunk's MFMAs
            const bool last_c = c + 1 == nch;
            const bool has_next = !last_c || ti + 1 < te;
            int nb_ = b, nt0 = t0;
            if (last_c) { nt0 = t0 + 64; if (nt0 >= a.seg_rows) { nt0 = 0; nb_ = b + 1; } }
            const Chunk nxt = chunk_of(last_c ? 0 : c + 1);
            if (has_next) issue_chunk(nxt, nb_, nt0);
            if constexpr (EK == 0) {
                if (last_c && a.res_tm) {
                    const int rr = l >> 4, cq = l & 15;
                    const int mt = t0 + wr * 32, nrows = min(32, a.M - mt);
                    const long long R0 = (long long)b * a.seg_rows + mt;
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = 16 * kh + 4 * ps + rr;
                        const bool on = cq < CW / 8 && row < nrows;
                        const u32x4 z = {0u, 0u, 0u, 0u};
                        rres[ps] = on ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const elt_t*>(a.res_tm) + (R0 + row) * a.ldr_tm + n0 + 8 * cq) : z;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk(cur);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                       // every wave is done with the A tile
            if (last_c && ti - tb < 4) clk_stamp_p(a.clk, w, l, 3 + 3 * (ti - tb));
            if (last_c) {
                const int m0 = ti * 64;
                if constexpr (KSPLIT) {
                    // ---- add the two K halves: wave (r, 1) parks its tiles, wave (r, 0) adds them
                    if (kh == 1) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) xr[(j * 16 + r) * 64 + l] = acc[j][r];
                    }
                    __syncthreads();
                    if (kh == 0) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[j][r] += xr[(j * 16 + r) * 64 + l];
                    }
                }
                if constexpr (EK == 4) {
                    if (kh == 0) {   // banded cross-attention: lane -> query token, one head per column tile
                        const int t = t0 + wr * 32 + (l & 31);
                        const bool tv = t < a.M;
                        const int tc = min(t, a.M - 1);
                        const int lo = a.band_lo[tc], hi = a.band_hi[tc];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) band_head<true>(a, acc[j], b, t, tv, lo, hi, n0 / 32 + j, l);
                    }
                } else if constexpr (EK == 2) {
                    // ---- GEGLU product of this wave's (value, gate) pair -> bf16 token-major; scratch [32][36] per wave
                    float* const sc = Af + w * (32 * 36);
                    const int mt = t0 + wr * 32, nrows = min(32, a.M - mt);
                    if (nrows > 0) {
                        const int n = n0 + kh * 64 + fr;
                        const float bv = a.bias ? a.bias[n] : 0.f, bg = a.bias ? a.bias[n + 32] : 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            sc[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + fr] = (acc[0][r] + bv) * gelu_f(acc[1][r] + bg);
                        __builtin_amdgcn_wave_barrier();
                        const int c0 = a.geglu_c0(n0 + kh * 64);
                        const long long R0 = (long long)b * a.seg_rows + mt;
                        const int rr = l >> 2, cq = l & 3;
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps) {
                            const int row = 16 * ps + rr;
                            if (row >= nrows) continue;
                            const f32x4t v0 = *reinterpret_cast<const f32x4t*>(sc + row * 36 + 8 * cq);
                            const f32x4t v1 = *reinterpret_cast<const f32x4t*>(sc + row * 36 + 8 * cq + 4);
                            const bf16x8 o = {(__bf16)v0[0], (__bf16)v0[1], (__bf16)v0[2], (__bf16)v0[3], (__bf16)v1[0], (__bf16)v1[1], (__bf16)v1[2], (__bf16)v1[3]};
                            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.yb) + (R0 + row) * a.ldy + c0 + 8 * cq) = o;
                        }
                    }
                } else if constexpr (EK == 0) {
                    // ---- token-major activation epilogue on all four waves (xgemm_kernel's)
                    float* const sc = xr;
                    const int mt = t0 + wr * 32;
                    const int nrows = min(32, a.M - mt);
                    if (kh == 0 && nrows > 0) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const int n = n0 + j * 32 + fr;
                            float add = a.bias ? a.bias[n] : 0.f;
                            if (a.emb) add += a.emb[(long long)n * a.emb_pitch + (a.step_ptr ? *a.step_ptr : 0) + b * a.emb_b_stride];
#pragma unroll
                            for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * lh) * CP + j * 32 + fr] = acc[j][r] + add;
                        }
                    }
                    __syncthreads();
                    const int rr = l >> 4, cq = l & 15;
                    const bool lane_on = cq < CW / 8;
                    const int n = n0 + 8 * min(cq, CW / 8 - 1);
                    float ref[8], s1[8], s2[8], rca[8], rcb[8], add2[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        ref[e] = sc[8 * min(cq, CW / 8 - 1) + e];
                        s1[e] = 0.f; s2[e] = 0.f;
                        rca[e] = a.res_gn ? coefS[2 * (n + e)] : 1.f;
                        rcb[e] = a.res_gn ? coefS[2 * (n + e) + 1] : 0.f;
                        add2[e] = (a.y2_tm && a.y2_add) ? a.y2_add[n + e] : 0.f;
                    }
                    const long long R0 = (long long)b * a.seg_rows + mt;
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = 16 * kh + 4 * ps + rr;
                        if (!lane_on || row >= nrows) continue;
                        const f32x4t v0 = *reinterpret_cast<const f32x4t*>(sc + row * CP + 8 * cq);
                        const f32x4t v1 = *reinterpret_cast<const f32x4t*>(sc + row * CP + 8 * cq + 4);
                        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        if (a.res_tm) {
                            const u32x4 rv = rres[ps];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[2 * e] += fmaf(__builtin_bit_cast(float, rv[e] << 16), rca[2 * e], rcb[2 * e]);
                                v[2 * e + 1] += fmaf(__builtin_bit_cast(float, rv[e] & 0xffff0000u), rca[2 * e + 1], rcb[2 * e + 1]);
                            }
                        }
                        const long long o = (R0 + row) * a.ldy + n;
                        const bf16x8 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.y_tm) + o) = ov;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (float)ov[e];   // the statistics are those of the stored values
                        if (a.y2_tm) {
                            const bf16x8 o2 = {(__bf16)(v[0] + add2[0]), (__bf16)(v[1] + add2[1]), (__bf16)(v[2] + add2[2]), (__bf16)(v[3] + add2[3]),
                                               (__bf16)(v[4] + add2[4]), (__bf16)(v[5] + add2[5]), (__bf16)(v[6] + add2[6]), (__bf16)(v[7] + add2[7])};
                            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.y2_tm) + o + a.y2_row_off * a.ldy) = o2;
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float d = v[e] - ref[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
                    }
                    if (a.stats) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            s1[e] += __shfl_xor(s1[e], 16); s2[e] += __shfl_xor(s2[e], 16);
                            s1[e] += __shfl_xor(s1[e], 32); s2[e] += __shfl_xor(s2[e], 32);
                        }
                        float* const ex = statx + wr * (2 * CW);
                        if (kh == 1 && l < CW / 8) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) { ex[2 * (8 * l + e)] = s1[e]; ex[2 * (8 * l + e) + 1] = s2[e]; }
                        }
                        __syncthreads();
                        if (kh == 0 && l < CW / 8 && nrows > 0) {
                            float* so = a.stats + (long long)b * a.stats_bs + ((long long)(mt >> 5) * a.N + n) * 2;   // [tile][channel][2]
                            const float cnt = (float)nrows, inv = 1.0f / cnt;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float S1 = s1[e] + ex[2 * (8 * l + e)], S2 = s2[e] + ex[2 * (8 * l + e) + 1];
                                const float md = S1 * inv;
                                so[2 * e] = ref[e] + md;
                                so[2 * e + 1] = fmaxf(S2 - cnt * md * md, 0.f);
                            }
                        }
                    }
                } else {
                    // ---- q/k/v split (EK 1), channel-major fp32 result (EK 3): tgemm_dev.h's epilogue on the K-half-0 waves
                    if (kh == 0) {
                        __builtin_amdgcn_wave_barrier();
                        tg_epilogue<NJ, 0, NJ, EK>(a, acc, 0, m0 + wr * 32, n0, l, xr, coefS);
                    }
                }
#pragma unroll
                for (int j = 0; j < NA; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
                __syncthreads();                   // the A tile (scratch) and the coefficients are free
                if (ti - tb < 4) clk_stamp_p(a.clk, w, l, 4 + 3 * (ti - tb));
                if (has_next && nb_ != b && (gn_src || a.res_gn)) sample_coefs(nb_);
            }
            if (has_next) {
                park_chunk(nxt, nt0, (!last_c && c + 1 < nres) ? c + 1 : 0);
                __syncthreads();
            }
            if (last_c && ti - tb < 4) clk_stamp_p(a.clk, w, l, 5 + 3 * (ti - tb));
            cur = nxt; b = nb_; t0 = nt0;
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct PgPlan { int nj, ek, t3, occ, S, per, rg8, lds; };
static bool pg_plan(const TGemmArgs& a, PgPlan& p) {
    if (a.f32 || a.seg_rows <= 0 || a.seg_rows % 64 || a.seg_rows != ((a.M + 63) & ~63) || a.M < 1) return false;
    if (a.ra[1]) return false;                                  // concatenated 3-tap input (K = 1152): its slice does not fit LDS beside the tile
    for (int i = 0; i < 3; ++i) if (a.sk[i] % 192) return false;
    if (a.ra[0] && a.rtaps != 1 && a.rtaps != 3) return false;
    const int kres = a.ra[0] ? a.rtaps * 192 : 0;
    if (kres + a.sk[0] + a.sk[1] + a.sk[2] != a.K) return false;
    if ((a.rmode == 1 || a.rmode == 3) && a.res_gn) return false;
    p.t3 = (a.ra[0] && a.rtaps == 3) ? 1 : 0;
    const int arows = p.t3 ? 66 : 64;
    if (a.band_k) { if (a.N % 96 || !a.ra[0] || a.sk[0] || !a.y_tm || a.band_wmax < 1 || a.band_wmax > 8) return false; p.nj = 3; p.ek = 4; }
    else if (a.geglu) { if (a.N % 256 || !a.yb || !a.ra[0] || a.sk[0]) return false; p.nj = 4; p.ek = 2; }
    else if (a.qk) { if (a.N % 96) return false; p.nj = 3; p.ek = 1; }
    else if (a.y_cm) { if (a.cm_pitch % 4 || a.cm_pitch < ((a.M + 3) & ~3)) return false; p.nj = a.K > 576 ? 2 : 3; p.ek = 3; }
    else if (a.y_tm) { p.nj = a.K > 576 ? 2 : 3; p.ek = 0; }
    else return false;
    if (a.N % (32 * p.nj)) return false;
    if ((long long)a.batch * a.seg_rows > 0x7fffffffLL / 768) return false;
    p.lds = p.nj == 2 ? pg_lds_bytes<2>(a.K, arows) : p.nj == 3 ? pg_lds_bytes<3>(a.K, arows) : pg_lds_bytes<4>(a.K, arows);
    if (p.lds > 160 * 1024) return false;
    p.occ = (p.lds <= 80 * 1024 && p.ek != 4) ? 2 : 1;
    p.S = a.N / (32 * p.nj);
    const int MT = a.batch * (a.seg_rows / 64);
    const int slots = 32 * p.occ;                              // workgroup slots per XCD
    if (p.S > slots) return false;
    int rg8 = slots / p.S;
    int per = (MT + 8 * rg8 - 1) / (8 * rg8);
    const int groups = (MT + per - 1) / per;
    rg8 = (groups + 7) / 8;
    p.per = per; p.rg8 = rg8;
    return true;
}
bool pgemm_supports(const TGemmArgs& a_in, int batch) {
    TGemmArgs a = a_in; a.batch = batch;
    PgPlan p;
    return pg_plan(a, p);
}
template <int NJ, int EK, bool T3>
static void launch_pg_one(const TGemmArgs& a, const PgPlan& p, hipStream_t s) {
    hipLaunchKernelGGL((pgemm_kernel<NJ, EK, T3>), dim3((unsigned)(8 * p.S * p.rg8)), dim3(256), p.lds, s, a);
}
template <int NJ, int EK, bool T3>
static void config_pg_one() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pgemm_kernel<NJ, EK, T3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
void configure_pgemm_kernels() {
    config_pg_one<3, 0, true>(); config_pg_one<3, 0, false>(); config_pg_one<2, 0, true>(); config_pg_one<2, 0, false>();
    config_pg_one<3, 1, false>(); config_pg_one<4, 2, false>(); config_pg_one<2, 3, false>(); config_pg_one<3, 3, false>(); config_pg_one<3, 4, false>();
}
bool launch_pgemm(const TGemmArgs& a_in, int batch, hipStream_t s) {
    TGemmArgs a = a_in;
    a.batch = batch;
    PgPlan p;
    if (!pg_plan(a, p)) return false;
    a.pg_s = p.S; a.pg_per = p.per;
    const bool t3 = p.t3 != 0;
    if (p.ek == 0 && p.nj == 3) { if (t3) launch_pg_one<3, 0, true>(a, p, s); else launch_pg_one<3, 0, false>(a, p, s); }
    else if (p.ek == 0 && p.nj == 2) { if (t3) launch_pg_one<2, 0, true>(a, p, s); else launch_pg_one<2, 0, false>(a, p, s); }
    else if (p.ek == 1 && !t3) launch_pg_one<3, 1, false>(a, p, s);
    else if (p.ek == 2 && !t3) launch_pg_one<4, 2, false>(a, p, s);
    else if (p.ek == 3 && !t3 && p.nj == 2) launch_pg_one<2, 3, false>(a, p, s);
    else if (p.ek == 3 && !t3 && p.nj == 3) launch_pg_one<3, 3, false>(a, p, s);
    else if (p.ek == 4 && !t3) launch_pg_one<3, 4, false>(a, p, s);
    else return false;
    return true;
}

}  // namespace said

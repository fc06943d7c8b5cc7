// tgemm.h — launch interface of the token-major GEMMs (bf16: tgemm_kernel / tgemm256_kernel, fp32: fgemm_kernel) and their
// companion kernels (tgemm.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace said {

struct TGemmArgs {
    const void* a;         // bf16 A: row m of batch b starts at a + b * a_bs + m * lda (elements), K contiguous.  A strided
    long long a_bs;        // Conv1d over token-major data is lda = stride * C, K = taps * C (overlapping rows).
    int lda;
    const void* w;         // bf16 W [N][K], K contiguous
    const float* bias;     // [N] or null
    const float* res;      // fp32 residual [b][m][ldr] or null, added after the activation
    long long res_bs;
    int ldr;
    float* yf;             // fp32 output [b][m][ldy] (nullable)
    void* yb;              // bf16 output, same indexing (nullable)
    long long y_bs;
    int ldy;
    // q/k/v split for attn.hip's operand layout (used when qk != null): outputs n < qk_n go to qk token-major per head
    // [b][heads2][rows][head_dim] (fp32), the rest to vt channel-major [b][N - qk_n][v_pitch] (fp32)
    float* qk;
    float* vt;
    long long v_bs;
    int qk_n, head_dim, rows, heads2, v_pitch;
    int kv_pack;           // fp32 q/k/v outputs: k and v elements are stored as packed split-fp16 pairs (split_f16.h pack_split_f16) for attn_kernel<PM = 3> (fgemm_kernel's epilogue)
    float q_scale;         // qkv_bf16: q is stored multiplied by this (battn_kernel wants scale * log2(e) folded in before the rounding)
    int qkv_bf16;          // rgemm_kernel: q / k / v are written as bf16 in battn_kernel's operand layout (attn.hip: launch_battn) — same indexing, v's
                           // tokens permuted inside every block of 16 ([0-3, 8-11, 4-7, 12-15]: a lane half's eight keys are one 16-byte piece)
    int M, N, K;
    int batch;             // filled in by launch_tgemm
    int act;               // 0 none, 1 GELU (erf)
    // ---- second K segment (optional): for k >= K1 the A operand comes from a2 (row m at a2 + b * a2_bs + m * lda2 + (k - K1)):
    // ResBlock conv + 1x1 skip over the concatenated input, proj_out o ff.net.2 over [h ; x2].  K1 % 64 == 0.
    const void* a2;
    long long a2_bs;
    int lda2, K1;
    // ---- channel-major fp32 epilogue (UNet activations; used when y_cm != null): y_cm[b][n][cm_pitch] = acc + bias + emb + res,
    // GroupNorm partial statistics of the result per (channel, 32-token tile), optional second copy (kernels.h GemmCommon::y2)
    float* y_cm;
    long long cm_bs;
    int cm_pitch;
    const float* emb;      // [n][emb_pitch] timestep-embedding term, row = *step_ptr + b * emb_b_stride (or null)
    const int* step_ptr;
    int emb_pitch, emb_b_stride;
    const float* res_cm;   // channel-major fp32 residual [b][n][cm_pitch] (or null)
    long long res_cm_bs;
    const float* res_cm_coef;   // != null: the residual is GroupNorm(res_cm) = res_cm * a + b with (a, b) = res_cm_coef[b][n][0..1] (prep_kernel's coef_out)
    long long res_cm_coef_bs;
    float* stats;          // [b][N][ceil(M/32)][2] (mean, M2) or null
    long long stats_bs;
    float* y2_cm;          // second copy of the result (or null)
    long long y2_bs;
    const float* y2_add_cm;     // != null: ... plus this per-channel constant (guidance: x2 of the unconditional half = x1 + c2, kernels.h y2_add)
    // ---- GEGLU epilogue (geglu != 0): the weight rows are tile-interleaved on the host so that a wave's two 32-column MFMA
    // tiles are (value, gate) of the same 32 channels; out[m][c] = value * gelu(gate) -> yb (bf16 token-major, N / 2 channels)
    int geglu;
    // ---- batch as rows (UNet): the A operand of ALL samples is one row axis, sample b's rows starting at b * seg_rows
    // (seg_rows % 32 == 0, >= M); row R is sample R / seg_rows, token R % seg_rows; tokens >= M are padding.  0: per-sample
    // operands addressed through a_bs (audio encoder).
    int seg_rows;
    int sb;                // tgemm_kernel<128> only: 1 = single-LDS-buffer variant (four workgroups per CU)
    int direct;            // 1: per-sample operands with N % 256 == 0 run tgemm256d_kernel (256 x 256 tile, operand tiles loaded straight into LDS; round 6)
    int grp;               // tgemm_kernel only: > 1 = grouped launch, the batch axis is (sample, group) [batch = samples x grp]; group g reads A at
    long long a_gs, w_gs;  //   a + sample a_bs + g a_gs and W at w + g w_gs (elements) and owns the output columns [g col_gs, g col_gs + n_store)
    int col_gs;            //   of bias / res / y (the wav2vec2 positional convolution: 16 groups of 48 channels)
    int n_store;           // token-major outputs: columns n >= n_store are not written (0: all N) — a column count padded to the tile
    int f32;               // 1: A and W are fp32 (fgemm_kernel on v_mfma_f32_32x32x2_f32; fp32 mode, large batches); K % 32 == 0
    int f32_split;         // f32 operands: run the products on split-fp16 operands (fgemm_kernel SP; split_f16.h) — fp32-equivalent results
    int f32_packed;        // ... and A, A2 and W ARRIVE split: every element one dword h | l << 16 (split_f16.h pack_split_f16: prep_kernel PrepArgs::pack, engine.cpp
                           // upload_tm_pair): the k loop unpacks with v_perm instead of splitting (round 6: the loop was VALU-bound on the splits); bit-identical results
    int dbg;               // timing experiments (SAID_TG_DBG): bit 0 = skip the epilogue, bit 1 = skip the K loop
    // ==== token-major ACTIVATION interface (round 3; xgemm_kernel only; large batches, both precisions) ====================
    // Between the UNet kernels the activations are token-major [sample][token][192] in the context's element type ET (bf16 in
    // bf16 mode, fp32 else), sample pitch seg_rows tokens (a multiple of 64, so a 64-row tile never straddles samples), plus
    // the producer's fp32 GroupNorm partials [sample][32-token tile][192][2].  Nothing prepares operands any more: the consuming
    // GEMM applies GroupNorm + SiLU / LayerNorm itself.
    // -- resident first K segment (ra[0] != null): the workgroup's source tile — 64 tokens (+ 2 halo tokens for taps == 3) x 192
    // channels of ra[0], then of ra[1] (concatenated input) — is loaded ONCE, transformed once per element and parked in LDS;
    // only the weights stream.  k order of W for this segment: [tap][source][channel] (Conv1d weight, tap-major).
    int ntw;               // column tiles per workgroup (0: chosen by launch_xgemm)
    int pg_s, pg_per;      // rgemm_kernel (filled in by the launch helper): column slices (groups) of the launch, row tiles per range
    int w_ld, w_k0, w_seg; // rgemm_kernel: the launch multiplies a COLUMN RANGE of the weight rows — row pitch (0: K), first column, distance between
                           // the 192-wide segments (taps; 0: 192) — so a long-K GEMM can run as several launches over one packed weight
    long long* clk;        // optional [4 waves][16] shader-clock stamps of workgroup 8 (-DSAID_CLK_STAMPS builds; scripts/xgemm_clocks.py)
    const void* ra[2];     // sources (row pitch 192), or null
    int rmode;             // 0: raw, 1: silu(GroupNorm(x)), 2: LayerNorm(x), 3: LayerNorm(GroupNorm(x))
    int rtaps;             // 1, or 3: output token t reads tokens t - 1 + tap, zero outside [0, M)
    const float* gn_part[2];       // GroupNorm partials of ra[0] / ra[1]
    long long gn_part_bs;
    int gn_cpg, gn_nparts; float gn_eps;
    const float* gn_gamma; const float* gn_beta;     // [192 per source]
    const float* ln_gamma; const float* ln_beta;     // [192]
    // -- streamed raw K segments AFTER the resident one (or the whole K when ra[0] == null): up to three sources, source i holding
    // sk[i] channels with row pitch sld[i]; row R (= sample * seg_rows + token) of source i starts at sa[i] + R * sld[i]
    const void* sa[3]; int sld[3]; int sk[3];
    // -- token-major activation epilogue (y_tm != null): y_tm[R][n] (ET, row pitch ldy) = acc + bias + emb + residual; `stats` then
    // receives the GroupNorm partials of the STORED (rounded) values
    void* y_tm;
    const void* res_tm; int ldr_tm;      // residual rows R (ET); with res_gn: GroupNorm'ed first (attention.py:227: x = norm(x))
    int res_gn; const float* res_part; const float* res_gamma; const float* res_beta; float res_eps;
    void* y2_tm; long long y2_row_off; const float* y2_add;   // second copy at rows R + y2_row_off (+ per-channel constant)
    // -- banded cross-attention epilogue (band_k != null; needs the transposed product: xgemm_kernel<.., TR = true>): the N = 192
    // outputs are the queries of 6 heads; softmax over keys [lo[t], hi[t]) of the precomputed K / V, result -> y_tm
    const float* band_k; const float* band_v;   // [b][192][kv_pitch] channel-major fp32
    const int* band_lo; const int* band_hi;
    long long band_kv_bs; int band_kv_pitch, band_wmax; float band_scale;
    // value channel of the first column of a GEGLU value tile starting at permuted column n (see tgemm_geglu_src_row)
    __host__ __device__ int geglu_c0(int n) const { return (n / 256) * 128 + ((n % 256) / 128) * 64 + ((n % 128) / 64) * 32; }
};
int tgemm_geglu_src_row(int n, int N);
bool tgemm_supports(const TGemmArgs& a);
// false: shape not served by any instantiation (nothing launched) — the caller reports it through the C ABI
bool launch_tgemm(const TGemmArgs& a, int batch, hipStream_t s);   // N % 128 == 0: 128-wide tiles, else N % 64 == 0: 64-wide
void configure_tgemm_kernel();
// GEMMs on token-major activations with the operand transform inside (TGemmArgs fields of round 3)
bool xgemm_supports(const TGemmArgs& a);
bool launch_xgemm(const TGemmArgs& a, int batch, hipStream_t s);
void configure_xgemm_kernels();
// round 4 (rgemm.hip): register-stationary weights, wave-specialised persistent workgroups: the 192-wide GEMMs with K <= 576 and q/k/v (bf16 mode)
bool rgemm_supports(const TGemmArgs& a, int batch);
bool launch_rgemm(const TGemmArgs& a, int batch, hipStream_t s);
void configure_rgemm_kernels();
// UNet operand preparation (bf16 mode, large batches): channel-major fp32 x[b][C][pitch] -> transform -> token-major bf16.
// mode 0: silu(GroupNorm(x)) into dst[b][1 + t][ldd] at column `coff` (rows 0 and T + 1 zero: Conv1d padding), mode 1:
// LayerNorm(GroupNorm(x)) -> dst[b][t][ldd], mode 2: LayerNorm(x) -> dst and raw x -> dst2 (both [b][t][*]), mode 3: raw x.
struct PrepArgs {
    const float* x; long long x_bs; int pitch, T, C;
    const float* coef; long long coef_bs;        // GroupNorm (a, b) per (sample, channel) from launch_gn_coef (modes 0, 1) — or,
    const float* part; long long part_bs;        // when part != null, the producer's Welford partials [b][tile][192][2]: every workgroup
    int gn_cpg, gn_nparts; float gn_eps;         // finalises the coefficients itself (same code as gn_coef_kernel; its loads fly with
    const float* gn_gamma; const float* gn_beta; // the tile's), which saves the 5 us dependent launch in front of each preparation
    const float* ln_gamma; const float* ln_beta;
    void* dst; long long dst_bs; int ldd, coff;
    void* dst2; long long dst2_bs; int ldd2, coff2;
    float* coef_out; long long coef_out_bs;      // != null: the first tile's workgroup of every sample also stores the finalised (a, b) [b][192][2]
    int mode;
    int f32;               // 1: dst / dst2 are fp32 (operands of the fp32 token-major GEMM), else bf16
    int pack;              // f32 only: dst / dst2 elements are written as split-fp16 pairs (h | l << 16: split_f16.h pack_split_f16) for fgemm_kernel's packed mode
};
bool launch_prep(const PrepArgs& a, int batch, hipStream_t s);
// GroupNorm coefficients of a 192-channel tensor from its Welford partials [b][192][nparts][2] -> coef_out[b][192][2]
void launch_gn_coef(const float* part, long long part_bs, int cpg, int nparts, int T, float eps, const float* gamma, const float* beta,
                    float* coef_out, long long coef_bs, int batch, hipStream_t s);
// token-major fp32 [b][T][G * CG] -> per-group bf16 [b][G][R][CG] with `lpad` zero rows in front and zeros behind (R >= lpad + T): the
// operand of a grouped Conv1d run as G GEMMs with overlapping rows (Wav2Vec2's positional convolution)
void launch_tm_to_group_bf16(const float* src, long long src_bs, void* dst, int B, int T, int G, int CG, int R, int lpad, hipStream_t s);
void launch_cm_to_tm_bf16(const float* src, long long src_bs, int pitch, void* dst, long long dst_bs, int B, int T, int C, hipStream_t s);
void launch_ln_tm(const float* x, const float* add, float* yf, void* yb, const float* gamma, const float* beta, long long ntok, int C, float eps,
                  hipStream_t s);
void launch_interp_ln_tm(const void* src, long long src_bs, int Tin, void* dst, long long dst_bs, int Tout, int B, int C, const float* gamma,
                         const float* beta, float eps, hipStream_t s);

}  // namespace said

// kernels.h — launch-level interface between engine.cpp and the gfx950 kernels.
//
// Activation layout ("channel-major"): X[b][c][t], t contiguous, row pitch `pitch`
// floats (multiple of 32, >= T).  This is the layout of the reference's Conv1d
// tensors (said/model/unet_1d_condition.py:73-75) and makes every MFMA operand
// fetch a coalesced 128-byte row segment: for v_mfma_f32_32x32x2_f32 the B operand
// of lane l is X[c0 + (l>>5)][t0 + (l&31)].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace said {

// Development knobs (A/B switches, experiment sizes) are environment variables that ONLY a build compiled with
// -DSAID_DEV_KNOBS reads (SAID_EXTRA_DEFS=-DSAID_DEV_KNOBS python -m said_amd.build --force; scripts/README.md).  The
// shipped library ignores the environment: every dev_env() below is a null pointer there.
inline const char* dev_env(const char* name) {
#ifdef SAID_DEV_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// operand transforms applied while loading X (fused producer-side elementwise work)
enum XForm : int {
    XF_NONE = 0,
    XF_GN_SILU = 1,  // silu(groupnorm(x))           ResBlock in_layers/out_layers, UNet `out`
    XF_LN = 2,       // layernorm over channels      norm2 / norm3
    XF_GN_LN = 3,    // layernorm(groupnorm(x))      SpatialTransformer.norm -> norm1
    XF_SILU = 4,     // silu(x)                      emb_layers
};

// ACT_LRELU_02 / _001: LeakyReLU(0.2) / LeakyReLU() (slope 0.01) of the VAE encoder (said/model/vae.py:41-64)
enum Act : int { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_LRELU_02 = 3, ACT_LRELU_001 = 4 };
enum Res : int { RES_NONE = 0, RES_PLAIN = 1, RES_GN = 2 };
enum Epi : int { EPI_STORE = 0, EPI_QKV = 1, EPI_GEGLU = 2, EPI_BAND = 3 };

// One K-segment of a GEMM: out[n][t] += sum_{tap,c} W[n][c][tap] * xform(X[c][t*stride + tap - pad])
struct SegFields {
    const float* x;        // source, channel 0 of this segment, batch 0
    const float* w;        // packed weights [groups][ntiles][taps][C/2][64]
    const float* w4;       // same weights packed for dwordx4 fetch [ntiles][taps][C/8][64 lanes][4 k-pairs] (or null)
    const float* w2;       // bf16 packing for v_mfma_f32_32x32x8_bf16_1k: [ntiles][taps][C/8][64 lanes][4 bf16], value j of
                           // lane l = W[tile*32 + (l&31)][8*cq + 4*(l>>5) + j][tap]; same tails as w4 (or null)
    const float* ws;       // round 5: split-fp16 packing for v_mfma_f32_32x32x16_f16 (gemm_lds.hip SP; w = h + 2^-11 l, fp16 planes h, l):
                           // [ntiles][C/24 blocks][NS steps][2 planes][64 lanes][8 halfs], NS = 5 (3 taps) / 2 (1 tap); the 8 halfs of lane l in step s are
                           // K-group g = 2 s + (l >> 5) of the block's tap-major K slice (tap = g / 3, channels 8 (g % 3) .. + 7; groups past the slice: zeros),
                           // row tile*32 + (l & 31); ws_flat: [ntiles][C/16 steps][2][64][8] without per-block padding (GEGLU); same tails as w4 (or null)
    const float* gn_part;  // GN partial stats of the source [B][C][nparts][2] (mean, M2), channel 0 of segment
    const float* gn_gamma; // GN affine (segment channel 0)
    const float* gn_beta;
    const float* ln_gamma; // LN affine
    const float* ln_beta;
    long long x_bstride;   // floats between batches of the source
    long long gn_part_bstride;
    int x_pitch;           // floats between channels of the source
    int C;                 // channels in this segment (per group)
    int taps, pad, stride;
    int Tin;               // valid source length
    int xform;
    int gn_cpg;            // channels per GN group
    int gn_nparts;         // partials per channel
    float gn_eps, ln_eps;
    int b_mod;             // source batch = b % b_mod when > 0 (CFG: both halves read the same latents)
    int c_group_stride;    // source channel offset per conv group (grouped conv), else 0
    int w4_gn_tail;        // 1: w4 is followed by gn_gamma[C], gn_beta[C] of this segment (copies; see FastHdr)
    int w4_ln_tail;        // 1: ... and then by ln_gamma[C], ln_beta[C]
    int ws_flat;           // 1: ws uses the flat step layout (one output tile per wave over the whole K: the GEGLU shape)
};
// padded to 256 bytes: the LDS-staged kernel fetches the argument block with one coalesced 256-B vector
// load per block and extracts fields with v_readlane (a by-value struct read field-by-field costs one
// serialized scalar-memory round trip per field group — measured ~8k clocks per kernel)
struct Seg : SegFields { char pad_[256 - sizeof(SegFields)]; };
static_assert(sizeof(Seg) == 256, "Seg must be one 256-byte block");

struct BandArgs {          // banded cross-attention fused behind the q projection
    const float* k;        // [Be][C][Sp] channel-major keys (precomputed per clip)
    const float* v;
    const int* lo;         // [T] first visible key of query t (ldm/attention.py:184-189)
    const int* hi;         // [T] one past the last visible key
    long long kv_bstride;
    int kv_pitch;
    int wmax;              // max(hi - lo)
    float scale;           // dim_head ** -0.5, applied after QK^T (ldm/attention.py:101)
};

struct GemmCommon {        // 8-byte members first, then 4-byte ones: no internal padding, <= 256 bytes
    const float* bias;     // [groups*N] or null
    // + emb[row][n]: per-(row, channel) additive term (ResBlock emb_layers output)
    const float* emb;      // [(n)][emb_pitch] channel-major table, or null
    const int* step_ptr;   // device step counter: row = *step_ptr (loop) + b * emb_b_stride (forward(): row = b)
    const float* res;      // residual [B][N][res_pitch]
    long long res_bstride;
    const float* res_gn_part;  // RES_GN: partial stats of `res`, gamma/beta/eps
    const float* res_gn_gamma;
    const float* res_gn_beta;
    float* y;              // output, channel-major [B][groups*N][y_pitch]
    long long y_bstride;
    float* stats_out;      // GN partials of y: [B][groups*N][ceil(T/32)][2], or null
    union {                // epilogue-specific arguments (one epilogue per launch)
        BandArgs band;     // EPI_BAND
        struct {           // EPI_QKV: tiles < tm_tiles (q and k) are written token-major: vt[b][h][t][d], h over vt_heads
            float* vt;     // = 2 * heads (q heads, then k heads); the remaining tiles (v) go to y channel-major with
            int vt_heads, vt_dim, vt_rows;   // channel index n - 32 * tm_tiles (rows = padded T of the vt buffer).  This is
            int tm_tiles;  // the operand layout attn.hip fetches with dwordx4.
            int kv_split;  // 1: k and v elements are stored as packed split-fp16 pairs (split_f16.h pack_split_f16) for attn_kernel<PM = 3>
        };
        struct {           // EPI_STORE: optional second copy of the result, y2[b][n][t] = y[b][n][t] + y2_add[n]
            float* y2;     // (same pitch as y).  Under classifier-free guidance the unconditional half's cross-attention
            const float* y2_add;   // output is a per-channel constant, so attn1's to_out also emits x2 = x1 + const for it;
            long long y2_bstride;  // and guidance-shared tensors are written once per clip into both halves' slots.
        };
    };
    float* gn_coef_out;    // EPI_QKV on ugemm_kernel, or null: the first workgroup of every sample also stores segment 0's finalised GroupNorm (a, b) [sample][C][2]
                           // (gn_coef_bs floats between samples) — stchain_kernel's GroupNorm'ed residual then needs no partials of its own (round 6)
    long long* clk;        // optional [KS][16] shader-clock stamps of workgroup (1,0,0) (debug)
    int* step_inc;         // if set, workgroup (0,0,0) increments *step_inc before anything else (loop step counter)
    int nseg;
    int T;                 // output length
    int N;                 // output channels per group
    int groups;            // conv groups (1 unless grouped conv)
    int ntiles_per_group;  // ceil(N / 32)
    int act;               // Act, applied after bias
    int emb_b_stride;
    int emb_pitch;
    int res_kind;
    int res_pitch;
    int res_gn_part_bstride;
    int res_gn_cpg, res_gn_nparts;
    float res_gn_eps;
    int y_pitch;
    int stats_bstride;
    int geglu_gate_tiles;  // EPI_GEGLU: gate tile = value tile + geglu_gate_tiles
    int b0;                // batch offset: this launch covers samples [b0, b0 + gridDim.z)
    int gn_coef_bs;
    int kconv_off;         // host only: 1 = the K-long ResBlock convolutions keep ugemm_body's block loop (said_debug_option "kconv" = 0; gemm_lds.hip kconv_body)
};
struct GemmArgs : GemmCommon {
    char pad_[256 - sizeof(GemmCommon)];
    Seg seg[3];
};
static_assert(sizeof(GemmCommon) <= 256, "common block must fit 256 bytes");
static_assert(sizeof(GemmArgs) == 1024, "GemmArgs = 4 blocks of 256 bytes");

// A launch helper that cannot serve its arguments (no instantiation for the shape, LDS budget exceeded, 32-bit stride overflow) records
// the reason here and launches nothing; the C-ABI entry point that drove it returns the message as its error (engine.cpp: LAUNCHCHK).
// Thread-local: two contexts may be driven from two host threads.  (Round 2 called abort() at these sites.)
void launch_fault(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* launch_fault_peek();    // nullptr when nothing is recorded
void launch_fault_clear();

struct AttnArgs {
    const float* qk;       // token-major [B][2*H][rows][D]: q heads, then k heads
    const float* v;        // channel-major [B][H*D][pitch]
    float* o;              // channel-major [B][H*D][pitch]
    long long v_bstride;   // floats between batches of v
    long long o_bstride;
    int pitch;
    int T;                 // queries == keys
    int heads;
    int rows;              // padded T of the token-major buffer
    float scale;
    int b0;                // batch offset
    int o_mode = 0;        // 0: o channel-major fp32; 1 / 2: token-major fp32 / bf16 [b * o_bstride + i][h * D + d], o_bstride = sample
                           // pitch in tokens (key-split-free shapes only: KS == 1 or -4)
};

// tile shape selection: NB 32-row tiles per workgroup, KS waves splitting K
void launch_gemm(const GemmArgs& a, int epi, int batch, int NB, int KS, hipStream_t s);
void launch_attn(const AttnArgs& a, int batch, int head_dim, int KS, hipStream_t s, int mode = 0);   // mode: 0 fp32 MFMA, 1 bf16 operands, 2 split-fp16 operands
// LDS-staged UNet GEMM (gemm_lds.hip): same arguments; only for shapes ugemm_supports() accepts
// pm (product mode): 0 = v_mfma_f32_32x32x2_f32 on fp32 operands; 1 = bf16 (v_mfma_f32_32x32x8_bf16_1k; needs Seg::w2), everything else stays fp32;
// 2 = split-fp16 operands (round 5: x = h + 2^-11 l, three v_mfma_f32_32x32x16_f16 per eight fp32 MFMAs, fp32 accumulation; needs Seg::ws; tt == 1 only)
// tt > 1: multi-tile workgroups (each walks over tt consecutive 32-token tiles keeping its weights in registers)
bool ugemm_supports(const GemmArgs& a, int epi, int NB, int KS, int pm = 0, int tt = 1);
void launch_ugemm(const GemmArgs& a, int epi, int batch, int NB, int KS, hipStream_t s, int pm = 0, int tt = 1);
void configure_ugemm_kernels();

// token-major (B,T,C) <-> channel-major [B][C][pitch]
void launch_tm_to_cm(const float* src, float* dst, int B, int T, int C, int pitch, long long dst_bstride, hipStream_t s);
void launch_cm_to_tm(const float* src, float* dst, int B, int T, int C, int pitch, long long src_bstride, hipStream_t s);
// broadcast one vector to every column of a channel-major tensor (null_cond_emb.repeat)
void launch_fill_cm_vec(const float* vec, float* dst, int B, int T, int C, int pitch, long long dst_bstride, hipStream_t s);
// sinusoidal timestep embedding, channel-major [dim][pitch], column r <- timesteps[r] (ldm/util.py:66-90)
// freqs_dev: the context's own [dim/2] frequency table (no process-global state: several contexts may coexist)
void launch_timestep_embedding(const long long* timesteps_dev, const float* freqs_dev, float* dst, int n, int dim, int pitch, hipStream_t s);
void launch_step_advance(int* step_ptr, hipStream_t s);
void launch_spin(long long ticks_100mhz, hipStream_t s);   // one wave busy for that long (stream-concurrency probe)
void configure_gemm_kernels();   // raise the dynamic-LDS limit of every instantiation (call once, outside capture)
void configure_attn_kernels();
// round 4: bf16 operands (written by rgemm.hip's q/k/v epilogue), a head's K / V resident in LDS; T <= 640, head_dim 32
bool battn_supports(const AttnArgs& a, int head_dim);
// banded cross-attention over alignment windows of any width, q / o in place (misc.hip: band_wide_kernel)
void launch_band_wide(float* qo, long long qo_bstride, int pitch, const float* k, const float* v, long long kv_bstride, int kv_pitch, const int* lo,
                      const int* hi, int T, int heads, int batch, float scale, hipStream_t s);
// round 6: pre-split K / V, head_dim 32, four key slices, three query tiles per wave (attn2q.hip: long sequences at small batch); channel-major output
void launch_attn2q(const AttnArgs& a, int batch, hipStream_t s);
void configure_attn2q_kernel();
void launch_battn(const AttnArgs& a, int batch, hipStream_t s, int qt = 8);   // qt: query tiles (waves) per workgroup, 4 or 8

struct SchedArgs {
    const float* eps;          // channel-major [Be][C][pitch] model output
    long long eps_bstride;
    int pitch;
    int B, T, C;
    int cfg;                   // 1: eps = e_c + s*(e_c - e_u), uncond half first (diffusion.py:430-434)
    float guidance_scale;
    float guidance_rescale;    // > 0: rescale_noise_cfg using rescale_stats
    const float* rescale_part; // [B][2][nblk][3] partial (n, mean, M2) of e_c and eps_cfg
    int rescale_nblk;
    int prediction_type;
    const float* coef;         // device [nsteps][8]
    const int* step_ptr;       // device step counter
    float* x;                  // latents, channel-major [B][C][pitch], updated in place
    long long x_bstride;
    const float* step_noise;   // channel-major [nsteps][B][C][pitch] or null
    const float* init;         // channel-major or null
    const float* edit_noise;
    const float* mask;
    float* inter;              // token-major (nsteps, B, T, C) or null: pre-step latents / latent_scale
    float latent_scale;
    const unsigned* noise_seed;   // device [2] Philox key: the eta noise is generated in the kernel (sched_math.h), or null
    unsigned noise_elem0;         // element index of this call's first clip in the whole batch (clip groups on concurrent streams draw the noise of ONE batch)
    int* status;                  // device [2] sticky numeric status (said_numeric_status): [0] <- 1 + the first step whose model output was not finite, or null
};
void launch_sched_step(const SchedArgs& a, hipStream_t s);

// out conv (GN -> SiLU -> Conv1d(192 -> Cout, k3)) + guidance + DDIM update in one kernel (out_sched.hip)
struct OutSchedArgs {
    const float* x;            // last hidden state, channel-major [Be][192][pitch]
    const float* gn_part;      // its GroupNorm partials [Be][192][gn_nparts][2]
    const float* gn_gamma;     // out.0
    const float* gn_beta;
    const float* w4;           // out.2 weights, dwordx4 packing [1][3][24][64][4] (+ out.0's gamma[192], beta[192] behind)
    const float* ws;           // the same in the split-fp16 packing (Seg::ws layout, one tile; same tail), or null: fp32 matrix instructions
    const float* bias;
    const float* coef;         // device [nsteps][8] scheduler coefficients
    const int* step_ptr;       // device step counter
    float* lat;                // latents, channel-major [B][Cout][pitch], updated in place
    const float* step_noise;   // channel-major [nsteps][B][Cout][pitch] or null
    const float* init;         // channel-major or null (editing)
    const float* edit_noise;
    const float* mask;
    float* inter;              // token-major (nsteps, B, T, Cout) or null: pre-step latents / latent_scale
    long long x_bstride, gn_part_bstride, lat_bstride;
    int pitch, T, B, Cin, Cout, gn_nparts;
    int cfg;                   // 1: samples [0, B) unconditional, [B, 2B) conditional
    int prediction_type;
    float guidance_scale, guidance_rescale, latent_scale;
    const unsigned* noise_seed;   // device [2] Philox key: eta noise generated in the kernel (sched_math.h), or null
    unsigned noise_elem0;         // see SchedArgs
    // out_sched_tm_kernel (bf16 large-batch schedule): the last hidden state token-major bf16 [Be][seg][192], out.2 weights bf16 [32][3 taps][192]
    const void* x_tm;
    const void* wb;
    int seg;
    int* status;               // see SchedArgs
};
bool out_sched_supports(const OutSchedArgs& a);
bool out_sched_tm_supports(const OutSchedArgs& a);
void launch_out_sched_tm(const OutSchedArgs& a, hipStream_t s);
void launch_out_sched(const OutSchedArgs& a, hipStream_t s);
void configure_out_sched_kernel();
// pre-pass for guidance_rescale > 0: per-block Welford partials of e_c and eps_cfg
void launch_rescale_partials(const SchedArgs& a, float* part_out, hipStream_t s);
// standalone elementwise scheduler step on token-major buffers (said_ddim_step)
void launch_ddim_flat(const float* eps, const float* eps_u, float gs, const float* x, const float* coef_dev, int pred,
                      const float* noise, const float* init, const float* edit_noise, const float* mask, float* out,
                      long long n, hipStream_t s);
// out[b][i] = a[b]*x[b][i] + c[b]*y[b][i] (explicitly rounded mul, mul, add); coefficients live in device memory
void launch_axpby(const float* a_dev, const float* x, const float* c_dev, const float* y, float* out, int B, long long n,
                  hipStream_t s);
// out (nsteps, B, T, C) token-major <- the standard normals the loop's kernels generate for (seed, step0 + k, element)
void launch_philox_normal(const unsigned* seed_dev, int step0, int nsteps, long long n_per_step, float* out, hipStream_t s);
// result = clamp(x / latent_scale, 0, 1), channel-major -> token-major; also copies latents out
// status (optional): [1] <- 1 when a final latent is not finite (said_numeric_status)
void launch_finish(const float* x_cm, long long x_bstride, int pitch, int B, int T, int C, float latent_scale,
                   float* latents_tm, float* result_tm, hipStream_t s, int* status = nullptr);
// status[1] <- 1 when any of x[b][c][t < T] (channel-major) is not finite: said_unet_forward's model output
void launch_nonfinite_check(const float* x_cm, long long x_bstride, int pitch, int B, int T, int C, int* status, hipStream_t s);

// UNet input conv (32 -> Cout, k3) + GroupNorm partials + step-counter increment (conv_in.hip): one result per clip,
// written to `copies` batch halves (clip b -> samples b, b + B, ...)
bool conv_in_supports(int Cin, int Cout, int taps, int T, int pitch, int copies);
void launch_conv_in(const float* x, const float* w4, const float* bias, float* y, float* stats, int* step_inc, int B, int copies, int T,
                    int pitch, int Cout, hipStream_t s);
void launch_conv_in_tm(const float* x, const float* w4, const float* bias, void* y_tm, int seg, float* stats, int* step_inc, int B, int copies, int T,
                       int pitch, hipStream_t s);

// ---- VAE encoder (said/model/vae.py:26-112) ----
// windows of (L, C) token-major coefficients, window w starting at src + w * win_stride floats (win_stride = L*C for a
// batch of separate windows, step*C for sliding windows over one sequence) -> channel-major [n][C][pitch]
void launch_windows_to_cm(const float* src, long long win_stride, float* dst, int n, int L, int C, int pitch, long long dst_bstride, hipStream_t s);
// nn.Flatten of the conv stack's (n, C, T) output into the feature-major FC operand: dst[c * T + t][w] = src[w][c][t]
void launch_flatten_cm(const float* src, long long src_bstride, int src_pitch, float* dst, int dst_pitch, int n, int C, int T, hipStream_t s);

// ---- audio-encoder specific kernels ----
// conv0: 1 -> C channels, kernel K, stride S, no bias (Wav2Vec2 feature extractor layer 0)
void launch_conv0(const float* wav, const float* w, float* y, int B, int Ta, int C, int K, int S, int Tout, int pitch,
                  long long y_bstride, hipStream_t s);
// per-row (b, c) mean/rstd over Tout, then y = gelu(gamma*(y-mean)*rstd+beta) in place
// bf16 encoder front end (conv0 + GroupNorm(512, 512) + GELU -> token-major bf16 [b][Tout][512]); scratch: B * (ceil(Tout / 256) + 1) * 1024 floats.
// false: shape not served (only wav2vec2-base's conv0: 512 channels, kernel 10, stride 5)
bool launch_conv0_gn_gelu_tm_bf16(const float* wav, const float* w, const float* gamma, const float* beta, float* scratch, void* dst, int B, int Ta,
                                  int C, int K, int S, int Tout, float eps, hipStream_t s);
void launch_rownorm_gelu(float* y, const float* gamma, const float* beta, int rows_per_batch, int B, int T, int pitch,
                         long long bstride, float eps, hipStream_t s);
// linear interpolation along t with align_corners=True (wav2vec2.py:41-44)
void launch_interp_linear(const float* src, float* dst, int B, int C, int Tin, int Tout, int src_pitch, int dst_pitch,
                          long long src_bstride, long long dst_bstride, hipStream_t s);
// materialising LayerNorm over channels of a channel-major tensor: y = LN(x (+ add))
void launch_layernorm_cm(const float* x, const float* add, float* y, const float* gamma, const float* beta, int B, int C,
                         int T, int pitch, long long bstride, float eps, hipStream_t s);

}  // namespace said

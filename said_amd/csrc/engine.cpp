// engine.cpp — host side of libsaid_hip.so: context, weight packing, the UNet1D / Wav2Vec2
// kernel schedules, the hipGraph-replayed denoising loop and the C ABI of include/said_hip.h.
// No math happens here: every tensor op is one of the gfx950 kernels in gemm/attn/misc.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/said_hip.h"
#include "said_hip_debug.h"
#include "kernels.h"
#include "stchain.h"
#include "tgemm.h"

using namespace said;

namespace {

constexpr int MC = 192;       // model_channels (unet_1d_condition.py:40)
constexpr int TE = 768;       // time_embed_dim = 4 * model_channels
constexpr int HEADS = 6;      // 192 / num_head_channels(32)
constexpr int HD = 32;
constexpr int FFI = 768;      // GEGLU inner dim (4 * 192)
constexpr int NRES = 5, NST = 4;
constexpr int W2V_H = 768, W2V_HEADS = 12, W2V_HD = 64, W2V_FFN = 3072, W2V_CONV = 512;

std::string g_create_err;

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto d : shape) n *= d; return n; }
};

struct PW {  // packed GEMM weight (up to 2 K-segments) + bias
    float* w[2] = {nullptr, nullptr};
    float* w4[2] = {nullptr, nullptr};   // dwordx4 packing for the LDS-staged kernel
    float* w2[2] = {nullptr, nullptr};   // bf16 packing for the LDS-staged kernel's bf16 MFMA mode (same tails as w4)
    float* ws[2] = {nullptr, nullptr};   // split-fp16 packing (kernels.h Seg::ws; same tails as w4) — UNet weights only (said_ctx::pw_split)
    bool ws_flat = false;                // ws in the flat step layout (GEGLU's one-tile-per-wave shape)
    float* bias = nullptr;
    int N = 0, C[2] = {0, 0}, taps = 1, nseg = 1;
    bool gn_tail = false;                // w4[s] is followed by the GroupNorm gamma[C] and beta[C] of its source segment
    bool ln_tail = false;                // ... and then by the LayerNorm gamma[C] and beta[C]
};
struct ResW { float *g1, *b1, *g2, *b2; PW conv1, conv2, skip; int cin; bool has_skip; float* bias2;
              void *t_conv1 = nullptr, *t_conv2 = nullptr; /* bf16 [192][taps * cin] (conv2: [576 | 384 skip]) for tgemm.hip */
              void *tf_conv1 = nullptr, *tf_conv2 = nullptr; /* the same matrices in fp32 (fgemm_kernel) */
              void *tp_conv1 = nullptr, *tp_conv2 = nullptr; /* ... and as packed split-fp16 pairs (h | l << 16 per element: fgemm_kernel's packed mode, round 6) */ };
struct STW { float *gn_g, *gn_b, *l1g, *l1b, *l2g, *l2b, *l3g, *l3b; PW qkv, out1, q2, out2, ff1, ff2, proj, ffproj;
             void *t_qkv = nullptr, *t_ff1 = nullptr, *t_ffproj = nullptr; float* t_ff1_bias = nullptr; /* bf16 weights for tgemm.hip */
             void *tf_qkv = nullptr, *tf_ff1 = nullptr, *tf_ffproj = nullptr; /* fp32 copies (fgemm_kernel) */
             void* tp_qkv = nullptr; /* q/k/v rows as packed split-fp16 pairs (fgemm_kernel's packed mode) */
             void *t_out1 = nullptr, *t_q2 = nullptr, *t_out2 = nullptr, *tf_out1 = nullptr, *tf_q2 = nullptr, *tf_out2 = nullptr; /* [192][192] (xgemm_kernel) */
             float *chain_w = nullptr, *chain_vec = nullptr; /* round 5: weight stream + vectors of the fused tail (stchain.hip) */
             float* chain_w3 = nullptr; /* round 6: the three-slice stream (small launches: three workgroups per token tile) */
             float* chain_w2 = nullptr; /* ... and the two-slice stream (launches of 86 .. 128 tiles) */
             void* chain_wb = nullptr; /* ... and the bf16 stream (1 KB units) of its bf16-mode variant */ };
struct W2VLayer { PW qkv, out, ff1, ff2; float *ln1g, *ln1b, *ln2g, *ln2b; };

struct ActBuf {  // channel-major activation + its GroupNorm partial statistics
    float* p = nullptr;
    float* st = nullptr;
    void* t = nullptr;   // token-major twin [sample][seg rows][192] in the precision mode's element type (round 3, large batches)
};

}  // namespace

struct said_ctx {
    int device = 0, maxBe = 0, maxT = 0, cin = 32, ctx_dim = 768;
    int maxTp = 0, maxNp = 0;
    std::string err;
    std::string launch_err;          // set by a schedule function whose kernel refused its shape; reported by the C-ABI entry point
    std::map<std::string, HostTensor> host_w;
    std::vector<void*> allocs;       // weights, tables and lazily grown buffers: live as long as the context
    std::vector<void*> ws_allocs;    // the (max_batch_eff, max_frames)-sized workspace: replaced as a whole by said_reserve
    std::map<void*, size_t> alloc_bytes;   // size of every live allocation made through dalloc (said_debug_ws_*)
    std::vector<void*>* alloc_list = &allocs;
    bool finalized = false, has_audio = false, has_audio_proj = false;
    bool is_clone = false;           // said_clone: the packed weights and tables belong to the parent context
    int w2v_layers = 0;
    long long n_audio_clips = 0;   // clips encoded by said_audio_encode so far (tests: identical rows of a batch are encoded once)
    int n_set_weight = 0;   // said_set_weight calls so far (tests: capacity growth must not re-upload the weights)
    int w2v_kernel[7] = {0}, w2v_stride[7] = {5, 2, 2, 2, 2, 2, 2};

    // ---- UNet weights ----
    PW conv_in, conv_out, te1, te2, emb_all, kv_all;
    float *out_g = nullptr, *out_b = nullptr;
    ResW res[NRES];
    STW st[NST];
    float* null_cond = nullptr;
    float* c2[NST] = {nullptr, nullptr, nullptr, nullptr};   // attn2 output of the unconditional half: to_out(to_v(null_cond_emb)) + bias, per block
    bool cfg_share = true;   // exploit the guidance structure (shared prefix, constant unconditional cross-attention); SAID_NO_CFG_SHARE=1 disables
    float* freqs = nullptr;
    bool freqs_set = false;

    // ---- audio encoder weights ----
    float *c0_w = nullptr, *c0_g = nullptr, *c0_b = nullptr;
    PW aconv[7];
    float *fp_lng = nullptr, *fp_lnb = nullptr, *enc_lng = nullptr, *enc_lnb = nullptr;
    PW fproj, posconv, aproj;
    std::vector<W2VLayer> layers;

    // ---- UNet workspace ----
    float *x_cm = nullptr, *eps_cm = nullptr;
    ActBuf H0, H1, P, Q, M;
    float *X1 = nullptr, *X2 = nullptr, *X3 = nullptr, *O = nullptr, *QK = nullptr, *VT = nullptr, *F = nullptr;
    float *KV = nullptr, *CTX = nullptr;
    float* KVT = nullptr;        // key-major copy of KV [sample][S][NST * 2 * MC] for the fused SpatialTransformer tail (stchain.hip), made by run_kv
    bool band_chain_ok = false;  // the alignment band fits stchain's window tile (set_band)
    int kvt_S = -1;              // key count of the key-major copy KVT as run_kv last made it (-1: not made — the fused tail then does not run: ADVICE r5)
    int kvt_bf16 = 0;            // ... and its element type
    bool st_chain_large = true;  // ... at large batches too, beside the token-major q / k / v GEMM (32 clips: 3.2 -> 2.4 ms per step; said_debug_option "st_chain_large")
    long long st_chain_max_tiles = 1LL << 40;   // ... while the launch is at most this many workgroups (sample x 32-token tiles; said_debug_option "st_chain_max_tiles")
    int st_chain_bf16 = -1;      // bf16 mode, large batches: the same fused tail on bf16 operands instead of rgemm's five launches; 0: off; -1 / 1: stchain_kernel<true> (one token tile per
                                 // workgroup, two workgroups per CU; said_debug_option "st_chain_bf16").  (Round 5's two-tiles-per-workgroup variant measured slower — 96 vs 83 us,
                                 // profiles/r05p_stchain2_two_tiles_ab.txt — and was removed in round 6.)
    bool st_chain_dbg = false;   // debug: the fused kernel also writes x1 / x2 to X1 / X2
    int attn_2q = -1;            // fp32 mode, pre-split K / V, four key slices: three query tiles per wave (attn2q_kernel) — -1: launches of >= 512 (sample, head, tile) triples, 0: never, 1: always (said_debug_option "attn_2q")
    int attn_ks_force = 0;       // development: != 0 forces the self-attention workgroup shape (8 / 4 / 1: key-split waves; -4: four query tiles per workgroup) — said_debug_option "attn_ks"
    int st_chain_slices = -1;    // fp32 mode: -1 / 3: launches of at most CHAIN3_MAX_TILES (sample, token tile) pairs run THREE workgroups per tile, of at most CHAIN2_MAX_TILES two
                                 // (stchain.hip CU<>); 2: two wherever slicing is possible; 1: never
                                 // (said_debug_option "st_chain_slices")
    float* chain_part = nullptr; // ... their partial sums [CHAIN3_MAX_TILES][3][6][16][64] (workspace)
    int* chain_ticket = nullptr; // ... and arrival counters [CHAIN3_MAX_TILES][6]: zero between launches (not part of the workspace: said_debug_ws_fill must not touch them)
    int st_chain = -1;           // fp32 mode, small batches: everything behind self-attention as ONE launch per block (stchain.hip); 0: the five launches
                                 // (said_debug_option "st_chain")
    float *E0 = nullptr, *E1 = nullptr, *E2 = nullptr, *EO = nullptr;
    long long* ts_dev = nullptr;
    float* coef_dev = nullptr;
    int* step_dev = nullptr;
    int *band_lo = nullptr, *band_hi = nullptr;
    int band_T = -1, band_S = -1, band_wmax = 0;
    float *init_cm = nullptr, *enoise_cm = nullptr, *mask_cm = nullptr, *rescale_part = nullptr;
    float* noise_cm = nullptr; size_t noise_cm_elems = 0;
    float* coef1_dev = nullptr;  // one row for said_ddim_step
    unsigned seed_host[2] = {0, 0};
    unsigned* seed_dev = nullptr;  // [2] Philox key of the loop's eta noise (said_loop_params::noise_seed)
    float* axpby_coef = nullptr;
    long long* clk_dev = nullptr;  // [64 launches][8 waves][8 slots]
    bool clk_on = false;
    int cur_b0 = 0;          // batch offset applied to every launch issued by run_unet (parallel graph branches)
    hipStream_t cap_stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool use_branches = false;  // SAID_BRANCHES=1: capture the two halves of the UNet batch as parallel graph branches
    bool bf16_mode = false;  // said_set_precision(SAID_PREC_BF16): multiply in bf16 wherever the LDS-staged kernel is used
    int prec_mode = 0;       // the SAID_PREC_* mode asked for (said_set_precision)
    bool split_unsafe = false;   // a weight tensor lies outside the split-fp16 representation's range (scan_split_range): SAID_PREC_FP32 then runs as SAID_PREC_FP32_STRICT
    std::string split_note;      // ... which tensor and why (said_precision_note)
    int* status_dev = nullptr;   // [2] sticky numeric status of the last loop / forward call (said_numeric_status)
    bool use_ugemm = true;   // SAID_NO_UGEMM=1 forces the generic kernel everywhere (A/B testing)

    // ---- bf16 audio encoder (tgemm.hip): bf16 weights [N][K] (convs: K = tap-major), token-major workspace ----
    void* bw_conv[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void *bw_fproj = nullptr, *bw_aproj = nullptr;
    void* bw_out = nullptr;   // out.2 weights bf16 [in_channels][3 taps][192] for out_sched_tm_kernel (round 4)
    int out_tm = -1;          // the token-major out + scheduler kernel behind the bf16 large-batch schedule (said_debug_option "out_tm": 0 off)
    void* bw_pos = nullptr;       // positional conv as 16 GEMMs: bf16 [16][64 (48 + 16 zero rows)][taps * 48], tap-major
    float* pos_bias_pad = nullptr; // its bias, 16 zeros appended (the last group's 64-wide tile reads past 768)
    void* bXg = nullptr;          // per-group operand [clips][16][R][48] bf16
    size_t bXg_elems = 0;
    bool pos_tgemm = true;        // SAID_NO_POSCONV_TGEMM=1: the grouped fp32 channel-major kernel also in bf16 mode
    struct BLayer { void *qkv, *out, *ff1, *ff2; };
    std::vector<BLayer> blayers;
    void *bA0 = nullptr, *bA1 = nullptr, *bX = nullptr, *bHb = nullptr, *bF = nullptr, *bO = nullptr;
    float *bH = nullptr, *bT = nullptr, *bPosT = nullptr;
    size_t b_conv_elems[2] = {0, 0}, b_tok = 0;
    bool audio_bf16 = true;   // SAID_NO_AUDIO_BF16=1: the fp32 audio encoder also in bf16 mode
    // bf16-mode UNet at large batch: token-major bf16 GEMM operands (tgemm.hip) prepared from the channel-major fp32 activations
    float* gn_coef = nullptr;   // [2 slots][maxBe][192][2] GroupNorm coefficients for prep_kernel
    void *uPA = nullptr, *uPB = nullptr, *uPL = nullptr, *uPH = nullptr, *uPX = nullptr;   // conv operand [Be][T+2][384], raw cat input
                                                                                            // [Be][T][384], LN'd [Be][T][192], GEGLU out [Be][T][768], raw x2 [Be][T][192]
    int tm_acts = -1;         // bf16 mode, large batches: token-major bf16 activations BETWEEN the UNet kernels, operand transforms inside the GEMMs (-1 / 1: on; 0: the
                              // channel-major schedule with preparation kernels — said_debug_option "tm_acts").  The fp32 twin of this schedule (round 3: 4.78 vs 4.41 ms per
                              // step at 32 clips) was removed in round 6 together with xgemm_kernel's fp32 instantiations.
    bool mt_mid = true;       // multi-tile workgroups for mid-size launches too (said_debug_option "mt_mid")
    int mt_wgs = 0;           // > 0: multi-tile workgroups from this many workgroups per token tile on (said_debug_option "mt_wgs"; default 1024)
    int tgemm_direct = -1;    // audio encoder (bf16): the projections on tgemm256d_kernel (256 x 256 tile, operand tiles loaded straight into LDS; -1 / 1: on, 0: tgemm_kernel<128> — said_debug_option "tgemm_direct")
    int tgemm_sb = 1;         // audio encoder (bf16): the single-LDS-buffer 128 x 128 GEMM variant, three workgroups per CU (said_debug_option "tgemm_sb"; 0: double buffer, two per CU)
    int unet_nb = 0;          // > 0: forces pick_unet's column tiles per workgroup (said_debug_option "unet_nb")
    bool unet_nb_model = true; // pick_unet by the busiest-CU model (0: round 2's rule; said_debug_option "unet_nb_model")
    bool f32_out1_tm = true;  // fp32 large batch: attn1.to_out on the token-major fp32 GEMM (said_debug_option "f32_out1_tm")
    bool hybrid = true;       // bf16 mode at large batch: SpatialTransformers from the attention output on use round 3's token-major kernels
    int xgemm_dbg = 0;
    bool xclk_on = false;
    int battn = -1;           // round 4: bf16-operand self-attention with a head's K / V resident in LDS (attn.hip: battn_kernel) behind rgemm's q/k/v;
                              // 0: attn_kernel on fp32 operands (said_debug_option "battn")
    int attn_split = -1;      // fp32 mode: both attention products on split-fp16 operands (attn.hip: PM == 2; x = h + 2^-11 l: 22-bit significands, fp32 accumulation,
                              // as close to a float64 evaluation as the fp32 MFMAs: tests/test_gpu_round4.py).  Default (-1) and 1: ON since round 5; 0: v_mfma_f32_32x32x2_f32
                              // on the fp32 operands (said_debug_option "attn_split").  Round 4 shipped it opt-in because runs beside other streams were not bit-stable;
                              // round 5 found the mechanism in OTHER kernels' packed-fp32 instructions (split_f16.h, build.py NO_SLP) and removed it.
    int gemm_presplit = -1;   // fp32 mode, large batches: the ResBlock convolutions' and q / k / v's operands reach fgemm_kernel already split (prep_kernel packs the activations,
                              // the weights have a packed copy): -1 / 1 on; 0: fp32 operands split in the k loop (said_debug_option "gemm_presplit"; bit-identical)
    int gemm_split = -1;      // fp32 mode: the large-batch token-major GEMMs (fgemm_kernel) on split-fp16 operands (tgemm.hip: SP).  Default (-1) and 1: ON since round 5
                              // (as above); 0: fp32 MFMAs (said_debug_option "gemm_split").
    int attn_presplit = -1;   // fp32 small batch: the q/k/v GEMM stores k and v as packed split-fp16 pairs and attn_kernel<PM = 3> unpacks them instead of splitting all of K and V
                              // again in each of a sample's query-tile workgroups (-1 / 1: on; 0: off — said_debug_option "attn_presplit")
    int out_split = -1;       // out_sched_kernel's convolution on split-fp16 operands (-1 / 1: on; 0: fp32 matrix instructions — said_debug_option "out_split")
    int kconv = -1;           // fp32 mode, small batch: the K-long ResBlock convolutions of the up path as straight-line two- / three-block waves (gemm_lds.hip kconv_body; -1 / 1: on,
                              // 0: ugemm_body's block loop — said_debug_option "kconv"; bit-identical)
    long long kconv_max_tiles = 4096;   // launches of at most this many (sample, token tile) pairs run the K-long convolutions as NB = 1 kconv_body workgroups when the chosen NB has no split shape
    int chain_coef = -1;      // fp32 small batch: stchain_kernel takes the block input's GroupNorm coefficients from the q/k/v GEMM (-1 / 1: on; 0: finalises them itself — said_debug_option "chain_coef")
    int ugemm_split = -1;     // fp32 mode: the small-batch channel-major GEMMs (ugemm_kernel) on split-fp16 operands too (gemm_lds.hip: SP; weights pre-split on the host:
                              // Seg::ws).  Default (-1) and 1: ON; 0: fp32 MFMAs (said_debug_option "ugemm_split")
    int pw_split = 0;         // make_pw: also build the split-fp16 packing (1: per-block layout, 2: flat) — set around the UNet weights only
    int rgemm = -1;           // round 4: register-stationary, wave-specialised persistent GEMMs (rgemm.hip) wherever launch_rgemm serves the shape
                              // (bf16 mode: 192-wide GEMMs with K <= 576, q/k/v); 0: off (said_debug_option "rgemm")
    long long n_rgemm = 0;
    long long n_stchain = 0;  // launches issued through stchain_kernel (said_debug_get)
    long long n_xgemm = 0;    // launches issued through round 3's xgemm_kernel (said_debug_get; n_rgemm: through rgemm_kernel)
    int xgemm_ntw = 0;        // test / measurement: column tiles per workgroup of the resident-source GEMMs (0: launch_xgemm decides)
    void *tX1 = nullptr, *tX2 = nullptr, *tO = nullptr, *tF = nullptr;   // token-major x1, x2, attention output [.][192], GEGLU product [.][768]
    bool unet_tgemm = true;   // SAID_NO_UNET_TGEMM=1 keeps the channel-major kernels in bf16 mode at every batch size
    bool unet_fgemm = true;   // SAID_NO_UNET_FGEMM=1: the same for the fp32 mode's token-major path (fgemm_kernel)
    // tokens per launch from which the token-major GEMM path is taken (measured crossovers, scripts/gpu_r2_w.sh: bf16 between 4800
    // and 6000 tokens, fp32 between 9600 and 10800); SAID_UNET_TGEMM_MIN overrides both
    // (bf16: 3000 since round 4 — with the persistent kernels the crossover sits between 2400 rows (2 clips x 600 frames under guidance: 21.9 vs 22.6 ms per
    //  50 steps) and 3600 (3 clips: 30.3 vs 22.7 ms); rounds 2-3: 5800)
    long long unet_tgemm_min_tokens = 3000, unet_fgemm_min_tokens = 10000;
    long long unet_fgemm_min_concurrent = 6000;   // fp32 threshold while other contexts' loops run beside this one (said_loop_params::concurrent)
    bool cur_concurrent = false;
    int spg_limit = 50;      // denoise steps captured per graph in loops of >= 400 steps, at most 10 below (steps_per_graph; round 6: 10 -> 50: at 0.245 ms per step the 100 graph boundaries of a 1000-step loop were 0.4-0.5 % of it — headline
                             // 2424-2429 -> 2435-2438 frames/s; 100 / 250 / 500 add nothing: profiles/r06l_steps_per_graph.txt)
    int audio_chunk = 32;    // clips per audio-encoder pass

    // ---- audio workspace (lazily sized) ----
    float *abufA = nullptr, *abufB = nullptr; size_t abuf_elems[2] = {0, 0};
    float *aH = nullptr, *aT = nullptr, *aO = nullptr, *aQK = nullptr, *aVT = nullptr, *aF = nullptr, *aPOS = nullptr, *aX = nullptr;
    size_t a_tok_elems = 0; int a_chunk = 0;

    // ---- per-step graph ----
    hipStream_t cap_stream = nullptr;  // private stream used only to capture the per-step graph
    hipStream_t own_stream = nullptr;  // a clone's stream (said_stream): from the process-wide pool below, never destroyed
    int n_clones = 0;                  // clones made of this context so far (picks the pool slot)
    hipGraph_t graph = nullptr, graph_rem = nullptr;       // `gspg` consecutive steps / the N % gspg remaining steps
    hipGraphExec_t gexec = nullptr, gexec_rem = nullptr;
    std::vector<long long> gkey;
    int gspg = 1;            // denoise steps captured per graph
    int gnodes = 0;
    int dbg_stop = -1, dbg_count = 0, dbg_only = -1;
    bool log_on = false;
    struct StageInfo { int kind, epi, NB, KS; double bytes, flops; };
    std::vector<StageInfo> stage_log;
};

namespace {

int fail(said_ctx* c, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return -1;
}

// fp32 mode multiplies on split-fp16 operands (split_f16.h) unless the caller asked for strict fp32 (said_set_precision) or the weights do not fit the
// representation (scan_split_range): `opt` is one of the per-kernel-family development switches (said_debug_option), all on by default.
inline bool strict_f32(const said_ctx* c) { return !c->bf16_mode && (c->prec_mode == SAID_PREC_FP32_STRICT || c->split_unsafe); }
inline bool sp_on(const said_ctx* c, int opt) { return opt != 0 && !strict_f32(c); }

// Range check of a tensor whose elements are split into fp16 planes (w = h + 2^-11 l, both fp16): h is finite for |w| < 65504 — a 2x margin is kept — and the
// pair resolves 2^-36 absolute, i.e. 2^-22 of the tensor's largest element (fp32's own resolution in a dot product) only while that element is >= 2^-14.
// Outside this range fp32 mode keeps the fp32 matrix instructions for EVERYTHING (one arithmetic per run), and said_precision_note says why.
void scan_split_range(said_ctx* ctx, const std::string& name, const float* w, size_t n) {
    if (ctx->split_unsafe) return;
    float mx = 0.f;
    bool finite = true;
    for (size_t i = 0; i < n; ++i) { const float a = std::fabs(w[i]); if (!(a <= 3.4028234663852886e38f)) finite = false; else if (a > mx) mx = a; }
    if (!finite || mx >= 32768.f || (mx > 0.f && mx < 6.103515625e-05f)) {
        char b[320];
        snprintf(b, sizeof b, "%s: max |w| = %.3g is outside [2^-14, 2^15): fp32 mode runs on v_mfma_f32_32x32x2_f32 (SAID_PREC_FP32_STRICT) instead of split-fp16 products",
                 name.c_str(), finite ? (double)mx : INFINITY);
        ctx->split_unsafe = true;
        ctx->split_note = b;
    }
}

static bool trace_on() { static int v = -1; if (v < 0) v = dev_env("SAID_TRACE") ? 1 : 0; return v == 1; }

#define TRACE(msg) do { if (trace_on()) { fprintf(stderr, "[said] %s:%d %s\n", __FILE__, __LINE__, msg); fflush(stderr); } } while (0)

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return fail(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// The capture streams exist only while a graph is being built: a live stream holds one of the device's few hardware queues
// (four by default), and streams beyond that count share queues, which serialises launches the caller meant to run side by
// side (the clip groups' loops).
static int ensure_cap_streams(said_ctx* ctx) {
    if (!ctx->cap_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
    if (ctx->use_branches) {
        if (!ctx->cap_stream2) HIPCHK(hipStreamCreateWithFlags(&ctx->cap_stream2, hipStreamNonBlocking));
        if (!ctx->ev_fork) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        if (!ctx->ev_join) HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    return 0;
}
static void release_cap_streams(said_ctx* ctx) {
    if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
    if (ctx->cap_stream2) (void)hipStreamDestroy(ctx->cap_stream2);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    ctx->cap_stream = ctx->cap_stream2 = nullptr; ctx->ev_fork = ctx->ev_join = nullptr;
}

// a schedule function's kernel refused its shape (nothing was launched for it): report instead of continuing
#define LAUNCHCHK()                                                                                   \
    do {                                                                                              \
        if (const char* f_ = launch_fault_peek()) { const std::string m_ = f_; launch_fault_clear(); ctx->launch_err.clear(); return fail(ctx, "%s", m_.c_str()); } \
        if (!ctx->launch_err.empty()) { const std::string m_ = ctx->launch_err; ctx->launch_err.clear(); return fail(ctx, "%s", m_.c_str()); } \
    } while (0)

inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// said_create / said_destroy must not change the calling thread's current device as a side effect
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

template <typename T>
int dalloc(said_ctx* ctx, T** out, size_t n, bool zero = true) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
    if (zero) HIPCHK(hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)));
    ctx->alloc_list->push_back(p);
    ctx->alloc_bytes[p] = std::max<size_t>(n, 1) * sizeof(T);
    *out = static_cast<T*>(p);
    return 0;
}
// grow-on-demand buffers: the buffer being replaced is released (the caller has synchronised the stream that used it)
template <typename T>
int drealloc(said_ctx* ctx, T** out, size_t n, bool zero = true) {
    if (*out) {
        auto it = std::find(ctx->allocs.begin(), ctx->allocs.end(), static_cast<void*>(*out));
        if (it != ctx->allocs.end()) ctx->allocs.erase(it);
        HIPCHK(hipFree(*out));
        *out = nullptr;
    }
    return dalloc(ctx, out, n, zero);
}
int upload(said_ctx* ctx, float** out, const float* h, size_t n) {
    if (dalloc(ctx, out, n, false)) return -1;
    HIPCHK(hipMemcpy(*out, h, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

const HostTensor* getw(said_ctx* ctx, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = ctx->host_w.find(name);
    if (it == ctx->host_w.end()) { fail(ctx, "missing key in state dict: %s", name.c_str()); return nullptr; }
    if (it->second.shape != std::vector<int64_t>(shape)) {
        std::string got;
        for (auto d : it->second.shape) got += std::to_string(d) + ",";
        fail(ctx, "size mismatch for %s: got (%s)", name.c_str(), got.c_str());
        return nullptr;
    }
    return &it->second;
}
int upvec(said_ctx* ctx, float** out, const std::string& name, int64_t n) {
    const HostTensor* t = getw(ctx, name, {n});
    if (!t) return -1;
    return upload(ctx, out, t->data.data(), (size_t)n);
}

// Pack W[Ntot][Ctot][taps] into MFMA A-fragment order: Wp[group][tile][tap][cpair][lane],
// lane l <-> (n = tile*32 + (l & 31), c = c_begin + 2*cpair + (l >> 5)); rows beyond N are zero.
std::vector<float> pack_rows(const float* W, int Ctot, int taps, const std::vector<int>& row_of /* per (group,tile,r): row or -1 */,
                             int ntiles_total, int c_begin, int C) {
    std::vector<float> out((size_t)ntiles_total * taps * (C / 2) * 64);
    size_t o = 0;
    for (int tile = 0; tile < ntiles_total; ++tile)
        for (int tap = 0; tap < taps; ++tap)
            for (int cp = 0; cp < C / 2; ++cp)
                for (int l = 0; l < 64; ++l) {
                    const int row = row_of[tile * 32 + (l & 31)];
                    const int c = c_begin + 2 * cp + (l >> 5);
                    out[o++] = row < 0 ? 0.f : W[((size_t)row * Ctot + c) * taps + tap];
                }
    return out;
}
// dwordx4 packing: Wq[tile][tap][c/8][lane][4]; value j of lane l = W[tile*32 + (l & 31)][c_begin + 8*cq + 2*j + (l >> 5)][tap]
std::vector<float> pack_rows4(const float* W, int Ctot, int taps, const std::vector<int>& row_of, int ntiles_total, int c_begin, int C) {
    std::vector<float> out((size_t)ntiles_total * taps * (C / 8) * 256);
    size_t o = 0;
    for (int tile = 0; tile < ntiles_total; ++tile)
        for (int tap = 0; tap < taps; ++tap)
            for (int cq = 0; cq < C / 8; ++cq)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 4; ++j) {
                        const int row = row_of[tile * 32 + (l & 31)];
                        const int c = c_begin + 8 * cq + 2 * j + (l >> 5);
                        out[o++] = row < 0 ? 0.f : W[((size_t)row * Ctot + c) * taps + tap];
                    }
    return out;
}
// round-to-nearest-even fp32 -> bf16 (finite inputs)
inline uint16_t bf16_rne(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}
// bf16 packing for v_mfma_f32_32x32x8_bf16_1k: Wb[tile][tap][c/8][lane][4]; value j of lane l =
// W[tile*32 + (l & 31)][c_begin + 8*cq + 4*(l >> 5) + j][tap].  Returned as float storage (2 bf16 per float).
std::vector<float> pack_rows_bf16(const float* W, int Ctot, int taps, const std::vector<int>& row_of, int ntiles_total, int c_begin, int C) {
    std::vector<uint16_t> h((size_t)ntiles_total * taps * (C / 8) * 256);
    size_t o = 0;
    for (int tile = 0; tile < ntiles_total; ++tile)
        for (int tap = 0; tap < taps; ++tap)
            for (int cq = 0; cq < C / 8; ++cq)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 4; ++j) {
                        const int row = row_of[tile * 32 + (l & 31)];
                        const int c = c_begin + 8 * cq + 4 * (l >> 5) + j;
                        h[o++] = row < 0 ? (uint16_t)0 : bf16_rne(W[((size_t)row * Ctot + c) * taps + tap]);
                    }
    std::vector<float> out(h.size() / 2);
    memcpy(out.data(), h.data(), h.size() * 2);
    return out;
}
// split-fp16 packing for v_mfma_f32_32x32x16_f16 (gemm_lds.hip SP; kernels.h Seg::ws): w = h + 2^-11 l with h = RN16(w), l = RN16((w - h) * 2^11) (split_f16.h).
// Per 24-channel block: NS = 5 (3 taps) / 2 (1 tap) k16 steps x 2 planes (h, l) x 64 lanes x 8 halfs; the 8 halfs of lane l in step s are K-group
// g = 2 s + (l >> 5) of the block's tap-major K slice (tap = g / 3, channels 8 (g % 3) .. + 7), zeros past the slice's 9 / 3 groups.  flat: C / 16 steps over
// the whole K (taps == 1), no padding.  Returned as float storage (2 halfs per float).
std::vector<float> pack_rows_split(const float* W, int Ctot, int taps, const std::vector<int>& row_of, int ntiles_total, int c_begin, int C, bool flat) {
    const int nblk = flat ? 1 : C / 24, ns = flat ? C / 16 : (taps == 3 ? 5 : 2), ng = flat ? C / 8 : 3 * taps;
    std::vector<_Float16> h((size_t)ntiles_total * nblk * ns * 2 * 512);
    size_t o = 0;
    for (int tile = 0; tile < ntiles_total; ++tile)
        for (int blk = 0; blk < nblk; ++blk)
            for (int st = 0; st < ns; ++st)
                for (int pl = 0; pl < 2; ++pl)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int row = row_of[tile * 32 + (l & 31)];
                            const int g = 2 * st + (l >> 5);
                            float v = 0.f;
                            if (row >= 0 && g < ng) {
                                const int tap = flat ? 0 : g / 3;
                                const int c = c_begin + (flat ? 8 * g : 24 * blk + 8 * (g % 3)) + j;
                                v = W[((size_t)row * Ctot + c) * taps + tap];
                            }
                            const _Float16 hv = (_Float16)v;
                            h[o++] = pl == 0 ? hv : (_Float16)((v - (float)hv) * 2048.f);
                        }
    std::vector<float> out(h.size() / 2);
    memcpy(out.data(), h.data(), h.size() * 2);
    return out;
}
// fp32 host matrix -> bf16 (RNE) device array; `perm` (optional) reorders the K axis of a Conv1d weight [N][C][taps] to
// tap-major [N][taps][C] (the token-major im2col order of tgemm.hip)
int upload_bf16(said_ctx* ctx, void** out, const float* W, size_t N, size_t C, size_t taps) {
    std::vector<uint16_t> h(N * C * taps);
    for (size_t n = 0; n < N; ++n)
        for (size_t t = 0; t < taps; ++t)
            for (size_t c = 0; c < C; ++c) h[(n * taps + t) * C + c] = bf16_rne(W[(n * C + c) * taps + t]);
    uint16_t* d = nullptr;
    if (dalloc(ctx, &d, h.size(), false)) return -1;
    HIPCHK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    *out = d;
    return 0;
}
// the same matrix in bf16 AND fp32 (UNet operands of the token-major GEMMs: the precision mode is chosen per call)
// out_packed (optional): a third copy whose elements are split-fp16 pairs, one dword h | l << 16 each (split_f16.h pack_split_f16: the same two conversions) —
// fgemm_kernel's packed mode unpacks them with v_perm instead of splitting fp32 weights in its k loop
int upload_tm_pair(said_ctx* ctx, void** out_bf, void** out_f32, const float* W, size_t N, size_t C, size_t taps, void** out_packed = nullptr) {
    if (upload_bf16(ctx, out_bf, W, N, C, taps)) return -1;
    std::vector<float> h(N * C * taps);
    for (size_t n = 0; n < N; ++n)
        for (size_t t = 0; t < taps; ++t)
            for (size_t c = 0; c < C; ++c) h[(n * taps + t) * C + c] = W[(n * C + c) * taps + t];
    float* d = nullptr;
    if (upload(ctx, &d, h.data(), h.size())) return -1;
    *out_f32 = d;
    if (out_packed) {
        std::vector<float> pk(h.size());
        for (size_t i = 0; i < h.size(); ++i) {
            const float v = h[i];
            const _Float16 hv = (_Float16)v;
            const _Float16 lv = (_Float16)((v - (float)hv) * 2048.f);
            uint16_t hb, lb;
            memcpy(&hb, &hv, 2); memcpy(&lb, &lv, 2);
            const uint32_t u = (uint32_t)hb | ((uint32_t)lb << 16);
            memcpy(&pk[i], &u, 4);
        }
        float* dp = nullptr;
        if (upload(ctx, &dp, pk.data(), pk.size())) return -1;
        *out_packed = dp;
    }
    return 0;
}
std::vector<int> rows_dense(int N, int row0 = 0) {
    const int nt = (N + 31) / 32;
    std::vector<int> r(nt * 32, -1);
    for (int i = 0; i < N; ++i) r[i] = row0 + i;
    return r;
}

// Linear/conv weight `name` (N, Ctot[, taps]) -> PW with the K range split into `nseg` equal segments.
// gn_gamma/gn_beta (optional): affine of the GroupNorm applied to this GEMM's source; appended to each segment's w4
// block so that the LDS-staged kernel can locate them from its preloaded header alone (gemm_lds.hip, FastHdr).
int make_pw(said_ctx* ctx, PW* pw, const std::string& wname, const std::string& bname, int N, int Ctot, int taps, int nseg = 1,
            const std::string& gn_gamma = "", const std::string& gn_beta = "", const std::string& ln_gamma = "",
            const std::string& ln_beta = "") {
    const HostTensor* t = taps > 0 && ctx->host_w.count(wname) && ctx->host_w[wname].shape.size() == 3
                              ? getw(ctx, wname, {N, Ctot, taps})
                              : getw(ctx, wname, {N, Ctot});
    if (!t) return -1;
    const int tp = t->shape.size() == 3 ? taps : 1;
    pw->N = N; pw->taps = tp; pw->nseg = nseg;
    const auto rows = rows_dense(N);
    for (int s = 0; s < nseg; ++s) {
        const int C = Ctot / nseg;
        pw->C[s] = C;
        auto packed = pack_rows(t->data.data(), Ctot, tp, rows, (N + 31) / 32, s * C, C);
        if (upload(ctx, &pw->w[s], packed.data(), packed.size())) return -1;
        if (C % 8 == 0 && (tp == 1 || tp == 3)) {
            auto p4 = pack_rows4(t->data.data(), Ctot, tp, rows, (N + 31) / 32, s * C, C);
            auto p2 = pack_rows_bf16(t->data.data(), Ctot, tp, rows, (N + 31) / 32, s * C, C);
            const size_t w4_floats = p4.size();
            if (!gn_gamma.empty()) {
                const HostTensor* gg = getw(ctx, gn_gamma, {Ctot});
                const HostTensor* gb = getw(ctx, gn_beta, {Ctot});
                if (!gg || !gb) return -1;
                p4.insert(p4.end(), gg->data.begin() + s * C, gg->data.begin() + (s + 1) * C);
                p4.insert(p4.end(), gb->data.begin() + s * C, gb->data.begin() + (s + 1) * C);
                pw->gn_tail = true;
            }
            if (!ln_gamma.empty()) {
                const HostTensor* lg = getw(ctx, ln_gamma, {Ctot});
                const HostTensor* lb = getw(ctx, ln_beta, {Ctot});
                if (!lg || !lb) return -1;
                p4.insert(p4.end(), lg->data.begin() + s * C, lg->data.begin() + (s + 1) * C);
                p4.insert(p4.end(), lb->data.begin() + s * C, lb->data.begin() + (s + 1) * C);
                pw->ln_tail = true;
            }
            p2.insert(p2.end(), p4.begin() + w4_floats, p4.end());   // the GroupNorm / LayerNorm tails, unchanged
            if (upload(ctx, &pw->w4[s], p4.data(), p4.size())) return -1;
            if (upload(ctx, &pw->w2[s], p2.data(), p2.size())) return -1;
            if (ctx->pw_split && C % 192 == 0) {   // (KS = 8 waves x whole 24-channel blocks)
                const bool flat = ctx->pw_split == 2;
                if (s == 0) scan_split_range(ctx, wname, t->data.data(), t->data.size());
                auto ps = pack_rows_split(t->data.data(), Ctot, tp, rows, (N + 31) / 32, s * C, C, flat);
                ps.insert(ps.end(), p4.begin() + w4_floats, p4.end());
                if (upload(ctx, &pw->ws[s], ps.data(), ps.size())) return -1;
                pw->ws_flat = flat;
            }
        }
    }
    if (!bname.empty()) { if (upvec(ctx, &pw->bias, bname, N)) return -1; }
    return 0;
}

struct LaunchCfg { int NB, KS; };
// experiment knob: SAID_BIG=cgemm restores the generic kernel's large-batch tile shapes
static bool big_cgemm() { static const bool v = dev_env("SAID_BIG") && !strcmp(dev_env("SAID_BIG"), "cgemm"); return v; }
LaunchCfg pick_cfg(long long t_tiles_total, int ntiles, bool allow6 = true) {
    // small problems: maximise workgroups (split K over 8 waves, one tile each);
    // large problems: amortise the operand transform over more tiles per workgroup.
    if (t_tiles_total * ntiles <= 1536 || ntiles % 2) return {1, 8};
    if (t_tiles_total * ntiles <= 4096) return {2, 8};
    if (allow6 && ntiles % 6 == 0) return {6, 4};
    if (ntiles % 4 == 0) return {4, 4};
    if (ntiles % 3 == 0) return {3, 4};
    return {2, 8};
}

// UNet GEMMs with 6 output tiles (192 channels): the LDS-staged kernel at every batch size — two tiles per workgroup
// as soon as that still fills the chip (measured at Be=32: 47 TFLOP/s against 34 for the generic NB=6 shape)
// Column tiles per workgroup (NB) of the UNet's 192-wide channel-major GEMMs.  A launch is t_tiles x 6 / NB workgroups on 256 CUs; its
// time is that of the busiest CU: ceil(workgroups / 256) workgroups of (F + NB) units each, F = 0.45 the per-workgroup fixed part
// (GroupNorm finalisation, operand tile; fitted on T = 1800: NB = 1 14.8 us, NB = 2 25.0 us at 342 workgroups each).  Round 2's rule
// (NB = 2 from 64 tiles) put 342 workgroups on 256 CUs at T = 1800: two rounds of three units where NB = 3 is one round of 3.45.
LaunchCfg pick_unet(const said_ctx* c, long long t_tiles_total) {
    if (big_cgemm()) return pick_cfg(t_tiles_total, 6);
    if (c->unet_nb > 0) return LaunchCfg{c->unet_nb, 8};
    if (!c->unet_nb_model || t_tiles_total * 6 > 1024) return t_tiles_total * 3 >= 192 ? LaunchCfg{2, 8} : LaunchCfg{1, 8};   // (large launches: multi-tile NB = 2 shapes)
    int best = 1;
    double cost = 1e30;
    for (int nb = 1; nb <= 3; ++nb) {
        const long long wgs = t_tiles_total * (6 / nb);
        const double k = (double)((wgs + 255) / 256) * (0.45 + nb);
        if (k < cost - 1e-9) { cost = k; best = nb; }
    }
    return LaunchCfg{best, 8};
}

Seg mkseg(const float* x, long long bstride, int pitch, int C, int taps, int pad, int stride, int Tin, int xform, const float* w) {
    Seg s;
    memset(&s, 0, sizeof s);
    s.x = x; s.w = w; s.x_bstride = bstride; s.x_pitch = pitch; s.C = C; s.taps = taps; s.pad = pad; s.stride = stride;
    s.Tin = Tin; s.xform = xform; s.gn_cpg = 1; s.gn_nparts = 1;
    return s;
}
inline Seg with_w4(Seg s, const float* w4, bool gn_tail = false, bool ln_tail = false) { s.w4 = w4; s.w4_gn_tail = gn_tail ? 1 : 0; s.w4_ln_tail = ln_tail ? 1 : 0; return s; }
// attach K-segment k of a packed weight: both packings for the LDS-staged kernel and what follows them
inline Seg with_pw(Seg s, const PW& pw, int k) {
    s.w4 = pw.w4[k]; s.w2 = pw.w2[k]; s.ws = pw.ws[k]; s.ws_flat = pw.ws_flat ? 1 : 0; s.w4_gn_tail = pw.gn_tail ? 1 : 0; s.w4_ln_tail = pw.ln_tail ? 1 : 0;
    return s;
}
void seg_gn(Seg& s, const float* part, long long part_bstride, int cpg, int nparts, float eps, const float* g, const float* b) {
    s.gn_part = part; s.gn_part_bstride = part_bstride; s.gn_cpg = cpg; s.gn_nparts = nparts; s.gn_eps = eps; s.gn_gamma = g; s.gn_beta = b;
}
GemmArgs mkargs(int T, int N) {
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.T = T; a.N = N; a.groups = 1; a.ntiles_per_group = (N + 31) / 32;
    return a;
}

// geometry of one UNet evaluation
struct UGeo {
    int Be, B_lat /* latents batch (b_mod) */, T, Tp, np /* GN partials per channel */, S, Sp;
    long long hs;      // batch stride of a 192-channel activation
    long long sts;     // batch stride of its stats
    const int* step_ptr;
    int emb_b_stride;
    int b0;            // first sample of this launch range (Be = number of samples in the range)
    int* step_inc;     // if set: the first kernel of the schedule increments this counter
    const OutSchedArgs* out_sched;   // if set: the `out` conv is fused with the scheduler update (loop only)
    int Bc;            // > 0: classifier-free guidance over Bc clips — samples [0, Bc) are the unconditional half (context =
                       // null_cond_emb repeated), [Bc, 2 Bc) the conditional half, both on the SAME latents and timestep
                       // (diffusion.py:397-400, 421-423).  Then (a) everything before the first cross-attention is computed
                       // once per clip, (b) the unconditional half's cross-attention output is the constant c2[blk].
};

inline bool dbg_go(said_ctx* c) {
    const int k = c->dbg_count++;
    if (c->dbg_only >= 0) return k == c->dbg_only;
    return c->dbg_stop < 0 || k < c->dbg_stop;
}

// (a helper launch that is not a node of the counted schedule: runs whenever the NEXT counted launch would)
inline bool dbg_go_peek(const said_ctx* c) { return c->dbg_only >= 0 ? c->dbg_count == c->dbg_only : (c->dbg_stop < 0 || c->dbg_count < c->dbg_stop); }

// algorithmic HBM bytes / flops of one launch: weights + operands in + residual + result out
// Multi-tile workgroups (gemm_lds.hip, MT): once a launch would have several thousand workgroups, each workgroup walks
// over `tt` consecutive token tiles instead, keeping its weights in registers; tt is chosen so that ~4 workgroups per
// CU remain.  Returns 1 when the launch is not eligible.
static int pick_tt(said_ctx* c, const GemmArgs& a, int epi, int batch, int& NB, int KS, bool bf) {
    static const bool mt_off = dev_env("SAID_NO_MT") != nullptr;
    if (mt_off || !c->use_ugemm || a.step_inc) return 1;
    if (epi == EPI_GEGLU || epi == EPI_BAND) return 1;   // measured slower multi-tile (B=32: GEGLU NB=2 x tt vs NB=4, band)
    int nb = NB;
    const long long ntt = (a.T + 31) / 32;
    const long long wgs = ntt * (a.ntiles_per_group / nb) * batch;
    static const long long wgs_env = dev_env("SAID_MT_WGS") ? std::max(64, atoi(dev_env("SAID_MT_WGS"))) : 1024;   // experiment knob
    const long long wgs_per_tile = c->mt_wgs > 0 ? c->mt_wgs : wgs_env;
    long long want = wgs / wgs_per_tile;
    // mid-size launches (one to four rounds of one workgroup per CU, e.g. q/k/v at T = 1800: 684 workgroups): as many token tiles per
    // workgroup as there would be rounds, so that ONE round of <= 256 workgroups remains (configs[4]: 16.87k -> 17.34k frames/s)
    if (want <= 1 && c->mt_mid && c->mt_wgs <= 0) want = wgs / 228;
    int tt = (int)std::min<long long>(std::min<long long>(8, want), ntt);
    if (tt <= 1 || !ugemm_supports(a, epi, nb, KS, bf, tt)) return 1;
    NB = nb;
    return tt;
}

// returns true when the launch went to ugemm_kernel (gemm_lds.hip) — the only kernel that honours GemmCommon::gn_coef_out / kv_split
bool do_gemm(said_ctx* c, const GemmArgs& a, int epi, int batch, int NB, int KS, hipStream_t s) {
    bool on_ugemm = false;
    GemmArgs a2 = a;
    a2.b0 = c->cur_b0;
    a2.kconv_off = (c->kconv == 0) ? 1 : 0;
    const bool bf = c->bf16_mode;
    if (NB == 3 && epi == EPI_STORE && c->use_ugemm && !ugemm_supports(a2, epi, 3, KS, bf)) NB = 2;   // (the two-segment fp32 shapes spill at NB = 3: not built)
    // K-long up-path convolutions (two / three K segments): the split-fp16 shapes exist for one column tile per workgroup only (kconv_body; with two the second accumulator
    // set spills).  Where that takes at most 1.5 x the rounds of the chosen shape on the fp32 matrix instructions it wins (a round of NB = 1 split ~11 us, of NB = 2 fp32 ~21 us:
    // configs[4] 24.3k -> 25.5k frames/s, 3 clips +6 %; 76 tiles — 2 rounds against 1 — loses 2 %: profiles/r06g_kconv_ab.txt)
    if (!bf && epi == EPI_STORE && NB > 1 && a2.nseg >= 2 && c->kconv != 0 && sp_on(c, c->ugemm_split) && c->use_ugemm && !a2.step_inc && !c->clk_on && (!c->cur_concurrent || c->kconv == 2)) {   // (concurrent clip groups share the CUs: fewer workgroups win there — 5 clips -1.9 %; "kconv" = 2: there too)
        const long long tiles = (long long)batch * ((a2.T + 31) / 32), r1 = (tiles * 6 + 255) / 256, rn = (tiles * (6 / NB) + 255) / 256;
        if (tiles <= c->kconv_max_tiles && 2 * r1 <= 3 * rn && !ugemm_supports(a2, epi, NB, KS, 2) && ugemm_supports(a2, epi, 1, 8, 2)) { NB = 1; KS = 8; }
    }
    const int tt = pick_tt(c, a2, epi, batch, NB, KS, bf);
    // fp32 mode, single-tile workgroups: split-fp16 products wherever the shape is built for them (gemm_lds.hip SP) and the weights carry the packing
    const bool sp = !bf && tt <= 1 && sp_on(c, c->ugemm_split) && c->use_ugemm && !a2.step_inc && ugemm_supports(a2, epi, NB, KS, 2);
    if (c->log_on) {
        double w = 0, in = 0, fl = 0;
        const double nout = (double)a.groups * a.N * (epi == EPI_GEGLU ? 2 : 1);
        for (int i = 0; i < a.nseg; ++i) {
            const Seg& sg = a.seg[i];
            w += nout * sg.C * sg.taps * 4.0;
            in += (double)batch * a.groups * sg.C * sg.Tin * 4.0;
            fl += 2.0 * batch * nout * sg.C * sg.taps * a.T;
        }
        double out = (double)batch * a.groups * a.N * a.T * 4.0;
        if (a.res_kind != RES_NONE) in += out;
        if (epi == EPI_BAND) in += 2.0 * batch * a.N * a.T * 4.0;  // this block's K and V rows
        c->stage_log.push_back({(c->use_ugemm && (tt > 1 || ugemm_supports(a, epi, NB, KS, bf) || ugemm_supports(a, epi, NB, KS))) ? 2 : 0, epi, NB, KS, w + in + out, fl});
    }
    if (c->clk_on && c->dbg_count < 64) a2.clk = c->clk_dev + (long long)c->dbg_count * 128;
    if (dbg_go(c)) {
        if (trace_on()) { fprintf(stderr, "[said] gemm #%d epi=%d NB=%d KS=%d T=%d N=%d batch=%d tt=%d\n", c->dbg_count - 1, epi, NB, KS, a.T, a.N, batch, tt); fflush(stderr); }
        const bool ug = c->use_ugemm && !a2.step_inc;
        on_ugemm = true;
        if (tt > 1) launch_ugemm(a2, epi, batch, NB, KS, s, bf, tt);
        else if (sp) launch_ugemm(a2, epi, batch, NB, KS, s, 2);
        else if (ug && bf && ugemm_supports(a2, epi, NB, KS, true)) launch_ugemm(a2, epi, batch, NB, KS, s, true);
        else if (ug && ugemm_supports(a2, epi, NB, KS)) launch_ugemm(a2, epi, batch, NB, KS, s);
        else on_ugemm = false;
        if (on_ugemm) {}
        else if (epi == EPI_QKV && a2.kv_split) c->launch_err = "q/k/v GEMM asked for pre-split k / v but does not run on ugemm_kernel";
        else launch_gemm(a2, epi, batch, NB, KS, s);
        if (trace_on()) { hipError_t e = hipStreamSynchronize(s); fprintf(stderr, "[said]   -> %s\n", hipGetErrorString(e)); fflush(stderr); }
    }
    return on_ugemm;
}
void do_attn(said_ctx* c, const AttnArgs& a, int batch, int head_dim, int KS, hipStream_t s, bool presplit = false) {
    if (c->log_on) {
        const double e = (double)batch * a.heads * head_dim * a.T;
        c->stage_log.push_back({1, -1, head_dim / 32, KS, 4.0 * e * 4.0, 4.0 * e * a.T});
    }
    if (dbg_go(c)) {
        if (trace_on()) { fprintf(stderr, "[said] attn #%d D=%d KS=%d T=%d batch=%d\n", c->dbg_count - 1, head_dim, KS, a.T, batch); fflush(stderr); }
        AttnArgs a2 = a;
        a2.b0 = c->cur_b0;
        launch_attn(a2, batch, head_dim, KS, s, c->bf16_mode ? 1 : (sp_on(c, c->attn_split) ? (presplit ? 3 : 2) : 0));
        if (trace_on()) { hipError_t e = hipStreamSynchronize(s); fprintf(stderr, "[said]   -> %s\n", hipGetErrorString(e)); fflush(stderr); }
    }
}

// ---- bf16 mode, large batches: UNet GEMMs on the token-major bf16 kernel (tgemm.hip) ------------------------------------
// The channel-major fp32 activations stay as they are between kernels (GroupNorm statistics, residuals, attention operands);
// what changes is the GEMM itself: a prep kernel applies the fused operand transform once and writes the operand token-major
// in bf16, and the GEMM runs as 128-token tiles on v_mfma_f32_32x32x16_bf16 without any split-K reduction.
bool use_tg(said_ctx* c, const UGeo& g, int nsamples) {
    // (the token-major kernels address their operands with 32-bit element offsets: beyond that the channel-major kernels run)
    const long long widest = ((long long)nsamples * rup(g.T + 2, 32) + 2) * FFI;
    if (c->band_wmax > 8) return false;   // wide alignment windows: the channel-major schedule (its generic band kernel)
    return (c->bf16_mode ? c->unet_tgemm : c->unet_fgemm) && !c->clk_on && widest < 0x7fffffffLL &&
           (long long)nsamples * g.T >= (c->bf16_mode ? c->unet_tgemm_min_tokens : (c->cur_concurrent ? std::min(c->unet_fgemm_min_tokens, c->unet_fgemm_min_concurrent) : c->unet_fgemm_min_tokens));
}
// weight of the token-major GEMM in the context's precision mode
inline const void* tw(const said_ctx* c, const void* bf, const void* f32) { return c->bf16_mode ? bf : f32; }
void do_prep(said_ctx* c, const PrepArgs& a, int batch, hipStream_t s) {
    if (c->log_on) c->stage_log.push_back({5, -2, 0, 0, (double)batch * a.C * a.T * (4.0 + (c->bf16_mode ? 2.0 : 4.0)) * (a.dst2 ? 1.5 : 1.0), 0.0});
    PrepArgs a2 = a;
    a2.f32 = c->bf16_mode ? 0 : 1;
    if (!a2.f32) a2.pack = 0;
    if (dbg_go(c) && !launch_prep(a2, batch, s)) c->launch_err = "operand preparation kernel: unsupported shape";
}
void do_tgemm(said_ctx* c, const TGemmArgs& a, int batch, hipStream_t s) {
    if (c->log_on) {
        const double eb = c->bf16_mode ? 2.0 : 4.0;   // operand element size; KS = 32 marks the fp32 kernel (fgemm_kernel) in the log
        const double out_b = a.geglu ? a.N / 2 * eb : (a.y_cm ? 4.0 * a.N * (a.res_cm ? 2 : 1) : 4.0 * a.N);
        const int tile_n = c->bf16_mode ? (a.N % 128 == 0 ? 128 : 64) : ((a.N % 128 == 0 && (a.geglu || a.N % 96)) ? 128 : 96);
        c->stage_log.push_back({4, a.geglu ? EPI_GEGLU : (a.qk ? EPI_QKV : EPI_STORE), tile_n, c->bf16_mode ? 4 : 32,
                                eb * a.N * a.K + (double)batch * a.M * (eb * a.K + out_b), 2.0 * batch * (double)a.M * a.N * a.K});
    }
    TGemmArgs a2 = a;
    a2.f32 = c->bf16_mode ? 0 : 1;
    if (a2.f32 && a2.yb) { a2.yf = reinterpret_cast<float*>(a2.yb); a2.yb = nullptr; }   // token-major intermediate (GEGLU product) in fp32
    a2.f32_split = (a2.f32 && sp_on(c, c->gemm_split)) ? 1 : 0;
    if (!a2.f32_split) a2.f32_packed = 0;
    if (dbg_go(c) && !launch_tgemm(a2, batch, s)) {
        char b[160]; snprintf(b, sizeof b, "token-major GEMM: shape M=%d N=%d K=%d (batch %d) is not served by any kernel", a.M, a.N, a.K, batch);
        c->launch_err = b;
    }
}
// per-sample row pitch of the token-major operands: a multiple of 32 that holds the T tokens plus the two Conv1d padding rows, so
// that all samples form ONE row axis for the 256-row GEMM tiles (tgemm.h: seg_rows) and 32-row MFMA tiles never straddle samples
inline int tg_rows(const UGeo& g) { return rup(g.T + 2, 32); }
PrepArgs mkprep(const UGeo& g, const float* x, int mode, void* dst, long long dst_bs, int ldd, int coff) {
    PrepArgs p;
    memset(&p, 0, sizeof p);
    p.x = x; p.x_bs = g.hs; p.pitch = g.Tp; p.T = g.T; p.C = MC; p.mode = mode;
    p.dst = dst; p.dst_bs = dst_bs; p.ldd = ldd; p.coff = coff;
    return p;
}
// GroupNorm coefficients of the operand, once per tensor (slot: one of two coefficient tables, so that the two sources of a
// concatenated input can be prepared back to back)
void prep_gn(said_ctx* c, PrepArgs& p, const UGeo& g, const float* part, int cpg, float eps, const float* gamma, const float* beta, int nb,
             int slot, hipStream_t s) {
    static const bool separate = dev_env("SAID_PREP_GN_SEPARATE") != nullptr;   // A/B: coefficients by their own launch (gn_coef_kernel)
    if (!separate) {   // finalised inside the preparation kernel from the producer's partials
        p.part = part; p.part_bs = g.sts; p.gn_cpg = cpg; p.gn_nparts = g.np; p.gn_eps = eps; p.gn_gamma = gamma; p.gn_beta = beta;
        return;
    }
    float* co = c->gn_coef + (size_t)slot * c->maxBe * 2 * MC;
    if (c->log_on) c->stage_log.push_back({5, -2, 0, 0, (double)nb * MC * g.np * 8.0, 0.0});
    if (dbg_go(c)) launch_gn_coef(part, g.sts, cpg, g.np, g.T, eps, gamma, beta, co, 2 * MC, nb, s);
    p.coef = co; p.coef_bs = 2 * MC;
}
TGemmArgs mktg(const UGeo& g, const void* a, int lda, const void* w, int N, int K) {
    TGemmArgs t;
    memset(&t, 0, sizeof t);
    t.a = a; t.a_bs = 0; t.lda = lda; t.w = w; t.M = g.T; t.N = N; t.K = K; t.seg_rows = tg_rows(g);
    return t;
}
void tg_cm_out(TGemmArgs& t, const UGeo& g, const ActBuf& out) {
    t.y_cm = out.p; t.cm_bs = g.hs; t.cm_pitch = g.Tp; t.stats = out.st; t.stats_bs = g.sts;
}

// ---- round 3, large batches: token-major activations between the kernels, operand transforms inside the GEMMs (xgemm_kernel) ----
inline int tm_seg(const UGeo& g) { return rup(g.T, 64); }   // sample pitch in tokens: a 64-row tile never straddles samples
// (32-bit element offsets inside the kernels: the widest token-major tensor here is the GEGLU product, sample pitch tm_seg — up to 4/3 of
// the rup(T + 2, 32) pitch use_tg's guard is written for)
inline bool use_tm(said_ctx* c, const UGeo& g) {
    return c->bf16_mode && c->tm_acts != 0 && use_tg(c, g, g.Be) && g.b0 == 0 && ((long long)g.Be * tm_seg(g) + 2) * FFI < 0x7fffffffLL;
}
// `rows` tokens further into a token-major tensor of row width `ld` (element size by precision mode)
inline void* tm_at(const said_ctx* c, void* base, long long rows, int ld) { return static_cast<char*>(base) + rows * ld * (c->bf16_mode ? 2 : 4); }
void do_xgemm(said_ctx* c, const TGemmArgs& a, int batch, hipStream_t s) {
    TGemmArgs a2 = a;
    a2.f32 = c->bf16_mode ? 0 : 1;
    if (c->xgemm_ntw > 0 && a2.ra[0]) a2.ntw = c->xgemm_ntw;
    a2.dbg = c->xgemm_dbg;
    if (c->xclk_on && c->dbg_count < 64) a2.clk = c->clk_dev + (long long)c->dbg_count * 128;
    if (a2.f32 && a2.yb) { a2.yf = reinterpret_cast<float*>(a2.yb); a2.yb = nullptr; }
    // kernel family: 7 = rgemm_kernel (round 4: register-stationary weights, helper waves), 6 = round 3's xgemm_kernel
    const int fam = (c->rgemm != 0 && !a2.f32 && rgemm_supports(a2, batch)) ? 7 : 6;
    if (c->log_on) {
        const double eb = c->bf16_mode ? 2.0 : 4.0;
        const double out_n = a.geglu ? a.N / 2 : a.N;
        const double out_b = a.y_cm ? 4.0 * out_n : (a.qk ? 4.0 * out_n : eb * out_n * (a.y2_tm ? 2 : 1));
        const double in_k = (a.ra[0] ? (a.ra[1] ? 384.0 : 192.0) : 0.0) + a.sk[0] + a.sk[1] + a.sk[2];   // source channels read per token (a conv reads its tile once)
        const double res_b = a.res_tm ? eb * a.N : 0.0;
        const double band_b = a.band_k ? 2.0 * 4.0 * a.N : 0.0;
        c->stage_log.push_back({fam, a.geglu ? EPI_GEGLU : (a.qk ? EPI_QKV : (a.band_k ? EPI_BAND : EPI_STORE)), a.N % 128 == 0 && (a.geglu || a.N % 96) ? 128 : 96, c->bf16_mode ? 4 : 32,
                                eb * a.N * a.K + (double)batch * a.M * (eb * in_k + out_b + res_b + band_b), 2.0 * batch * (double)a.M * a.N * a.K});
    }
    if (fam == 7) {
        if (dbg_go(c)) {
            if (!launch_rgemm(a2, batch, s)) c->launch_err = "rgemm_kernel refused a launch its own rgemm_supports() had accepted";
            ++c->n_rgemm;
        }
        return;
    }
    // a launch over a COLUMN RANGE of the packed weights (the split concatenated-input ResBlocks / folded proj_out probe only their first launch) is
    // something only rgemm_kernel understands: falling through to xgemm_kernel would silently multiply the wrong columns
    if (a2.w_k0 != 0 || (a2.w_ld != 0 && a2.w_ld != a2.K) || a2.w_seg != 0) {
        c->launch_err = "token-major GEMM over a weight column range is not served by rgemm_kernel for this shape";
        return;
    }
    ++c->n_xgemm;
    if (dbg_go(c) && !launch_xgemm(a2, batch, s)) {
        char b[160]; snprintf(b, sizeof b, "token-major activation GEMM: shape M=%d N=%d K=%d (batch %d) is not served by any kernel", a.M, a.N, a.K, batch);
        c->launch_err = b;
    }
}
TGemmArgs mkx(const UGeo& g, const void* w, int N, int K) {
    TGemmArgs t;
    memset(&t, 0, sizeof t);
    t.w = w; t.M = g.T; t.N = N; t.K = K; t.seg_rows = tm_seg(g);
    t.gn_part_bs = g.sts; t.gn_nparts = g.np; t.stats_bs = g.sts; t.ldy = MC; t.ldr_tm = MC;
    return t;
}
void run_resblock_tm(said_ctx* c, const UGeo& g, const ResW& rw, int rb_index, const ActBuf& in0, const ActBuf* in1, const ActBuf& out, hipStream_t s, bool shared) {
    const int nb = shared ? g.Bc : g.Be;
    // Round 4 (bf16 mode): a concatenated-input ResBlock (K = 1152 / 576 + 384) runs as launches of rgemm_kernel over COLUMN RANGES of the
    // same packed weights — its register-stationary weight fragments hold 576 columns per wave — with a bf16 partial sum in between (one extra
    // rounding of a quantity the next kernel rounds anyway): conv1 = [source 0 -> P] + [source 1 + bias + emb + P], and the 1x1 skip over
    // the raw input becomes the residual of conv2.  The partial / skip tensor lives in tX1 (idle until the block's SpatialTransformer).
    bool split = false;
    if (in1 && c->rgemm != 0 && c->bf16_mode && rw.has_skip) {
        TGemmArgs t = mkx(g, rw.t_conv1, MC, 3 * MC);
        t.ra[0] = in0.t; t.rmode = 1; t.rtaps = 3; t.gn_part[0] = in0.st; t.gn_cpg = rw.cin / 32; t.gn_eps = 1e-5f; t.gn_gamma = rw.g1; t.gn_beta = rw.b1;
        t.y_tm = c->tX1; t.w_ld = 3 * rw.cin; t.w_seg = rw.cin;
        split = rgemm_supports(t, nb);
    }
    if (split) {
        {   // in_layers, source 0: GN -> SiLU -> conv3 over the first 192 input channels -> P
            TGemmArgs t = mkx(g, rw.t_conv1, MC, 3 * MC);
            t.ra[0] = in0.t; t.rmode = 1; t.rtaps = 3;
            t.gn_part[0] = in0.st; t.gn_cpg = rw.cin / 32; t.gn_eps = 1e-5f; t.gn_gamma = rw.g1; t.gn_beta = rw.b1;
            t.w_ld = 3 * rw.cin; t.w_k0 = 0; t.w_seg = rw.cin;
            t.y_tm = c->tX1;
            do_xgemm(c, t, nb, s);
        }
        {   // in_layers, source 1 (the skip connection's channels) + bias + emb term + P
            TGemmArgs t = mkx(g, rw.t_conv1, MC, 3 * MC);
            t.ra[0] = in1->t; t.rmode = 1; t.rtaps = 3;
            t.gn_part[0] = in1->st; t.gn_cpg = rw.cin / 32; t.gn_eps = 1e-5f; t.gn_gamma = rw.g1 + MC; t.gn_beta = rw.b1 + MC;
            t.w_ld = 3 * rw.cin; t.w_k0 = MC; t.w_seg = rw.cin;
            t.bias = rw.conv1.bias;
            t.emb = c->EO + (long long)rb_index * MC * c->maxNp; t.emb_pitch = c->maxNp; t.step_ptr = g.step_ptr; t.emb_b_stride = g.emb_b_stride;
            t.res_tm = c->tX1;
            t.y_tm = c->M.t; t.stats = c->M.st;
            do_xgemm(c, t, nb, s);
        }
        {   // skip_connection: Conv1d 1x1 over the concatenated raw input (openaimodel.py:194) -> tX1
            TGemmArgs t = mkx(g, rw.t_conv2, MC, 2 * MC);
            t.sa[0] = in0.t; t.sld[0] = MC; t.sk[0] = MC;
            t.sa[1] = in1->t; t.sld[1] = MC; t.sk[1] = MC;
            t.w_ld = 3 * MC + 2 * MC; t.w_k0 = 3 * MC;
            t.y_tm = c->tX1;
            do_xgemm(c, t, nb, s);
        }
        {   // out_layers: GN -> SiLU -> conv3 + (both biases) + skip(x)
            TGemmArgs t = mkx(g, rw.t_conv2, MC, 3 * MC);
            t.ra[0] = c->M.t; t.rmode = 1; t.rtaps = 3;
            t.gn_part[0] = c->M.st; t.gn_cpg = 6; t.gn_eps = 1e-5f; t.gn_gamma = rw.g2; t.gn_beta = rw.b2;
            t.w_ld = 3 * MC + 2 * MC; t.w_k0 = 0;
            t.bias = rw.bias2;
            t.res_tm = c->tX1;
            t.y_tm = out.t; t.stats = out.st;
            do_xgemm(c, t, nb, s);
        }
        return;
    }
    {   // in_layers: GN -> SiLU -> conv3 + emb term   (openaimodel.py:205-225)
        TGemmArgs t = mkx(g, tw(c, rw.t_conv1, rw.tf_conv1), MC, 3 * rw.cin);
        t.ra[0] = in0.t; t.ra[1] = in1 ? in1->t : nullptr; t.rmode = 1; t.rtaps = 3;
        t.gn_part[0] = in0.st; t.gn_part[1] = in1 ? in1->st : nullptr; t.gn_cpg = rw.cin / 32; t.gn_eps = 1e-5f; t.gn_gamma = rw.g1; t.gn_beta = rw.b1;
        t.bias = rw.conv1.bias;
        t.emb = c->EO + (long long)rb_index * MC * c->maxNp; t.emb_pitch = c->maxNp; t.step_ptr = g.step_ptr; t.emb_b_stride = g.emb_b_stride;
        t.y_tm = c->M.t; t.stats = c->M.st;
        do_xgemm(c, t, nb, s);
    }
    {   // out_layers: GN -> SiLU -> conv3 ; + skip(x)   (openaimodel.py:226-227)
        TGemmArgs t = mkx(g, tw(c, rw.t_conv2, rw.tf_conv2), MC, 3 * MC + (rw.has_skip ? 2 * MC : 0));
        t.ra[0] = c->M.t; t.rmode = 1; t.rtaps = 3;
        t.gn_part[0] = c->M.st; t.gn_cpg = 6; t.gn_eps = 1e-5f; t.gn_gamma = rw.g2; t.gn_beta = rw.b2;
        if (rw.has_skip) {   // 1x1 conv over the concatenated raw input: two streamed K segments behind the resident one
            t.sa[0] = in0.t; t.sld[0] = MC; t.sk[0] = MC;
            t.sa[1] = in1->t; t.sld[1] = MC; t.sk[1] = MC;
            t.bias = rw.bias2;
        } else {
            t.bias = rw.conv2.bias;
            t.res_tm = in0.t;
        }
        t.y_tm = out.t; t.stats = out.st;
        if (shared) { t.y2_tm = out.t; t.y2_row_off = (long long)g.Bc * tm_seg(g); }
        do_xgemm(c, t, nb, s);
    }
}
// last: this block feeds the `out` convolution, which reads channel-major fp32 + GroupNorm partials (out_sched.hip)
void run_transformer_tm(said_ctx* c, const UGeo& g, const STW& sw, int blk, const ActBuf& in, const ActBuf& out, hipStream_t s, bool shared, bool last) {
    const int n1 = shared ? g.Bc : g.Be;
    const int n2 = g.Bc > 0 ? g.Bc : g.Be;
    const int x_off = (g.Bc > 0 && !shared) ? g.Bc : 0;
    const int kv_off = g.Bc > 0 ? g.Bc : 0;
    const long long seg = tm_seg(g);
    const int vt_rows = rup(g.T, 32);
    bool battn = false;
    {   // x = norm(x); q, k, v = to_{q,k,v}(norm1(x)) into attn.hip's operand layout   (attention.py:227, 168, 93-97)
        TGemmArgs t = mkx(g, tw(c, sw.t_qkv, sw.tf_qkv), 3 * MC, MC);
        t.ra[0] = in.t; t.rmode = 3; t.rtaps = 1;
        t.gn_part[0] = in.st; t.gn_cpg = 6; t.gn_eps = 1e-6f; t.gn_gamma = sw.gn_g; t.gn_beta = sw.gn_b;
        t.ln_gamma = sw.l1g; t.ln_beta = sw.l1b;
        t.qk = c->QK; t.vt = c->VT; t.v_bs = (long long)MC * g.Tp; t.qk_n = 2 * MC; t.head_dim = HD; t.rows = vt_rows; t.heads2 = 2 * HEADS; t.v_pitch = g.Tp;
        // round 4 (bf16 mode, T <= 640): the projection writes bf16 q / k / v in battn_kernel's operand layout (attn.hip)
        AttnArgs pa;
        pa.qk = c->QK; pa.v = c->VT; pa.o = static_cast<float*>(c->tO); pa.v_bstride = (long long)MC * g.Tp; pa.o_bstride = seg;
        pa.pitch = g.Tp; pa.T = g.T; pa.heads = HEADS; pa.rows = vt_rows; pa.b0 = 0; pa.scale = 0.17677669529663687f;
        t.qkv_bf16 = (c->battn != 0 && c->bf16_mode && c->rgemm != 0 && battn_supports(pa, HD)) ? 1 : 0;
        if (t.qkv_bf16 && !rgemm_supports(t, n1)) t.qkv_bf16 = 0;
        battn = t.qkv_bf16 != 0;
        t.q_scale = 0.17677669529663687f * 1.4426950408889634f;   // dim_head ** -0.5 (attention.py:101) x log2(e)
        do_xgemm(c, t, n1, s);
    }
    {   // softmax(q k^T * scale) v -> token-major   (attention.py:99-126)
        AttnArgs a;
        a.qk = c->QK; a.v = c->VT; a.o = static_cast<float*>(c->tO);
        a.v_bstride = (long long)MC * g.Tp; a.o_bstride = seg; a.o_mode = c->bf16_mode ? 2 : 1;
        a.pitch = g.Tp; a.T = g.T; a.heads = HEADS; a.rows = vt_rows; a.b0 = 0;
        a.scale = 0.17677669529663687f;
        if (battn) {
            if (c->log_on) { const double e = (double)n1 * HEADS * HD * g.T; c->stage_log.push_back({9, -1, 1, 1, 4.0 * e * 2.0, 4.0 * e * g.T}); }
            if (dbg_go(c)) launch_battn(a, n1, s, c->battn == 4 ? 4 : (c->battn == 10 ? 10 : 8));
        } else {
            do_attn(c, a, n1, HD, -4, s);
        }
    }
    // round 5: everything behind the attention as ONE launch on bf16 operands (stchain_kernel<true>; the token-major bf16 tensors are its operands as they are)
    if (c->bf16_mode && c->st_chain_bf16 != 0 && !last && sw.chain_wb && sw.chain_vec && c->band_chain_ok && c->kvt_S == g.S && c->kvt_bf16 && c->cur_b0 == 0 && g.S == c->band_S && g.T == c->band_T &&
        (long long)g.Be * seg * MC < 0x7fffffffLL) {
        ChainArgs ca;
        memset(&ca, 0, sizeof ca);
        ca.wstream = static_cast<const float*>(sw.chain_wb); ca.vec = sw.chain_vec;
        ca.xin_part = in.st; ca.part_bs = g.sts; ca.gn_gamma = sw.gn_g; ca.gn_beta = sw.gn_b;
        ca.kvt = c->KVT; ca.kvt_bs = (long long)g.S * (NST * 2 * MC); ca.lo = c->band_lo; ca.hi = c->band_hi;
        ca.y = static_cast<float*>(out.t); ca.y_bs = seg * MC; ca.stats_out = out.st; ca.stats_bs = g.sts;
        ca.S = g.S; ca.np = g.np; ca.koff = blk * 2 * MC; ca.wmax = c->band_wmax; ca.scale = 0.17677669529663687f;
        if (c->log_on) {
            const double w = 2.0 * ((double)3 * MC * MC + 2.0 * FFI * MC + (double)(FFI + MC) * MC);
            const double io = 2.0 * MC * g.T * ((double)n1 * 2 + g.Be) + 4.0 * 2 * MC * g.T * (double)n2;
            const double fl = 2.0 * g.T * ((double)n1 * MC * MC + (double)n2 * 2 * MC * MC + (double)g.Be * (2.0 * FFI * MC + (double)(FFI + MC) * MC)) + 4.0 * n2 * MC * g.T * c->band_wmax;
            c->stage_log.push_back({10, EPI_STORE, 6, 8, w + io, fl});
        }
        if (c->xclk_on && c->dbg_count < 64) ca.clk = c->clk_dev + (long long)c->dbg_count * 128;   // (said_debug_option "xgemm_clk": -DSAID_CLK_STAMPS builds)
        if (dbg_go(c)) {
            launch_stchain(ca, static_cast<const float*>(c->tO), static_cast<const float*>(in.t), g.T, g.Tp, seg * MC, seg * MC, shared ? g.Bc : 0, g.Bc > 0 ? g.Bc : 0, g.Be, s, true);
            ++c->n_stchain;
        }
        return;
    }
    {   // x1 = to_out(attn) + GroupNorm(x_in)   (attention.py:127, 168); under guidance also x2 of the unconditional half = x1 + c2
        TGemmArgs t = mkx(g, tw(c, sw.t_out1, sw.tf_out1), MC, MC);
        t.sa[0] = c->tO; t.sld[0] = MC; t.sk[0] = MC;
        t.bias = sw.out1.bias;
        t.res_tm = in.t; t.res_gn = 1; t.res_part = in.st; t.res_gamma = sw.gn_g; t.res_beta = sw.gn_b; t.res_eps = 1e-6f; t.gn_cpg = 6;
        t.y_tm = c->tX1;
        if (g.Bc > 0) { t.y2_tm = c->tX2; t.y2_row_off = 0; t.y2_add = c->c2[blk]; }
        do_xgemm(c, t, n1, s);
    }
    {   // attn2: q = to_q(norm2(x1)); banded softmax over the precomputed audio K/V   (attention.py:170-191)
        TGemmArgs t = mkx(g, tw(c, sw.t_q2, sw.tf_q2), MC, MC);
        t.ra[0] = tm_at(c, c->tX1, x_off * seg, MC); t.rmode = 2; t.rtaps = 1;
        t.ln_gamma = sw.l2g; t.ln_beta = sw.l2b;
        const long long kvbs = (long long)NST * 2 * MC * g.Sp;
        t.band_k = c->KV + (long long)(blk * 2 * MC) * g.Sp + (long long)kv_off * kvbs;
        t.band_v = c->KV + (long long)(blk * 2 * MC + MC) * g.Sp + (long long)kv_off * kvbs;
        t.band_kv_bs = kvbs; t.band_kv_pitch = g.Sp; t.band_lo = c->band_lo; t.band_hi = c->band_hi; t.band_wmax = c->band_wmax;
        t.band_scale = 0.17677669529663687f;
        t.y_tm = tm_at(c, c->tO, x_off * seg, MC);
        do_xgemm(c, t, n2, s);
    }
    {   // x2 = to_out(attn2) + x1   (conditional half only under guidance: its rows are [Bc, 2 Bc) of X2)
        TGemmArgs t = mkx(g, tw(c, sw.t_out2, sw.tf_out2), MC, MC);
        t.sa[0] = tm_at(c, c->tO, x_off * seg, MC); t.sld[0] = MC; t.sk[0] = MC;
        t.bias = sw.out2.bias;
        t.res_tm = tm_at(c, c->tX1, x_off * seg, MC);
        t.y_tm = tm_at(c, c->tX2, kv_off * seg, MC);
        do_xgemm(c, t, n2, s);
    }
    {   // GEGLU: proj(norm3(x2)) -> a * gelu(gate)   (attention.py:25-32)
        TGemmArgs t = mkx(g, tw(c, sw.t_ff1, sw.tf_ff1), 2 * FFI, MC);
        t.ra[0] = c->tX2; t.rmode = 2; t.rtaps = 1;
        t.ln_gamma = sw.l3g; t.ln_beta = sw.l3b;
        t.bias = sw.t_ff1_bias; t.geglu = 1;
        t.yb = c->tF; t.y_bs = seg * FFI; t.ldy = FFI;
        do_xgemm(c, t, g.Be, s);
    }
    bool split = false;
    if (!last && c->rgemm != 0 && c->bf16_mode) {   // round 4: as two rgemm launches over column ranges of the folded weight (see run_resblock_tm)
        TGemmArgs t = mkx(g, sw.t_ffproj, MC, FFI);
        t.sa[0] = c->tF; t.sld[0] = FFI; t.sk[0] = FFI; t.w_ld = FFI + MC;
        t.res_tm = in.t; t.y_tm = c->tX1;
        split = rgemm_supports(t, g.Be);
    }
    if (split) {
        {   // (P F2) h + x_in -> tX1 (x1 is dead by now)
            TGemmArgs t = mkx(g, sw.t_ffproj, MC, FFI);
            t.sa[0] = c->tF; t.sld[0] = FFI; t.sk[0] = FFI; t.w_ld = FFI + MC; t.w_k0 = 0;
            t.res_tm = in.t;
            t.y_tm = c->tX1;
            do_xgemm(c, t, g.Be, s);
        }
        {   // + P x2 + (P b2 + bp)
            TGemmArgs t = mkx(g, sw.t_ffproj, MC, MC);
            t.sa[0] = c->tX2; t.sld[0] = MC; t.sk[0] = MC; t.w_ld = FFI + MC; t.w_k0 = FFI;
            t.bias = sw.ffproj.bias;
            t.res_tm = c->tX1;
            t.y_tm = out.t; t.stats = out.st;
            do_xgemm(c, t, g.Be, s);
        }
        return;
    }
    {   // proj_out o ff.net.2 over [h ; x2] + x_in   (attention.py:193, 232-234)
        TGemmArgs t = mkx(g, tw(c, sw.t_ffproj, sw.tf_ffproj), MC, FFI + MC);
        t.sa[0] = c->tF; t.sld[0] = FFI; t.sk[0] = FFI;
        t.sa[1] = c->tX2; t.sld[1] = MC; t.sk[1] = MC;
        t.bias = sw.ffproj.bias;
        t.res_tm = in.t;
        if (last) { t.y_cm = out.p; t.cm_bs = g.hs; t.cm_pitch = g.Tp; t.stats = out.st; }
        else { t.y_tm = out.t; t.stats = out.st; }
        do_xgemm(c, t, g.Be, s);
    }
}

// ---- round 3, bf16 mode at large batch (default): the HYBRID SpatialTransformer.  The ResBlocks and q/k/v keep round 2's kernels
// (channel-major fp32 between them, prep_kernel + token-major GEMM), everything from the attention output on runs on round 3's
// token-major-activation kernels, where they are faster in situ: attention writes token-major bf16, attn1.to_out / band / attn2.to_out /
// GEGLU / folded proj_out are xgemm launches (no second preparation kernel: GEGLU applies norm3 itself), and the folded proj_out writes
// channel-major fp32 + GroupNorm partials again for the next ResBlock.  The token-major copy of the block's input (the residual of
// attn1.to_out and of proj_out) is a by-product of the q/k/v operand preparation.  8 launches per block instead of 9.
void run_transformer_hybrid(said_ctx* c, const UGeo& g, const STW& sw, int blk, const ActBuf& in, const ActBuf& out, hipStream_t s, bool shared) {
    const int n1 = shared ? g.Bc : g.Be;
    const int n2 = g.Bc > 0 ? g.Bc : g.Be;
    const int x_off = (g.Bc > 0 && !shared) ? g.Bc : 0;
    const int kv_off = g.Bc > 0 ? g.Bc : 0;
    const long long seg = tm_seg(g);
    const int vt_rows = rup(g.T, 32);
    {   // q, k, v on round 2's token-major GEMM; the preparation kernel also leaves the raw input token-major (in.t)
        PrepArgs p = mkprep(g, in.p, 1, c->uPL, (long long)tg_rows(g) * MC, MC, 0);
        prep_gn(c, p, g, in.st, 6, 1e-6f, sw.gn_g, sw.gn_b, g.Be, 0, s);
        p.ln_gamma = sw.l1g; p.ln_beta = sw.l1b;
        p.dst2 = in.t; p.dst2_bs = seg * MC; p.ldd2 = MC; p.coff2 = 0;
        do_prep(c, p, g.Be, s);   // (all samples: the second half's rows are the residual of proj_out)
        TGemmArgs t = mktg(g, c->uPL, MC, tw(c, sw.t_qkv, sw.tf_qkv), 3 * MC, MC);
        t.qk = c->QK; t.vt = c->VT; t.v_bs = (long long)MC * g.Tp; t.qk_n = 2 * MC; t.head_dim = HD; t.rows = vt_rows; t.heads2 = 2 * HEADS; t.v_pitch = g.Tp;
        do_tgemm(c, t, n1, s);
    }
    {   // softmax(q k^T * scale) v -> token-major   (attention.py:99-126)
        AttnArgs a;
        a.qk = c->QK; a.v = c->VT; a.o = static_cast<float*>(c->tO);
        a.v_bstride = (long long)MC * g.Tp; a.o_bstride = seg; a.o_mode = c->bf16_mode ? 2 : 1;
        a.pitch = g.Tp; a.T = g.T; a.heads = HEADS; a.rows = vt_rows; a.b0 = 0;
        a.scale = 0.17677669529663687f;
        do_attn(c, a, n1, HD, -4, s);
    }
    {   // x1 = to_out(attn) + GroupNorm(x_in); under guidance also x2 of the unconditional half = x1 + c2
        TGemmArgs t = mkx(g, tw(c, sw.t_out1, sw.tf_out1), MC, MC);
        t.sa[0] = c->tO; t.sld[0] = MC; t.sk[0] = MC;
        t.bias = sw.out1.bias;
        t.res_tm = in.t; t.res_gn = 1; t.res_part = in.st; t.res_gamma = sw.gn_g; t.res_beta = sw.gn_b; t.res_eps = 1e-6f; t.gn_cpg = 6;
        t.y_tm = c->tX1;
        if (g.Bc > 0) { t.y2_tm = c->tX2; t.y2_row_off = 0; t.y2_add = c->c2[blk]; }
        do_xgemm(c, t, n1, s);
    }
    {   // attn2: q = to_q(norm2(x1)); banded softmax over the precomputed audio K/V
        TGemmArgs t = mkx(g, tw(c, sw.t_q2, sw.tf_q2), MC, MC);
        t.ra[0] = tm_at(c, c->tX1, x_off * seg, MC); t.rmode = 2; t.rtaps = 1;
        t.ln_gamma = sw.l2g; t.ln_beta = sw.l2b;
        const long long kvbs = (long long)NST * 2 * MC * g.Sp;
        t.band_k = c->KV + (long long)(blk * 2 * MC) * g.Sp + (long long)kv_off * kvbs;
        t.band_v = c->KV + (long long)(blk * 2 * MC + MC) * g.Sp + (long long)kv_off * kvbs;
        t.band_kv_bs = kvbs; t.band_kv_pitch = g.Sp; t.band_lo = c->band_lo; t.band_hi = c->band_hi; t.band_wmax = c->band_wmax;
        t.band_scale = 0.17677669529663687f;
        t.y_tm = tm_at(c, c->tO, x_off * seg, MC);
        do_xgemm(c, t, n2, s);
    }
    {   // x2 = to_out(attn2) + x1   (conditional half only under guidance: its rows are [Bc, 2 Bc) of X2)
        TGemmArgs t = mkx(g, tw(c, sw.t_out2, sw.tf_out2), MC, MC);
        t.sa[0] = tm_at(c, c->tO, x_off * seg, MC); t.sld[0] = MC; t.sk[0] = MC;
        t.bias = sw.out2.bias;
        t.res_tm = tm_at(c, c->tX1, x_off * seg, MC);
        t.y_tm = tm_at(c, c->tX2, kv_off * seg, MC);
        do_xgemm(c, t, n2, s);
    }
    {   // GEGLU: proj(norm3(x2)) -> a * gelu(gate): the GEMM normalises its operand itself
        TGemmArgs t = mkx(g, tw(c, sw.t_ff1, sw.tf_ff1), 2 * FFI, MC);
        t.ra[0] = c->tX2; t.rmode = 2; t.rtaps = 1;
        t.ln_gamma = sw.l3g; t.ln_beta = sw.l3b;
        t.bias = sw.t_ff1_bias; t.geglu = 1;
        t.yb = c->tF; t.y_bs = seg * FFI; t.ldy = FFI;
        do_xgemm(c, t, g.Be, s);
    }
    {   // proj_out o ff.net.2 over [h ; x2] + x_in -> channel-major fp32 + GroupNorm partials for the next ResBlock
        TGemmArgs t = mkx(g, tw(c, sw.t_ffproj, sw.tf_ffproj), MC, FFI + MC);
        t.sa[0] = c->tF; t.sld[0] = FFI; t.sk[0] = FFI;
        t.sa[1] = c->tX2; t.sld[1] = MC; t.sk[1] = MC;
        t.bias = sw.ffproj.bias;
        t.res_tm = in.t;
        t.y_cm = out.p; t.cm_bs = g.hs; t.cm_pitch = g.Tp; t.stats = out.st;
        do_xgemm(c, t, g.Be, s);
    }
}

// shared: guidance-shared prefix — only the first g.Bc samples are computed, and the result is ALSO written into the
// conditional half's slots (values only; its statistics are consumed by kernels that run on the first half alone)
void run_resblock(said_ctx* c, const UGeo& g, const ResW& rw, int rb_index, const ActBuf& in0, const ActBuf* in1, const ActBuf& out, hipStream_t s,
                  bool shared = false) {
    const int cpg = rw.cin / 32;
    const int nb = shared ? g.Bc : g.Be;
    const long long tt = (long long)nb * ((g.T + 31) / 32);
    if (use_tg(c, g, nb)) {
        const long long T2 = tg_rows(g);
        // fp32 mode: the operands reach fgemm_kernel already split (packed h | l pairs: prep_kernel's pack mode + the packed weight copies)
        const int pk = (!c->bf16_mode && sp_on(c, c->gemm_split) && c->gemm_presplit != 0 && rw.tp_conv1 && rw.tp_conv2) ? 1 : 0;
        {   // in_layers: GN -> SiLU -> conv3 + emb term   (openaimodel.py:205-225)
            PrepArgs p = mkprep(g, in0.p, 0, c->uPA, T2 * rw.cin, rw.cin, 0);
            p.pack = pk;
            prep_gn(c, p, g, in0.st, cpg, 1e-5f, rw.g1, rw.b1, nb, 0, s);
            if (rw.has_skip) { p.dst2 = c->uPB; p.dst2_bs = T2 * 2 * MC; p.ldd2 = 2 * MC; p.coff2 = 0; }   // raw copy for the 1x1 skip conv
            do_prep(c, p, nb, s);
            if (in1) {
                PrepArgs q = mkprep(g, in1->p, 0, c->uPA, T2 * rw.cin, rw.cin, MC);
                q.pack = pk;
                prep_gn(c, q, g, in1->st, cpg, 1e-5f, rw.g1 + MC, rw.b1 + MC, nb, 1, s);
                if (rw.has_skip) { q.dst2 = c->uPB; q.dst2_bs = T2 * 2 * MC; q.ldd2 = 2 * MC; q.coff2 = MC; }
                do_prep(c, q, nb, s);
            }
            TGemmArgs t = mktg(g, c->uPA, rw.cin, pk ? rw.tp_conv1 : tw(c, rw.t_conv1, rw.tf_conv1), MC, 3 * rw.cin);
            t.f32_packed = pk;
            t.bias = rw.conv1.bias;
            t.emb = c->EO + (long long)rb_index * MC * c->maxNp; t.emb_pitch = c->maxNp; t.step_ptr = g.step_ptr; t.emb_b_stride = g.emb_b_stride;
            tg_cm_out(t, g, c->M);
            do_tgemm(c, t, nb, s);
        }
        {   // out_layers: GN -> SiLU -> conv3 ; + skip(x)   (openaimodel.py:226-227)
            PrepArgs p = mkprep(g, c->M.p, 0, c->uPA, T2 * MC, MC, 0);
            p.pack = pk;
            prep_gn(c, p, g, c->M.st, 6, 1e-5f, rw.g2, rw.b2, nb, 0, s);
            do_prep(c, p, nb, s);
            TGemmArgs t = mktg(g, c->uPA, MC, pk ? rw.tp_conv2 : tw(c, rw.t_conv2, rw.tf_conv2), MC, 3 * MC);
            t.f32_packed = pk;
            if (rw.has_skip) {   // 1x1 conv over the concatenated raw input as a second K segment
                // (the raw copies of the two inputs were written into uPB by the in_layers operand preparation above)
                t.a2 = c->uPB; t.a2_bs = 0; t.lda2 = 2 * MC; t.K1 = 3 * MC; t.K = 5 * MC;
                t.bias = rw.bias2;
            } else {
                t.bias = rw.conv2.bias;
                t.res_cm = in0.p; t.res_cm_bs = g.hs;
            }
            tg_cm_out(t, g, out);
            if (shared) { t.y2_cm = out.p + (long long)g.Bc * g.hs; t.y2_bs = g.hs; }
            do_tgemm(c, t, nb, s);
        }
        return;
    }
    {   // in_layers: GN -> SiLU -> conv3 ; + emb_layers(emb)   (openaimodel.py:205-225)
        GemmArgs a = mkargs(g.T, MC);
        a.nseg = in1 ? 2 : 1;
        a.seg[0] = with_pw(mkseg(in0.p, g.hs, g.Tp, MC, 3, 1, 1, g.T, XF_GN_SILU, rw.conv1.w[0]), rw.conv1, 0);
        seg_gn(a.seg[0], in0.st, g.sts, cpg, g.np, 1e-5f, rw.g1, rw.b1);
        if (in1) {
            a.seg[1] = with_pw(mkseg(in1->p, g.hs, g.Tp, MC, 3, 1, 1, g.T, XF_GN_SILU, rw.conv1.w[1]), rw.conv1, 1);
            seg_gn(a.seg[1], in1->st, g.sts, cpg, g.np, 1e-5f, rw.g1 + MC, rw.b1 + MC);
        }
        a.bias = rw.conv1.bias;
        a.emb = c->EO + (long long)rb_index * MC * c->maxNp; a.emb_pitch = c->maxNp; a.step_ptr = g.step_ptr; a.emb_b_stride = g.emb_b_stride;
        a.y = c->M.p; a.y_bstride = g.hs; a.y_pitch = g.Tp;
        a.stats_out = c->M.st; a.stats_bstride = g.sts;
        const LaunchCfg lc = pick_unet(c, tt);
        do_gemm(c, a, EPI_STORE, nb, lc.NB, lc.KS, s);
    }
    {   // out_layers: GN -> SiLU -> conv3 ; + skip(x)   (openaimodel.py:226-227)
        GemmArgs a = mkargs(g.T, MC);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->M.p, g.hs, g.Tp, MC, 3, 1, 1, g.T, XF_GN_SILU, rw.conv2.w[0]), rw.conv2, 0);
        seg_gn(a.seg[0], c->M.st, g.sts, 6, g.np, 1e-5f, rw.g2, rw.b2);
        if (rw.has_skip) {  // 1x1 conv over the concatenated input folded in as two extra K segments
            a.seg[1] = with_pw(mkseg(in0.p, g.hs, g.Tp, MC, 1, 0, 1, g.T, XF_NONE, rw.skip.w[0]), rw.skip, 0);
            a.seg[2] = with_pw(mkseg(in1->p, g.hs, g.Tp, MC, 1, 0, 1, g.T, XF_NONE, rw.skip.w[1]), rw.skip, 1);
            a.nseg = 3;
            a.bias = rw.bias2;
        } else {
            a.bias = rw.conv2.bias;
            a.res_kind = RES_PLAIN; a.res = in0.p; a.res_bstride = g.hs; a.res_pitch = g.Tp;
        }
        a.y = out.p; a.y_bstride = g.hs; a.y_pitch = g.Tp;
        a.stats_out = out.st; a.stats_bstride = g.sts;
        if (shared) { a.y2 = out.p + (long long)g.Bc * g.hs; a.y2_bstride = g.hs; a.y2_add = nullptr; }
        const LaunchCfg lc = pick_unet(c, tt);
        do_gemm(c, a, EPI_STORE, nb, lc.NB, lc.KS, s);
    }
}

// shared: guidance-shared prefix (first transformer under classifier-free guidance): self-attention and everything
// before it run once per clip on samples [0, g.Bc).  With g.Bc > 0 the cross-attention (q projection, band softmax,
// to_out) runs on the conditional half only — the unconditional half attends to one repeated key/value, so its
// attn2 output is the per-channel constant c2[blk] and attn1's to_out emits its x2 = x1 + c2 directly.
void run_transformer(said_ctx* c, const UGeo& g, const STW& sw, int blk, const ActBuf& in, const ActBuf& out, hipStream_t s,
                     bool shared = false) {
    const int n1 = shared ? g.Bc : g.Be;          // samples through self-attention
    const int n2 = g.Bc > 0 ? g.Bc : g.Be;        // samples through cross-attention
    // first sample of the cross-attention range in the activations (X1, O): the conditional half, unless the prefix is
    // shared (then X1 exists once per clip, in slots [0, Bc)); its K/V are always the conditional half's
    const int x_off = (g.Bc > 0 && !shared) ? g.Bc : 0;
    const int kv_off = g.Bc > 0 ? g.Bc : 0;
    const long long tt1 = (long long)n1 * ((g.T + 31) / 32), tt2 = (long long)n2 * ((g.T + 31) / 32);
    const long long tt = (long long)g.Be * ((g.T + 31) / 32);
    const bool big = big_cgemm() && tt * 6 > 1536;
    const bool big_qkv = tt1 * 6 > 1536 && dev_env("SAID_NO_MT") && !c->bf16_mode;   // without multi-tile workgroups the generic NB=6 shape wins in fp32
    const int vt_rows = rup(g.T, 32);
    const long long obs = 2LL * MC * g.Tp;   // batch stride of O (shared with QK so attention uses one stride)
    const bool tg = use_tg(c, g, n1);
    if (c->hybrid && c->bf16_mode && tg && use_tg(c, g, g.Be) && g.b0 == 0) { run_transformer_hybrid(c, g, sw, blk, in, out, s, shared); return; }
    // fp32 mode: attn1.to_out on the token-major fp32 GEMM too (the attention kernel writes its operand token-major into the free q/k/v
    // operand buffer; the GroupNorm'ed residual uses the coefficients the q/k/v preparation finalised): 59 -> ~30 us per launch at Be = 64
    // fp32 mode: everything behind the self-attention as ONE launch (stchain.hip) — at small batches beside the channel-major GEMMs, at large ones (st_chain_large)
    // beside the token-major GEMMs' q / k / v (the attention kernel then writes channel-major, as the small-batch schedule has it)
    const bool chain = !c->bf16_mode && sp_on(c, c->st_chain) && c->use_ugemm && sw.chain_w && sw.chain_vec && c->band_chain_ok && c->kvt_S == g.S && !c->kvt_bf16 && c->cur_b0 == 0 && !c->use_branches &&
                       ((!tg && !use_tg(c, g, g.Be)) || c->st_chain_large) && tt <= c->st_chain_max_tiles && g.S == c->band_S && g.T == c->band_T;
    const bool out1_tm = tg && !chain && !c->bf16_mode && c->f32_out1_tm && tt1 * HEADS >= 2048 && sw.tf_out1;
    bool presplit = false;   // k and v stored as packed split-fp16 pairs for attn_kernel<PM = 3> (see below)
    bool coef_ready = false; // the q/k/v GEMM left the block input's GroupNorm coefficients in gn_coef (small-batch ugemm_kernel only)
    if (tg) {   // q, k, v on the bf16 token-major GEMM: operand = LayerNorm(GroupNorm(x)) prepared once
        const int pk = (!c->bf16_mode && sp_on(c, c->gemm_split) && c->gemm_presplit != 0 && sw.tp_qkv) ? 1 : 0;   // operands arrive split (see run_resblock)
        PrepArgs p = mkprep(g, in.p, 1, c->uPL, (long long)tg_rows(g) * MC, MC, 0);
        p.pack = pk;
        prep_gn(c, p, g, in.st, 6, 1e-6f, sw.gn_g, sw.gn_b, n1, 0, s);
        p.ln_gamma = sw.l1g; p.ln_beta = sw.l1b;
        if ((out1_tm || (chain && c->chain_coef != 0)) && p.part) { p.coef_out = c->gn_coef; p.coef_out_bs = 2 * MC; coef_ready = chain; }   // (the fp32 fused tail reads them too: round 6)
        do_prep(c, p, n1, s);
        TGemmArgs t = mktg(g, c->uPL, MC, pk ? sw.tp_qkv : tw(c, sw.t_qkv, sw.tf_qkv), 3 * MC, MC);
        t.f32_packed = pk;
        t.qk = c->QK; t.vt = c->VT; t.v_bs = (long long)MC * g.Tp; t.qk_n = 2 * MC; t.head_dim = HD; t.rows = vt_rows; t.heads2 = 2 * HEADS; t.v_pitch = g.Tp;
        // round 6: this GEMM's epilogue stores k and v pre-split too (as ugemm_kernel's does at small batch), for every attention shape that unpacks them: the
        // key-split ones and the four-query-tile one of large batches (each wave of which used to split all of K and V for itself)
        presplit = !c->bf16_mode && sp_on(c, c->attn_split) && c->attn_presplit != 0 && sp_on(c, c->gemm_split) && !c->attn_ks_force && !dev_env("SAID_ATTN_KS") &&
                   !dev_env("SAID_NO_ATTN_QW") && !c->clk_on;   // (then the attention shape below is -4, 8 or 4: all unpack)
        t.kv_pack = presplit ? 1 : 0;
        do_tgemm(c, t, n1, s);
    } else
    {   // x = norm(x) (GroupNorm eps 1e-6); q,k,v = to_{q,k,v}(norm1(x))   (attention.py:227, 168, 93-97)
        GemmArgs a = mkargs(g.T, 3 * MC);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(in.p, g.hs, g.Tp, MC, 1, 0, 1, g.T, XF_GN_LN, sw.qkv.w[0]), sw.qkv, 0);
        seg_gn(a.seg[0], in.st, g.sts, 6, g.np, 1e-6f, sw.gn_g, sw.gn_b);
        a.seg[0].ln_gamma = sw.l1g; a.seg[0].ln_beta = sw.l1b; a.seg[0].ln_eps = 1e-5f;
        // q and k tiles (0..11) token-major into QK [Be][2*heads][rows][32]; v tiles channel-major into VT [Be][192][Tp]
        // (y is biased so that output channel n = 384 + c lands on row c)
        a.tm_tiles = 2 * MC / 32;
        a.vt = c->QK; a.vt_heads = 2 * HEADS; a.vt_dim = HD; a.vt_rows = vt_rows;
        a.y = c->VT - (long long)a.tm_tiles * 32 * g.Tp; a.y_bstride = (long long)MC * g.Tp; a.y_pitch = g.Tp;
        // tiles per workgroup: the largest shape that still gives every CU a workgroup in ONE round (at Be=2, T=600:
        // NB=3 -> 228 workgroups, 27.5 -> 13.8 us per launch against NB=1's 684 workgroups in 2.7 rounds)
        static const int qkv_env = dev_env("SAID_QKV_NB") ? atoi(dev_env("SAID_QKV_NB")) : 0;
        int qkv_nb = qkv_env ? qkv_env : (tt1 * 6 >= 192 ? 3 : (tt1 * 9 >= 192 ? 2 : 1));
        if (!qkv_env && c->unet_nb_model && tt1 * 18 <= 1024) {   // pick_unet's busiest-CU model over the 18 column tiles (19 tiles: NB 1 -> 2, 342 -> 171 workgroups)
            double cost = 1e30;
            for (int nb = 1; nb <= 3; ++nb) {
                const double k = (double)((tt1 * (18 / nb) + 255) / 256) * (0.45 + nb);
                if (k < cost - 1e-9) { cost = k; qkv_nb = nb; }
            }
        }
        const LaunchCfg lc = big_qkv ? LaunchCfg{6, 4} : LaunchCfg{qkv_nb, 8};
        // k and v pre-split for the key-split attention shapes (same rule as below), when this GEMM runs on ugemm_kernel (the only epilogue that packs)
        {
            const bool key_split = (c->attn_ks_force == 4 || c->attn_ks_force == 8) || (!(tt1 * HEADS >= 2048) && !(tt1 * HEADS > 8192) && !dev_env("SAID_ATTN_KS") && !dev_env("SAID_NO_ATTN_QW") && !c->attn_ks_force);
            GemmArgs probe = a;
            probe.b0 = c->cur_b0;
            presplit = !c->bf16_mode && sp_on(c, c->attn_split) && c->attn_presplit != 0 && key_split && c->use_ugemm && !big_qkv && !c->clk_on &&
                       (ugemm_supports(probe, EPI_QKV, lc.NB, lc.KS, 2) || ugemm_supports(probe, EPI_QKV, lc.NB, lc.KS));
            a.kv_split = presplit ? 1 : 0;
        }
        // round 6: this GEMM finalises the block input's GroupNorm coefficients anyway (its operand is LayerNorm(GroupNorm(x))): the first workgroup of every sample
        // leaves them for stchain_kernel, whose GroupNorm'ed residual then needs 2 loads per wave instead of 23 and no finalisation in front of its first barrier
        if (chain && c->chain_coef != 0 && c->gn_coef) { a.gn_coef_out = c->gn_coef; a.gn_coef_bs = 2 * MC; }
        coef_ready = do_gemm(c, a, EPI_QKV, n1, lc.NB, lc.KS, s) && a.gn_coef_out != nullptr;
    }
    bool out1_done = false;
    {   // softmax(q k^T * scale) v   (attention.py:99-126)
        AttnArgs a;
        a.qk = c->QK; a.v = c->VT; a.o = c->O;
        a.v_bstride = (long long)MC * g.Tp; a.o_bstride = obs;
        a.pitch = g.Tp; a.T = g.T; a.heads = HEADS; a.rows = vt_rows; a.b0 = 0;
        a.scale = 0.17677669529663687f;  // 32 ** -0.5
        static const int attn_ks_env0 = dev_env("SAID_ATTN_KS") ? atoi(dev_env("SAID_ATTN_KS")) : 0;   // experiment knob
        const int attn_ks_env = c->attn_ks_force ? c->attn_ks_force : attn_ks_env0;
        // waves per workgroup = ways the key tiles are split: 8 only pays while a wave would otherwise hold a single
        // tile (T <= 256); from there 4 waves with ~5 tiles each merge half as many partial states (B=1: -0.5 % per step)
        // large batches: four query tiles per workgroup sharing each K / V tile through the CU's L1 (-4), see attn.hip
        static const bool no_qw = dev_env("SAID_NO_ATTN_QW") != nullptr;
        const int attn_ks = (!no_qw && tt1 * HEADS >= 2048) ? -4 : ((tt1 * HEADS > 8192) ? 1 : ((g.T <= 256 && tt1 * HEADS <= 2048) ? 8 : 4));
        if (out1_tm && !attn_ks_env && attn_ks == -4) { a.o = static_cast<float*>(c->uPL); a.o_bstride = tg_rows(g); a.o_mode = 1; }
        int ks_run = attn_ks_env ? attn_ks_env : attn_ks;
        // long sequences at small batch (configs[4]): three query tiles per wave share every K / V fragment (attn2q.hip; bit-identical) where a third of the
        // launch is most of ONE round of workgroups (101 KB of LDS, 264 registers: one workgroup per CU)
        if (presplit && ks_run == 4 && a.o_mode == 0 && (c->attn_2q > 0 || (c->attn_2q < 0 && tt1 * HEADS >= 512 && (long long)n1 * HEADS * ((((g.T + 31) / 32) + 2) / 3) <= 256))) ks_run = 34;   // (one round of one workgroup per CU)
        do_attn(c, a, n1, HD, ks_run, s, presplit && (ks_run == 4 || ks_run == 8 || ks_run == -4 || ks_run == 34));
        out1_done = a.o_mode == 1;
    }
    if (chain && !out1_done) {
        ChainArgs ca;
        memset(&ca, 0, sizeof ca);
        ca.wstream = sw.chain_w; ca.vec = sw.chain_vec;
        ca.xin_part = in.st; ca.part_bs = g.sts; ca.gn_gamma = sw.gn_g; ca.gn_beta = sw.gn_b;
        ca.kvt = c->KVT; ca.kvt_bs = (long long)g.S * (NST * 2 * MC); ca.lo = c->band_lo; ca.hi = c->band_hi;
        ca.y = out.p; ca.y_bs = g.hs; ca.stats_out = out.st; ca.stats_bs = g.sts;
        ca.S = g.S; ca.np = g.np; ca.koff = blk * 2 * MC; ca.wmax = c->band_wmax; ca.scale = 0.17677669529663687f;
        // small launches: three workgroups per token tile, each streaming a third of the GEGLU / folded proj_out weights (38 workgroups on 256 CUs were bound by
        // one CU's L2 port each: 2.36 MB per workgroup) — while the launch still is one round of the chip
        ca.slices = 1;
        if (c->st_chain_slices != 1 && sw.chain_w3 && sw.chain_w2 && !c->st_chain_dbg) {
            if (tt <= CHAIN3_MAX_TILES && c->st_chain_slices != 2) ca.slices = 3;
            else if (tt <= CHAIN2_MAX_TILES) ca.slices = 2;
        }
        if (ca.slices > 1) { ca.wstream = ca.slices == 3 ? sw.chain_w3 : sw.chain_w2; ca.part = c->chain_part; ca.ticket = c->chain_ticket; }
        // the fp32 kernels read the block input's GroupNorm coefficients (round 6): left by the q/k/v GEMM (ugemm_kernel) or its operand preparation (prep_kernel), else made here
        if (!coef_ready && dbg_go_peek(c)) launch_gn_coef(in.st, g.sts, 6, g.np, g.T, 1e-6f, sw.gn_g, sw.gn_b, c->gn_coef, 2 * MC, n1, s);
        ca.gn_coef = c->gn_coef; ca.coef_bs = 2 * MC;
        if (c->clk_on && c->dbg_count < 64) ca.clk = c->clk_dev + (long long)c->dbg_count * 128;
        if (c->st_chain_dbg) { ca.dbg_x1 = c->X1; ca.dbg_x2 = c->X2; ca.dbg_o2 = c->X3; }
        if (c->log_on) {
            const double w = 4.0 * ((double)3 * MC * MC + 2.0 * FFI * MC + (double)(FFI + MC) * MC);
            const double io = 4.0 * MC * g.T * ((double)n1 * 2 + g.Be) + 4.0 * 2 * MC * g.T * (double)n2;
            const double fl = 2.0 * g.T * ((double)n1 * MC * MC + (double)n2 * 2 * MC * MC + (double)g.Be * (2.0 * FFI * MC + (double)(FFI + MC) * MC)) + 4.0 * n2 * MC * g.T * c->band_wmax;
            c->stage_log.push_back({10, EPI_STORE, 6, 8, w + io, fl});
        }
        if (dbg_go(c)) {
            if (trace_on()) { fprintf(stderr, "[said] stchain #%d T=%d samples=%d shared=%d\n", c->dbg_count - 1, g.T, g.Be, (int)shared); fflush(stderr); }
            launch_stchain(ca, c->O, in.p, g.T, g.Tp, obs, g.hs, shared ? g.Bc : 0, g.Bc > 0 ? g.Bc : 0, g.Be, s, false);
            ++c->n_stchain;
            if (trace_on()) { hipError_t e = hipStreamSynchronize(s); fprintf(stderr, "[said]   -> %s\n", hipGetErrorString(e)); fflush(stderr); }
        }
        return;
    }
    if (out1_done) {   // x1 = to_out(attn) + GroupNorm(x_in) on fgemm_kernel; under guidance also x2 of the unconditional half = x1 + c2
        TGemmArgs t = mktg(g, c->uPL, MC, sw.tf_out1, MC, MC);
        t.bias = sw.out1.bias;
        t.res_cm = in.p; t.res_cm_bs = g.hs; t.res_cm_coef = c->gn_coef; t.res_cm_coef_bs = 2 * MC;
        t.y_cm = c->X1; t.cm_bs = g.hs; t.cm_pitch = g.Tp;
        if (g.Bc > 0) { t.y2_cm = c->X2; t.y2_bs = g.hs; t.y2_add_cm = c->c2[blk]; }
        do_tgemm(c, t, n1, s);
    } else
    {   // x1 = to_out(attn) + x, with x = GroupNorm(in) recomputed on the fly   (attention.py:127, 168)
        GemmArgs a = mkargs(g.T, MC);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->O, obs, g.Tp, MC, 1, 0, 1, g.T, XF_NONE, sw.out1.w[0]), sw.out1, 0);
        a.bias = sw.out1.bias;
        a.res_kind = RES_GN; a.res = in.p; a.res_bstride = g.hs; a.res_pitch = g.Tp;
        a.res_gn_part = in.st; a.res_gn_part_bstride = g.sts; a.res_gn_cpg = 6; a.res_gn_nparts = g.np; a.res_gn_eps = 1e-6f;
        a.res_gn_gamma = sw.gn_g; a.res_gn_beta = sw.gn_b;
        a.y = c->X1; a.y_bstride = g.hs; a.y_pitch = g.Tp;
        if (g.Bc > 0) {   // x2 of the unconditional half: x1 + const (written for every sample of this launch; the
                          // conditional slots — when this launch covers them — are overwritten by attn2's to_out below)
            a.y2 = c->X2; a.y2_bstride = g.hs; a.y2_add = c->c2[blk];
        }
        const LaunchCfg lc = big_cgemm() ? pick_unet(c, tt1) : LaunchCfg{1, 8};   // the GroupNorm'ed-residual variant exists for NB = 1
        do_gemm(c, a, EPI_STORE, n1, lc.NB, lc.KS, s);
    }
    {   // attn2: q = to_q(norm2(x1)); banded softmax over the precomputed audio K/V   (attention.py:170-191)
        GemmArgs a = mkargs(g.T, MC);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->X1 + (long long)x_off * g.hs, g.hs, g.Tp, MC, 1, 0, 1, g.T, XF_LN, sw.q2.w[0]), sw.q2, 0);
        a.seg[0].ln_gamma = sw.l2g; a.seg[0].ln_beta = sw.l2b; a.seg[0].ln_eps = 1e-5f;
        a.y = c->O + (long long)x_off * obs; a.y_bstride = obs; a.y_pitch = g.Tp;
        const long long kvbs = (long long)NST * 2 * MC * g.Sp;
        a.band.k = c->KV + (long long)(blk * 2 * MC) * g.Sp + (long long)kv_off * kvbs;
        a.band.v = c->KV + (long long)(blk * 2 * MC + MC) * g.Sp + (long long)kv_off * kvbs;
        a.band.kv_bstride = kvbs; a.band.kv_pitch = g.Sp;
        a.band.lo = c->band_lo; a.band.hi = c->band_hi; a.band.wmax = c->band_wmax; a.band.scale = 0.17677669529663687f;
        if (c->band_wmax > 8) {   // windows wider than the fused epilogue's eight keys (S >> T through SAID.forward): plain q projection, then the
                                  // generic band kernel in place (misc.hip; never on SAID.inference's path)
            const BandArgs bd = a.band;
            memset(&a.band, 0, sizeof a.band);
            do_gemm(c, a, EPI_STORE, n2, 1, 8, s);
            if (c->log_on) c->stage_log.push_back({5, -2, 0, 0, 0.0, 0.0});
            if (dbg_go(c)) launch_band_wide(a.y, obs, g.Tp, bd.k, bd.v, bd.kv_bstride, bd.kv_pitch, bd.lo, bd.hi, g.T, HEADS, n2, bd.scale, s);
        } else
        do_gemm(c, a, EPI_BAND, n2, 1, big ? 4 : 8, s);
    }
    {   // x2 = to_out(attn2) + x1   (conditional half only under guidance: its slots are [Bc, 2 Bc) of X2)
        GemmArgs a = mkargs(g.T, MC);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->O + (long long)x_off * obs, obs, g.Tp, MC, 1, 0, 1, g.T, XF_NONE, sw.out2.w[0]), sw.out2, 0);
        a.bias = sw.out2.bias;
        a.res_kind = RES_PLAIN; a.res = c->X1 + (long long)x_off * g.hs; a.res_bstride = g.hs; a.res_pitch = g.Tp;
        a.y = c->X2 + (long long)kv_off * g.hs; a.y_bstride = g.hs; a.y_pitch = g.Tp;
        const LaunchCfg lc = pick_unet(c, tt2);
        do_gemm(c, a, EPI_STORE, n2, lc.NB, lc.KS, s);
    }
    if (use_tg(c, g, g.Be)) {
        {   // GEGLU: operand norm3(x2) (and raw x2 for the folded proj_out), value/gate pairs multiplied in the epilogue
            const long long P = tg_rows(g);
            PrepArgs p = mkprep(g, c->X2, 2, c->uPL, P * MC, MC, 0);
            p.ln_gamma = sw.l3g; p.ln_beta = sw.l3b;
            p.dst2 = c->uPX; p.dst2_bs = P * MC; p.ldd2 = MC; p.coff2 = 0;
            do_prep(c, p, g.Be, s);
            TGemmArgs t = mktg(g, c->uPL, MC, tw(c, sw.t_ff1, sw.tf_ff1), 2 * FFI, MC);
            t.bias = sw.t_ff1_bias; t.geglu = 1;
            t.yb = c->uPH; t.y_bs = P * FFI; t.ldy = FFI;
            do_tgemm(c, t, g.Be, s);
        }
        {   // proj_out o ff.net.2 over [h ; x2] + x_in, channel-major result + GroupNorm partials
            TGemmArgs t = mktg(g, c->uPH, FFI, tw(c, sw.t_ffproj, sw.tf_ffproj), MC, FFI + MC);
            t.a2 = c->uPX; t.a2_bs = 0; t.lda2 = MC; t.K1 = FFI;
            t.bias = sw.ffproj.bias;
            t.res_cm = in.p; t.res_cm_bs = g.hs;
            tg_cm_out(t, g, out);
            do_tgemm(c, t, g.Be, s);
        }
        return;
    }
    {   // GEGLU: proj(norm3(x2)) -> a * gelu(gate)   (attention.py:25-32)
        GemmArgs a = mkargs(g.T, FFI);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->X2, g.hs, g.Tp, MC, 1, 0, 1, g.T, XF_LN, sw.ff1.w[0]), sw.ff1, 0);
        a.seg[0].ln_gamma = sw.l3g; a.seg[0].ln_beta = sw.l3b; a.seg[0].ln_eps = 1e-5f;
        a.bias = sw.ff1.bias; a.geglu_gate_tiles = FFI / 32;
        a.y = c->F; a.y_bstride = (long long)FFI * g.Tp; a.y_pitch = g.Tp;
        static const int geglu_env = dev_env("SAID_GEGLU_NB") ? atoi(dev_env("SAID_GEGLU_NB")) : 0;
        const int geglu_nb = geglu_env ? geglu_env : (tt * 6 >= 192 ? 4 : (tt * 12 >= 192 ? 2 : 1));   // one round of workgroups, as for qkv
        do_gemm(c, a, EPI_GEGLU, g.Be, big ? 3 : geglu_nb, big ? 4 : 8, s);
    }
    static const bool no_fold = dev_env("SAID_NO_FFPROJ_FOLD") != nullptr;   // A/B knob: the two unfused launches
    if (!no_fold) {
        // x3 = net.2(h) + x2 and out = proj_out(x3) + x_in as ONE GEMM over the K segments [h ; x2] with the host-folded
        // weights (P F2 | P) — both maps are per-token linear, so nothing but rounding order changes
        GemmArgs a = mkargs(g.T, MC);
        a.nseg = 2;
        a.seg[0] = with_pw(mkseg(c->F, (long long)FFI * g.Tp, g.Tp, FFI, 1, 0, 1, g.T, XF_NONE, sw.ffproj.w[0]), sw.ffproj, 0);
        a.seg[1] = with_pw(mkseg(c->X2, g.hs, g.Tp, MC, 1, 0, 1, g.T, XF_NONE, sw.ffproj.w[1]), sw.ffproj, 1);
        a.bias = sw.ffproj.bias;
        a.res_kind = RES_PLAIN; a.res = in.p; a.res_bstride = g.hs; a.res_pitch = g.Tp;
        a.y = out.p; a.y_bstride = g.hs; a.y_pitch = g.Tp;
        a.stats_out = out.st; a.stats_bstride = g.sts;
        const LaunchCfg lc = pick_unet(c, tt);
        do_gemm(c, a, EPI_STORE, g.Be, lc.NB, lc.KS, s);
        return;
    }
    {   // x3 = net.2(h) + x2
        GemmArgs a = mkargs(g.T, MC);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->F, (long long)FFI * g.Tp, g.Tp, FFI, 1, 0, 1, g.T, XF_NONE, sw.ff2.w[0]), sw.ff2, 0);
        a.bias = sw.ff2.bias;
        a.res_kind = RES_PLAIN; a.res = c->X2; a.res_bstride = g.hs; a.res_pitch = g.Tp;
        a.y = c->X3; a.y_bstride = g.hs; a.y_pitch = g.Tp;
        const LaunchCfg lc = pick_unet(c, tt);
        do_gemm(c, a, EPI_STORE, g.Be, lc.NB, lc.KS, s);
    }
    {   // proj_out (1x1 conv) + x_in   (attention.py:232-234)
        GemmArgs a = mkargs(g.T, MC);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->X3, g.hs, g.Tp, MC, 1, 0, 1, g.T, XF_NONE, sw.proj.w[0]), sw.proj, 0);
        a.bias = sw.proj.bias;
        a.res_kind = RES_PLAIN; a.res = in.p; a.res_bstride = g.hs; a.res_pitch = g.Tp;
        a.y = out.p; a.y_bstride = g.hs; a.y_pitch = g.Tp;
        a.stats_out = out.st; a.stats_bstride = g.sts;
        const LaunchCfg lc = pick_unet(c, tt);
        do_gemm(c, a, EPI_STORE, g.Be, lc.NB, lc.KS, s);
    }
}

// UNetModel.forward (openaimodel.py:677-709): x_cm (latents) -> eps_cm.  Needs KV, band tables and EO ready.
void run_unet(said_ctx* c, const UGeo& g, hipStream_t s) {
    c->cur_b0 = g.b0;
    const long long tt = (long long)g.Be * ((g.T + 31) / 32);
    const int in_copies = (g.B_lat > 0 && g.Be % g.B_lat == 0) ? g.Be / g.B_lat : 0;   // samples sharing one clip's latents
    static const bool no_conv_in = dev_env("SAID_NO_CONV_IN") != nullptr;
    bool conv_in_tm = false;
    if (!no_conv_in && g.b0 == 0 && in_copies >= 1 && c->conv_in.w4[0] && !c->clk_on &&
        conv_in_supports(c->cin, MC, c->conv_in.taps, g.T, g.Tp, in_copies)) {
        // input_blocks.0: Conv1d(32 -> 192, k3), computed once per clip (conv_in.hip)
        // (large batches in bf16 mode: straight into the token-major bf16 layout the persistent GEMMs read — no channel-major copy, no transposition)
        conv_in_tm = c->bf16_mode && use_tm(c, g) && tm_seg(g) <= 0xffff;
        if (dbg_go(c)) {
            if (conv_in_tm) launch_conv_in_tm(c->x_cm, c->conv_in.w4[0], c->conv_in.bias, c->H0.t, tm_seg(g), c->H0.st, g.step_inc, g.B_lat, in_copies, g.T, g.Tp, s);
            else launch_conv_in(c->x_cm, c->conv_in.w4[0], c->conv_in.bias, c->H0.p, c->H0.st, g.step_inc, g.B_lat, in_copies, g.T, g.Tp, MC, s);
        }
        if (c->log_on)
            c->stage_log.push_back({11 /* conv_in_kernel */, EPI_STORE, 1, 4, 4.0 * ((double)MC * c->cin * 3 + (double)g.B_lat * c->cin * g.T + (double)g.Be * MC * g.T),
                                    2.0 * g.B_lat * MC * c->cin * 3 * g.T});
    } else {   // input_blocks.0: Conv1d(32 -> 192, k3)
        GemmArgs a = mkargs(g.T, MC);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->x_cm, (long long)c->cin * g.Tp, g.Tp, c->cin, 3, 1, 1, g.T, XF_NONE, c->conv_in.w[0]), c->conv_in, 0);
        a.seg[0].b_mod = g.B_lat;
        a.step_inc = g.step_inc;   // the loop's device step counter is advanced by the first kernel of the step
        a.bias = c->conv_in.bias;
        a.y = c->H0.p; a.y_bstride = g.hs; a.y_pitch = g.Tp; a.stats_out = c->H0.st; a.stats_bstride = g.sts;
        const LaunchCfg lc = pick_unet(c, tt);
        do_gemm(c, a, EPI_STORE, g.Be, lc.NB, lc.KS, s);
    }
    const bool sh = g.Bc > 0;   // guidance-shared prefix: the two halves first differ at input_blocks.1.1's cross-attention
    // round 4: the loop's last kernel reads the token-major bf16 hidden state itself (out_sched.hip: out_sched_tm_kernel)
    OutSchedArgs osa_tm;
    bool out_tm = false;
    if (g.out_sched && c->out_tm != 0 && c->bf16_mode && c->rgemm != 0 && use_tm(c, g) && c->bw_out) {
        osa_tm = *g.out_sched;
        osa_tm.x_tm = c->P.t; osa_tm.wb = c->bw_out; osa_tm.seg = tm_seg(g);
        out_tm = out_sched_tm_supports(osa_tm);
    }
    if (use_tm(c, g)) {
        if (!conv_in_tm) {   // conv_in's result (channel-major fp32 + GroupNorm partials) -> token-major, raw
            PrepArgs p = mkprep(g, c->H0.p, 3, c->H0.t, (long long)tm_seg(g) * MC, MC, 0);
            do_prep(c, p, g.Be, s);
        }
        run_resblock_tm(c, g, c->res[0], 0, c->H0, nullptr, c->P, s, sh);            // input_blocks.1.0
        run_transformer_tm(c, g, c->st[0], 0, c->P, c->H1, s, sh, false);           // input_blocks.1.1
        run_resblock_tm(c, g, c->res[1], 1, c->H1, nullptr, c->P, s, false);         // middle_block.0
        run_transformer_tm(c, g, c->st[1], 1, c->P, c->Q, s, false, false);         // middle_block.1
        run_resblock_tm(c, g, c->res[2], 2, c->Q, nullptr, c->P, s, false);          // middle_block.2
        run_resblock_tm(c, g, c->res[3], 3, c->P, &c->H1, c->Q, s, false);           // output_blocks.0.0  cat([h, H1])
        run_transformer_tm(c, g, c->st[2], 2, c->Q, c->P, s, false, false);         // output_blocks.0.1
        run_resblock_tm(c, g, c->res[4], 4, c->P, &c->H0, c->Q, s, false);           // output_blocks.1.0  cat([h, H0])
        run_transformer_tm(c, g, c->st[3], 3, c->Q, c->P, s, false, !out_tm);       // output_blocks.1.1 (-> channel-major for out_sched_kernel unless out_tm)
    } else {
    run_resblock(c, g, c->res[0], 0, c->H0, nullptr, c->P, s, sh);   // input_blocks.1.0
    run_transformer(c, g, c->st[0], 0, c->P, c->H1, s, sh);          // input_blocks.1.1   (hs: H0, H1)
    run_resblock(c, g, c->res[1], 1, c->H1, nullptr, c->P, s);       // middle_block.0
    run_transformer(c, g, c->st[1], 1, c->P, c->Q, s);               // middle_block.1
    run_resblock(c, g, c->res[2], 2, c->Q, nullptr, c->P, s);        // middle_block.2
    run_resblock(c, g, c->res[3], 3, c->P, &c->H1, c->Q, s);         // output_blocks.0.0  cat([h, H1])
    run_transformer(c, g, c->st[2], 2, c->Q, c->P, s);               // output_blocks.0.1
    run_resblock(c, g, c->res[4], 4, c->P, &c->H0, c->Q, s);         // output_blocks.1.0  cat([h, H0])
    run_transformer(c, g, c->st[3], 3, c->Q, c->P, s);               // output_blocks.1.1
    }
    if (g.out_sched) {   // out conv + guidance + DDIM update in one kernel (out_sched.hip)
        if (dbg_go(c)) { if (out_tm) launch_out_sched_tm(osa_tm, s); else launch_out_sched(*g.out_sched, s); }
    } else {   // out: GN -> SiLU -> Conv1d(192 -> 32, k3)
        GemmArgs a = mkargs(g.T, c->cin);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->P.p, g.hs, g.Tp, MC, 3, 1, 1, g.T, XF_GN_SILU, c->conv_out.w[0]), c->conv_out, 0);
        seg_gn(a.seg[0], c->P.st, g.sts, 6, g.np, 1e-5f, c->out_g, c->out_b);
        a.bias = c->conv_out.bias;
        a.y = c->eps_cm; a.y_bstride = (long long)c->cin * g.Tp; a.y_pitch = g.Tp;
        do_gemm(c, a, EPI_STORE, g.Be, 1, 8, s);
        if (c->log_on && !c->stage_log.empty()) c->stage_log.back().kind = 12;   // the `out` convolution alone (forward(), profiling): in the loop it is out_sched_kernel's first half
    }
}

// time_embed + the five emb_layers for `n` timesteps already in ts_dev -> EO [5*192][Np]
void run_time_embed(said_ctx* c, int n, hipStream_t s) {
    const int Np = c->maxNp;
    launch_timestep_embedding(c->ts_dev, c->freqs, c->E0, n, MC, Np, s);
    const long long tt = (n + 31) / 32;
    {
        GemmArgs a = mkargs(n, TE);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->E0, 0, Np, MC, 1, 0, 1, n, XF_NONE, c->te1.w[0]), c->te1, 0);
        a.bias = c->te1.bias; a.act = ACT_SILU;
        a.y = c->E1; a.y_pitch = Np;
        const LaunchCfg lc = pick_cfg(tt, TE / 32);
        launch_gemm(a, EPI_STORE, 1, lc.NB, lc.KS, s);
    }
    {
        GemmArgs a = mkargs(n, TE);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->E1, 0, Np, TE, 1, 0, 1, n, XF_NONE, c->te2.w[0]), c->te2, 0);
        a.bias = c->te2.bias;
        a.y = c->E2; a.y_pitch = Np;
        const LaunchCfg lc = pick_cfg(tt, TE / 32);
        launch_gemm(a, EPI_STORE, 1, lc.NB, lc.KS, s);
    }
    {
        GemmArgs a = mkargs(n, NRES * MC);
        a.nseg = 1;
        a.seg[0] = with_pw(mkseg(c->E2, 0, Np, TE, 1, 0, 1, n, XF_SILU, c->emb_all.w[0]), c->emb_all, 0);
        a.bias = c->emb_all.bias;
        a.y = c->EO; a.y_pitch = Np;
        const LaunchCfg lc = pick_cfg(tt, NRES * MC / 32);
        launch_gemm(a, EPI_STORE, 1, lc.NB, lc.KS, s);
    }
}

// cross-attention K/V of all four transformer blocks from the channel-major context (step-invariant)
// for samples [b0, b0 + nb): under guidance with the shared schedule only the conditional half's are ever read
void run_kv(said_ctx* c, int b0, int nb, int S, int Sp, hipStream_t s) {
    GemmArgs a = mkargs(S, NST * 2 * MC);
    a.nseg = 1;
    const long long cbs = (long long)c->ctx_dim * Sp, ybs = (long long)NST * 2 * MC * Sp;
    a.seg[0] = with_pw(mkseg(c->CTX + b0 * cbs, cbs, Sp, c->ctx_dim, 1, 0, 1, S, XF_NONE, c->kv_all.w[0]), c->kv_all, 0);
    a.y = c->KV + b0 * ybs; a.y_bstride = ybs; a.y_pitch = Sp;
    const LaunchCfg lc = pick_cfg((long long)nb * ((S + 31) / 32), NST * 2 * MC / 32);
    launch_gemm(a, EPI_STORE, nb, lc.NB, lc.KS, s);
    // the key-major copy the fused SpatialTransformer tail reads its window tiles from (once per loop; fp32 mode's small-batch schedule only)
    c->kvt_S = -1;
    if ((c->bf16_mode ? c->st_chain_bf16 != 0 : sp_on(c, c->st_chain)) && (long long)S * NST * 2 * MC <= (long long)NST * 2 * MC * c->maxTp)
    {
        c->kvt_S = S; c->kvt_bf16 = c->bf16_mode ? 1 : 0;
        // (bf16 mode: the copy itself is bf16 — the kernels round the window tiles to bf16 anyway, and at 32 clips the windows are 64 of a launch's 117 MB of HBM traffic in fp32)
        if (c->bf16_mode) launch_cm_to_tm_bf16(c->KV + b0 * ybs, ybs, Sp, reinterpret_cast<unsigned short*>(c->KVT) + (long long)b0 * S * (NST * 2 * MC), (long long)S * (NST * 2 * MC), nb, S, NST * 2 * MC, s);
        else launch_cm_to_tm(c->KV + b0 * ybs, c->KVT + (long long)b0 * S * (NST * 2 * MC), nb, S, NST * 2 * MC, Sp, ybs, s);
    }
}

// alignment band of ldm/attention.py:170-189 with Python's banker's rounding on doubles
int set_band(said_ctx* ctx, int T, int S, hipStream_t s) {
    if (ctx->band_T == T && ctx->band_S == S) return 0;
    std::vector<int> lo(T), hi(T);
    const double ratio = (double)S / (double)T, kh = ratio / 2 + 1;
    int wmax = 0;
    for (int i = 0; i < T; ++i) {
        const double mid = (i + 0.5) * ratio;
        const int a = std::max((int)std::nearbyint(mid - kh), 0);
        const int b = std::min((int)std::nearbyint(mid + kh), S);
        lo[i] = a; hi[i] = b;
        wmax = std::max(wmax, b - a);
        if (b <= a) return fail(ctx, "empty alignment window at query %d (T=%d, S=%d)", i, T, S);
    }
    // work queued earlier on the caller's stream may still read the previous tables: drain it before they change
    // (only happens when the clip length changes)
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(ctx->band_lo, lo.data(), T * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->band_hi, hi.data(), T * sizeof(int), hipMemcpyHostToDevice));
    ctx->band_T = T; ctx->band_S = S; ctx->band_wmax = wmax;
    {   // stchain.hip keeps the windows of a 32-token tile as CHAIN_KW key rows starting at lo[first token]: needs a non-decreasing band of that span
        bool ok = wmax <= 8;
        for (int i = 1; i < T && ok; ++i) ok = lo[i] >= lo[i - 1];
        for (int t0 = 0; t0 < T && ok; t0 += 32) {
            int hmax = 0;
            for (int i = t0; i < std::min(T, t0 + 32); ++i) hmax = std::max(hmax, hi[i]);
            ok = hmax - lo[t0] <= CHAIN_KW;
        }
        ctx->band_chain_ok = ok;
    }
    return 0;
}

UGeo make_geo(said_ctx* c, int Be, int B_lat, int T, int S) {
    UGeo g;
    g.Be = Be; g.B_lat = B_lat; g.T = T; g.Tp = rup(T, 32); g.np = (T + 31) / 32; g.S = S; g.Sp = rup(S, 32);
    g.hs = (long long)MC * g.Tp; g.sts = (long long)MC * g.np * 2;
    g.step_ptr = nullptr; g.emb_b_stride = 0; g.b0 = 0; g.step_inc = nullptr; g.out_sched = nullptr; g.Bc = 0;
    return g;
}

// The workspace of a context: every buffer whose size depends on (max_batch_eff, max_frames).  Allocated by said_create and
// replaced as a whole by said_reserve; weights, packed tables and the lazily grown audio / noise buffers are not touched.
int alloc_workspace(said_ctx* ctx, int max_batch_eff, int max_frames) {
    ctx->maxBe = max_batch_eff; ctx->maxT = max_frames;
    ctx->maxTp = rup(max_frames, 32);
    ctx->maxNp = rup(std::max(1024, max_batch_eff), 32);
    ctx->alloc_list = &ctx->ws_allocs;
    const size_t Be = max_batch_eff, Tp = ctx->maxTp, np = Tp / 32;
    const int ctx_dim = ctx->ctx_dim;
    int rc = 0;
    const size_t act = Be * MC * Tp, stt = Be * MC * np * 2;
    rc |= dalloc(ctx, &ctx->x_cm, Be * 32 * Tp);
    rc |= dalloc(ctx, &ctx->eps_cm, Be * 32 * Tp);
    for (ActBuf* a : {&ctx->H0, &ctx->H1, &ctx->P, &ctx->Q, &ctx->M}) { rc |= dalloc(ctx, &a->p, act); rc |= dalloc(ctx, &a->st, stt); }
    rc |= dalloc(ctx, &ctx->X1, act); rc |= dalloc(ctx, &ctx->X2, act); rc |= dalloc(ctx, &ctx->X3, act);
    rc |= dalloc(ctx, &ctx->O, 2 * act);  // same batch stride as QK so attention can share one stride
    rc |= dalloc(ctx, &ctx->QK, 2 * act);
    rc |= dalloc(ctx, &ctx->VT, Be * HEADS * Tp * HD);
    rc |= dalloc(ctx, &ctx->F, Be * FFI * Tp);
    rc |= dalloc(ctx, &ctx->KV, Be * NST * 2 * MC * Tp);
    rc |= dalloc(ctx, &ctx->KVT, Be * NST * 2 * MC * Tp);
    rc |= dalloc(ctx, &ctx->chain_part, (size_t)256 * 6 * 16 * 64);   // (85 x 3 or 128 x 2 workgroups)
    rc |= dalloc(ctx, &ctx->CTX, Be * (size_t)ctx_dim * Tp);
    const size_t Np = ctx->maxNp;
    rc |= dalloc(ctx, &ctx->E0, MC * Np); rc |= dalloc(ctx, &ctx->E1, TE * Np); rc |= dalloc(ctx, &ctx->E2, TE * Np);
    rc |= dalloc(ctx, &ctx->EO, NRES * MC * Np);
    rc |= dalloc(ctx, &ctx->ts_dev, Np); rc |= dalloc(ctx, &ctx->coef_dev, Np * 8);
    rc |= dalloc(ctx, &ctx->axpby_coef, 2 * Np);
    rc |= dalloc(ctx, &ctx->band_lo, Tp); rc |= dalloc(ctx, &ctx->band_hi, Tp);
    rc |= dalloc(ctx, &ctx->init_cm, Be * 32 * Tp); rc |= dalloc(ctx, &ctx->enoise_cm, Be * 32 * Tp); rc |= dalloc(ctx, &ctx->mask_cm, Be * 32 * Tp);
    rc |= dalloc(ctx, &ctx->rescale_part, Be * 2 * 64 * 3);
    {   // operand buffers of the large-batch token-major path (sized for fp32 elements; zero-initialised, so padding rows start at 0)
        const size_t Tm = (size_t)rup(max_frames + 2, 32);   // tg_rows(): per-sample row pitch
        rc |= dalloc(ctx, reinterpret_cast<uint16_t**>(&ctx->uPA), 2 * (Be * Tm * 2 * MC + 4096));
        rc |= dalloc(ctx, reinterpret_cast<uint16_t**>(&ctx->uPB), 2 * (Be * Tm * 2 * MC + 4096));
        rc |= dalloc(ctx, reinterpret_cast<uint16_t**>(&ctx->uPL), 2 * (Be * Tm * MC + 4096));
        rc |= dalloc(ctx, reinterpret_cast<uint16_t**>(&ctx->uPH), 2 * (Be * Tm * FFI + 4096));
        rc |= dalloc(ctx, reinterpret_cast<uint16_t**>(&ctx->uPX), 2 * (Be * Tm * MC + 4096));
        rc |= dalloc(ctx, &ctx->gn_coef, 2 * Be * 2 * MC);
    }
    {   // token-major activations of the round-3 large-batch path: sample pitch = frames rounded up to 64 tokens, fp32-sized
        const size_t seg = (size_t)rup(max_frames, 64), tm = Be * seg * MC + 4096;
        for (ActBuf* a : {&ctx->H0, &ctx->H1, &ctx->P, &ctx->Q, &ctx->M}) rc |= dalloc(ctx, reinterpret_cast<float**>(&a->t), tm);
        rc |= dalloc(ctx, reinterpret_cast<float**>(&ctx->tX1), tm);
        rc |= dalloc(ctx, reinterpret_cast<float**>(&ctx->tX2), tm);
        rc |= dalloc(ctx, reinterpret_cast<float**>(&ctx->tO), tm);
        rc |= dalloc(ctx, reinterpret_cast<float**>(&ctx->tF), Be * seg * FFI + 4096);
    }
    ctx->alloc_list = &ctx->allocs;
    ctx->band_T = ctx->band_S = -1;   // the band tables are part of the workspace
    ctx->kvt_S = -1;                  // ... and so is the key-major K / V copy
    return rc;
}

void drop_graphs(said_ctx* ctx) {
    if (ctx->gexec) { (void)hipGraphExecDestroy(ctx->gexec); ctx->gexec = nullptr; }
    if (ctx->graph) { (void)hipGraphDestroy(ctx->graph); ctx->graph = nullptr; }
    if (ctx->gexec_rem) { (void)hipGraphExecDestroy(ctx->gexec_rem); ctx->gexec_rem = nullptr; }
    if (ctx->graph_rem) { (void)hipGraphDestroy(ctx->graph_rem); ctx->graph_rem = nullptr; }
    ctx->gkey.clear();
    ctx->gnodes = 0;
}

int check_ready(said_ctx* ctx) {
    if (!ctx) return -1;
    if (!ctx->finalized) return fail(ctx, "weights not finalized: call said_finalize_weights first");
    return 0;
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" {

int said_abi_version(void) { return 9; }   // 9: said_set_precision modes (SAID_PREC_FP32_STRICT), said_effective_precision, said_precision_note, said_numeric_status; 8: said_loop_progress_reset; 7: said_debug_ws_*; 6: said_loop_progress; 5: said_clone, said_loop_params::noise_batch_offset; 4: said_reserve, noise_seed, said_philox_normal, said_debug_option

const char* said_last_error(const said_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int said_create(said_ctx** out, int device, int max_batch_eff, int max_frames, int in_channels, int ctx_dim) {
    said_ctx* ctx = nullptr;
    if (!out) return fail(nullptr, "said_create: out is null");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, "said_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(nullptr, "said_create: device %d out of range (%d visible)", device, ndev);
    if (max_batch_eff < 1 || max_frames < 1) return fail(nullptr, "said_create: bad sizes");
    if (in_channels != 32) return fail(nullptr, "said_create: in_channels must be 32 (got %d)", in_channels);
    if (ctx_dim < 16 || ctx_dim % 16) return fail(nullptr, "said_create: ctx_dim must be a positive multiple of 16 (got %d)", ctx_dim);
    DeviceRestore restore_device;
    {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return fail(nullptr, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess) return fail(nullptr, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(nullptr, "said_create: device is %s; this library is built for gfx950 only", prop.gcnArchName);
    }
    ctx = new said_ctx();
    ctx->device = device; ctx->cin = in_channels; ctx->ctx_dim = ctx_dim;
    configure_gemm_kernels();
    configure_ugemm_kernels();
    configure_stchain_kernel();
    configure_attn_kernels();
    configure_out_sched_kernel();
    ctx->use_ugemm = dev_env("SAID_NO_UGEMM") == nullptr;
    ctx->use_branches = dev_env("SAID_BRANCHES") != nullptr;
    ctx->cfg_share = dev_env("SAID_NO_CFG_SHARE") == nullptr;
    ctx->audio_bf16 = dev_env("SAID_NO_AUDIO_BF16") == nullptr;
    ctx->pos_tgemm = dev_env("SAID_NO_POSCONV_TGEMM") == nullptr;
    ctx->unet_tgemm = dev_env("SAID_NO_UNET_TGEMM") == nullptr;
    ctx->unet_fgemm = dev_env("SAID_NO_UNET_FGEMM") == nullptr;
    if (dev_env("SAID_UNET_TGEMM_MIN")) ctx->unet_tgemm_min_tokens = ctx->unet_fgemm_min_tokens = atoll(dev_env("SAID_UNET_TGEMM_MIN"));
    configure_tgemm_kernel();
    configure_xgemm_kernels();
    configure_rgemm_kernels();

    int rc = 0;
    rc |= dalloc(ctx, &ctx->coef1_dev, 8);
    rc |= dalloc(ctx, &ctx->step_dev, 4);
    rc |= dalloc(ctx, &ctx->status_dev, 4);
    rc |= dalloc(ctx, &ctx->chain_ticket, CHAIN2_MAX_TILES * 6);
    rc |= dalloc(ctx, &ctx->seed_dev, 4);
    rc |= dalloc(ctx, &ctx->clk_dev, 64 * 128);
    rc |= dalloc(ctx, &ctx->freqs, MC / 2);
    rc |= alloc_workspace(ctx, max_batch_eff, max_frames);
    if (rc) { g_create_err = ctx->err; said_destroy(ctx); return -1; }
    *out = ctx;
    return 0;
}

int said_destroy(said_ctx* ctx) {
    if (!ctx) return 0;
    DeviceRestore restore_device;
    (void)hipSetDevice(ctx->device);
    drop_graphs(ctx);
    release_cap_streams(ctx);   // (own_stream belongs to the pool)
    for (void* p : ctx->allocs) (void)hipFree(p);
    for (void* p : ctx->ws_allocs) (void)hipFree(p);
    delete ctx;
    return 0;
}

int said_reserve(said_ctx* ctx, int max_batch_eff, int max_frames) {
    if (!ctx) return -1;
    if (max_batch_eff < 1 || max_frames < 1) return fail(ctx, "said_reserve: bad sizes");
    if (max_batch_eff <= ctx->maxBe && max_frames <= ctx->maxT) return 0;
    DeviceRestore restore_device;
    HIPCHK(hipSetDevice(ctx->device));
    // earlier calls may still be running on any stream: the buffers they use are about to be freed
    HIPCHK(hipDeviceSynchronize());
    drop_graphs(ctx);   // the captured step graphs hold the old buffers' addresses
    for (void* p : ctx->ws_allocs) (void)hipFree(p);
    ctx->ws_allocs.clear();
    const int want_be = std::max(max_batch_eff, ctx->maxBe), want_t = std::max(max_frames, ctx->maxT);
    if (alloc_workspace(ctx, want_be, want_t)) {
        // Out of memory part-way through: the old workspace is gone and some members still hold its (freed) addresses.  Release what was
        // obtained and advertise NO capacity — every entry point then refuses its sizes instead of running on freed memory — until a
        // later said_reserve (e.g. with a smaller batch) succeeds and re-assigns every workspace pointer.
        const std::string why = ctx->err;
        for (void* p : ctx->ws_allocs) (void)hipFree(p);
        ctx->ws_allocs.clear();
        (void)hipGetLastError();
        ctx->maxBe = 0; ctx->maxT = 0; ctx->maxTp = 0;
        ctx->band_T = ctx->band_S = -1;
        return fail(ctx, "said_reserve(%d, %d): workspace allocation failed (%s); the context has no workspace until a smaller said_reserve succeeds",
                    want_be, want_t, why.c_str());
    }
    return 0;
}

// The clones' streams come from ONE pool per device for the whole process (three streams, chosen on first use, never destroyed): a live
// stream is mapped onto one of the device's few hardware queues (four by default), streams on the same queue serialise, and the mapping
// follows creation order — with two other streams alive in the application the pool's second stream landed on the CALLER's queue and a
// three-group loop ran 37 % slower (2.43 vs 1.77 ms per step), while with four others it was fine again.  There is no API to ask which
// queue a stream sits on, so the pool is chosen by measurement: candidates are created one by one and probed with a 0.3 ms spinning wave —
// two streams that finish two such waves in the time of one are on different queues — until three are found that run beside the default
// stream and beside each other; the other candidates are destroyed.  Two models with a stream per clone were enough to exhaust the queues
// (the second model's groups ran 16 % SLOWER than its unsplit batch): clones of different parents share the pool.
constexpr int POOL_STREAMS = 3, POOL_DEVICES = 64, POOL_CANDIDATES = 12;
static std::mutex g_pool_mu;
static hipStream_t g_pool[POOL_DEVICES][POOL_STREAMS];
static bool g_pool_ready[POOL_DEVICES];
static int g_pool_probed[POOL_DEVICES];   // candidates looked at (said_debug_get "pool_probed")

// true when a wave on `b` runs beside one on `a` (a == nullptr: the default stream)
static bool streams_overlap(hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t e1) {
    const long long ticks = 30000;   // 0.3 ms of the 100 MHz wall clock
    if (hipEventRecord(e0, a) != hipSuccess) return false;
    launch_spin(ticks, a);
    launch_spin(ticks, b);
    if (hipEventRecord(e1, b) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return false;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return false;
    return ms < 0.48f;               // serialised: >= 0.6 ms
}
static void pool_init(int dev) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int found = 0;
    if (hipDeviceSynchronize() == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        launch_spin(1, nullptr);     // first launch of the probe kernel (module load) outside the measurements
        (void)hipDeviceSynchronize();
        hipStream_t rejected[POOL_CANDIDATES];
        int nrej = 0;
        for (int c = 0; c < POOL_CANDIDATES && found < POOL_STREAMS; ++c) {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
            ++g_pool_probed[dev];
            bool ok = streams_overlap(nullptr, s, e0, e1);
            for (int k = 0; ok && k < found; ++k) ok = streams_overlap(g_pool[dev][k], s, e0, e1);
            if (ok) g_pool[dev][found++] = s;
            else rejected[nrej++] = s;   // kept alive until the end: a destroyed candidate's queue is the next candidate's again
        }
        for (int i = 0; i < nrej; ++i) (void)hipStreamDestroy(rejected[i]);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipGetLastError();
    for (; found < POOL_STREAMS; ++found)   // fewer independent queues than groups (or the probe failed): any stream will do
        if (hipStreamCreateWithFlags(&g_pool[dev][found], hipStreamNonBlocking) != hipSuccess) g_pool[dev][found] = nullptr;
    g_pool_ready[dev] = true;
}
static hipStream_t pool_stream(int dev, int idx) {
    if (dev < 0 || dev >= POOL_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_pool_ready[dev]) pool_init(dev);
    return g_pool[dev][idx];
}

int said_clone(said_ctx* parent, said_ctx** out, int max_batch_eff, int max_frames) {
    if (!out) return fail(parent, "said_clone: out is null");
    *out = nullptr;
    if (check_ready(parent)) return -1;
    if (max_batch_eff < 1 || max_frames < 1) return fail(parent, "said_clone: bad sizes");
    DeviceRestore restore_device;
    {
        said_ctx* ctx = parent;
        HIPCHK(hipSetDevice(parent->device));
    }
    // a copy of the parent: every packed weight / table pointer is SHARED (read-only on the device); everything a run writes is its own
    said_ctx* c = new said_ctx(*parent);
    c->err.clear(); c->launch_err.clear(); c->host_w.clear();
    c->allocs.clear(); c->ws_allocs.clear(); c->alloc_list = &c->allocs;
    c->is_clone = true;
    c->graph = c->graph_rem = nullptr; c->gexec = c->gexec_rem = nullptr; c->gkey.clear(); c->gnodes = 0;
    c->cap_stream = c->cap_stream2 = nullptr; c->ev_fork = c->ev_join = nullptr; c->own_stream = nullptr;
    c->stage_log.clear(); c->log_on = false; c->dbg_stop = -1; c->dbg_only = -1; c->dbg_count = 0; c->clk_on = false; c->xclk_on = false;
    // lazily grown buffers start empty
    c->abufA = c->abufB = nullptr; c->abuf_elems[0] = c->abuf_elems[1] = 0;
    c->aH = c->aT = c->aO = c->aQK = c->aVT = c->aF = c->aPOS = c->aX = nullptr; c->a_tok_elems = 0;
    c->bA0 = c->bA1 = c->bX = c->bHb = c->bF = c->bO = nullptr; c->bH = c->bT = c->bPosT = nullptr; c->b_conv_elems[0] = c->b_conv_elems[1] = 0; c->b_tok = 0;
    c->bXg = nullptr; c->bXg_elems = 0;
    c->noise_cm = nullptr; c->noise_cm_elems = 0;
    c->kvt_S = -1;   // (the clone's own workspace holds no key-major copy yet)
    said_ctx* ctx = c;
    auto bail = [&](const char* what) { parent->err = std::string("said_clone: ") + what + (c->err.empty() ? "" : ": " + c->err); said_destroy(c); return -1; };
    c->own_stream = pool_stream(parent->device, parent->n_clones++ % POOL_STREAMS);
    if (!c->own_stream) return bail("hipStreamCreateWithFlags failed");
    c->n_clones = 0;
    int rc = 0;
    rc |= dalloc(ctx, &c->coef1_dev, 8);
    rc |= dalloc(ctx, &c->step_dev, 4);
    rc |= dalloc(ctx, &c->status_dev, 4);
    rc |= dalloc(ctx, &c->chain_ticket, CHAIN2_MAX_TILES * 6);
    rc |= dalloc(ctx, &c->seed_dev, 4);
    rc |= dalloc(ctx, &c->clk_dev, 64 * 128);
    rc |= alloc_workspace(c, max_batch_eff, max_frames);
    if (rc) return bail("workspace allocation failed");
    *out = c;
    return 0;
}

void* said_stream(const said_ctx* ctx) { return ctx ? (void*)ctx->own_stream : nullptr; }

int said_capacity(const said_ctx* ctx, int* max_batch_eff, int* max_frames) {
    if (!ctx) return -1;
    if (max_batch_eff) *max_batch_eff = ctx->maxBe;
    if (max_frames) *max_frames = ctx->maxT;
    return 0;
}

int said_set_weight(said_ctx* ctx, const char* name, const float* data_host, const int64_t* shape, int ndim) {
    if (!ctx) return -1;
    if (ctx->finalized) return fail(ctx, "said_set_weight after finalize");
    if (!name || !data_host || !shape || ndim < 1 || ndim > 8) return fail(ctx, "said_set_weight: bad arguments");
    for (int i = 0; i < ndim; ++i) if (shape[i] < 0) return fail(ctx, "said_set_weight(%s): negative dimension", name);
    ++ctx->n_set_weight;
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    t.data.assign(data_host, data_host + t.numel());
    ctx->host_w[name] = std::move(t);
    return 0;
}

int said_set_timestep_freqs(said_ctx* ctx, const float* f, int n) {
    if (!ctx) return -1;
    if (n != MC / 2) return fail(ctx, "said_set_timestep_freqs: expected %d entries, got %d", MC / 2, n);
    HIPCHK(hipMemcpy(ctx->freqs, f, n * sizeof(float), hipMemcpyHostToDevice));
    ctx->freqs_set = true;
    return 0;
}

// Operands of the fused SpatialTransformer tail (stchain.hip) for block `b` (= "...transformer_blocks.0"):
//  * one weight stream: for each of the kernel's eight waves, 2 KB units (a k16 step's h and l fragments of v_mfma_f32_32x32x16_f16 A operands: lane l holds
//    row 32 tile + (l & 31), k = 16 step + 8 (l >> 5) .. + 7) in the order the wave consumes them — waves 0-5 (column owner j = output columns [32 j, 32 j + 32)):
//    to_out1 12 steps, to_q 12, to_out2 12, GEGLU pairs j, j + 8, j + 16 (12 steps each, value then gate unit per step), folded proj_out over [h ; x2]: 60 steps for
//    j < 4, steps 0 .. 29 for j = 4, 5; waves 6, 7: GEGLU pairs w, w + 8, w + 16, then steps 30 .. 59 of the folded proj_out's column tiles 4, 5.  LayerNorm2 / LayerNorm3 affines are folded in: W' = W diag(gamma) (formed in double), bias' = bias + W beta.
//  * the vectors b1, bq = Wq beta2, bo2, c2, bffp, bff' (CHAIN_VEC_FLOATS).
static int pack_chain(said_ctx* ctx, STW& sw, const std::string& b, const std::vector<float>& c2v) {
    const HostTensor* W1 = getw(ctx, b + ".attn1.to_out.0.weight", {MC, MC});
    const HostTensor* B1 = getw(ctx, b + ".attn1.to_out.0.bias", {MC});
    const HostTensor* Wq = getw(ctx, b + ".attn2.to_q.weight", {MC, MC});
    const HostTensor* G2 = getw(ctx, b + ".norm2.weight", {MC});
    const HostTensor* Be2 = getw(ctx, b + ".norm2.bias", {MC});
    const HostTensor* W3 = getw(ctx, b + ".attn2.to_out.0.weight", {MC, MC});
    const HostTensor* B3 = getw(ctx, b + ".attn2.to_out.0.bias", {MC});
    const HostTensor* Wf = getw(ctx, b + ".ff.net.0.proj.weight", {2 * FFI, MC});
    const HostTensor* Bf = getw(ctx, b + ".ff.net.0.proj.bias", {2 * FFI});
    const HostTensor* G3 = getw(ctx, b + ".norm3.weight", {MC});
    const HostTensor* Be3 = getw(ctx, b + ".norm3.bias", {MC});
    const HostTensor* PF = getw(ctx, "__ffproj.w0", {MC, FFI});
    const HostTensor* PX = getw(ctx, "__ffproj.w1", {MC, MC});
    const HostTensor* PB = getw(ctx, "__ffproj.b", {MC});
    if (!W1 || !B1 || !Wq || !G2 || !Be2 || !W3 || !B3 || !Wf || !Bf || !G3 || !Be3 || !PF || !PX || !PB || (int)c2v.size() != MC) return -1;
    // element (row n, k) of the matrix a unit multiplies, in double (the folds) -> split into fp16 planes
    auto elem = [&](int kind, int n, int k) -> double {
        switch (kind) {
            case 0: return W1->data[(size_t)n * MC + k];
            case 1: return (double)Wq->data[(size_t)n * MC + k] * (double)G2->data[k];
            case 2: return W3->data[(size_t)n * MC + k];
            case 3: return (double)Wf->data[(size_t)n * MC + k] * (double)G3->data[k];
            default: return k < FFI ? PF->data[(size_t)n * FFI + k] : PX->data[(size_t)n * MC + (k - FFI)];
        }
    };
    {   // the matrices the stream holds that make_pw has not seen: the LayerNorm-folded to_q and GEGLU projections
        std::vector<float> f((size_t)2 * FFI * MC);
        for (int n = 0; n < MC; ++n) for (int k = 0; k < MC; ++k) f[(size_t)n * MC + k] = (float)elem(1, n, k);
        scan_split_range(ctx, b + ".attn2.to_q.weight * norm2.weight", f.data(), (size_t)MC * MC);
        for (int n = 0; n < 2 * FFI; ++n) for (int k = 0; k < MC; ++k) f[(size_t)n * MC + k] = (float)elem(3, n, k);
        scan_split_range(ctx, b + ".ff.net.0.proj.weight * norm3.weight", f.data(), f.size());
    }
    std::vector<_Float16> st(CHAIN_STREAM_BYTES / 2);
    size_t o = 0;
    auto put_unit = [&](int kind, int row0, int step) {
        for (int pl = 0; pl < 2; ++pl)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 8; ++i) {
                    const float v = (float)elem(kind, row0 + (l & 31), 16 * step + 8 * (l >> 5) + i);
                    const _Float16 hv = (_Float16)v;
                    st[o++] = pl == 0 ? hv : (_Float16)((v - (float)hv) * 2048.f);
                }
    };
    auto put_geglu = [&](int w) {
        for (int pi = 0; pi < 3; ++pi)
            for (int s = 0; s < 12; ++s) {
                put_unit(3, 32 * (w + 8 * pi), s);
                put_unit(3, FFI + 32 * (w + 8 * pi), s);
            }
    };
    for (int w = 0; w < 6; ++w) {
        for (int kind = 0; kind < 3; ++kind)
            for (int s = 0; s < 12; ++s) put_unit(kind, 32 * w, s);
        put_geglu(w);
        for (int s = 0; s < (w < 4 ? 60 : 30); ++s) put_unit(4, 32 * w, s);   // (column tiles 4, 5: steps 30 .. 59 belong to waves 6, 7)
    }
    for (int w = 6; w < 8; ++w) {
        put_geglu(w);
        for (int s = 30; s < 60; ++s) put_unit(4, 32 * (w - 2), s);
    }
    if (o != st.size()) return fail(ctx, "pack_chain: stream size mismatch");
    {   // the bf16 stream: the same units in the same order, ONE plane of bf16 (RNE) — 1 KB per unit
        std::vector<uint16_t> sb(CHAIN_STREAM_UNITS * 512);
        size_t ob = 0;
        auto put_unit_b = [&](int kind, int row0, int step) {
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 8; ++i) sb[ob++] = bf16_rne((float)elem(kind, row0 + (l & 31), 16 * step + 8 * (l >> 5) + i));
        };
        auto put_geglu_b = [&](int w) {
            for (int pi = 0; pi < 3; ++pi)
                for (int s = 0; s < 12; ++s) {
                    put_unit_b(3, 32 * (w + 8 * pi), s);
                    put_unit_b(3, FFI + 32 * (w + 8 * pi), s);
                }
        };
        for (int w = 0; w < 6; ++w) {
            for (int kind = 0; kind < 3; ++kind)
                for (int s = 0; s < 12; ++s) put_unit_b(kind, 32 * w, s);
            put_geglu_b(w);
            for (int s = 0; s < (w < 4 ? 60 : 30); ++s) put_unit_b(4, 32 * w, s);
        }
        for (int w = 6; w < 8; ++w) {
            put_geglu_b(w);
            for (int s = 30; s < 60; ++s) put_unit_b(4, 32 * (w - 2), s);
        }
        if (ob != sb.size()) return fail(ctx, "pack_chain: bf16 stream size mismatch");
        uint16_t* d = nullptr;
        if (dalloc(ctx, &d, sb.size(), false)) return -1;
        HIPCHK(hipMemcpy(d, sb.data(), sb.size() * 2, hipMemcpyHostToDevice));
        sw.chain_wb = d;
    }
    std::vector<float> stf(st.size() / 2);
    memcpy(stf.data(), st.data(), st.size() * 2);
    if (upload(ctx, &sw.chain_w, stf.data(), stf.size())) return -1;
    {   // the three-slice stream (stchain.h CHAIN3_*): slice c = GEGLU pairs 8 c .. 8 c + 7 (one per wave) + k16 steps 16 c .. 16 c + 15 of the folded proj_out's GEGLU
        // segment + steps 4 c .. 4 c + 3 of its x2 segment (steps 48 .. 59 of the 60); the three 192 x 192 projections in front are in every slice
        st.assign(CHAIN3_STREAM_BYTES / 2, (_Float16)0.f);
        o = 0;
        for (int c = 0; c < 3; ++c) {
            int ff[20];
            for (int i = 0; i < 16; ++i) ff[i] = 16 * c + i;
            for (int i = 0; i < 4; ++i) ff[16 + i] = 48 + 4 * c + i;
            auto put_pair = [&](int p) {
                for (int s2 = 0; s2 < 12; ++s2) { put_unit(3, 32 * p, s2); put_unit(3, FFI + 32 * p, s2); }
            };
            for (int w = 0; w < 6; ++w) {
                for (int kind = 0; kind < 3; ++kind)
                    for (int s2 = 0; s2 < 12; ++s2) put_unit(kind, 32 * w, s2);
                put_pair(8 * c + w);
                for (int i = 0; i < (w < 4 ? 20 : 10); ++i) put_unit(4, 32 * w, ff[i]);
            }
            for (int w = 6; w < 8; ++w) {
                put_pair(8 * c + w);
                for (int i = 10; i < 20; ++i) put_unit(4, 32 * (w - 2), ff[i]);
            }
        }
        if (o != st.size()) return fail(ctx, "pack_chain: three-slice stream size mismatch");
        stf.resize(st.size() / 2);
        memcpy(stf.data(), st.data(), st.size() * 2);
        if (upload(ctx, &sw.chain_w3, stf.data(), stf.size())) return -1;
    }
    {   // the two-slice stream (stchain.h CHAIN2_*): slice c = GEGLU pairs 12 c .. 12 c + 11 — waves 0-3 two each (local pairs w, w + 4), waves 4-7 one (w + 4) —
        // + k16 steps 24 c .. 24 c + 23 of the folded proj_out's GEGLU segment + steps 6 c .. 6 c + 5 of its x2 segment
        st.assign(CHAIN2_STREAM_BYTES / 2, (_Float16)0.f);
        o = 0;
        for (int c = 0; c < 2; ++c) {
            int ff[30];
            for (int i = 0; i < 24; ++i) ff[i] = 24 * c + i;
            for (int i = 0; i < 6; ++i) ff[24 + i] = 48 + 6 * c + i;
            auto put_pair = [&](int p) {
                for (int s2 = 0; s2 < 12; ++s2) { put_unit(3, 32 * p, s2); put_unit(3, FFI + 32 * p, s2); }
            };
            for (int w = 0; w < 6; ++w) {
                for (int kind = 0; kind < 3; ++kind)
                    for (int s2 = 0; s2 < 12; ++s2) put_unit(kind, 32 * w, s2);
                if (w < 4) { put_pair(12 * c + w); put_pair(12 * c + w + 4); } else put_pair(12 * c + w + 4);
                for (int i = 0; i < (w < 4 ? 30 : 15); ++i) put_unit(4, 32 * w, ff[i]);
            }
            for (int w = 6; w < 8; ++w) {
                put_pair(12 * c + w + 4);
                for (int i = 15; i < 30; ++i) put_unit(4, 32 * (w - 2), ff[i]);
            }
        }
        if (o != st.size()) return fail(ctx, "pack_chain: two-slice stream size mismatch");
        stf.resize(st.size() / 2);
        memcpy(stf.data(), st.data(), st.size() * 2);
        if (upload(ctx, &sw.chain_w2, stf.data(), stf.size())) return -1;
    }
    std::vector<float> vec(CHAIN_VEC_FLOATS);
    for (int n = 0; n < MC; ++n) {
        double bq = 0.0;
        for (int k = 0; k < MC; ++k) bq += (double)Wq->data[(size_t)n * MC + k] * (double)Be2->data[k];
        vec[n] = B1->data[n];
        vec[MC + n] = (float)bq;
        vec[2 * MC + n] = B3->data[n];
        vec[3 * MC + n] = c2v[n];
        vec[4 * MC + n] = PB->data[n];
    }
    for (int n = 0; n < 2 * FFI; ++n) {
        double bf = Bf->data[n];
        for (int k = 0; k < MC; ++k) bf += (double)Wf->data[(size_t)n * MC + k] * (double)Be3->data[k];
        vec[5 * MC + n] = (float)bf;
    }
    return upload(ctx, &sw.chain_vec, vec.data(), vec.size());
}

int said_finalize_weights(said_ctx* ctx, void* stream) {
    (void)stream;
    if (!ctx) return -1;
    if (ctx->finalized) return fail(ctx, "weights already finalized");
    HIPCHK(hipSetDevice(ctx->device));
    const std::string D = "denoiser.model.";
    const int CD = ctx->ctx_dim;
    size_t used = 0;
    auto count_prefix = [&](const std::string& p) { size_t n = 0; for (auto& kv : ctx->host_w) if (kv.first.rfind(p, 0) == 0) ++n; return n; };

    // ---- UNet ----
    if (make_pw(ctx, &ctx->te1, D + "time_embed.0.weight", D + "time_embed.0.bias", TE, MC, 0)) return -1;
    if (make_pw(ctx, &ctx->te2, D + "time_embed.2.weight", D + "time_embed.2.bias", TE, TE, 0)) return -1;
    if (make_pw(ctx, &ctx->conv_in, D + "input_blocks.0.0.weight", D + "input_blocks.0.0.bias", MC, ctx->cin, 3)) return -1;
    ctx->pw_split = 1;   // (out_sched_kernel's split-fp16 products)
    if (make_pw(ctx, &ctx->conv_out, D + "out.2.weight", D + "out.2.bias", ctx->cin, MC, 3, 1, D + "out.0.weight", D + "out.0.bias")) return -1;
    ctx->pw_split = 0;
    if (upload_bf16(ctx, &ctx->bw_out, ctx->host_w[D + "out.2.weight"].data.data(), (size_t)ctx->cin, MC, 3)) return -1;
    if (upvec(ctx, &ctx->out_g, D + "out.0.weight", MC) || upvec(ctx, &ctx->out_b, D + "out.0.bias", MC)) return -1;
    used += 8;
    const char* res_names[NRES] = {"input_blocks.1.0", "middle_block.0", "middle_block.2", "output_blocks.0.0", "output_blocks.1.0"};
    const char* st_names[NST] = {"input_blocks.1.1", "middle_block.1", "output_blocks.0.1", "output_blocks.1.1"};
    std::vector<float> emb_w((size_t)NRES * MC * TE), emb_b((size_t)NRES * MC);
    ctx->pw_split = 1;   // the ResBlock / SpatialTransformer weights also in the split-fp16 packing (small-batch fp32 products: gemm_lds.hip SP)
    for (int r = 0; r < NRES; ++r) {
        const std::string p = D + res_names[r];
        ResW& rw = ctx->res[r];
        rw.cin = r >= 3 ? 2 * MC : MC;
        rw.has_skip = r >= 3;
        if (upvec(ctx, &rw.g1, p + ".in_layers.0.weight", rw.cin) || upvec(ctx, &rw.b1, p + ".in_layers.0.bias", rw.cin)) return -1;
        if (make_pw(ctx, &rw.conv1, p + ".in_layers.2.weight", p + ".in_layers.2.bias", MC, rw.cin, 3, rw.has_skip ? 2 : 1, p + ".in_layers.0.weight", p + ".in_layers.0.bias")) return -1;
        if (upvec(ctx, &rw.g2, p + ".out_layers.0.weight", MC) || upvec(ctx, &rw.b2, p + ".out_layers.0.bias", MC)) return -1;
        if (make_pw(ctx, &rw.conv2, p + ".out_layers.3.weight", p + ".out_layers.3.bias", MC, MC, 3, 1, p + ".out_layers.0.weight", p + ".out_layers.0.bias")) return -1;
        {   // tgemm.hip operands: conv1 [192][3 * cin] tap-major; conv2 [192][576 (+ 384 skip columns)]
            if (upload_tm_pair(ctx, &rw.t_conv1, &rw.tf_conv1, ctx->host_w[p + ".in_layers.2.weight"].data.data(), MC, (size_t)rw.cin, 3, &rw.tp_conv1)) return -1;
            const HostTensor& c2w = ctx->host_w[p + ".out_layers.3.weight"];
            std::vector<float> cat((size_t)MC * (3 * MC + (rw.has_skip ? 2 * MC : 0)));
            const size_t Kc = 3 * MC + (rw.has_skip ? 2 * MC : 0);
            const HostTensor* sk = rw.has_skip ? getw(ctx, p + ".skip_connection.weight", {MC, 2 * MC, 1}) : nullptr;
            if (rw.has_skip && !sk) return -1;
            for (int n = 0; n < MC; ++n) {
                for (int t = 0; t < 3; ++t)
                    for (int cc = 0; cc < MC; ++cc) cat[n * Kc + t * MC + cc] = c2w.data[((size_t)n * MC + cc) * 3 + t];
                if (sk) for (int cc = 0; cc < 2 * MC; ++cc) cat[n * Kc + 3 * MC + cc] = sk->data[(size_t)n * 2 * MC + cc];
            }
            if (upload_tm_pair(ctx, &rw.t_conv2, &rw.tf_conv2, cat.data(), MC, Kc, 1, &rw.tp_conv2)) return -1;
        }
        const HostTensor* ew = getw(ctx, p + ".emb_layers.1.weight", {MC, TE});
        const HostTensor* eb = getw(ctx, p + ".emb_layers.1.bias", {MC});
        if (!ew || !eb) return -1;
        std::copy(ew->data.begin(), ew->data.end(), emb_w.begin() + (size_t)r * MC * TE);
        std::copy(eb->data.begin(), eb->data.end(), emb_b.begin() + (size_t)r * MC);
        used += 10;
        rw.bias2 = nullptr;
        if (rw.has_skip) {
            if (make_pw(ctx, &rw.skip, p + ".skip_connection.weight", p + ".skip_connection.bias", MC, 2 * MC, 1, 2)) return -1;
            const HostTensor* b2 = getw(ctx, p + ".out_layers.3.bias", {MC});
            const HostTensor* bs = getw(ctx, p + ".skip_connection.bias", {MC});
            std::vector<float> sum(MC);
            for (int i = 0; i < MC; ++i) sum[i] = b2->data[i] + bs->data[i];
            if (upload(ctx, &rw.bias2, sum.data(), MC)) return -1;
            used += 2;
        }
    }
    ctx->pw_split = 0;
    {   // all five emb_layers as one GEMM (960 x 768)
        ctx->host_w["__emb_all.w"] = HostTensor{emb_w, {NRES * MC, TE}};
        ctx->host_w["__emb_all.b"] = HostTensor{emb_b, {NRES * MC}};
        if (make_pw(ctx, &ctx->emb_all, "__emb_all.w", "__emb_all.b", NRES * MC, TE, 0)) return -1;
    }
    std::vector<float> kv_w((size_t)NST * 2 * MC * CD);
    for (int i = 0; i < NST; ++i) {
        const std::string p = D + st_names[i], b = p + ".transformer_blocks.0";
        STW& sw = ctx->st[i];
        ctx->pw_split = 1;
        if (upvec(ctx, &sw.gn_g, p + ".norm.weight", MC) || upvec(ctx, &sw.gn_b, p + ".norm.bias", MC)) return -1;
        if (upvec(ctx, &sw.l1g, b + ".norm1.weight", MC) || upvec(ctx, &sw.l1b, b + ".norm1.bias", MC)) return -1;
        if (upvec(ctx, &sw.l2g, b + ".norm2.weight", MC) || upvec(ctx, &sw.l2b, b + ".norm2.bias", MC)) return -1;
        if (upvec(ctx, &sw.l3g, b + ".norm3.weight", MC) || upvec(ctx, &sw.l3b, b + ".norm3.bias", MC)) return -1;
        const HostTensor* wq = getw(ctx, b + ".attn1.to_q.weight", {MC, MC});
        const HostTensor* wk = getw(ctx, b + ".attn1.to_k.weight", {MC, MC});
        const HostTensor* wv = getw(ctx, b + ".attn1.to_v.weight", {MC, MC});
        if (!wq || !wk || !wv) return -1;
        std::vector<float> qkv;
        qkv.insert(qkv.end(), wq->data.begin(), wq->data.end());
        qkv.insert(qkv.end(), wk->data.begin(), wk->data.end());
        qkv.insert(qkv.end(), wv->data.begin(), wv->data.end());
        ctx->host_w["__qkv"] = HostTensor{qkv, {3 * MC, MC}};
        if (make_pw(ctx, &sw.qkv, "__qkv", "", 3 * MC, MC, 0, 1, p + ".norm.weight", p + ".norm.bias", b + ".norm1.weight", b + ".norm1.bias")) return -1;
        if (make_pw(ctx, &sw.out1, b + ".attn1.to_out.0.weight", b + ".attn1.to_out.0.bias", MC, MC, 0)) return -1;
        if (make_pw(ctx, &sw.q2, b + ".attn2.to_q.weight", "", MC, MC, 0, 1, "", "", b + ".norm2.weight", b + ".norm2.bias")) return -1;
        const HostTensor* k2 = getw(ctx, b + ".attn2.to_k.weight", {MC, CD});
        const HostTensor* v2 = getw(ctx, b + ".attn2.to_v.weight", {MC, CD});
        if (!k2 || !v2) return -1;
        std::copy(k2->data.begin(), k2->data.end(), kv_w.begin() + (size_t)(i * 2) * MC * CD);
        std::copy(v2->data.begin(), v2->data.end(), kv_w.begin() + (size_t)(i * 2 + 1) * MC * CD);
        if (make_pw(ctx, &sw.out2, b + ".attn2.to_out.0.weight", b + ".attn2.to_out.0.bias", MC, MC, 0)) return -1;
        {   // the three 192 x 192 projections around the banded cross-attention as token-major GEMM operands (xgemm_kernel)
            const HostTensor* w1 = getw(ctx, b + ".attn1.to_out.0.weight", {MC, MC});
            const HostTensor* wq2 = getw(ctx, b + ".attn2.to_q.weight", {MC, MC});
            const HostTensor* w2 = getw(ctx, b + ".attn2.to_out.0.weight", {MC, MC});
            if (!w1 || !wq2 || !w2) return -1;
            if (upload_tm_pair(ctx, &sw.t_out1, &sw.tf_out1, w1->data.data(), MC, MC, 1) || upload_tm_pair(ctx, &sw.t_q2, &sw.tf_q2, wq2->data.data(), MC, MC, 1) ||
                upload_tm_pair(ctx, &sw.t_out2, &sw.tf_out2, w2->data.data(), MC, MC, 1))
                return -1;
        }
        std::vector<float> c2_host;
        {   // attn2 output for the unconditional context (null_cond_emb repeated: every key / value identical, softmax
            // uniform => output = to_v(null)), pushed through to_out: c2 = W_out (W_v null) + b_out, in double
            const HostTensor* nc = getw(ctx, "null_cond_emb", {1, 1, CD});
            const HostTensor* wo = getw(ctx, b + ".attn2.to_out.0.weight", {MC, MC});
            const HostTensor* bo = getw(ctx, b + ".attn2.to_out.0.bias", {MC});
            if (!nc || !wo || !bo) return -1;
            std::vector<double> vn(MC, 0.0);
            for (int r = 0; r < MC; ++r) {
                double a = 0.0;
                for (int k = 0; k < CD; ++k) a += (double)v2->data[(size_t)r * CD + k] * nc->data[k];
                vn[r] = (double)(float)a;   // the reference's value rows are fp32
            }
            std::vector<float> c2v(MC);
            for (int n = 0; n < MC; ++n) {
                double a = bo->data[n];
                for (int r = 0; r < MC; ++r) a += (double)wo->data[(size_t)n * MC + r] * vn[r];
                c2v[n] = (float)a;
            }
            if (upload(ctx, &ctx->c2[i], c2v.data(), MC)) return -1;
            c2_host = c2v;
        }
        ctx->pw_split = 2;   // GEGLU runs as one output tile per wave over the whole K (NB = 4): flat step layout
        if (make_pw(ctx, &sw.ff1, b + ".ff.net.0.proj.weight", b + ".ff.net.0.proj.bias", 2 * FFI, MC, 0, 1, "", "", b + ".norm3.weight", b + ".norm3.bias")) return -1;
        ctx->pw_split = 1;
        if (make_pw(ctx, &sw.ff2, b + ".ff.net.2.weight", b + ".ff.net.2.bias", MC, FFI, 0)) return -1;
        if (make_pw(ctx, &sw.proj, p + ".proj_out.weight", p + ".proj_out.bias", MC, MC, 1)) return -1;
        {   // proj_out o ff.net.2 folded into ONE GEMM over [h (768) ; x2 (192)]  (attention.py:193 `ff(norm3(x)) + x`, :232-234):
            //   proj(F2 h + b2 + x2) + bp = (P F2) h + P x2 + (P b2 + bp).  The product is formed in double on the host.
            const HostTensor* F2 = getw(ctx, b + ".ff.net.2.weight", {MC, FFI});
            const HostTensor* b2 = getw(ctx, b + ".ff.net.2.bias", {MC});
            const HostTensor* Pw = ctx->host_w.count(p + ".proj_out.weight") && ctx->host_w[p + ".proj_out.weight"].shape.size() == 3
                                       ? getw(ctx, p + ".proj_out.weight", {MC, MC, 1}) : getw(ctx, p + ".proj_out.weight", {MC, MC});
            const HostTensor* bp = getw(ctx, p + ".proj_out.bias", {MC});
            if (!F2 || !b2 || !Pw || !bp) return -1;
            HostTensor PF, PX, PB;
            PF.shape = {MC, FFI}; PF.data.resize((size_t)MC * FFI);
            PX.shape = {MC, MC}; PX.data = Pw->data;
            PB.shape = {MC}; PB.data.resize(MC);
            std::vector<double> row(FFI);
            for (int n = 0; n < MC; ++n) {
                std::fill(row.begin(), row.end(), 0.0);
                double bb = bp->data[n];
                for (int k = 0; k < MC; ++k) {
                    const double pk = Pw->data[(size_t)n * MC + k];
                    bb += pk * b2->data[k];
                    const float* f2 = &F2->data[(size_t)k * FFI];
                    for (int c = 0; c < FFI; ++c) row[c] += pk * f2[c];
                }
                for (int c = 0; c < FFI; ++c) PF.data[(size_t)n * FFI + c] = (float)row[c];
                PB.data[n] = (float)bb;
            }
            ctx->host_w["__ffproj.w0"] = std::move(PF);
            ctx->host_w["__ffproj.w1"] = std::move(PX);
            ctx->host_w["__ffproj.b"] = std::move(PB);
            PW t0, t1;
            if (make_pw(ctx, &t0, "__ffproj.w0", "__ffproj.b", MC, FFI, 0) || make_pw(ctx, &t1, "__ffproj.w1", "", MC, MC, 0)) return -1;
            {   // tgemm.hip operands of this block: q/k/v rows, GEGLU rows tile-interleaved (value, gate), [P F2 | P]
                if (upload_tm_pair(ctx, &sw.t_qkv, &sw.tf_qkv, qkv.data(), 3 * MC, MC, 1, &sw.tp_qkv)) return -1;
                const HostTensor* f1 = getw(ctx, b + ".ff.net.0.proj.weight", {2 * FFI, MC});
                const HostTensor* f1b = getw(ctx, b + ".ff.net.0.proj.bias", {2 * FFI});
                if (!f1 || !f1b) return -1;
                std::vector<float> pw((size_t)2 * FFI * MC), pb((size_t)2 * FFI);
                for (int np = 0; np < 2 * FFI; ++np) {   // tile-interleaved (value, gate) rows for tgemm.hip's 256-wide tile
                    const int src = tgemm_geglu_src_row(np, 2 * FFI);
                    std::copy(f1->data.begin() + (size_t)src * MC, f1->data.begin() + (size_t)(src + 1) * MC, pw.begin() + (size_t)np * MC);
                    pb[np] = f1b->data[src];
                }
                if (upload_tm_pair(ctx, &sw.t_ff1, &sw.tf_ff1, pw.data(), 2 * FFI, MC, 1) || upload(ctx, &sw.t_ff1_bias, pb.data(), pb.size())) return -1;
                const HostTensor& w0 = ctx->host_w["__ffproj.w0"];
                const HostTensor& w1 = ctx->host_w["__ffproj.w1"];
                std::vector<float> cat((size_t)MC * (FFI + MC));
                for (int n = 0; n < MC; ++n) {
                    std::copy(w0.data.begin() + (size_t)n * FFI, w0.data.begin() + (size_t)(n + 1) * FFI, cat.begin() + (size_t)n * (FFI + MC));
                    std::copy(w1.data.begin() + (size_t)n * MC, w1.data.begin() + (size_t)(n + 1) * MC, cat.begin() + (size_t)n * (FFI + MC) + FFI);
                }
                if (upload_tm_pair(ctx, &sw.t_ffproj, &sw.tf_ffproj, cat.data(), MC, FFI + MC, 1)) return -1;
            }
            PW& fp = sw.ffproj;
            fp.N = MC; fp.taps = 1; fp.nseg = 2; fp.bias = t0.bias;
            fp.w[0] = t0.w[0]; fp.w4[0] = t0.w4[0]; fp.w2[0] = t0.w2[0]; fp.ws[0] = t0.ws[0]; fp.C[0] = FFI;
            fp.w[1] = t1.w[0]; fp.w4[1] = t1.w4[0]; fp.w2[1] = t1.w2[0]; fp.ws[1] = t1.ws[0]; fp.C[1] = MC;
        }
        if (pack_chain(ctx, sw, b, c2_host)) return -1;
        used += 24;
        ctx->pw_split = 0;
    }
    ctx->host_w["__kv_all"] = HostTensor{kv_w, {NST * 2 * MC, CD}};
    if (make_pw(ctx, &ctx->kv_all, "__kv_all", "", NST * 2 * MC, CD, 0)) return -1;
    {
        const HostTensor* nc = getw(ctx, "null_cond_emb", {1, 1, CD});
        if (!nc) return -1;
        if (upload(ctx, &ctx->null_cond, nc->data.data(), CD)) return -1;
        used += 1;
    }
    if (count_prefix("denoiser.") != 160) return fail(ctx, "unexpected key(s) in state dict: %zu denoiser.* tensors, expected 160", count_prefix("denoiser."));

    // ---- audio encoder (optional as a group: absent => said_audio_encode is unavailable) ----
    const std::string A = "audio_encoder.";
    const size_t n_audio = count_prefix(A);
    if (n_audio > 0) {
        int cin = 1;
        for (int i = 0; i < 7; ++i) {
            const std::string wn = A + "feature_extractor.conv_layers." + std::to_string(i) + ".conv.weight";
            auto it = ctx->host_w.find(wn);
            if (it == ctx->host_w.end() || it->second.shape.size() != 3) return fail(ctx, "missing key in state dict: %s", wn.c_str());
            const int k = (int)it->second.shape[2];
            ctx->w2v_kernel[i] = k;
            if (!getw(ctx, wn, {W2V_CONV, cin, k})) return -1;
            if (i == 0) {
                if (upload(ctx, &ctx->c0_w, it->second.data.data(), (size_t)W2V_CONV * k)) return -1;
                if (upvec(ctx, &ctx->c0_g, A + "feature_extractor.conv_layers.0.layer_norm.weight", W2V_CONV)) return -1;
                if (upvec(ctx, &ctx->c0_b, A + "feature_extractor.conv_layers.0.layer_norm.bias", W2V_CONV)) return -1;
            } else {
                if (make_pw(ctx, &ctx->aconv[i], wn, "", W2V_CONV, W2V_CONV, k)) return -1;
                if (upload_bf16(ctx, &ctx->bw_conv[i], it->second.data.data(), W2V_CONV, W2V_CONV, (size_t)k)) return -1;
            }
            cin = W2V_CONV;
        }
        if (upvec(ctx, &ctx->fp_lng, A + "feature_projection.layer_norm.weight", W2V_CONV)) return -1;
        if (upvec(ctx, &ctx->fp_lnb, A + "feature_projection.layer_norm.bias", W2V_CONV)) return -1;
        if (make_pw(ctx, &ctx->fproj, A + "feature_projection.projection.weight", A + "feature_projection.projection.bias", W2V_H, W2V_CONV, 0)) return -1;
        if (upload_bf16(ctx, &ctx->bw_fproj, ctx->host_w[A + "feature_projection.projection.weight"].data.data(), W2V_H, W2V_CONV, 1)) return -1;
        if (!getw(ctx, A + "masked_spec_embed", {W2V_H})) return -1;
        {   // positional conv: weight_norm(dim=2) folded on the host, then grouped packing (16 groups of 48)
            auto ig = ctx->host_w.find(A + "encoder.pos_conv_embed.conv.weight_g");
            if (ig == ctx->host_w.end() || ig->second.shape.size() != 3) return fail(ctx, "missing key in state dict: %sencoder.pos_conv_embed.conv.weight_g", A.c_str());
            const int K = (int)ig->second.shape[2];
            const int G = 16, CG = W2V_H / G;
            const HostTensor* wg = getw(ctx, A + "encoder.pos_conv_embed.conv.weight_g", {1, 1, K});
            const HostTensor* wv = getw(ctx, A + "encoder.pos_conv_embed.conv.weight_v", {W2V_H, CG, K});
            if (!wg || !wv) return -1;
            std::vector<double> nrm(K, 0.0);
            for (size_t i = 0; i < wv->data.size(); ++i) nrm[i % K] += (double)wv->data[i] * wv->data[i];
            std::vector<float> wfull(wv->data.size());
            for (size_t i = 0; i < wv->data.size(); ++i) {
                const float nk = (float)std::sqrt(nrm[i % K]);
                wfull[i] = wv->data[i] * (wg->data[i % K] / nk);
            }
            PW& pw = ctx->posconv;
            pw.N = CG; pw.taps = K; pw.nseg = 1; pw.C[0] = CG;
            std::vector<float> packed;
            for (int g = 0; g < G; ++g) {
                auto rows = rows_dense(CG, g * CG);
                auto part = pack_rows(wfull.data(), CG, K, rows, (CG + 31) / 32, 0, CG);
                packed.insert(packed.end(), part.begin(), part.end());
            }
            if (upload(ctx, &pw.w[0], packed.data(), packed.size())) return -1;
            if (upvec(ctx, &pw.bias, A + "encoder.pos_conv_embed.conv.bias", W2V_H)) return -1;
            if (CG % 8 == 0 && CG <= 64 && (K * CG) % 64 == 0) {   // bf16 mode: group g as a GEMM, rows padded to one 64-wide tile
                std::vector<float> wg64((size_t)G * 64 * K * CG, 0.f);
                for (int g = 0; g < G; ++g)
                    for (int n = 0; n < CG; ++n)
                        for (int c = 0; c < CG; ++c)
                            for (int k = 0; k < K; ++k)
                                wg64[(((size_t)g * 64 + n) * K + k) * CG + c] = wfull[((size_t)(g * CG + n) * CG + c) * K + k];
                if (upload_bf16(ctx, &ctx->bw_pos, wg64.data(), (size_t)G * 64, (size_t)K * CG, 1)) return -1;
                std::vector<float> bp(W2V_H + 64, 0.f);
                const HostTensor* pb = getw(ctx, A + "encoder.pos_conv_embed.conv.bias", {W2V_H});
                if (!pb) return -1;
                std::copy(pb->data.begin(), pb->data.end(), bp.begin());
                if (upload(ctx, &ctx->pos_bias_pad, bp.data(), bp.size())) return -1;
            }
        }
        if (upvec(ctx, &ctx->enc_lng, A + "encoder.layer_norm.weight", W2V_H) || upvec(ctx, &ctx->enc_lnb, A + "encoder.layer_norm.bias", W2V_H)) return -1;
        int L = 0;
        while (ctx->host_w.count(A + "encoder.layers." + std::to_string(L) + ".layer_norm.weight")) ++L;
        ctx->w2v_layers = L;
        ctx->layers.resize(L);
        ctx->blayers.resize(L);
        for (int l = 0; l < L; ++l) {
            const std::string p = A + "encoder.layers." + std::to_string(l);
            W2VLayer& ly = ctx->layers[l];
            std::vector<float> qkv, qb;
            for (const char* n : {"q_proj", "k_proj", "v_proj"}) {
                const HostTensor* w = getw(ctx, p + ".attention." + n + ".weight", {W2V_H, W2V_H});
                const HostTensor* b = getw(ctx, p + ".attention." + n + ".bias", {W2V_H});
                if (!w || !b) return -1;
                qkv.insert(qkv.end(), w->data.begin(), w->data.end());
                qb.insert(qb.end(), b->data.begin(), b->data.end());
            }
            ctx->host_w["__aqkv.w"] = HostTensor{qkv, {3 * W2V_H, W2V_H}};
            ctx->host_w["__aqkv.b"] = HostTensor{qb, {3 * W2V_H}};
            if (make_pw(ctx, &ly.qkv, "__aqkv.w", "__aqkv.b", 3 * W2V_H, W2V_H, 0)) return -1;
            if (make_pw(ctx, &ly.out, p + ".attention.out_proj.weight", p + ".attention.out_proj.bias", W2V_H, W2V_H, 0)) return -1;
            if (make_pw(ctx, &ly.ff1, p + ".feed_forward.intermediate_dense.weight", p + ".feed_forward.intermediate_dense.bias", W2V_FFN, W2V_H, 0)) return -1;
            if (make_pw(ctx, &ly.ff2, p + ".feed_forward.output_dense.weight", p + ".feed_forward.output_dense.bias", W2V_H, W2V_FFN, 0)) return -1;
            {   // bf16 copies for the bf16-mode encoder (the shapes were validated by make_pw above)
                said_ctx::BLayer& bl = ctx->blayers[l];
                if (upload_bf16(ctx, &bl.qkv, qkv.data(), 3 * W2V_H, W2V_H, 1)) return -1;
                if (upload_bf16(ctx, &bl.out, ctx->host_w[p + ".attention.out_proj.weight"].data.data(), W2V_H, W2V_H, 1)) return -1;
                if (upload_bf16(ctx, &bl.ff1, ctx->host_w[p + ".feed_forward.intermediate_dense.weight"].data.data(), W2V_FFN, W2V_H, 1)) return -1;
                if (upload_bf16(ctx, &bl.ff2, ctx->host_w[p + ".feed_forward.output_dense.weight"].data.data(), W2V_H, W2V_FFN, 1)) return -1;
            }
            if (upvec(ctx, &ly.ln1g, p + ".layer_norm.weight", W2V_H) || upvec(ctx, &ly.ln1b, p + ".layer_norm.bias", W2V_H)) return -1;
            if (upvec(ctx, &ly.ln2g, p + ".final_layer_norm.weight", W2V_H) || upvec(ctx, &ly.ln2b, p + ".final_layer_norm.bias", W2V_H)) return -1;
        }
        const size_t expect = 1 + 7 + 2 + 4 + 3 + 2 + (size_t)L * 16;
        if (n_audio != expect) return fail(ctx, "unexpected key(s) in state dict: %zu audio_encoder.* tensors, expected %zu", n_audio, expect);
        ctx->has_audio = true;
    }
    if (ctx->host_w.count("audio_proj_layer.weight")) {
        if (make_pw(ctx, &ctx->aproj, "audio_proj_layer.weight", "audio_proj_layer.bias", CD, W2V_H, 0)) return -1;
        if (upload_bf16(ctx, &ctx->bw_aproj, ctx->host_w["audio_proj_layer.weight"].data.data(), (size_t)CD, W2V_H, 1)) return -1;
        ctx->has_audio_proj = true;
    }
    for (auto& kv : ctx->host_w) {
        const std::string& k = kv.first;
        if (k.rfind("denoiser.", 0) == 0 || k.rfind(A, 0) == 0 || k.rfind("__", 0) == 0 || k == "null_cond_emb" ||
            k == "audio_proj_layer.weight" || k == "audio_proj_layer.bias")
            continue;
        return fail(ctx, "unexpected key(s) in state dict: %s", k.c_str());
    }
    if (!ctx->freqs_set) {
        std::vector<float> f(MC / 2);
        for (int k = 0; k < MC / 2; ++k) f[k] = (float)std::exp(-std::log(10000.0) * k / (MC / 2));
        HIPCHK(hipMemcpy(ctx->freqs, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    ctx->host_w.clear();
    ctx->finalized = true;
    (void)used;
    return 0;
}

// --------------------------------------------------------------------------------------------
// SAID.forward
// --------------------------------------------------------------------------------------------
int said_unet_forward(said_ctx* ctx, const float* sample_dev, const int64_t* timesteps_host, const float* context_dev,
                      int Be, int T, int S, float* out_dev, void* stream) {
    if (check_ready(ctx)) return -1;
    ctx->cur_concurrent = false;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (Be < 1 || Be > ctx->maxBe) return fail(ctx, "batch %d exceeds the context's max_batch_eff %d", Be, ctx->maxBe);
    if (T < 1 || T > ctx->maxT || S < 1 || S > ctx->maxT) return fail(ctx, "frames %d / context length %d exceed max_frames %d", T, S, ctx->maxT);
    if (set_band(ctx, T, S, s)) return -1;
    UGeo g = make_geo(ctx, Be, 0, T, S);
    g.emb_b_stride = 1;
    std::vector<long long> ts(timesteps_host, timesteps_host + Be);
    HIPCHK(hipMemcpyAsync(ctx->ts_dev, ts.data(), Be * sizeof(long long), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));  // ts is a stack-local staging buffer
    ctx->dbg_count = 0;
    HIPCHK(hipMemsetAsync(ctx->status_dev, 0, 2 * sizeof(int), s));
    run_time_embed(ctx, Be, s);
    launch_tm_to_cm(context_dev, ctx->CTX, Be, S, ctx->ctx_dim, g.Sp, (long long)ctx->ctx_dim * g.Sp, s);
    run_kv(ctx, 0, Be, S, g.Sp, s);
    launch_tm_to_cm(sample_dev, ctx->x_cm, Be, T, ctx->cin, g.Tp, (long long)ctx->cin * g.Tp, s);
    run_unet(ctx, g, s);
    if (ctx->dbg_stop < 0) launch_nonfinite_check(ctx->eps_cm, (long long)ctx->cin * g.Tp, g.Tp, Be, T, ctx->cin, ctx->status_dev, s);
    launch_cm_to_tm(ctx->eps_cm, out_dev, Be, T, ctx->cin, g.Tp, (long long)ctx->cin * g.Tp, s);
    LAUNCHCHK();
    HIPCHK(hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------------------------
// SAID.inference loop
// --------------------------------------------------------------------------------------------
// What the captured step graph depends on (everything else it reads from device memory at replay time).
// Denoise steps captured per graph.  A graph launch costs its host-side enqueue (~1 us per node) before the GPU sees its first kernel: behind a running graph that is hidden,
// for the FIRST graph of a loop it is not — 1200-1400 nodes (50 steps) in front of a 50- or 100-step loop cost configs[2] / configs[4] 2-6 % where the long loops gain 0.4-0.5 %
// from five times fewer graph boundaries (profiles/r06l_steps_per_graph.txt).  So: spg_limit (50) from 400 steps on, at most 10 below.
static int steps_per_graph(const said_ctx* ctx, int N) {
    if (ctx->use_branches) return 1;
    const int lim = N >= 400 ? ctx->spg_limit : std::min(ctx->spg_limit, 10);
    return std::max(1, std::min(lim, N));
}
static std::vector<long long> loop_graph_key(const said_ctx* ctx, const said_loop_params* p, const float* noise_cm) {
    const bool cfg = p->guidance_scale > 1.0f;
    float gs = p->guidance_scale, gr = (cfg && p->guidance_rescale > 0.f) ? p->guidance_rescale : 0.f, ls = p->latent_scale;
    int gsi, gri, lsi;
    memcpy(&gsi, &gs, 4); memcpy(&gri, &gr, 4); memcpy(&lsi, &ls, 4);
    const int spg = p->num_steps > 0 ? steps_per_graph(ctx, p->num_steps) : 0;
    const int rem = spg > 0 ? p->num_steps % spg : 0;
    return {spg, rem, p->batch, p->frames, cfg, gsi, gri, lsi, p->prediction_type, p->use_mask, p->use_step_noise, ctx->bf16_mode ? 1 : (strict_f32(ctx) ? 2 : 0), p->noise_batch_offset, p->concurrent != 0,
            (long long)(uintptr_t)(p->save_intermediate ? p->intermediates_dev : nullptr), (long long)(uintptr_t)(p->use_step_noise == 1 ? noise_cm : nullptr)};
}

static int loop_impl(said_ctx* ctx, const said_loop_params* p, void* stream, bool prepare_only) {
    if (check_ready(ctx)) return -1;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->cur_concurrent = p->concurrent != 0;
    ctx->dbg_count = 0;   // (said_debug_stop_after counts launches from the start of the call: the eager warm-up step then runs the first n launches, the captured steps none)
    const int B = p->batch, T = p->frames, N = p->num_steps, C = ctx->cin;
    const bool cfg = p->guidance_scale > 1.0f;
    const int Be = cfg ? 2 * B : B;
    if (B < 1 || Be > ctx->maxBe) return fail(ctx, "effective batch %d exceeds the context's max_batch_eff %d", Be, ctx->maxBe);
    if (T < 1 || T > ctx->maxT) return fail(ctx, "frames %d exceed max_frames %d", T, ctx->maxT);
    if (N < 0 || N > ctx->maxNp) return fail(ctx, "num_steps %d exceeds %d", N, ctx->maxNp);
    if (p->prediction_type < 0 || p->prediction_type > 2) return fail(ctx, "bad prediction_type %d", p->prediction_type);
    if (!p->latents_dev || !p->context_dev) return fail(ctx, "latents_dev/context_dev must not be null");
    if (p->use_mask && !(p->init_latents_dev && p->edit_noise_dev && p->mask_dev)) return fail(ctx, "use_mask needs init_latents_dev, edit_noise_dev and mask_dev");
    if (p->use_step_noise == 1 && !p->step_noise_dev) return fail(ctx, "use_step_noise = 1 needs step_noise_dev");
    if (p->use_step_noise < 0 || p->use_step_noise > 2) return fail(ctx, "bad use_step_noise %d", p->use_step_noise);
    if (p->save_intermediate && !p->intermediates_dev) return fail(ctx, "save_intermediate needs intermediates_dev");
    if (prepare_only) {   // nothing to do when the step graph of this configuration exists
        if (N == 0) return 0;
        const bool grow = p->use_step_noise == 1 && (size_t)N * B * C * (size_t)rup(T, 32) > ctx->noise_cm_elems;
        if (!grow && ctx->gexec && loop_graph_key(ctx, p, ctx->noise_cm) == ctx->gkey) return 0;
    }
    if (set_band(ctx, T, T, s)) return -1;
    UGeo g = make_geo(ctx, Be, cfg ? B : 0, T, T);
    g.step_ptr = ctx->step_dev;
    if (cfg && ctx->cfg_share && !ctx->use_branches) g.Bc = B;
    const long long xs = (long long)C * g.Tp;

    TRACE("loop: begin");
    // per-clip, step-invariant work
    if (N > 0) {
        HIPCHK(hipMemcpyAsync(ctx->ts_dev, p->timesteps_host, N * sizeof(long long), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(ctx->coef_dev, p->coef_host, (size_t)N * 8 * sizeof(float), hipMemcpyHostToDevice, s));
        run_time_embed(ctx, N, s);
    }
    const long long cs = (long long)ctx->ctx_dim * g.Sp;
    const bool uncond_const = g.Bc > 0;   // the unconditional half's cross-attention is the constant c2[blk]: its context and K/V are never read
    if (cfg) {  // uncond FIRST (diffusion.py:397-400): null_cond_emb repeated over (B, S)
        if (!uncond_const) launch_fill_cm_vec(ctx->null_cond, ctx->CTX, B, T, ctx->ctx_dim, g.Sp, cs, s);
        launch_tm_to_cm(p->context_dev, ctx->CTX + (long long)B * cs, B, T, ctx->ctx_dim, g.Sp, cs, s);
    } else {
        launch_tm_to_cm(p->context_dev, ctx->CTX, B, T, ctx->ctx_dim, g.Sp, cs, s);
    }
    TRACE("loop: ctx done");
    if (uncond_const) run_kv(ctx, B, B, T, g.Sp, s); else run_kv(ctx, 0, Be, T, g.Sp, s);
    TRACE("loop: kv launched");
    launch_tm_to_cm(p->latents_dev, ctx->x_cm, B, T, C, g.Tp, xs, s);
    if (p->use_mask) {
        launch_tm_to_cm(p->init_latents_dev, ctx->init_cm, B, T, C, g.Tp, xs, s);
        launch_tm_to_cm(p->edit_noise_dev, ctx->enoise_cm, B, T, C, g.Tp, xs, s);
        launch_tm_to_cm(p->mask_dev, ctx->mask_cm, B, T, C, g.Tp, xs, s);
    }
    if (p->use_step_noise == 2) {   // Philox key of this call's eta noise (the graph reads it from device memory)
        ctx->seed_host[0] = (unsigned)(p->noise_seed & 0xffffffffu); ctx->seed_host[1] = (unsigned)(p->noise_seed >> 32);
        HIPCHK(hipMemcpyAsync(ctx->seed_dev, ctx->seed_host, 2 * sizeof(unsigned), hipMemcpyHostToDevice, s));
    }
    if (p->use_step_noise == 1 && N > 0) {
        const size_t need = (size_t)N * B * xs;
        if (need > ctx->noise_cm_elems) {
            HIPCHK(hipStreamSynchronize(s));
            if (drealloc(ctx, &ctx->noise_cm, need)) return -1;
            ctx->noise_cm_elems = need;
        }
        launch_tm_to_cm(p->step_noise_dev, ctx->noise_cm, N * B, T, C, g.Tp, xs, s);
    }
    HIPCHK(hipMemsetAsync(ctx->step_dev, 0xFF, sizeof(int), s));  // step = -1
    HIPCHK(hipMemsetAsync(ctx->status_dev, 0, 2 * sizeof(int), s));   // numeric status of THIS loop (the eager warm-up step below sees step 0's own data)

    // per-step graph
    SchedArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.eps = ctx->eps_cm; sa.eps_bstride = xs; sa.pitch = g.Tp; sa.B = B; sa.T = T; sa.C = C; sa.cfg = cfg ? 1 : 0;
    sa.guidance_scale = p->guidance_scale; sa.guidance_rescale = (cfg && p->guidance_rescale > 0.f) ? p->guidance_rescale : 0.f;
    sa.rescale_part = ctx->rescale_part; sa.rescale_nblk = 16; sa.prediction_type = p->prediction_type;
    sa.coef = ctx->coef_dev; sa.step_ptr = ctx->step_dev; sa.x = ctx->x_cm; sa.x_bstride = xs;
    sa.step_noise = p->use_step_noise == 1 ? ctx->noise_cm : nullptr;
    sa.noise_seed = p->use_step_noise == 2 ? ctx->seed_dev : nullptr;
    sa.noise_elem0 = (unsigned)((long long)p->noise_batch_offset * T * C);
    sa.init = p->use_mask ? ctx->init_cm : nullptr; sa.edit_noise = p->use_mask ? ctx->enoise_cm : nullptr;
    sa.mask = p->use_mask ? ctx->mask_cm : nullptr;
    sa.inter = p->save_intermediate ? p->intermediates_dev : nullptr; sa.latent_scale = p->latent_scale;
    sa.status = ctx->status_dev;

    OutSchedArgs osa;
    memset(&osa, 0, sizeof osa);
    osa.x = ctx->P.p; osa.gn_part = ctx->P.st; osa.gn_gamma = ctx->out_g; osa.gn_beta = ctx->out_b;
    osa.w4 = ctx->conv_out.w4[0]; osa.ws = sp_on(ctx, ctx->out_split) ? ctx->conv_out.ws[0] : nullptr; osa.bias = ctx->conv_out.bias; osa.coef = ctx->coef_dev; osa.step_ptr = ctx->step_dev;
    osa.lat = ctx->x_cm; osa.step_noise = sa.step_noise; osa.noise_seed = sa.noise_seed; osa.noise_elem0 = sa.noise_elem0; osa.init = sa.init; osa.edit_noise = sa.edit_noise; osa.mask = sa.mask;
    osa.inter = sa.inter; osa.x_bstride = g.hs; osa.gn_part_bstride = g.sts; osa.lat_bstride = xs;
    osa.pitch = g.Tp; osa.T = T; osa.B = B; osa.Cin = MC; osa.Cout = C; osa.gn_nparts = g.np; osa.cfg = cfg ? 1 : 0;
    osa.prediction_type = p->prediction_type; osa.guidance_scale = p->guidance_scale; osa.guidance_rescale = sa.guidance_rescale;
    osa.latent_scale = p->latent_scale;
    osa.status = ctx->status_dev;
    static const bool no_fuse = dev_env("SAID_NO_FUSE_SCHED") != nullptr;
    const bool fused = !no_fuse && !ctx->use_branches && ctx->conv_out.w4[0] && out_sched_supports(osa);
    if (fused) g.out_sched = &osa;

    if (N > 0) {
        // steps per graph: consecutive denoise steps captured back to back in ONE graph (the device-side step counter
        // makes the copies distinct): N / spg launches of the spg-step graph cover the loop, and a second graph holding
        // the N % spg remaining steps finishes it (prime N, e.g. 997 = 99 x 10 + 7)
        const int spg = steps_per_graph(ctx, N);   // measured: ~6 us per graph launch boundary; 10 steps per graph recovered 1.2 % at B=1 (round 1), 50 another 0.4-0.5 % of a long loop (round 6)
        const int rem = N % spg;
        const std::vector<long long> key = loop_graph_key(ctx, p, sa.step_noise);
        if (!ctx->gexec || key != ctx->gkey) {
            drop_graphs(ctx);
            const bool fold_step = !(ctx->use_branches && Be >= 2);   // conv_in advances the step counter itself
            // Eager warm-up of the exact step sequence first: the first launch of a kernel inside a
            // stream capture hangs on ROCm 7.2 (lazy per-kernel initialisation is not capturable).
            // The step is run on step index 0 and its effect on the latents is undone afterwards.
            TRACE("loop: warmup begin");
            if (fold_step) g.step_inc = ctx->step_dev; else launch_step_advance(ctx->step_dev, s);
            if (trace_on()) { hipError_t e = hipStreamSynchronize(s); fprintf(stderr, "[said] step_advance -> %s\n", hipGetErrorString(e)); fflush(stderr); }
            run_unet(ctx, g, s);
            TRACE("loop: warmup unet launched");
            if (!fused) {
                if (sa.guidance_rescale > 0.f) launch_rescale_partials(sa, ctx->rescale_part, s);
                launch_sched_step(sa, s);
            }
            launch_tm_to_cm(p->latents_dev, ctx->x_cm, B, T, C, g.Tp, xs, s);
            HIPCHK(hipMemsetAsync(ctx->step_dev, 0xFF, sizeof(int), s));
            HIPCHK(hipStreamSynchronize(s));
            TRACE("loop: warmup synced");
            // A launch helper that refused a shape launched NOTHING (launch_fault / launch_err): a graph captured from the same schedule would be
            // missing that kernel and — cached under this key — replay silently wrong on every later call.  So the faults of the warm-up and of
            // each capture are consumed HERE, on the thread that drove them, in the prepare-only path too, and nothing is cached.
            auto loop_fault = [&]() -> bool { return launch_fault_peek() != nullptr || !ctx->launch_err.empty(); };
            if (loop_fault()) { drop_graphs(ctx); LAUNCHCHK(); }
            // capture on a private stream: the caller's stream may be the legacy default stream,
            // which cannot be captured; the instantiated graph is then replayed on the caller's stream
            if (ensure_cap_streams(ctx)) return -1;
            hipStream_t cs = ctx->cap_stream;
            auto capture_steps = [&](int nsteps, hipGraph_t* gr, hipGraphExec_t* ge) -> int {
                HIPCHK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
                for (int rep = 0; rep < nsteps; ++rep) {
                    if (!fold_step) launch_step_advance(ctx->step_dev, cs);
                    if (ctx->use_branches && Be >= 2) {
                        // The two halves of the UNet batch (unconditional / conditional under CFG) are independent until
                        // the scheduler: capture them as parallel branches so their per-kernel latencies overlap.
                        UGeo ga = g, gb = g;
                        ga.Be = Be / 2; ga.b0 = 0;
                        gb.Be = Be - Be / 2; gb.b0 = Be / 2;
                        HIPCHK(hipEventRecord(ctx->ev_fork, cs));
                        HIPCHK(hipStreamWaitEvent(ctx->cap_stream2, ctx->ev_fork, 0));
                        run_unet(ctx, ga, cs);
                        run_unet(ctx, gb, ctx->cap_stream2);
                        HIPCHK(hipEventRecord(ctx->ev_join, ctx->cap_stream2));
                        HIPCHK(hipStreamWaitEvent(cs, ctx->ev_join, 0));
                    } else {
                        run_unet(ctx, g, cs);
                    }
                    if (!fused) {
                        if (sa.guidance_rescale > 0.f) launch_rescale_partials(sa, ctx->rescale_part, cs);
                        launch_sched_step(sa, cs);
                    }
                }
                TRACE("loop: capture recorded");
                hipError_t e = hipStreamEndCapture(cs, gr);
                if (e != hipSuccess) return fail(ctx, "hipStreamEndCapture: %s", hipGetErrorString(e));
                HIPCHK(hipGraphInstantiate(ge, *gr, nullptr, nullptr, 0));
                return 0;
            };
            if (capture_steps(spg, &ctx->graph, &ctx->gexec)) { drop_graphs(ctx); release_cap_streams(ctx); return -1; }
            if (loop_fault()) { drop_graphs(ctx); release_cap_streams(ctx); LAUNCHCHK(); }
            if (rem > 0 && capture_steps(rem, &ctx->graph_rem, &ctx->gexec_rem)) { drop_graphs(ctx); release_cap_streams(ctx); return -1; }
            if (loop_fault()) { drop_graphs(ctx); release_cap_streams(ctx); LAUNCHCHK(); }
            size_t nn = 0;
            (void)hipGraphGetNodes(ctx->graph, nullptr, &nn);
            ctx->gnodes = (int)nn / spg;
            ctx->gspg = spg;
            ctx->gkey = key;
            release_cap_streams(ctx);
            TRACE("loop: graph instantiated");
        }
        if (prepare_only) return 0;
        for (int k = 0; k < N / ctx->gspg; ++k) HIPCHK(hipGraphLaunch(ctx->gexec, s));
        if (rem > 0) HIPCHK(hipGraphLaunch(ctx->gexec_rem, s));
    }
    TRACE("loop: graphs launched");
    launch_finish(ctx->x_cm, xs, g.Tp, B, T, C, p->latent_scale, p->latents_dev, p->result_dev, s, ctx->dbg_stop < 0 ? ctx->status_dev : nullptr);
    LAUNCHCHK();
    HIPCHK(hipGetLastError());
    return 0;
}

int said_denoise_loop(said_ctx* ctx, const said_loop_params* p, void* stream) { return loop_impl(ctx, p, stream, false); }

int said_loop_prepare(said_ctx* ctx, const said_loop_params* p, void* stream) { return loop_impl(ctx, p, stream, true); }

int said_ddim_step(said_ctx* ctx, const float* eps_dev, const float* eps_uncond_dev, float guidance_scale,
                   const float* sample_dev, const float* coef_host, int prediction_type, const float* step_noise_dev,
                   const float* init_latents_dev, const float* edit_noise_dev, const float* mask_dev, float* prev_sample_dev,
                   int64_t n, void* stream) {
    if (!ctx) return -1;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (prediction_type < 0 || prediction_type > 2) return fail(ctx, "bad prediction_type %d", prediction_type);
    if (mask_dev && !(init_latents_dev && edit_noise_dev)) return fail(ctx, "mask needs init_latents_dev and edit_noise_dev");
    HIPCHK(hipMemcpyAsync(ctx->coef1_dev, coef_host, 8 * sizeof(float), hipMemcpyHostToDevice, s));
    if (n > 0)
        launch_ddim_flat(eps_dev, eps_uncond_dev, guidance_scale, sample_dev, ctx->coef1_dev, prediction_type, step_noise_dev,
                         init_latents_dev, edit_noise_dev, mask_dev, prev_sample_dev, n, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int said_axpby(said_ctx* ctx, const float* a_host, const float* x_dev, const float* c_host, const float* y_dev, float* out_dev,
               int batch, int64_t n_per_batch, void* stream) {
    if (!ctx) return -1;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (batch < 1 || batch > ctx->maxNp) return fail(ctx, "said_axpby: batch %d out of range", batch);
    // coefficient staging shares the (otherwise idle here) E0 table: 2 * batch floats
    std::vector<float> ac(2 * (size_t)batch, 0.f);
    for (int b = 0; b < batch; ++b) { ac[b] = a_host[b]; ac[batch + b] = (y_dev && c_host) ? c_host[b] : 0.f; }
    HIPCHK(hipMemcpyAsync(ctx->axpby_coef, ac.data(), ac.size() * sizeof(float), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    if (n_per_batch > 0) launch_axpby(ctx->axpby_coef, x_dev, ctx->axpby_coef + batch, y_dev, out_dev, batch, n_per_batch, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int said_debug_clocks(said_ctx* ctx, int enable, long long* out_host /* [64][8][8] or null */) {
    if (!ctx) return -1;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->clk_on = enable != 0;
    if (out_host) {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(out_host, ctx->clk_dev, 64 * 128 * sizeof(long long), hipMemcpyDeviceToHost));
    }
    return 0;
}

int said_debug_option(said_ctx* ctx, const char* name, long long value) {
    if (!ctx || !name) return -1;
    const std::string k = name;
    if (k == "unet_tgemm_min_tokens") {
        ctx->unet_tgemm_min_tokens = value < 0 ? 3000 : value;
        ctx->unet_fgemm_min_tokens = value < 0 ? 10000 : value;
    } else if (k == "audio_chunk") {
        if (value < 1) return fail(ctx, "audio_chunk must be >= 1");
        ctx->audio_chunk = (int)value;
    } else if (k == "steps_per_graph") {
        if (value < 1) return fail(ctx, "steps_per_graph must be >= 1");
        ctx->spg_limit = (int)value;
    } else if (k == "hybrid") {
        ctx->hybrid = value != 0;
    } else if (k == "out_tm") {
        ctx->out_tm = (int)value;
    } else if (k == "mt_mid") {
        ctx->mt_mid = value != 0;
    } else if (k == "mt_wgs") {
        ctx->mt_wgs = (int)value;
    } else if (k == "tgemm_sb") {
        ctx->tgemm_sb = value != 0;
    } else if (k == "unet_nb") {
        ctx->unet_nb = (int)value;
    } else if (k == "unet_nb_model") {
        ctx->unet_nb_model = value != 0;
    } else if (k == "f32_out1_tm") {
        ctx->f32_out1_tm = value != 0;
    } else if (k == "xgemm_clk") {   // shader-clock stamps of the token-major-activation GEMMs (-DSAID_CLK_STAMPS builds); read with said_debug_clocks
        ctx->xclk_on = value != 0;
    } else if (k == "xgemm_ntw") {
        ctx->xgemm_ntw = (int)value;
    } else if (k == "xgemm_dbg") {   // knock-out timing experiments (1: no epilogue, 2: no k loop, 4: no residual, 8: no source tile): WRONG results,
        if (!dev_env("SAID_DEV")) return fail(ctx, "xgemm_dbg needs a -DSAID_DEV_KNOBS build with SAID_DEV=1");   // so refused by the shipped library
        ctx->xgemm_dbg = (int)value;
    } else if (k == "tm_acts") {
        ctx->tm_acts = value < 0 ? -1 : (value != 0);
    } else if (k == "rgemm") {
        ctx->rgemm = value < 0 ? -1 : (value != 0);
    } else if (k == "gemm_presplit") {
        ctx->gemm_presplit = value < 0 ? -1 : (value != 0);
    } else if (k == "gemm_split") {
        ctx->gemm_split = value < 0 ? -1 : (value != 0);
    } else if (k == "attn_split") {
        ctx->attn_split = value < 0 ? -1 : (value != 0);
    } else if (k == "attn_presplit") {
        ctx->attn_presplit = value < 0 ? -1 : (value != 0);
    } else if (k == "out_split") {
        ctx->out_split = value < 0 ? -1 : (value != 0);
    } else if (k == "ugemm_split") {
        ctx->ugemm_split = value < 0 ? -1 : (value != 0);
    } else if (k == "chain_coef") {
        ctx->chain_coef = value < 0 ? -1 : (value != 0);
    } else if (k == "tgemm_direct") {
        ctx->tgemm_direct = value < 0 ? -1 : (value != 0);
    } else if (k == "kconv") {
        ctx->kconv = value < 0 ? -1 : (value == 2 ? 2 : (value != 0));
    } else if (k == "kconv_max_tiles") {
        ctx->kconv_max_tiles = value;
    } else if (k == "st_chain") {
        ctx->st_chain = value < 0 ? -1 : (value != 0);
    } else if (k == "attn_2q") {
        ctx->attn_2q = value < 0 ? -1 : (value != 0);
    } else if (k == "attn_ks") {
        ctx->attn_ks_force = (int)value;
    } else if (k == "st_chain_slices") {
        ctx->st_chain_slices = value < 0 ? -1 : ((value == 3 || value == 2) ? (int)value : 1);   // 2: two slices wherever a sliced launch is possible (<= 128 tiles)
    } else if (k == "st_chain_dbg") {
        ctx->st_chain_dbg = value != 0;
    } else if (k == "st_chain_bf16") {
        ctx->st_chain_bf16 = value < 0 ? -1 : (value != 0);
    } else if (k == "st_chain_large") {
        ctx->st_chain_large = value != 0;
    } else if (k == "st_chain_max_tiles") {
        ctx->st_chain_max_tiles = value;
    } else if (k == "battn") {
        ctx->battn = value < 0 ? -1 : (int)value;   // 0: off, 4 / 8: query tiles per workgroup (experiments), else on
    } else {
        return fail(ctx, "said_debug_option: unknown option %s", name);
    }
    ctx->gkey.clear();   // the captured step graph may hold the other schedule
    return 0;
}

long long said_debug_get(const said_ctx* ctx, const char* name) {
    if (!ctx || !name) return -1;
    const std::string k = name;
    if (k == "n_set_weight") return ctx->n_set_weight;
    if (k == "n_audio_clips") return (int)ctx->n_audio_clips;
    if (k == "unet_tgemm_min_tokens") return ctx->bf16_mode ? ctx->unet_tgemm_min_tokens : ctx->unet_fgemm_min_tokens;
    if (k == "audio_chunk") return ctx->audio_chunk;
    if (k == "steps_per_graph") return ctx->spg_limit;
    if (k == "tm_acts") return ctx->tm_acts;
    if (k == "gemm_split") return (!ctx->bf16_mode && sp_on(ctx, ctx->gemm_split)) ? 1 : 0;
    if (k == "ugemm_split") return (!ctx->bf16_mode && sp_on(ctx, ctx->ugemm_split)) ? 1 : 0;
    if (k == "st_chain") return (!ctx->bf16_mode && sp_on(ctx, ctx->st_chain)) ? 1 : 0;
    if (k == "n_stchain") return ctx->n_stchain;
    if (k == "st_chain_bf16") return (ctx->bf16_mode && ctx->st_chain_bf16 != 0) ? 1 : 0;
    if (k == "attn_split") return (!ctx->bf16_mode && sp_on(ctx, ctx->attn_split)) ? 1 : 0;   // 1: fp32-mode attention products run on split-fp16 operands
    if (k == "rgemm") return ctx->rgemm;
    if (k == "n_rgemm") return ctx->n_rgemm;
    if (k == "n_xgemm") return ctx->n_xgemm;
    if (k == "pool_probed") return (ctx->device >= 0 && ctx->device < POOL_DEVICES) ? g_pool_probed[ctx->device] : -1;
    return -1;
}

int said_philox_normal(said_ctx* ctx, uint64_t seed, int step0, int nsteps, int64_t n_per_step, float* out_dev, void* stream) {
    if (!ctx) return -1;
    if (nsteps < 0 || n_per_step < 0 || n_per_step > 0xffffffffLL || !out_dev) return fail(ctx, "said_philox_normal: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->seed_host[0] = (unsigned)(seed & 0xffffffffu); ctx->seed_host[1] = (unsigned)(seed >> 32);
    HIPCHK(hipMemcpyAsync(ctx->seed_dev, ctx->seed_host, 2 * sizeof(unsigned), hipMemcpyHostToDevice, s));
    launch_philox_normal(ctx->seed_dev, step0, nsteps, n_per_step, out_dev, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int said_debug_stop_after(said_ctx* ctx, int n) {
    if (!ctx) return -1;
    ctx->dbg_stop = n;
    ctx->dbg_count = 0;
    ctx->gkey.clear();   // a cached step graph holds the other launch count
    return 0;
}

int said_debug_read(said_ctx* ctx, const char* name, float* out_host, int64_t n) {
    if (!ctx) return -1;
    HIPCHK(hipSetDevice(ctx->device));
    const std::map<std::string, const float*> m = {
        {"H0", ctx->H0.p}, {"H1", ctx->H1.p}, {"P", ctx->P.p}, {"Q", ctx->Q.p}, {"M", ctx->M.p},
        {"stH0", ctx->H0.st}, {"stH1", ctx->H1.st}, {"stP", ctx->P.st}, {"stQ", ctx->Q.st}, {"stM", ctx->M.st},
        {"X1", ctx->X1}, {"X2", ctx->X2}, {"X3", ctx->X3}, {"O", ctx->O}, {"QK", ctx->QK}, {"VT", ctx->VT}, {"F", ctx->F},
        {"KV", ctx->KV}, {"CTX", ctx->CTX}, {"EO", ctx->EO}, {"E0", ctx->E0}, {"E1", ctx->E1}, {"E2", ctx->E2},
        {"x", ctx->x_cm}, {"eps", ctx->eps_cm}, {"aH", ctx->aH}, {"aT", ctx->aT}, {"aX", ctx->aX}, {"aPOS", ctx->aPOS},
        {"aQK", ctx->aQK}, {"aO", ctx->aO}, {"aF", ctx->aF}, {"abufA", ctx->abufA}, {"abufB", ctx->abufB}};
    auto it = m.find(name);
    if (it == m.end() || !it->second) return fail(ctx, "said_debug_read: unknown or unallocated buffer %s", name);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_host, it->second, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// ---- workspace inspection (tests / race hunting): every (max_batch_eff, max_frames)-sized buffer by index --------------------------------
static const char* ws_name(const said_ctx* c, const void* p) {
    const std::pair<const void*, const char*> t[] = {
        {c->x_cm, "x"}, {c->eps_cm, "eps"}, {c->H0.p, "H0"}, {c->H1.p, "H1"}, {c->P.p, "P"}, {c->Q.p, "Q"}, {c->M.p, "M"},
        {c->H0.st, "stH0"}, {c->H1.st, "stH1"}, {c->P.st, "stP"}, {c->Q.st, "stQ"}, {c->M.st, "stM"},
        {c->X1, "X1"}, {c->X2, "X2"}, {c->X3, "X3"}, {c->O, "O"}, {c->QK, "QK"}, {c->VT, "VT"}, {c->F, "F"}, {c->KV, "KV"}, {c->KVT, "KVT"}, {c->chain_part, "chain_part"}, {c->CTX, "CTX"},
        {c->E0, "E0"}, {c->E1, "E1"}, {c->E2, "E2"}, {c->EO, "EO"}, {c->ts_dev, "ts"}, {c->coef_dev, "coef"}, {c->axpby_coef, "axpby_coef"},
        {c->band_lo, "band_lo"}, {c->band_hi, "band_hi"}, {c->init_cm, "init_cm"}, {c->enoise_cm, "enoise_cm"}, {c->mask_cm, "mask_cm"},
        {c->rescale_part, "rescale_part"}, {c->uPA, "uPA"}, {c->uPB, "uPB"}, {c->uPL, "uPL"}, {c->uPH, "uPH"}, {c->uPX, "uPX"}, {c->gn_coef, "gn_coef"},
        {c->H0.t, "tH0"}, {c->H1.t, "tH1"}, {c->P.t, "tP"}, {c->Q.t, "tQ"}, {c->M.t, "tM"}, {c->tX1, "tX1"}, {c->tX2, "tX2"}, {c->tO, "tO"}, {c->tF, "tF"}};
    for (const auto& e : t) if (e.first == p) return e.second;
    return "?";
}
int said_debug_ws_count(const said_ctx* ctx) { return ctx ? (int)ctx->ws_allocs.size() : -1; }
int said_debug_ws_info(said_ctx* ctx, int idx, void** ptr_out, long long* bytes_out, const char** name_out) {
    if (!ctx || idx < 0 || idx >= (int)ctx->ws_allocs.size()) return -1;
    void* p = ctx->ws_allocs[idx];
    if (ptr_out) *ptr_out = p;
    if (bytes_out) { auto it = ctx->alloc_bytes.find(p); *bytes_out = it == ctx->alloc_bytes.end() ? 0 : (long long)it->second; }
    if (name_out) *name_out = ws_name(ctx, p);
    return 0;
}
int said_debug_ws_fill(said_ctx* ctx, int byte_value) {
    if (!ctx) return -1;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    for (void* p : ctx->ws_allocs) {
        auto it = ctx->alloc_bytes.find(p);
        if (it != ctx->alloc_bytes.end()) HIPCHK(hipMemset(p, byte_value, it->second));
    }
    HIPCHK(hipDeviceSynchronize());
    ctx->band_T = ctx->band_S = -1;   // the band tables were part of it
    drop_graphs(ctx);
    return 0;
}
int said_debug_ws_copy(said_ctx* ctx, int idx, void* dst_dev, long long bytes, void* stream) {
    if (!ctx || idx < 0 || idx >= (int)ctx->ws_allocs.size() || !dst_dev) return -1;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(dst_dev, ctx->ws_allocs[idx], (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int said_profile_unet(said_ctx* ctx, int Be, int T, int cfg_clips, int reps, int max_stages, float* us_out, double* bytes_out, double* flops_out,
                      int* kind_out, int* epi_out, int* nb_out, int* ks_out, int* n_stages_out, void* stream) {
    if (check_ready(ctx)) return -1;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    if (Be < 1 || Be > ctx->maxBe || T < 1 || T > ctx->maxT) return fail(ctx, "said_profile_unet: shape out of range");
    if (set_band(ctx, T, T, s)) return -1;
    if (cfg_clips < 0 || (cfg_clips > 0 && Be != 2 * cfg_clips)) return fail(ctx, "said_profile_unet: cfg_clips must be 0 or batch_eff / 2");
    UGeo g = make_geo(ctx, Be, cfg_clips, T, T);
    g.emb_b_stride = 0;
    if (cfg_clips > 0 && ctx->cfg_share && !ctx->use_branches) g.Bc = cfg_clips;   // the schedule the guided loop runs
    run_unet(ctx, g, s);  // eager warm-up: kernels must not see their first launch inside a capture
    HIPCHK(hipStreamSynchronize(s));
    if (ensure_cap_streams(ctx)) return -1;
    // pass 0: log the schedule without launching anything
    ctx->stage_log.clear();
    ctx->log_on = true; ctx->dbg_only = -1; ctx->dbg_stop = 0; ctx->dbg_count = 0;
    run_unet(ctx, g, ctx->cap_stream);
    ctx->log_on = false; ctx->dbg_stop = -1;
    const int n = (int)ctx->stage_log.size();
    if (n_stages_out) *n_stages_out = n;
    if (n > max_stages) return fail(ctx, "said_profile_unet: %d stages > max_stages %d", n, max_stages);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int k = 0; k < n; ++k) {
        // `reps` back-to-back launches of stage k captured in one graph, timed with HIP events on `s`
        hipGraph_t gr = nullptr;
        hipGraphExec_t ge = nullptr;
        ctx->dbg_only = k;
        HIPCHK(hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeThreadLocal));
        for (int r = 0; r < reps; ++r) { ctx->dbg_count = 0; run_unet(ctx, g, ctx->cap_stream); }
        hipError_t e = hipStreamEndCapture(ctx->cap_stream, &gr);
        ctx->dbg_only = -1;
        if (e != hipSuccess) return fail(ctx, "profile capture: %s", hipGetErrorString(e));
        HIPCHK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        HIPCHK(hipGraphLaunch(ge, s));  // warm-up
        HIPCHK(hipEventRecord(e0, s));
        HIPCHK(hipGraphLaunch(ge, s));
        HIPCHK(hipEventRecord(e1, s));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        const auto& si = ctx->stage_log[k];
        us_out[k] = ms * 1000.f / reps;
        bytes_out[k] = si.bytes; flops_out[k] = si.flops; kind_out[k] = si.kind; epi_out[k] = si.epi; nb_out[k] = si.NB; ks_out[k] = si.KS;
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(gr);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    release_cap_streams(ctx);   // a live stream pins a hardware queue (the clip groups need them): DESIGN.md 5.1
    return 0;
}

int said_graph_num_nodes(const said_ctx* ctx) { return ctx ? ctx->gnodes : 0; }

int said_loop_progress(said_ctx* ctx, int* steps_done) {
    if (!ctx || !steps_done) return -1;
    DeviceRestore restore_device;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t ps = nullptr;   // (created per call: a live stream holds one of the device's few hardware queues — DESIGN.md 5.1)
    HIPCHK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    int v = -1;
    hipError_t e = hipMemcpyAsync(&v, ctx->step_dev, sizeof(int), hipMemcpyDeviceToHost, ps);
    if (e == hipSuccess) e = hipStreamSynchronize(ps);
    (void)hipStreamDestroy(ps);
    if (e != hipSuccess) return -2;   // (no fail(): this is called from a polling thread while the loop's thread may be writing ctx->err — ADVICE r4)
    *steps_done = v + 1;        // the counter is -1 before the first step and k - 1 once step k - 1 has STARTED; its kernels complete in order
    return 0;
}

int said_loop_progress_reset(said_ctx* ctx) {
    if (!ctx) return -1;
    DeviceRestore restore_device;
    HIPCHK(hipSetDevice(ctx->device));
    const int m1 = -1;
    HIPCHK(hipMemcpy(ctx->step_dev, &m1, sizeof(int), hipMemcpyHostToDevice));   // synchronous: a poller started after this call never sees the previous loop's count
    return 0;
}

int said_set_precision(said_ctx* ctx, int mode) {
    if (!ctx) return -1;
    if (mode != SAID_PREC_FP32 && mode != SAID_PREC_BF16 && mode != SAID_PREC_FP32_STRICT) return fail(ctx, "said_set_precision: unknown mode %d", mode);
    if (mode != ctx->prec_mode) {
        ctx->prec_mode = mode;
        ctx->bf16_mode = mode == SAID_PREC_BF16;
        ctx->gkey.clear();   // the captured step graph holds the other kernels
    }
    return 0;
}
int said_get_precision(const said_ctx* ctx) { return ctx ? ctx->prec_mode : 0; }
int said_effective_precision(const said_ctx* ctx) { return !ctx ? 0 : (ctx->bf16_mode ? SAID_PREC_BF16 : (strict_f32(ctx) ? SAID_PREC_FP32_STRICT : SAID_PREC_FP32)); }
const char* said_precision_note(const said_ctx* ctx) { return ctx ? ctx->split_note.c_str() : ""; }

int said_numeric_status(said_ctx* ctx, void* stream, int* first_bad_step, int* result_nonfinite) {
    if (!ctx) return -1;
    DeviceRestore restore_device;
    HIPCHK(hipSetDevice(ctx->device));
    int v[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(v, ctx->status_dev, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (first_bad_step) *first_bad_step = v[0] - 1;
    if (result_nonfinite) *result_nonfinite = v[1];
    return 0;
}

double said_unet_algorithmic_bytes(int Be, int T, int bytes_per_elem) {
    // W + Be*T*A, A = 63,232 B/token at fp32
    return 6992672.0 * bytes_per_elem + (double)Be * T * (63232.0 * bytes_per_elem / 4.0);
}
double said_unet_algorithmic_flops(int Be, int T) {
    // banded cross-attention, cross K/V precomputed
    const double lin = 5492736.0 - 1179648.0;
    return 2.0 * Be * (lin * T + 1536.0 * T * (double)T + 4608.0 * T) + 2.0 * Be * 1474560.0;
}

// --------------------------------------------------------------------------------------------
// Audio encoder
// --------------------------------------------------------------------------------------------
int said_audio_encode(said_ctx* ctx, const float* wav_dev, int B, int Ta, int num_frames, int apply_proj, float* out_dev,
                      int* out_frames, void* stream) {
    if (check_ready(ctx)) return -1;
    if (!ctx->has_audio) return fail(ctx, "audio_encoder.* weights were not loaded");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->n_audio_clips += B;
    int L[7];
    {
        int len = Ta;
        for (int i = 0; i < 7; ++i) {
            len = (len - ctx->w2v_kernel[i]) / ctx->w2v_stride[i] + 1;
            if (len < 1) return fail(ctx, "waveform of %d samples is too short for the feature extractor", Ta);
            L[i] = len;
        }
    }
    const int Fr = num_frames > 0 ? num_frames : L[6];
    if (out_frames) *out_frames = Fr;
    const int Fp = rup(Fr, 32);
    if (apply_proj && !ctx->has_audio_proj) return fail(ctx, "apply_proj requested but audio_proj_layer.* was not loaded");
    const int out_dim = apply_proj ? ctx->ctx_dim : W2V_H;
    // workspace: ping-pong conv buffers + token-domain buffers, for `chunk` clips at a time
    // clips per pass: the 65 MB/clip conv0 activation is what bounds it (32 clips = 2.1 GB of 288 GB); larger launches
    // amortise the 377 MB of encoder weights over more tokens
    const int chunk = std::min(B, ctx->audio_chunk);
    const size_t eA = (size_t)chunk * W2V_CONV * rup(L[0], 32), eB = (size_t)chunk * W2V_CONV * rup(L[1], 32);
    const size_t tok = (size_t)chunk * Fp;
    const size_t tw = std::max<size_t>(W2V_H, (size_t)ctx->ctx_dim);   // aT also receives the audio_proj_layer output (ctx_dim wide)
    if (eA > ctx->abuf_elems[0] || eB > ctx->abuf_elems[1] || tok > ctx->a_tok_elems) {
        HIPCHK(hipStreamSynchronize(s));   // the buffers being replaced may still be in use by an earlier call
        if (eA > ctx->abuf_elems[0]) { if (drealloc(ctx, &ctx->abufA, eA)) return -1; ctx->abuf_elems[0] = eA; }
        if (eB > ctx->abuf_elems[1]) { if (drealloc(ctx, &ctx->abufB, eB)) return -1; ctx->abuf_elems[1] = eB; }
        if (tok > ctx->a_tok_elems) {
            if (drealloc(ctx, &ctx->aX, tok * W2V_CONV) || drealloc(ctx, &ctx->aH, tok * W2V_H) || drealloc(ctx, &ctx->aT, tok * tw) ||
                drealloc(ctx, &ctx->aO, tok * 2 * W2V_H) || drealloc(ctx, &ctx->aQK, tok * 2 * W2V_H) || drealloc(ctx, &ctx->aVT, tok * W2V_H) ||
                drealloc(ctx, &ctx->aF, tok * W2V_FFN) || drealloc(ctx, &ctx->aPOS, tok * W2V_H))
                return -1;
            ctx->a_tok_elems = tok;
        }
    }
    // bf16 mode (said_set_precision): token-major bf16 encoder on v_mfma_f32_32x32x16_bf16 (tgemm.hip).  conv0 + its
    // per-channel GroupNorm, the grouped positional convolution and the attention kernel are shared with the fp32 path.
    const bool bfa = ctx->bf16_mode && ctx->audio_bf16 && (!apply_proj || ctx->ctx_dim % 128 == 0) && (int)ctx->blayers.size() == ctx->w2v_layers;
    if (bfa) {
        const size_t e0 = (size_t)chunk * L[0] * W2V_CONV, e1 = (size_t)chunk * L[1] * W2V_CONV, tk = (size_t)chunk * Fr;
        const size_t xg = (size_t)chunk * 16 * (size_t)rup(Fr + ctx->posconv.taps, 8) * (W2V_H / 16) + 4096;   // per-group positional-conv operand
        if (e0 > ctx->b_conv_elems[0] || e1 > ctx->b_conv_elems[1] || tk > ctx->b_tok || xg > ctx->bXg_elems) {
            HIPCHK(hipStreamSynchronize(s));
            uint16_t** u;
            if (e0 > ctx->b_conv_elems[0]) { u = reinterpret_cast<uint16_t**>(&ctx->bA0); if (drealloc(ctx, u, e0 + 64)) return -1; ctx->b_conv_elems[0] = e0; }
            if (e1 > ctx->b_conv_elems[1]) { u = reinterpret_cast<uint16_t**>(&ctx->bA1); if (drealloc(ctx, u, e1 + 64)) return -1; ctx->b_conv_elems[1] = e1; }
            if (tk > ctx->b_tok) {
                if (drealloc(ctx, reinterpret_cast<uint16_t**>(&ctx->bX), tk * W2V_CONV) || drealloc(ctx, reinterpret_cast<uint16_t**>(&ctx->bHb), tk * W2V_H) ||
                    drealloc(ctx, reinterpret_cast<uint16_t**>(&ctx->bF), tk * W2V_FFN) || drealloc(ctx, reinterpret_cast<uint16_t**>(&ctx->bO), tk * W2V_H) ||
                    drealloc(ctx, &ctx->bH, tk * W2V_H) || drealloc(ctx, &ctx->bT, tk * std::max<size_t>(W2V_H, (size_t)ctx->ctx_dim)) ||
                    drealloc(ctx, &ctx->bPosT, tk * W2V_H))
                    return -1;
                ctx->b_tok = tk;
            }
            if (xg > ctx->bXg_elems) {
                if (drealloc(ctx, reinterpret_cast<uint16_t**>(&ctx->bXg), xg)) return -1;
                ctx->bXg_elems = xg;
            }
        }
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = std::min(chunk, B - b0);
            const int pitch0 = rup(L[0], 32);
            const long long bs0 = (long long)W2V_CONV * pitch0;
            // conv0 + GroupNorm + GELU straight to token-major bf16 (abufA, sized for the fp32 activation, serves as its scratch)
            if (!launch_conv0_gn_gelu_tm_bf16(wav_dev + (long long)b0 * Ta, ctx->c0_w, ctx->c0_g, ctx->c0_b, ctx->abufA, ctx->bA0, nb, Ta, W2V_CONV,
                                              ctx->w2v_kernel[0], ctx->w2v_stride[0], L[0], 1e-5f, s)) {
                launch_conv0(wav_dev + (long long)b0 * Ta, ctx->c0_w, ctx->abufA, nb, Ta, W2V_CONV, ctx->w2v_kernel[0], ctx->w2v_stride[0], L[0], pitch0, bs0, s);
                launch_rownorm_gelu(ctx->abufA, ctx->c0_g, ctx->c0_b, W2V_CONV, nb, L[0], pitch0, bs0, 1e-5f, s);
                launch_cm_to_tm_bf16(ctx->abufA, bs0, pitch0, ctx->bA0, (long long)L[0] * W2V_CONV, nb, L[0], W2V_CONV, s);
            }
            void* src = ctx->bA0;
            void* dst = ctx->bA1;
            for (int i = 1; i < 7; ++i) {   // Conv1d(512, 512, k, stride 2, no bias) + GELU as a GEMM with overlapping rows
                TGemmArgs a;
                memset(&a, 0, sizeof a);
                a.a = src; a.a_bs = (long long)L[i - 1] * W2V_CONV; a.lda = ctx->w2v_stride[i] * W2V_CONV;
                a.w = ctx->bw_conv[i]; a.act = 1;
                a.yb = dst; a.y_bs = (long long)L[i] * W2V_CONV; a.ldy = W2V_CONV;
                a.M = L[i]; a.N = W2V_CONV; a.K = ctx->w2v_kernel[i] * W2V_CONV;
                if (!launch_tgemm(a, nb, s)) ctx->launch_err = "audio encoder: token-major GEMM shape not served";
                std::swap(src, dst);
            }
            // interpolation to the frame count (wav2vec2.py:41-44) + feature_projection.layer_norm
            launch_interp_ln_tm(src, (long long)L[6] * W2V_CONV, L[6], ctx->bX, (long long)Fr * W2V_CONV, Fr, nb, W2V_CONV, ctx->fp_lng, ctx->fp_lnb, 1e-5f, s);
            const long long hsT = (long long)Fr * W2V_H;            // token-major batch stride
            const long long hs = (long long)W2V_H * Fp;             // channel-major batch stride (positional conv, attention operands)
            const long long tt = (long long)nb * ((Fr + 31) / 32);
            {   // feature_projection.projection
                TGemmArgs a;
                memset(&a, 0, sizeof a);
                a.a = ctx->bX; a.a_bs = (long long)Fr * W2V_CONV; a.lda = W2V_CONV; a.w = ctx->bw_fproj; a.bias = ctx->fproj.bias;
                a.yf = ctx->bH; a.y_bs = hsT; a.ldy = W2V_H; a.M = Fr; a.N = W2V_H; a.K = W2V_CONV;
                if (!launch_tgemm(a, nb, s)) ctx->launch_err = "audio encoder: token-major GEMM shape not served";
            }
            const int PK = ctx->posconv.taps, PG = 16, PCG = W2V_H / PG;
            if (ctx->pos_tgemm && ctx->bw_pos && PK % 2 == 0) {
                // positional conv embedding (wav2vec2: Conv1d(768, 768, k=128, padding=64, groups=16), last output dropped, GELU) as 16
                // GEMMs on the bf16 token-major kernel: group g's input channels laid out [R][48] with 64 zero rows in front, so
                // that output token t is row t's 128 x 48 contiguous elements times W_g (tap-major) — 150 GFLOP per 32 clips that
                // the grouped fp32 kernel ran at 40 TFLOP/s (7.5 of the encoder's 21 ms).  Epilogue: + bias, GELU, + hidden state.
                const int R = rup(Fr + PK, 8);
                launch_tm_to_group_bf16(ctx->bH, hsT, ctx->bXg, nb, Fr, PG, PCG, R, PK / 2, s);
                {   // ONE grouped launch (batch axis = (clip, group)): 16 launches of 160 workgroups left 40 % of the CUs idle (16 x 60 us)
                    TGemmArgs a;
                    memset(&a, 0, sizeof a);
                    a.a = ctx->bXg; a.a_bs = (long long)PG * R * PCG; a.a_gs = (long long)R * PCG; a.lda = PCG;
                    a.w = ctx->bw_pos; a.w_gs = (long long)64 * PK * PCG; a.bias = ctx->pos_bias_pad; a.act = 1;
                    a.res = ctx->bH; a.res_bs = hsT; a.ldr = W2V_H;
                    a.yf = ctx->bT; a.y_bs = hsT; a.ldy = W2V_H; a.n_store = PCG;
                    a.grp = PG; a.col_gs = PCG;
                    a.M = Fr; a.N = 64; a.K = PK * PCG;
                    if (!launch_tgemm(a, nb * PG, s)) ctx->launch_err = "audio encoder: token-major GEMM shape not served";
                }
                launch_ln_tm(ctx->bT, nullptr, ctx->bH, ctx->bHb, ctx->enc_lng, ctx->enc_lnb, (long long)nb * Fr, W2V_H, 1e-5f, s);
            } else {
            {   // positional conv embedding (grouped, fp32 channel-major kernel) on the projected features
                launch_tm_to_cm(ctx->bH, ctx->aH, nb, Fr, W2V_H, Fp, hs, s);
                GemmArgs a = mkargs(Fr, W2V_H / 16);
                a.groups = 16; a.ntiles_per_group = 2;
                a.nseg = 1;
                a.seg[0] = mkseg(ctx->aH, hs, Fp, W2V_H / 16, ctx->posconv.taps, ctx->posconv.taps / 2, 1, Fr, XF_NONE, ctx->posconv.w[0]);
                a.seg[0].c_group_stride = W2V_H / 16;
                a.bias = ctx->posconv.bias; a.act = ACT_GELU;
                a.y = ctx->aPOS; a.y_bstride = hs; a.y_pitch = Fp;
                launch_gemm(a, EPI_STORE, nb, tt * 32 <= 2048 ? 1 : 2, 8, s);
                launch_cm_to_tm(ctx->aPOS, ctx->bPosT, nb, Fr, W2V_H, Fp, hs, s);
            }
            launch_ln_tm(ctx->bH, ctx->bPosT, ctx->bH, ctx->bHb, ctx->enc_lng, ctx->enc_lnb, (long long)nb * Fr, W2V_H, 1e-5f, s);
            }
            for (int l = 0; l < ctx->w2v_layers; ++l) {
                const W2VLayer& ly = ctx->layers[l];
                const said_ctx::BLayer& bl = ctx->blayers[l];
                {   // q, k, v projections -> attn.hip's operand layout
                    TGemmArgs a;
                    memset(&a, 0, sizeof a);
                    a.a = ctx->bHb; a.a_bs = hsT; a.lda = W2V_H; a.w = bl.qkv; a.bias = ly.qkv.bias;
                    a.qk = ctx->aQK; a.vt = ctx->aVT; a.v_bs = hs; a.qk_n = 2 * W2V_H; a.head_dim = W2V_HD; a.rows = Fp; a.heads2 = 2 * W2V_HEADS;
                    a.v_pitch = Fp; a.M = Fr; a.N = 3 * W2V_H; a.K = W2V_H;
                    a.sb = ctx->tgemm_sb; a.direct = ctx->tgemm_direct != 0;
                    if (!launch_tgemm(a, nb, s)) ctx->launch_err = "audio encoder: token-major GEMM shape not served";
                }
                {
                    AttnArgs a;
                    a.qk = ctx->aQK; a.v = ctx->aVT; a.o = ctx->aO;
                    a.v_bstride = hs; a.o_bstride = 2 * hs; a.b0 = 0;
                    a.pitch = Fp; a.T = Fr; a.heads = W2V_HEADS; a.rows = Fp; a.scale = 0.125f;
                    const int aks = tt * W2V_HEADS <= 2048 ? 8 : -4;
                    if (aks == -4) {   // the key-split-free variant writes the out_proj operand itself: token-major bf16 [clip][frame][768]
                        a.o = reinterpret_cast<float*>(ctx->bO); a.o_bstride = Fr; a.o_mode = 2;
                    }
                    launch_attn(a, nb, W2V_HD, aks, s, 1);
                    if (aks != -4) launch_cm_to_tm_bf16(ctx->aO, 2 * hs, Fp, ctx->bO, hsT, nb, Fr, W2V_H, s);
                }
                {   // out_proj + residual, then layer_norm
                    TGemmArgs a;
                    memset(&a, 0, sizeof a);
                    // (row-wise GEMMs see the pass's clips as ONE [nb * frames][768] matrix: no per-clip tile padding, 600 = 4.7 tiles of 128)
                    a.a = ctx->bO; a.a_bs = hsT; a.lda = W2V_H; a.w = bl.out; a.bias = ly.out.bias;
                    a.res = ctx->bH; a.res_bs = hsT; a.ldr = W2V_H;
                    a.yf = ctx->bT; a.y_bs = hsT; a.ldy = W2V_H; a.M = nb * Fr; a.N = W2V_H; a.K = W2V_H;
                    a.sb = ctx->tgemm_sb; a.direct = ctx->tgemm_direct != 0;
                    if (!launch_tgemm(a, 1, s)) ctx->launch_err = "audio encoder: token-major GEMM shape not served";
                }
                launch_ln_tm(ctx->bT, nullptr, ctx->bH, ctx->bHb, ly.ln1g, ly.ln1b, (long long)nb * Fr, W2V_H, 1e-5f, s);
                {   // feed_forward.intermediate_dense + GELU
                    TGemmArgs a;
                    memset(&a, 0, sizeof a);
                    a.a = ctx->bHb; a.a_bs = hsT; a.lda = W2V_H; a.w = bl.ff1; a.bias = ly.ff1.bias; a.act = 1;
                    a.yb = ctx->bF; a.y_bs = (long long)Fr * W2V_FFN; a.ldy = W2V_FFN; a.M = nb * Fr; a.N = W2V_FFN; a.K = W2V_H;
                    a.sb = ctx->tgemm_sb; a.direct = ctx->tgemm_direct != 0;
                    if (!launch_tgemm(a, 1, s)) ctx->launch_err = "audio encoder: token-major GEMM shape not served";
                }
                {   // feed_forward.output_dense + residual, then final_layer_norm
                    TGemmArgs a;
                    memset(&a, 0, sizeof a);
                    a.a = ctx->bF; a.a_bs = (long long)Fr * W2V_FFN; a.lda = W2V_FFN; a.w = bl.ff2; a.bias = ly.ff2.bias;
                    a.res = ctx->bH; a.res_bs = hsT; a.ldr = W2V_H;
                    a.yf = ctx->bT; a.y_bs = hsT; a.ldy = W2V_H; a.M = nb * Fr; a.N = W2V_H; a.K = W2V_FFN;
                    a.sb = ctx->tgemm_sb; a.direct = ctx->tgemm_direct != 0;
                    if (!launch_tgemm(a, 1, s)) ctx->launch_err = "audio encoder: token-major GEMM shape not served";
                }
                const bool last = l + 1 == ctx->w2v_layers && !apply_proj;   // the last LayerNorm writes the (B, frames, 768) result itself
                launch_ln_tm(ctx->bT, nullptr, last ? out_dev + (long long)b0 * Fr * W2V_H : ctx->bH, ctx->bHb, ly.ln2g, ly.ln2b, (long long)nb * Fr, W2V_H, 1e-5f, s);
            }
            if (apply_proj) {   // diffusion.py:228-229
                TGemmArgs a;
                memset(&a, 0, sizeof a);
                a.a = ctx->bHb; a.a_bs = hsT; a.lda = W2V_H; a.w = ctx->bw_aproj; a.bias = ctx->aproj.bias;
                a.yf = out_dev + (long long)b0 * Fr * out_dim; a.y_bs = (long long)Fr * out_dim; a.ldy = out_dim; a.M = Fr; a.N = out_dim; a.K = W2V_H;
                if (!launch_tgemm(a, nb, s)) ctx->launch_err = "audio encoder: token-major GEMM shape not served";
            } else if (ctx->w2v_layers == 0) {
                HIPCHK(hipMemcpyAsync(out_dev + (long long)b0 * Fr * W2V_H, ctx->bH, (size_t)nb * Fr * W2V_H * sizeof(float), hipMemcpyDeviceToDevice, s));
            }
        }
        LAUNCHCHK();
        HIPCHK(hipGetLastError());
        return 0;
    }
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        // ---- feature extractor ----
        int pitch = rup(L[0], 32);
        long long bs = (long long)W2V_CONV * pitch;
        launch_conv0(wav_dev + (long long)b0 * Ta, ctx->c0_w, ctx->abufA, nb, Ta, W2V_CONV, ctx->w2v_kernel[0], ctx->w2v_stride[0], L[0], pitch, bs, s);
        launch_rownorm_gelu(ctx->abufA, ctx->c0_g, ctx->c0_b, W2V_CONV, nb, L[0], pitch, bs, 1e-5f, s);
        float* src = ctx->abufA;
        float* dst = ctx->abufB;
        for (int i = 1; i < 7; ++i) {
            const int po = rup(L[i], 32);
            const long long bo = (long long)W2V_CONV * po;
            GemmArgs a = mkargs(L[i], W2V_CONV);
            a.nseg = 1;
            a.seg[0] = mkseg(src, bs, pitch, W2V_CONV, ctx->w2v_kernel[i], 0, ctx->w2v_stride[i], L[i - 1], XF_NONE, ctx->aconv[i].w[0]);
            a.act = ACT_GELU;
            a.y = dst; a.y_bstride = bo; a.y_pitch = po;
            const LaunchCfg lc = pick_cfg((long long)nb * ((L[i] + 31) / 32), W2V_CONV / 32, false);
            launch_gemm(a, EPI_STORE, nb, lc.NB, lc.KS, s);
            std::swap(src, dst);
            pitch = po; bs = bo;
        }
        // ---- interpolation to the frame count (wav2vec2.py:41-44) ----
        const long long xs = (long long)W2V_CONV * Fp, hs = (long long)W2V_H * Fp;
        const float* feat = src; long long feat_bs = bs; int feat_pitch = pitch;
        if (num_frames > 0) {
            launch_interp_linear(src, ctx->aX, nb, W2V_CONV, L[6], Fr, pitch, Fp, bs, xs, s);
            feat = ctx->aX; feat_bs = xs; feat_pitch = Fp;
        }
        const long long tt = (long long)nb * ((Fr + 31) / 32);
        {   // feature_projection: LayerNorm(512) -> Linear(512, 768)
            GemmArgs a = mkargs(Fr, W2V_H);
            a.nseg = 1;
            a.seg[0] = mkseg(feat, feat_bs, feat_pitch, W2V_CONV, 1, 0, 1, Fr, XF_LN, ctx->fproj.w[0]);
            a.seg[0].ln_gamma = ctx->fp_lng; a.seg[0].ln_beta = ctx->fp_lnb; a.seg[0].ln_eps = 1e-5f;
            a.bias = ctx->fproj.bias;
            a.y = ctx->aH; a.y_bstride = hs; a.y_pitch = Fp;
            const LaunchCfg lc = pick_cfg(tt, W2V_H / 32);
            launch_gemm(a, EPI_STORE, nb, lc.NB, lc.KS, s);
        }
        {   // positional conv embedding: grouped Conv1d(k=128, pad=64, groups=16) + GELU; last frame dropped
            GemmArgs a = mkargs(Fr, W2V_H / 16);
            a.groups = 16; a.ntiles_per_group = 2;
            a.nseg = 1;
            a.seg[0] = mkseg(ctx->aH, hs, Fp, W2V_H / 16, ctx->posconv.taps, ctx->posconv.taps / 2, 1, Fr, XF_NONE, ctx->posconv.w[0]);
            a.seg[0].c_group_stride = W2V_H / 16;
            a.bias = ctx->posconv.bias; a.act = ACT_GELU;
            a.y = ctx->aPOS; a.y_bstride = hs; a.y_pitch = Fp;
            launch_gemm(a, EPI_STORE, nb, tt * 32 <= 2048 ? 1 : 2, 8, s);
        }
        launch_layernorm_cm(ctx->aH, ctx->aPOS, ctx->aH, ctx->enc_lng, ctx->enc_lnb, nb, W2V_H, Fr, Fp, hs, 1e-5f, s);
        const int vt_rows = Fp;
        for (int l = 0; l < ctx->w2v_layers; ++l) {
            const W2VLayer& ly = ctx->layers[l];
            {
                GemmArgs a = mkargs(Fr, 3 * W2V_H);
                a.nseg = 1;
                a.seg[0] = mkseg(ctx->aH, hs, Fp, W2V_H, 1, 0, 1, Fr, XF_NONE, ly.qkv.w[0]);
                a.bias = ly.qkv.bias;
                a.tm_tiles = 2 * W2V_H / 32;
                a.vt = ctx->aQK; a.vt_heads = 2 * W2V_HEADS; a.vt_dim = W2V_HD; a.vt_rows = vt_rows;
                a.y = ctx->aVT - (long long)a.tm_tiles * 32 * Fp; a.y_bstride = hs; a.y_pitch = Fp;
                const bool big = tt * 72 > 4096;
                launch_gemm(a, EPI_QKV, nb, big ? 6 : 2, big ? 4 : 8, s);
            }
            {
                AttnArgs a;
                a.qk = ctx->aQK; a.v = ctx->aVT; a.o = ctx->aO;
                a.v_bstride = hs; a.o_bstride = 2 * hs; a.b0 = 0;
                a.pitch = Fp; a.T = Fr; a.heads = W2V_HEADS; a.rows = vt_rows; a.scale = 0.125f;
                launch_attn(a, nb, W2V_HD, tt * W2V_HEADS <= 2048 ? 8 : (tt * W2V_HEADS <= 8192 ? 4 : 1), s, sp_on(ctx, ctx->attn_split) ? 2 : 0);
            }
            {
                GemmArgs a = mkargs(Fr, W2V_H);
                a.nseg = 1;
                a.seg[0] = mkseg(ctx->aO, 2 * hs, Fp, W2V_H, 1, 0, 1, Fr, XF_NONE, ly.out.w[0]);
                a.bias = ly.out.bias;
                a.res_kind = RES_PLAIN; a.res = ctx->aH; a.res_bstride = hs; a.res_pitch = Fp;
                a.y = ctx->aT; a.y_bstride = hs; a.y_pitch = Fp;
                const LaunchCfg lc = pick_cfg(tt, W2V_H / 32);
                launch_gemm(a, EPI_STORE, nb, lc.NB, lc.KS, s);
            }
            launch_layernorm_cm(ctx->aT, nullptr, ctx->aH, ly.ln1g, ly.ln1b, nb, W2V_H, Fr, Fp, hs, 1e-5f, s);
            {
                GemmArgs a = mkargs(Fr, W2V_FFN);
                a.nseg = 1;
                a.seg[0] = mkseg(ctx->aH, hs, Fp, W2V_H, 1, 0, 1, Fr, XF_NONE, ly.ff1.w[0]);
                a.bias = ly.ff1.bias; a.act = ACT_GELU;
                a.y = ctx->aF; a.y_bstride = (long long)W2V_FFN * Fp; a.y_pitch = Fp;
                const LaunchCfg lc = pick_cfg(tt, W2V_FFN / 32);
                launch_gemm(a, EPI_STORE, nb, lc.NB, lc.KS, s);
            }
            {
                GemmArgs a = mkargs(Fr, W2V_H);
                a.nseg = 1;
                a.seg[0] = mkseg(ctx->aF, (long long)W2V_FFN * Fp, Fp, W2V_FFN, 1, 0, 1, Fr, XF_NONE, ly.ff2.w[0]);
                a.bias = ly.ff2.bias;
                a.res_kind = RES_PLAIN; a.res = ctx->aH; a.res_bstride = hs; a.res_pitch = Fp;
                a.y = ctx->aT; a.y_bstride = hs; a.y_pitch = Fp;
                const LaunchCfg lc = pick_cfg(tt, W2V_H / 32);
                launch_gemm(a, EPI_STORE, nb, lc.NB, lc.KS, s);
            }
            launch_layernorm_cm(ctx->aT, nullptr, ctx->aH, ly.ln2g, ly.ln2b, nb, W2V_H, Fr, Fp, hs, 1e-5f, s);
        }
        const float* fin = ctx->aH; long long fin_bs = hs;
        if (apply_proj) {  // diffusion.py:228-229
            GemmArgs a = mkargs(Fr, out_dim);
            a.nseg = 1;
            a.seg[0] = mkseg(ctx->aH, hs, Fp, W2V_H, 1, 0, 1, Fr, XF_NONE, ctx->aproj.w[0]);
            a.bias = ctx->aproj.bias;
            a.y = ctx->aT; a.y_bstride = (long long)out_dim * Fp; a.y_pitch = Fp;
            launch_gemm(a, EPI_STORE, nb, 1, 8, s);
            fin = ctx->aT; fin_bs = (long long)out_dim * Fp;
        }
        launch_cm_to_tm(fin, out_dev + (long long)b0 * Fr * out_dim, nb, Fr, out_dim, Fp, fin_bs, s);
    }
    LAUNCHCHK();
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"

// --------------------------------------------------------------------------------------------
// VAE encoder (said/model/vae.py:26-112; caller script/test_evaluate.py:53-106) — SURVEY §8(f)4
// --------------------------------------------------------------------------------------------
struct said_vae {
    said_ctx c;            // weight table, allocation list, error string, device (the UNet members stay unused)
    int seq_len = 120, cin = 32, zdim = 64;
    int L[5] = {0, 0, 0, 0, 0};           // sequence lengths through the conv stack
    PW conv[4], fc1, fc2, fc3, head;
    int conv_k[4] = {3, 3, 4, 3}, conv_s[4] = {1, 1, 2, 1}, conv_n[4] = {32, 64, 64, 32};
    bool finalized = false;
    // workspace for `cap` windows at a time
    int cap = 0;
    float *X0 = nullptr, *Y[4] = {nullptr, nullptr, nullptr, nullptr}, *F = nullptr, *G1 = nullptr, *G2 = nullptr, *G3 = nullptr, *G4 = nullptr;
};

namespace {
// eval-mode BatchNorm1d folded into the preceding conv / linear layer (host, double): y = (W x + b - mean) * g / sqrt(var + eps) + beta
int fold_bn(said_ctx* ctx, const std::string& wname, const std::string& bname, const std::string& bn, const std::string& out_w,
            const std::string& out_b) {
    auto iw = ctx->host_w.find(wname), ib = ctx->host_w.find(bname);
    if (iw == ctx->host_w.end() || ib == ctx->host_w.end()) return fail(ctx, "missing key in state dict: %s", (iw == ctx->host_w.end() ? wname : bname).c_str());
    HostTensor W = iw->second, B = ib->second;
    const int64_t N = W.shape[0];
    const int64_t per = W.numel() / N;
    if (!bn.empty()) {
        const HostTensor* g = getw(ctx, bn + ".weight", {N});
        const HostTensor* be = getw(ctx, bn + ".bias", {N});
        const HostTensor* mu = getw(ctx, bn + ".running_mean", {N});
        const HostTensor* var = getw(ctx, bn + ".running_var", {N});
        if (!g || !be || !mu || !var) return -1;
        for (int64_t n = 0; n < N; ++n) {
            const double sc = (double)g->data[n] / std::sqrt((double)var->data[n] + 1e-5);   // nn.BatchNorm1d default eps
            for (int64_t i = 0; i < per; ++i) W.data[n * per + i] = (float)((double)W.data[n * per + i] * sc);
            B.data[n] = (float)(((double)B.data[n] - (double)mu->data[n]) * sc + (double)be->data[n]);
        }
    }
    ctx->host_w[out_w] = std::move(W);
    ctx->host_w[out_b] = std::move(B);
    return 0;
}
}  // namespace

extern "C" {

int said_vae_create(said_vae** out, int device, int in_channels, int seq_len, int z_dim) {
    if (!out) return fail(nullptr, "said_vae_create: out is null");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, "said_vae_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(nullptr, "said_vae_create: device %d out of range (%d visible)", device, ndev);
    if (in_channels != 32 || seq_len != 120 || z_dim != 64)
        return fail(nullptr, "said_vae_create: only BCVAE(channels=32, seq_len=120, z_dim=64) is supported (the FC stack is sized for it, vae.py:52)");
    DeviceRestore restore_device;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, "said_vae_create: cannot query device %d", device);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(nullptr, "said_vae_create: device is %s; this library is built for gfx950 only", prop.gcnArchName);
    said_vae* v = new said_vae();
    v->c.device = device;
    v->seq_len = seq_len; v->cin = in_channels; v->zdim = z_dim;
    v->L[0] = seq_len;
    for (int i = 0; i < 4; ++i) v->L[i + 1] = (v->L[i] - v->conv_k[i]) / v->conv_s[i] + 1;   // 118, 116, 57, 55
    configure_gemm_kernels();
    *out = v;
    return 0;
}

int said_vae_destroy(said_vae* v) {
    if (!v) return 0;
    DeviceRestore restore_device;
    (void)hipSetDevice(v->c.device);
    for (void* p : v->c.allocs) (void)hipFree(p);
    delete v;
    return 0;
}

const char* said_vae_last_error(const said_vae* v) { return v ? v->c.err.c_str() : g_create_err.c_str(); }

int said_vae_set_weight(said_vae* v, const char* name, const float* data_host, const int64_t* shape, int ndim) {
    if (!v) return -1;
    said_ctx* ctx = &v->c;
    if (v->finalized) return fail(ctx, "said_vae_set_weight after finalize");
    if (!name || !data_host || !shape || ndim < 1 || ndim > 8) return fail(ctx, "said_vae_set_weight: bad arguments");
    for (int i = 0; i < ndim; ++i) if (shape[i] < 0) return fail(ctx, "said_vae_set_weight(%s): negative dimension", name);
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    t.data.assign(data_host, data_host + t.numel());
    ctx->host_w[name] = std::move(t);
    return 0;
}

int said_vae_finalize_weights(said_vae* v) {
    if (!v) return -1;
    said_ctx* ctx = &v->c;
    if (v->finalized) return fail(ctx, "weights already finalized");
    HIPCHK(hipSetDevice(ctx->device));
    const std::string E = "encoder.";
    const char* conv_idx[4] = {"0", "3", "6", "9"};
    const char* conv_bn[4] = {"1", "4", "7", ""};
    int cin = v->cin;
    for (int i = 0; i < 4; ++i) {
        const std::string w = E + "conv_layers." + conv_idx[i];
        if (!getw(ctx, w + ".weight", {v->conv_n[i], cin, v->conv_k[i]})) return -1;
        const std::string bn = conv_bn[i][0] ? E + "conv_layers." + conv_bn[i] : "";
        const std::string ow = "__vconv" + std::to_string(i) + ".w", ob = "__vconv" + std::to_string(i) + ".b";
        if (fold_bn(ctx, w + ".weight", w + ".bias", bn, ow, ob)) return -1;
        if (make_pw(ctx, &v->conv[i], ow, ob, v->conv_n[i], cin, v->conv_k[i])) return -1;
        cin = v->conv_n[i];
    }
    const int flat = v->conv_n[3] * v->L[4];   // 32 * 55 = 1760
    if (!getw(ctx, E + "fc_layers.0.weight", {256, flat}) || !getw(ctx, E + "fc_layers.3.weight", {128, 256}) ||
        !getw(ctx, E + "fc_layers.6.weight", {v->zdim, 128}))
        return -1;
    if (fold_bn(ctx, E + "fc_layers.0.weight", E + "fc_layers.0.bias", E + "fc_layers.1", "__vfc1.w", "__vfc1.b")) return -1;
    if (fold_bn(ctx, E + "fc_layers.3.weight", E + "fc_layers.3.bias", E + "fc_layers.4", "__vfc2.w", "__vfc2.b")) return -1;
    if (make_pw(ctx, &v->fc1, "__vfc1.w", "__vfc1.b", 256, flat, 0)) return -1;
    if (make_pw(ctx, &v->fc2, "__vfc2.w", "__vfc2.b", 128, 256, 0)) return -1;
    if (make_pw(ctx, &v->fc3, E + "fc_layers.6.weight", E + "fc_layers.6.bias", v->zdim, 128, 0)) return -1;
    {   // fc_mu and fc_logvar as one GEMM: rows [0, z) = mean, [z, 2z) = log_var
        const HostTensor* wm = getw(ctx, E + "fc_mu.weight", {v->zdim, v->zdim});
        const HostTensor* bm = getw(ctx, E + "fc_mu.bias", {v->zdim});
        const HostTensor* wl = getw(ctx, E + "fc_logvar.weight", {v->zdim, v->zdim});
        const HostTensor* bl = getw(ctx, E + "fc_logvar.bias", {v->zdim});
        if (!wm || !bm || !wl || !bl) return -1;
        HostTensor W, B;
        W.shape = {2 * v->zdim, v->zdim}; B.shape = {2 * v->zdim};
        W.data = wm->data; W.data.insert(W.data.end(), wl->data.begin(), wl->data.end());
        B.data = bm->data; B.data.insert(B.data.end(), bl->data.begin(), bl->data.end());
        ctx->host_w["__vhead.w"] = std::move(W);
        ctx->host_w["__vhead.b"] = std::move(B);
        if (make_pw(ctx, &v->head, "__vhead.w", "__vhead.b", 2 * v->zdim, v->zdim, 0)) return -1;
    }
    // strict key check, like load_state_dict(strict=True) of the encoder half; decoder.* keys are accepted and ignored
    // (the decoder is not on this path), num_batches_tracked counters are metadata
    size_t enc = 0;
    for (auto& kv : ctx->host_w) {
        const std::string& k = kv.first;
        if (k.rfind("__", 0) == 0 || k.rfind("decoder.", 0) == 0) continue;
        if (k.rfind(E, 0) != 0) return fail(ctx, "unexpected key(s) in state dict: %s", k.c_str());
        ++enc;
    }
    const size_t expect_min = 4 * 2 + 3 * 4 + 3 * 2 + 2 * 4 + 4;   // convs, conv BNs, fcs, fc BNs, mu/logvar (+ optional num_batches_tracked)
    if (enc < expect_min || enc > expect_min + 5) return fail(ctx, "unexpected key(s) in state dict: %zu encoder.* tensors, expected %zu (+5 num_batches_tracked)", enc, expect_min);
    ctx->host_w.clear();
    v->finalized = true;
    return 0;
}

// BCVAE.encode (vae.py:66-83, 228-243) for `n` windows of (seq_len, 32) coefficients; window w starts at
// coeffs_dev + w * window_stride floats.  mean_dev / logvar_dev: (n, 64) row-major; logvar_dev may be null.
int said_vae_encode(said_vae* v, const float* coeffs_dev, long long window_stride, int n, float* mean_dev, float* logvar_dev, void* stream) {
    if (!v) return -1;
    said_ctx* ctx = &v->c;
    if (!v->finalized) return fail(ctx, "weights not finalized: call said_vae_finalize_weights first");
    if (n < 0 || !coeffs_dev || !mean_dev) return fail(ctx, "said_vae_encode: bad arguments");
    if (window_stride < 1) return fail(ctx, "said_vae_encode: window_stride must be positive");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->device));
    const int Z = v->zdim, C = v->cin;
    const int chunk = std::min(n, 4096);
    const int Np = rup(chunk, 32);
    const int p0 = rup(v->L[0], 32), p12 = rup(v->L[1], 32), p34 = rup(v->L[3], 32);   // 128, 128, 64
    const int flat = v->conv_n[3] * v->L[4];
    if (chunk > v->cap) {
        HIPCHK(hipStreamSynchronize(s));
        const size_t c = (size_t)chunk;
        if (drealloc(ctx, &v->X0, c * C * p0) || drealloc(ctx, &v->Y[0], c * 32 * p12) || drealloc(ctx, &v->Y[1], c * 64 * p12) ||
            drealloc(ctx, &v->Y[2], c * 64 * p34) || drealloc(ctx, &v->Y[3], c * 32 * p34) || drealloc(ctx, &v->F, (size_t)flat * Np) ||
            drealloc(ctx, &v->G1, (size_t)256 * Np) || drealloc(ctx, &v->G2, (size_t)128 * Np) || drealloc(ctx, &v->G3, (size_t)Z * Np) ||
            drealloc(ctx, &v->G4, (size_t)2 * Z * Np))
            return -1;
        v->cap = chunk;
    }
    for (int w0 = 0; w0 < n; w0 += chunk) {
        const int nb = std::min(chunk, n - w0);
        const int np = rup(v->cap, 32);   // feature-major pitch of the FC operands
        launch_windows_to_cm(coeffs_dev + (long long)w0 * window_stride, window_stride, v->X0, nb, v->L[0], C, p0, (long long)C * p0, s);
        const float* src = v->X0;
        int cin = C, pin = p0;
        for (int i = 0; i < 4; ++i) {   // Conv1d [+ BatchNorm1d folded] [+ LeakyReLU(0.2)]   (vae.py:41-51)
            const int po = i < 2 ? p12 : p34;
            GemmArgs a = mkargs(v->L[i + 1], v->conv_n[i]);
            a.nseg = 1;
            a.seg[0] = mkseg(src, (long long)cin * pin, pin, cin, v->conv_k[i], 0, v->conv_s[i], v->L[i], XF_NONE, v->conv[i].w[0]);
            a.bias = v->conv[i].bias;
            if (i < 3) a.act = ACT_LRELU_02;
            a.y = v->Y[i]; a.y_bstride = (long long)v->conv_n[i] * po; a.y_pitch = po;
            launch_gemm(a, EPI_STORE, nb, v->conv_n[i] == 64 ? 2 : 1, 8, s);
            src = v->Y[i]; cin = v->conv_n[i]; pin = po;
        }
        launch_flatten_cm(v->Y[3], (long long)32 * p34, p34, v->F, np, nb, v->conv_n[3], v->L[4], s);   // nn.Flatten (vae.py:51)
        auto fc = [&](const PW& pw, const float* x, int Cin, float* y, int N, bool act, int NB, int KS) {
            GemmArgs a = mkargs(nb, N);
            a.nseg = 1;
            a.seg[0] = mkseg(x, 0, np, Cin, 1, 0, 1, nb, XF_NONE, pw.w[0]);
            a.bias = pw.bias;
            if (act) a.act = ACT_LRELU_001;   // nn.LeakyReLU() default slope 0.01 (vae.py:57, 60)
            a.y = y; a.y_pitch = np;
            launch_gemm(a, EPI_STORE, 1, NB, KS, s);
        };
        fc(v->fc1, v->F, flat, v->G1, 256, true, 4, 4);
        fc(v->fc2, v->G1, 256, v->G2, 128, true, 4, 4);
        fc(v->fc3, v->G2, 128, v->G3, Z, false, 2, 8);
        fc(v->head, v->G3, Z, v->G4, 2 * Z, false, 4, 4);
        launch_cm_to_tm(v->G4, mean_dev + (long long)w0 * Z, 1, nb, Z, np, 0, s);
        if (logvar_dev) launch_cm_to_tm(v->G4 + (long long)Z * np, logvar_dev + (long long)w0 * Z, 1, nb, Z, np, 0, s);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"

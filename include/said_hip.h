/*
 * said_hip.h — C ABI of the MI355X (gfx950) engine behind said_amd.
 *
 * The reference (yunik1004/SAiD) has no FFI/plugin interface: its boundary for the denoising path is the Python class surface of said/model/diffusion.py.  This
 * header is the C-ABI seam the build's own `said_amd.model.*` classes bind with ctypes; every entry point names the reference method / lines whose
 * device work it replaces (paths relative to /root/reference).  Development and test entry points live in said_amd/csrc/said_hip_debug.h, not here.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; said_last_error(ctx) (said_last_error(NULL) for create failures) gives the message.  Nothing here
 *     ever falls back to a CPU path.
 *   - `*_dev` pointers are device pointers (e.g. torch.Tensor.data_ptr()) owned by the caller; `*_host` pointers are host memory.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is enqueued on it, no call synchronises unless documented.
 *   - tensors crossing the ABI use the reference's layouts: coefficients (B, T, 32), audio features (B, S, D), waveforms (B, Ta), all fp32 row-major.
 */
#ifndef SAID_HIP_H
#define SAID_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct said_ctx said_ctx;

/* prediction_type of the noise scheduler (diffusion.py:100-104) */
enum { SAID_PRED_EPSILON = 0, SAID_PRED_SAMPLE = 1, SAID_PRED_V = 2 };

/* ---- lifecycle ---------------------------------------------------------- */

/* Replaces the device-side part of SAID_UNet1D.__init__ (diffusion.py:478-527): allocates every activation/workspace buffer once, sized for `max_batch_eff`
 * samples through the UNet at a time (= 2*B under classifier-free guidance) of at most `max_frames` frames.  `ctx_dim` is the cross-attention feature size
 * (768, or feature_dim when > 0); `in_channels` the coefficient width (32). */
int said_create(said_ctx** out, int device, int max_batch_eff, int max_frames, int in_channels, int ctx_dim);
int said_destroy(said_ctx* ctx);
/* Grow the workspace to (max_batch_eff, max_frames) (never shrinks).  Only the activation buffers are re-allocated: the packed weights stay, so a longer clip
 * or a larger batch (script/test_inference.py:160-186 walks clips of varying length with ONE model) does not pay the upload again.  Synchronises the device. */
int said_reserve(said_ctx* ctx, int max_batch_eff, int max_frames);
int said_capacity(const said_ctx* ctx, int* max_batch_eff, int* max_frames);
/* A second context on the same device that SHARES the parent's packed weights (read-only) and owns its workspace and step graph: contexts run
 * said_denoise_loop concurrently on different streams (the host wrapper runs large fp32 batches as two or three concurrent clip groups: DESIGN.md 5).
 * Inherits the parent's precision mode and options as they are now.  No reference counterpart.  Destroy the clone BEFORE its parent. */
int said_clone(said_ctx* parent, said_ctx** out, int max_batch_eff, int max_frames);

/* The non-blocking stream a clone's loops are meant to run on (NULL for a said_create context): from a per-device pool of three, picked once by a 5 ms timing
 * probe so that they sit on different hardware queues (the mapping cannot be queried); never destroyed. */
void* said_stream(const said_ctx* ctx);
const char* said_last_error(const said_ctx* ctx);
/* ABI version of this library (bumped on any signature change). */
int said_abi_version(void);

/* ---- weights ------------------------------------------------------------ */

/* Replaces `.load_state_dict(...)` + `.to(device)` (script/inference.py:157-158). `name` is the reference state-dict key: "null_cond_emb",
 * "denoiser.model.*", "audio_encoder.*" (transformers-4.30.2 naming, i.e. "...pos_conv_embed.conv.weight_g/weight_v"), optional "audio_proj_layer.*". Data is
 * copied; fp32, C-contiguous. */
int said_set_weight(said_ctx* ctx, const char* name, const float* data_host, const int64_t* shape, int ndim);
/* Validates the key set strictly (like load_state_dict(strict=True)), packs every GEMM weight into MFMA fragment order and uploads.  Must be called once after
 * the last said_set_weight and before any compute call. */
int said_finalize_weights(said_ctx* ctx, void* stream);

/* Optional: the sinusoid frequency table of ldm/util.py:78-82 (exp(-ln(10000) * arange(half) / half), fp32) as computed by the host in the reference's own op
 * order.  If never called the engine derives it from a double-precision exp rounded to fp32. */
int said_set_timestep_freqs(said_ctx* ctx, const float* freqs_host, int n);

/* ---- audio path (once per clip) ----------------------------------------- */

/* Replaces SAID.get_audio_embedding (diffusion.py:209-230) → ModifiedWav2Vec2Model.forward (said/model/wav2vec2.py:13-82): conv feature extractor, linear
 * interpolation to `num_frames` (<=0: none), projection, positional conv, transformer encoder; with apply_proj != 0 also the audio_proj_layer Linear(768,
 * feature_dim) of diffusion.py:228-229. waveform_dev (B, Ta) → out_dev (B, F, D), D = 768 or feature_dim.  *out_frames receives F. */
int said_audio_encode(said_ctx* ctx, const float* waveform_dev, int batch, int num_samples, int num_frames,
                      int apply_proj, float* out_dev, int* out_frames, void* stream);

/* ---- denoiser: one evaluation ------------------------------------------- */

/* Replaces SAID.forward (diffusion.py:127-155) → UNet1DConditionModel.forward (said/model/unet_1d_condition.py:51-77) → UNetModel.forward
 * (said/model/ldm/openaimodel.py:677-709).  sample_dev (Be, T, C_in), timesteps_host (Be) int64, context_dev (Be, S, ctx_dim) → out_dev (Be, T, C_in). Any (T,
 * S): the cross-attention's alignment windows (ldm/attention.py:170-189) of up to 8 keys — all that SAID.inference produces — run inside the q projection's
 * epilogue, wider ones (S >> T) through a generic band kernel on the channel-major schedule (slower; round 4). */
int said_unet_forward(said_ctx* ctx, const float* sample_dev, const int64_t* timesteps_host, const float* context_dev,
                      int batch_eff, int frames, int ctx_len, float* out_dev, void* stream);

/* ---- denoising loop (diffusion.py:354-472) ------------------------------ */

typedef struct said_loop_params {
    int batch;               /* B clips (UNet batch is 2*B when guidance_scale > 1) */
    int frames;              /* T = window_size */
    int num_steps;           /* number of scheduler steps actually run (len(timesteps[t_start:])) */
    int prediction_type;     /* SAID_PRED_* */
    float guidance_scale;    /* CFG active iff > 1.0 (diffusion.py:358) */
    float guidance_rescale;  /* rescale_noise_cfg phi, active iff > 0 (diffusion.py:436-439) */
    float latent_scale;      /* diffusion.py:370, 470 */
    int use_step_noise;      /* eta > 0: add sigma_t * noise_k (scheduler.step).  1: noise_k = step_noise_dev[k] (injected by the
                              * caller); 2: generated inside the step's last kernel from `noise_seed` — counter-based Philox4x32-10,
                              * one standard normal per (step, element), nothing materialised (see said_philox_normal) */
    int use_mask;            /* editing: blend with re-noised init each step (diffusion.py:446-456) */
    int save_intermediate;   /* write pre-step latents / latent_scale per step (diffusion.py:417-419) */
    /* host tables, num_steps entries each, produced by the host scheduler in fp32 */
    const int64_t* timesteps_host;   /* scheduler.timesteps[t_start:] */
    const float* coef_host;          /* [num_steps][SAID_NCOEF], see SAID_COEF_* */
    /* device tensors, reference layouts */
    const float* context_dev;        /* (B, S=T, ctx_dim) conditional audio embedding */
    float* latents_dev;              /* (B, T, C) in: start latents (already noised/scaled); out: final latents */
    const float* step_noise_dev;     /* (num_steps, B, T, C) or NULL */
    const float* init_latents_dev;   /* (B, T, C) clean init (editing) or NULL */
    const float* edit_noise_dev;     /* (B, T, C) noise drawn by add_noise (diffusion.py:383-385) or NULL */
    const float* mask_dev;           /* (B, T, C) or NULL */
    float* intermediates_dev;        /* (num_steps, B, T, C) or NULL */
    float* result_dev;               /* (B, T, C): clamp(latents / latent_scale, 0, 1) (diffusion.py:470) */
    uint64_t noise_seed;             /* use_step_noise == 2: Philox key of this call's eta noise */
    int noise_batch_offset;          /* ... and the index of this call's first clip in the batch the noise is drawn for (clip groups) */
    int concurrent;                  /* 1: this loop runs beside other contexts' loops (clip groups): launches that would under-fill the chip
                                      * alone overlap with the neighbours', so the fp32 token-major GEMM path starts at 6000 rows per launch
                                      * instead of 10000 (measured: 4.457 -> 4.41 ms per step at 32 clips in three groups) */
} said_loop_params;

/* columns of coef_host (all fp32, computed on the host in the scheduler's op order) */
enum {
    SAID_COEF_SQRT_ALPHA_T = 0,   /* alpha_prod_t ** 0.5                       */
    SAID_COEF_SQRT_BETA_T = 1,    /* (1 - alpha_prod_t) ** 0.5                 */
    SAID_COEF_SQRT_ALPHA_PREV = 2,/* alpha_prod_t_prev ** 0.5                  */
    SAID_COEF_DIR = 3,            /* (1 - alpha_prod_t_prev - sigma**2) ** 0.5 */
    SAID_COEF_SIGMA = 4,          /* eta * variance ** 0.5                     */
    SAID_COEF_NEXT_SQRT_ALPHA = 5,/* add_noise coefficient at t_next (mask blend), 1 on the last step */
    SAID_COEF_NEXT_SQRT_BETA = 6, /* add_noise coefficient at t_next, 0 on the last step               */
    SAID_NCOEF = 8
};

/* Runs the whole loop: per step one hipGraph replay covering the UNet, the CFG combine, the scheduler update and the mask blend.  Asynchronous on `stream`. */
int said_denoise_loop(said_ctx* ctx, const said_loop_params* p, void* stream);

/* Builds (warm-up, stream capture, instantiation) the step graph said_denoise_loop(ctx, p, stream) would need and launches none of the loop; returns at once
 * when that graph exists.  Building synchronises `stream` and captures, which must not run while another host thread drives a second context: the host wrapper
 * prepares both clip groups' contexts in turn before it starts their loops from two threads.  Same parameters (same intermediates_dev) as the loop call that
 * follows. */
int said_loop_prepare(said_ctx* ctx, const said_loop_params* p, void* stream);

/* Progress of the loop running (or last run) on this context: *steps_done <- number of denoise steps completed so far, read from the device-side step counter
 * through a private stream — never blocks, nor waits for, the stream the loop runs on.  Host wrapper: `show_process=True` polls it from a thread to drive the
 * progress line the reference prints with tqdm around its Python loop (/root/reference/said/model/diffusion.py:412-415). */
int said_loop_progress(said_ctx* ctx, int* steps_done);
/* Sets the step counter back to "no step started" synchronously.  The host wrapper calls it before it starts the polling thread, so that a poll that runs
 * before the new loop's own reset has executed cannot report the PREVIOUS loop's final count (ABI 8). */
int said_loop_progress_reset(said_ctx* ctx);

/* The standard normals the loop generates with use_step_noise == 2: out_dev (nsteps, B*T*C) <- noise of steps step0 .. step0 + nsteps - 1 for `seed` (element
 * index = ((b*T + t)*C + c)).  Replaces the `randn` drawn inside DDIMScheduler.step for eta > 0 (diffusion.py:441-443); lets a test feed the identical noise
 * to the CPU oracle. */
int said_philox_normal(said_ctx* ctx, uint64_t seed, int step0, int nsteps, int64_t n_per_step, float* out_dev, void* stream);

/* ---- scheduler arithmetic on its own (bit-exactness tests) -------------- */

/* One DDIMScheduler.step (+ optional CFG combine and mask blend) as a plain elementwise kernel over n = B*T*C values, using the same device function the loop
 * uses.  eps_uncond_dev may be NULL (no guidance).  coef = one row of coef_host.  Replaces diffusers' DDIMScheduler.step / add_noise as called at
 * diffusion.py:441-443, 451-454. */
int said_ddim_step(said_ctx* ctx, const float* eps_dev, const float* eps_uncond_dev, float guidance_scale,
                   const float* sample_dev, const float* coef_host, int prediction_type,
                   const float* step_noise_dev, const float* init_latents_dev, const float* edit_noise_dev,
                   const float* mask_dev, float* prev_sample_dev, int64_t n, void* stream);

/* out[b, i] = a[b] * x[b, i] + c[b] * y[b, i] with each product and the sum rounded separately (no FMA): DDIMScheduler.add_noise / get_velocity as called at
 * diffusion.py:271-272, 383-385 (velocity: a = sqrt_alpha, x = noise, c = -sqrt_beta, y = sample).  a_host/c_host have `batch` entries; y_dev may be NULL (c
 * ignored). */
int said_axpby(said_ctx* ctx, const float* a_host, const float* x_dev, const float* c_host, const float* y_dev,
               float* out_dev, int batch, int64_t n_per_batch, void* stream);

/* ---- precision ----------------------------------------------------------- */

/* How the matrix products of the UNet and the audio encoder multiply.  Accumulation, statistics, normalisations, softmax, residual sums and the scheduler are
 * fp32 in every mode.  The reference itself (diffusion.py) only runs fp32.
 * SAID_PREC_FP32 (default): fp32 tensors; products on SPLIT-fp16 operands — x = h + 2^-11 l, h = RN16(x), l = RN16((x - h) 2^11): 22-bit significands, three
 *   v_mfma_f32_32x32x16_f16 per eight v_mfma_f32_32x32x2_f32.  DOMAIN of every product operand: |x| < 65504, resolution 2^-36 absolute (a tensor
 *   whose largest element is below 2^-14 loses relative precision).  Weights are checked against [2^-14, 2^15) at said_finalize_weights: outside it this mode runs as
 *   SAID_PREC_FP32_STRICT (said_effective_precision / said_precision_note).  Activations are not checked per element: most are GroupNorm / LayerNorm outputs, but the
 *   residual stream, the GEGLU product, attention outputs and the skip input are not normalised — one beyond 65504 becomes inf / NaN, reaches the step's model output
 *   and is recorded there (said_numeric_status); the host wrapper then re-runs the call in SAID_PREC_FP32_STRICT.  Final clamps keep NaN like torch.clamp.
 * SAID_PREC_FP32_STRICT: every product on v_mfma_f32_32x32x2_f32 with fp32 operands (rounds 1-4's arithmetic; the fused SpatialTransformer tail as five launches).
 * SAID_PREC_BF16: operands rounded to bfloat16 (weights once, activations after their fused normalisation).  Below 3000 UNet rows per launch all tensors in HBM stay
 *   fp32; above, activations between the UNet's kernels and q / k / v are token-major bf16: a clip's result depends on which side its batch is
 *   (bounded in tests/).  BASELINE.json configs[2]'s "bf16"; the closest analogue in the reference is torch.autocast(bfloat16).
 * Takes effect at the next call. */
enum { SAID_PREC_FP32 = 0, SAID_PREC_BF16 = 1, SAID_PREC_FP32_STRICT = 2 };
int said_set_precision(said_ctx* ctx, int mode);
int said_get_precision(const said_ctx* ctx);         /* the mode asked for */
int said_effective_precision(const said_ctx* ctx);   /* the mode that runs: SAID_PREC_FP32 becomes _STRICT when a weight tensor is out of the split range */
const char* said_precision_note(const said_ctx* ctx);/* "" or which tensor forced strict fp32 */

/* Sticky numeric status of the last said_denoise_loop / said_unet_forward on this context; synchronises `stream` (the one the call ran on).  *first_bad_step
 * <- first denoise step whose model output held an inf / NaN (-1: none); *result_nonfinite <- 1 when the final latents (loop) / the model output (forward)
 * hold one. */
int said_numeric_status(said_ctx* ctx, void* stream, int* first_bad_step, int* result_nonfinite);

/* ---- introspection ------------------------------------------------------- */

/* Number of kernel launches captured in the current per-step graph (0 if none). */
int said_graph_num_nodes(const said_ctx* ctx);
/* Algorithmic bytes / flops of one UNet evaluation for (batch_eff, frames), SURVEY.md §8(d) formulas; used by bench.py's roofline block. */
double said_unet_algorithmic_bytes(int batch_eff, int frames, int bytes_per_elem);
double said_unet_algorithmic_flops(int batch_eff, int frames);

/* Per-launch timing of one UNet evaluation's schedule at (batch_eff, frames): stage k replayed `reps` times back to back (one hipGraph) between two HIP events
 * on `stream`.  Per stage: average us, algorithmic bytes, flops, kind (0 generic GEMM, 1 attention, 2 LDS-staged UNet GEMM, 4-9 large-batch kernels, 10
 * stchain_kernel, 11 conv_in_kernel, 12 the `out` convolution), epilogue id, tile shape.  cfg_clips > 0 (= batch_eff / 2): the guided loop's schedule (shared
 * prefix), else SAID.forward's.  bench.py's roofline block; internal buffers must hold finite data (run a forward first). */
int said_profile_unet(said_ctx* ctx, int batch_eff, int frames, int cfg_clips, int reps, int max_stages, float* us_out, double* bytes_out,
                      double* flops_out, int* kind_out, int* epi_out, int* nb_out, int* ks_out, int* n_stages_out, void* stream);

/* ---- VAE encoder ----------------------------------------------
 * Replaces the device work of BCVAE.encode -> BCEncoder.forward (said/model/vae.py:26-83,
 * 228-243) as driven by generate_latents_info (script/test_evaluate.py:53-106): sliding
 * windows of 120 frames of (T, 32) coefficients -> the 64-d latent mean used by the FD /
 * WInD / multimodality metrics.  Eval-mode only: BatchNorm1d layers use their running
 * statistics and are folded into the preceding Conv1d / Linear on the host.  The decoder
 * (vae.py:115-178) is not on this path; its state-dict keys are accepted and ignored. */
typedef struct said_vae said_vae;
/* BCVAE(channels=32, seq_len=120, z_dim=64) (vae.py:181-207); other sizes are refused. */
int said_vae_create(said_vae** out, int device, int in_channels, int seq_len, int z_dim);
int said_vae_destroy(said_vae* vae);
const char* said_vae_last_error(const said_vae* vae);
/* `name` is the BCVAE state-dict key ("encoder.conv_layers.0.weight", ..., "encoder.fc_mu.bias", BatchNorm "running_mean"/"running_var" included;
 * "num_batches_tracked" may be passed as a float scalar or omitted).  Replaces said_vae.load_state_dict(...) + .to(device) (test_evaluate.py:551-553). */
int said_vae_set_weight(said_vae* vae, const char* name, const float* data_host, const int64_t* shape, int ndim);
int said_vae_finalize_weights(said_vae* vae);
/* encode `n_windows` windows of (120, 32) fp32 coefficients.  Window w starts at coeffs_dev + w * window_stride_floats: 120*32 for a (N, 120, 32) batch of
 * windows, window_step_size*32 for sliding windows over one (T, 32) sequence (test_evaluate.py:90-95: n = (T - 120) // step + 1 - padding).  mean_dev and
 * logvar_dev (nullable) are (n_windows, 64) row-major: BCLatent.mean / .log_var (vae.py:79-83). */
int said_vae_encode(said_vae* vae, const float* coeffs_dev, long long window_stride_floats, int n_windows, float* mean_dev,
                    float* logvar_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAID_HIP_H */

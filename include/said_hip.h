/*
 * said_hip.h — C ABI of the MI355X (gfx950) engine behind said_amd.
 *
 * The reference (yunik1004/SAiD) has no FFI/plugin interface: its boundary for
 * the denoising path is the Python class surface of said/model/diffusion.py.
 * This header is the C-ABI seam the build's own `said_amd.model.*` classes bind
 * with ctypes; every entry point names the reference
 * method/lines whose device work it replaces.  All paths are relative to
 * /root/reference.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error;
 *     said_last_error(ctx) (or said_last_error(NULL) for create failures) gives
 *     the message.  Nothing here ever falls back to a CPU path.
 *   - `*_dev` pointers are device pointers (e.g. torch.Tensor.data_ptr()) owned
 *     by the caller; `*_host` pointers are host memory.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream()
 *     .cuda_stream); all work is enqueued on it, no call synchronises the
 *     device unless documented.
 *   - tensors crossing the ABI use the reference's layouts: coefficients
 *     (B, T, 32) fp32 row-major, audio features (B, S, D) fp32 row-major,
 *     waveforms (B, Ta) fp32.
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

/* Replaces the device-side part of SAID_UNet1D.__init__ (diffusion.py:478-527):
 * allocates every activation/workspace buffer once, sized for `max_batch_eff`
 * samples through the UNet at a time (= 2*B under classifier-free guidance) of
 * at most `max_frames` frames.  `ctx_dim` is the cross-attention feature size
 * (768, or feature_dim when > 0); `in_channels` the coefficient width (32). */
int said_create(said_ctx** out, int device, int max_batch_eff, int max_frames, int in_channels, int ctx_dim);
int said_destroy(said_ctx* ctx);
/* Grow the workspace to hold `max_batch_eff` samples of `max_frames` frames (never shrinks; no-op when both already fit).
 * Only the (batch, frames)-sized activation buffers are re-allocated: the packed weights stay on the device, so a
 * caller that meets a longer clip or a larger batch (script/test_inference.py:160-186 walks clips of varying length
 * with ONE model) does not pay load_state_dict + .to(device) again.  Synchronises the device. */
int said_reserve(said_ctx* ctx, int max_batch_eff, int max_frames);
int said_capacity(const said_ctx* ctx, int* max_batch_eff, int* max_frames);
/* A second context on the same device that SHARES the parent's packed weights (read-only) and owns its own workspace, capture
 * streams and step graph: contexts can run said_denoise_loop concurrently on different streams.  The host wrapper uses it to run a
 * large fp32 batch as two or three concurrent clip groups (a launch's phases — tile loads, MFMAs, result stores — then overlap across
 * the groups: -7 % per pass at 32 clips x 600 frames with round 5's schedule; bf16 large batches run as ONE group since round 4's persistent
 * kernels).  The clone inherits the parent's precision
 * mode and debug options as they are at this call.  No reference counterpart.  Destroy the clone BEFORE its parent. */
int said_clone(said_ctx* parent, said_ctx** out, int max_batch_eff, int max_frames);

/* The non-blocking stream a clone's loops are meant to run on (NULL for a context made by said_create).  It comes from a pool of three
 * streams per device shared by every clone in the process and never destroyed.  Streams that share one of the device's few hardware queues
 * serialise and the mapping cannot be queried, so the pool is picked by a one-off timing probe (about 5 ms, synchronises the device) at the
 * first said_clone: three streams that run beside the default stream and beside each other. */
void* said_stream(const said_ctx* ctx);
const char* said_last_error(const said_ctx* ctx);
/* ABI version of this library (bumped on any signature change). */
int said_abi_version(void);

/* ---- weights ------------------------------------------------------------ */

/* Replaces `.load_state_dict(...)` + `.to(device)` (script/inference.py:157-158).
 * `name` is the reference state-dict key: "null_cond_emb",
 * "denoiser.model.*", "audio_encoder.*" (transformers-4.30.2 naming, i.e.
 * "...pos_conv_embed.conv.weight_g/weight_v"), optional "audio_proj_layer.*".
 * Data is copied; fp32, C-contiguous. */
int said_set_weight(said_ctx* ctx, const char* name, const float* data_host, const int64_t* shape, int ndim);
/* Validates the key set strictly (like load_state_dict(strict=True)), packs
 * every GEMM weight into MFMA fragment order and uploads.  Must be called once
 * after the last said_set_weight and before any compute call. */
int said_finalize_weights(said_ctx* ctx, void* stream);

/* Optional: the sinusoid frequency table of ldm/util.py:78-82
 * (exp(-ln(10000) * arange(half) / half), fp32) as computed by the host in the
 * reference's own op order.  If never called the engine derives it from a
 * double-precision exp rounded to fp32. */
int said_set_timestep_freqs(said_ctx* ctx, const float* freqs_host, int n);

/* ---- audio path (once per clip) ----------------------------------------- */

/* Replaces SAID.get_audio_embedding (diffusion.py:209-230) →
 * ModifiedWav2Vec2Model.forward (said/model/wav2vec2.py:13-82): conv feature
 * extractor, linear interpolation to `num_frames` (<=0: none), projection,
 * positional conv, transformer encoder; with apply_proj != 0 also the
 * audio_proj_layer Linear(768, feature_dim) of diffusion.py:228-229.
 * waveform_dev (B, Ta) → out_dev (B, F, D), D = 768 or feature_dim.  *out_frames receives F. */
int said_audio_encode(said_ctx* ctx, const float* waveform_dev, int batch, int num_samples, int num_frames,
                      int apply_proj, float* out_dev, int* out_frames, void* stream);

/* ---- denoiser: one evaluation ------------------------------------------- */

/* Replaces SAID.forward (diffusion.py:127-155) → UNet1DConditionModel.forward
 * (said/model/unet_1d_condition.py:51-77) → UNetModel.forward
 * (said/model/ldm/openaimodel.py:677-709).  sample_dev (Be, T, C_in),
 * timesteps_host (Be) int64, context_dev (Be, S, ctx_dim) → out_dev (Be, T, C_in).
 * Any (T, S): the cross-attention's alignment windows (ldm/attention.py:170-189) of up to 8 keys —
 * all that SAID.inference produces — run inside the q projection's epilogue, wider ones (S >> T)
 * through a generic band kernel on the channel-major schedule (slower; round 4). */
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

/* Runs the whole loop: per step one hipGraph replay covering the UNet, the CFG
 * combine, the scheduler update and the mask blend.  Asynchronous on `stream`. */
int said_denoise_loop(said_ctx* ctx, const said_loop_params* p, void* stream);

/* Builds (warm-up, stream capture, instantiation) the step graph said_denoise_loop(ctx, p, stream) would need and launches
 * none of the loop; returns at once when that graph exists.  Building synchronises `stream` and captures, which must not run
 * while another host thread drives a second context: the host wrapper prepares both clip groups' contexts in turn before it
 * starts their loops from two threads.  Same parameters (same intermediates_dev) as the loop call that follows. */
int said_loop_prepare(said_ctx* ctx, const said_loop_params* p, void* stream);

/* Progress of the loop running (or last run) on this context: *steps_done <- number of denoise steps completed so far, read from the
 * device-side step counter through a private stream — never blocks, nor waits for, the stream the loop runs on.  Host wrapper:
 * `show_process=True` polls it from a thread to drive the progress line the reference prints with tqdm around its Python loop
 * (/root/reference/said/model/diffusion.py:412-415). */
int said_loop_progress(said_ctx* ctx, int* steps_done);
/* Sets the step counter back to "no step started" synchronously.  The host wrapper calls it before it starts the polling thread, so that a
 * poll that runs before the new loop's own reset has executed cannot report the PREVIOUS loop's final count (ABI 8). */
int said_loop_progress_reset(said_ctx* ctx);

/* The standard normals the loop generates with use_step_noise == 2: out_dev (nsteps, B*T*C) <- noise of steps
 * step0 .. step0 + nsteps - 1 for `seed` (element index = ((b*T + t)*C + c)).  Replaces the `randn` drawn inside
 * DDIMScheduler.step for eta > 0 (diffusion.py:441-443); lets a test feed the identical noise to the CPU oracle. */
int said_philox_normal(said_ctx* ctx, uint64_t seed, int step0, int nsteps, int64_t n_per_step, float* out_dev, void* stream);

/* ---- scheduler arithmetic on its own (bit-exactness tests) -------------- */

/* One DDIMScheduler.step (+ optional CFG combine and mask blend) as a plain
 * elementwise kernel over n = B*T*C values, using the same device function the
 * loop uses.  eps_uncond_dev may be NULL (no guidance).  coef = one row of
 * coef_host.  Replaces diffusers' DDIMScheduler.step / add_noise as called at
 * diffusion.py:441-443, 451-454. */
int said_ddim_step(said_ctx* ctx, const float* eps_dev, const float* eps_uncond_dev, float guidance_scale,
                   const float* sample_dev, const float* coef_host, int prediction_type,
                   const float* step_noise_dev, const float* init_latents_dev, const float* edit_noise_dev,
                   const float* mask_dev, float* prev_sample_dev, int64_t n, void* stream);

/* out[b, i] = a[b] * x[b, i] + c[b] * y[b, i] with each product and the sum rounded
 * separately (no FMA): DDIMScheduler.add_noise / get_velocity as called at
 * diffusion.py:271-272, 383-385 (velocity: a = sqrt_alpha, x = noise, c = -sqrt_beta,
 * y = sample).  a_host/c_host have `batch` entries; y_dev may be NULL (c ignored). */
int said_axpby(said_ctx* ctx, const float* a_host, const float* x_dev, const float* c_host, const float* y_dev,
               float* out_dev, int batch, int64_t n_per_batch, void* stream);

/* ---- precision ----------------------------------------------------------- */

/* How the GEMMs / convolutions and both attention products of the UNet (and of the audio encoder) multiply.  Statistics, normalisations,
 * softmax, residual sums and the scheduler are fp32 in every mode; accumulation is fp32 in every mode.  The reference itself
 * (diffusion.py) only runs fp32.
 *
 * SAID_PREC_FP32 (default): fp32 tensors everywhere; products on SPLIT-fp16 operands — x = h + 2^-11 l with h = RN16(x),
 *   l = RN16((x - h) 2^11): 22-bit significands, three v_mfma_f32_32x32x16_f16 per eight v_mfma_f32_32x32x2_f32, the l.l term
 *   (2^-22 relative) dropped.  DOMAIN of every product operand: |x| < 65504 (fp16's range), and an operand
 *   resolves 2^-36 absolute (fp16 denormals), i.e. a tensor whose largest element is below 2^-14 loses relative precision.
 *   Weights are checked against [2^-14, 2^15) at said_finalize_weights: outside it this mode runs as SAID_PREC_FP32_STRICT
 *   (said_effective_precision / said_precision_note say so).  Activations are not range-checked per element; most operands are
 *   GroupNorm / LayerNorm outputs (bounded by sqrt(channels) x gain), but the residual stream, the GEGLU product, the attention
 *   output and the concatenated skip input are not normalised: one beyond 65504 becomes inf / NaN in its product, reaches the
 *   step's model output through every later layer, and is recorded by the step's last kernel — said_numeric_status reports it;
 *   the host wrapper (SAID.inference) then re-runs the call in SAID_PREC_FP32_STRICT.  Nothing is silently clamped: final
 *   clamps keep NaN like torch.clamp.
 * SAID_PREC_FP32_STRICT: fp32 tensors, every product on v_mfma_f32_32x32x2_f32 with fp32 operands (the arithmetic of rounds
 *   1-4; the fused SpatialTransformer tail runs as five launches).  About 1.25x slower per step at batch 1 (bench.py
 *   secondary.cfg1_strict_fp32).
 * SAID_PREC_BF16: operands rounded to bfloat16 (weights once at said_finalize_weights, activations after their fused
 *   normalisation).  At small batches (< 3000 UNet rows per launch) all tensors in HBM stay fp32; at large batches the activations
 *   BETWEEN the UNet's kernels — and q / k / v — are stored token-major in bf16, i.e. a clip's result then
 *   depends on whether its batch crosses that threshold (bounded in tests/test_gpu_parity.py / test_gpu_round4.py).  This is
 *   the "bf16" of BASELINE.json configs[2]; the closest analogue in the reference is torch.autocast(bfloat16).
 * Takes effect at the next call. */
enum { SAID_PREC_FP32 = 0, SAID_PREC_BF16 = 1, SAID_PREC_FP32_STRICT = 2 };
int said_set_precision(said_ctx* ctx, int mode);
int said_get_precision(const said_ctx* ctx);         /* the mode asked for */
int said_effective_precision(const said_ctx* ctx);   /* the mode that runs: SAID_PREC_FP32 becomes _STRICT when a weight tensor is out of the split range */
const char* said_precision_note(const said_ctx* ctx);/* "" or which tensor forced strict fp32 */

/* Sticky numeric status of the last said_denoise_loop / said_unet_forward on this context.  Synchronises `stream` (the stream the
 * call was enqueued on).  *first_bad_step <- index of the first denoise step whose model output held an inf / NaN (-1: none);
 * *result_nonfinite <- 1 when the final latents (loop) / the model output (forward) hold one.  The reference would return the same
 * inf / NaN from an fp32 overflow of its own; in SAID_PREC_FP32 the cause can also be an operand beyond the split-fp16 domain,
 * which SAID_PREC_FP32_STRICT does not have: the host wrapper retries there. */
int said_numeric_status(said_ctx* ctx, void* stream, int* first_bad_step, int* result_nonfinite);

/* ---- introspection ------------------------------------------------------- */

/* Number of kernel launches captured in the current per-step graph (0 if none). */
int said_graph_num_nodes(const said_ctx* ctx);
/* Algorithmic bytes / flops of one UNet evaluation for (batch_eff, frames),
 * SURVEY.md §8(d) formulas; used by bench.py's roofline block. */
double said_unet_algorithmic_bytes(int batch_eff, int frames, int bytes_per_elem);
double said_unet_algorithmic_flops(int batch_eff, int frames);

/* Per-launch timing of one UNet evaluation's kernel schedule at (batch_eff, frames): stage k is
 * replayed `reps` times back to back (one hipGraph) between two HIP events on `stream`.
 * Outputs per stage: average microseconds, algorithmic bytes (weights + operands + result),
 * flops, kind (0 = generic GEMM/conv kernel, 1 = attention, 2 = LDS-staged UNet GEMM, 4-9 = the large-batch kernels, 10 = stchain_kernel,
 * 11 = conv_in_kernel, 12 = the `out` convolution alone — in the loop it is the first half of out_sched_kernel), epilogue id, tile shape (NB, KS).
 * cfg_clips > 0 (= batch_eff / 2): the schedule of the classifier-free-guidance loop (unconditional half first; the
 * prefix shared by the two halves runs once per clip), else the schedule of SAID.forward.
 * Used by bench.py's roofline block; internal buffers must hold finite data (run a forward first). */
int said_profile_unet(said_ctx* ctx, int batch_eff, int frames, int cfg_clips, int reps, int max_stages, float* us_out, double* bytes_out,
                      double* flops_out, int* kind_out, int* epi_out, int* nb_out, int* ks_out, int* n_stages_out, void* stream);

/* ---- debugging aids (used by tests/ only) ---------------------------------- */

/* Test-only switches of one context (the shipped library reads no environment variables):
 *   "unet_tgemm_min_tokens"  tokens per launch from which the UNet takes the token-major GEMM path, both precisions
 *                            (< 0: restore the measured defaults 3000 bf16 / 10000 fp32)
 *   "audio_chunk"            clips per audio-encoder pass (default 32)
 *   "steps_per_graph"        denoise steps captured per hipGraph (default 10)
 *   "tm_acts"                large batches: token-major activations (bf16 / fp32) between the UNet kernels, GroupNorm / LayerNorm applied inside
 *                            the consuming GEMM (xgemm_kernel), 41 launches per step, no preparation kernels (NOTEBOOK.md 7.3).  -1 (default): on
 *                            in bf16 mode, off in fp32 mode (measured slower there); 0 / 1 force it
 *   "xgemm_ntw"              column tiles per workgroup of the resident-source GEMMs (0: chosen per launch)
 *   "hybrid"                 0: bf16 mode at large batch keeps round 2's SpatialTransformer schedule throughout (default 1: from the
 *                            attention output on, the block runs on round 3's token-major kernels — NOTEBOOK.md 7.3)
 *   "audio_front_fused"      0: the bf16 audio encoder stores conv0's fp32 activation and runs GroupNorm + GELU and the transpose as separate
 *                            kernels (round 2); default 1: one recomputing pass writes token-major bf16 directly
 *   "mt_mid"                 0: multi-tile workgroups (several token tiles per workgroup, weights kept in registers) only from 1024 workgroups per
 *                            launch on (round 2); default 1: also for launches of 2-4 rounds of one workgroup per CU (NOTEBOOK.md 7.2)
 *   "mt_wgs"                 > 0: workgroups per token tile from which a launch goes multi-tile (overrides both rules)
 *   "tgemm_sb"               0: the bf16 audio encoder's 128 x 128 GEMM tiles keep two LDS operand buffers (two workgroups per CU; round 2); default 1: one
 *                            buffer, three workgroups per CU (11.12 -> 10.90 ms per 32 clips, bit-identical)
 *   "f32_out1_tm"            0: fp32 mode at large batch runs attn1.to_out on the channel-major kernel (round 2); default 1: on the token-major fp32 GEMM
 *   "unet_nb_model"          0: round 2's rule for the column tiles per workgroup of the 192-wide channel-major GEMMs (default 1: busiest-CU model)
 *   "unet_nb"                > 0: forces that number of column tiles per workgroup (1, 2 or 3)
 *   "hybrid_f32"             1: the hybrid SpatialTransformer schedule in fp32 mode too (measured slower: default 0)
 *   "out_tm"                 0: bf16 large batches end the step with round 3's channel-major out conv + scheduler kernel (default -1: out_sched_tm_kernel)
 *   "rgemm"                  0: bf16 large batches without round 4's persistent register-stationary GEMMs (default -1: on)
 *   "battn"                  0: bf16 large batches with attn_kernel on fp32 operands instead of battn_kernel (default -1: on; 4 / 8: query tiles per workgroup)
 *   "gemm_split"             fp32 mode, large batches: 0 puts fgemm_kernel back on v_mfma_f32_32x32x2_f32 (default -1 / 1: split-fp16 products)
 *   "attn_split"             fp32 mode: 0 puts both self-attention products back on fp32 MFMAs (default -1 / 1: split-fp16 products)
 *   "ugemm_split"            fp32 mode, channel-major GEMMs (ugemm_kernel): 0 = fp32 MFMAs (default -1 / 1: split-fp16 products, round 5)
 *   "attn_presplit"          fp32 small batch: 0 = attention splits K / V itself (default -1 / 1: the q/k/v GEMM stores them pre-split, attn_kernel<PM = 3>)
 *   "out_split"              out_sched_kernel's convolution: 0 = fp32 MFMAs (default -1 / 1: split-fp16 products, round 5)
 *   "st_chain"               fp32 mode: 0 runs everything behind a SpatialTransformer's self-attention as five launches (rounds 1-4); default -1 / 1:
 *                            one launch per block (stchain_kernel, round 5).  "st_chain_large" 0: only below the token-major threshold;
 *                            "st_chain_max_tiles" n: only while a launch has at most n (sample, 32-token tile) workgroups; "st_chain_dbg" 1: the fused
 *                            kernel also writes x1 / x2 / the cross-attention input to X1 / X2 / X3 (bring-up)
 *   "st_chain_bf16"          bf16 mode, large batches: 0 = rgemm's six launches behind self-attention; -1 / 1 (default) = stchain_kernel<bf16>, one token tile per workgroup,
 *                            two workgroups per CU; 2 = stchain2_kernel, two tiles per workgroup sharing every weight fragment (bit-identical, measured slower)
 *   "xgemm_clk"              1: shader-clock stamps of the token-major schedule's kernels (-DSAID_CLK_STAMPS builds; read with said_debug_clocks)
 * said_debug_get additionally knows "n_set_weight" (said_set_weight calls so far), "n_stchain" / "n_rgemm" / "n_xgemm" (launches issued through those kernels). */
int said_debug_option(said_ctx* ctx, const char* name, long long value);
long long said_debug_get(const said_ctx* ctx, const char* name);
/* Stop the UNet schedule after `n_launches` kernel launches, counted from the start of each said_unet_forward / said_denoise_loop call (< 0: run
 * everything).  In a loop call only the eager warm-up step then runs (its first n launches); the workspace afterwards holds that step's intermediates
 * (tests/test_gpu_round5.py reads the last hidden state this way). */
int said_debug_stop_after(said_ctx* ctx, int n_launches);
/* Enable/disable per-phase shader-clock stamps in the GEMM kernels of the next UNet evaluations and
 * (if out_host != NULL) read back the [64 launches][8 waves][8 slots] stamp table. */
int said_debug_clocks(said_ctx* ctx, int enable, long long* out_host);
/* Synchronously copy `n` floats from the start of the named internal buffer
 * ("H0","H1","P","Q","M","X1","X2","X3","O","QK","VT","F","KV","CTX","EO","E0","E1","E2",
 *  "x","eps","stH0","stP","stM", ...) to host memory. */
int said_debug_read(said_ctx* ctx, const char* name, float* out_host, int64_t n);
/* Workspace inspection (race hunting, round 5): the context's (max_batch_eff, max_frames)-sized buffers by allocation index.
 * _info: device pointer, size in bytes and a short name ("H0", "uPA", ..., "?" if unnamed) of buffer `idx`;
 * _fill: synchronises the device and sets EVERY byte of every workspace buffer to `byte_value` (0xFF: NaN patterns; drops the step graph
 *        and the cached band tables) — a result that changes with the fill value is a read of memory nobody wrote;
 * _copy: enqueues a device-to-device copy of the first `bytes` bytes of buffer `idx` to `dst_dev` on `stream`. */
int said_debug_ws_count(const said_ctx* ctx);
int said_debug_ws_info(said_ctx* ctx, int idx, void** ptr_out, long long* bytes_out, const char** name_out);
int said_debug_ws_fill(said_ctx* ctx, int byte_value);
int said_debug_ws_copy(said_ctx* ctx, int idx, void* dst_dev, long long bytes, void* stream);

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
/* `name` is the BCVAE state-dict key ("encoder.conv_layers.0.weight", ..., "encoder.fc_mu.bias",
 * BatchNorm "running_mean"/"running_var" included; "num_batches_tracked" may be passed as a float
 * scalar or omitted).  Replaces said_vae.load_state_dict(...) + .to(device) (test_evaluate.py:551-553). */
int said_vae_set_weight(said_vae* vae, const char* name, const float* data_host, const int64_t* shape, int ndim);
int said_vae_finalize_weights(said_vae* vae);
/* encode `n_windows` windows of (120, 32) fp32 coefficients.  Window w starts at
 * coeffs_dev + w * window_stride_floats: 120*32 for a (N, 120, 32) batch of windows,
 * window_step_size*32 for sliding windows over one (T, 32) sequence
 * (test_evaluate.py:90-95: n = (T - 120) // step + 1 - padding).  mean_dev and logvar_dev
 * (nullable) are (n_windows, 64) row-major: BCLatent.mean / .log_var (vae.py:79-83). */
int said_vae_encode(said_vae* vae, const float* coeffs_dev, long long window_stride_floats, int n_windows, float* mean_dev,
                    float* logvar_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAID_HIP_H */

/*
 * said_hip_debug.h — development / test entry points of libsaid_hip.so.  NOT part of the reference-facing boundary (include/said_hip.h): nothing in
 * said_amd/model, script/ or bench.py's measured path needs them; tests/ and scripts/ use them to pin a schedule, read internal buffers, stamp clocks.
 * Same conventions as said_hip.h (0 = success, said_last_error for the message).
 */
#ifndef SAID_HIP_DEBUG_H
#define SAID_HIP_DEBUG_H

#include "../../include/said_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test-only switches of one context (the shipped library reads no environment variables):
 *   "unet_tgemm_min_tokens"  tokens per launch from which the UNet takes the token-major GEMM path, both precisions
 *                            (< 0: restore the measured defaults 3000 bf16 / 10000 fp32)
 *   "audio_chunk"            clips per audio-encoder pass (default 32)
 *   "steps_per_graph"        denoise steps captured per hipGraph in loops of >= 400 steps (default 50; until round 6: 10); shorter loops: min(this, 10)
 *   "tm_acts"                bf16 mode, large batches: token-major bf16 activations between the UNet kernels, GroupNorm / LayerNorm applied inside the consuming
 *                            GEMMs (-1 / 1, default); 0: channel-major fp32 activations with preparation kernels (round 2).  (The fp32 twin of the schedule, measured
 *                            slower in round 3, was removed in round 6.)
 *   "xgemm_ntw"              column tiles per workgroup of the resident-source GEMMs (0: chosen per launch)
 *   "hybrid"                 0: bf16 mode at large batch keeps round 2's SpatialTransformer schedule throughout (default 1: from the
 *                            attention output on, the block runs on round 3's token-major kernels — NOTEBOOK.md 7.3)
 *   "mt_mid"                 0: multi-tile workgroups (several token tiles per workgroup, weights kept in registers) only from 1024 workgroups per
 *                            launch on (round 2); default 1: also for launches of 2-4 rounds of one workgroup per CU (NOTEBOOK.md 7.2)
 *   "mt_wgs"                 > 0: workgroups per token tile from which a launch goes multi-tile (overrides both rules)
 *   "tgemm_sb"               0: the bf16 audio encoder's 128 x 128 GEMM tiles keep two LDS operand buffers (two workgroups per CU; round 2); default 1: one
 *                            buffer, three workgroups per CU (11.12 -> 10.90 ms per 32 clips, bit-identical)
 *   "tgemm_direct"           0: the bf16 audio encoder's projections on tgemm_kernel<128> (rounds 2-5); -1 / 1 (default): on tgemm256d_kernel — 256 x 256 x 64 tile, operand tiles
 *                            loaded global -> LDS directly, XOR-swizzled chunks, one barrier per k-tile (round 6; bit-identical)
 *   "f32_out1_tm"            0: fp32 mode at large batch runs attn1.to_out on the channel-major kernel (round 2); default 1: on the token-major fp32 GEMM
 *   "unet_nb_model"          0: round 2's rule for the column tiles per workgroup of the 192-wide channel-major GEMMs (default 1: busiest-CU model)
 *   "unet_nb"                > 0: forces that number of column tiles per workgroup (1, 2 or 3)
 *   "out_tm"                 0: bf16 large batches end the step with round 3's channel-major out conv + scheduler kernel (default -1: out_sched_tm_kernel)
 *   "rgemm"                  0: bf16 large batches without round 4's persistent register-stationary GEMMs (default -1: on)
 *   "battn"                  0: bf16 large batches with attn_kernel on fp32 operands instead of battn_kernel (default -1: on; 4 / 8: query tiles per workgroup)
 *   "gemm_presplit"          fp32 mode, large batches: 0 = fgemm_kernel splits fp32 operands in its k loop (round 5); -1 / 1 (default): the ResBlock convolutions' and q / k / v's
 *                            operands arrive split (prep_kernel packs h | l pairs, packed weight copies; bit-identical, round 6)
 *   "gemm_split"             fp32 mode, large batches: 0 puts fgemm_kernel back on v_mfma_f32_32x32x2_f32 (default -1 / 1: split-fp16 products)
 *   "attn_split"             fp32 mode: 0 puts both self-attention products back on fp32 MFMAs (default -1 / 1: split-fp16 products)
 *   "ugemm_split"            fp32 mode, channel-major GEMMs (ugemm_kernel): 0 = fp32 MFMAs (default -1 / 1: split-fp16 products, round 5)
 *   "chain_coef"             fp32 mode, small batch: 0 = stchain_kernel finalises the block input's GroupNorm coefficients from the partials itself; -1 / 1 (default): it reads the
 *                            coefficients the q/k/v GEMM of the same block finalised and left behind (GemmCommon::gn_coef_out, round 6)
 *   "kconv"                  fp32 mode, small batch: 0 = the K-long ResBlock convolutions of the up path (two / three K segments) keep ugemm_body's block loop; -1 / 1 (default):
 *                            kconv_body's straight-line blocks (round 6; bit-identical); 2: also under concurrent clip groups where the chosen column-tile count has no split shape
 *                            (4 / 5 / 6 / 7 / 8 clips: -0.3 / -1.1 / +4.6 / -0.1 / +0.2 %: not the default)
 *   "attn_2q"                fp32 mode, pre-split K / V, four key slices: 0 = one query tile per wave always; 1 = three always; -1 (default) = three from 512 (sample, head, query tile)
 *                            triples per launch on (attn2q_kernel: long sequences at small batch; bit-identical)
 *   "attn_presplit"          fp32 small batch: 0 = attention splits K / V itself (default -1 / 1: the q/k/v GEMM stores them pre-split, attn_kernel<PM = 3>)
 *   "out_split"              out_sched_kernel's convolution: 0 = fp32 MFMAs (default -1 / 1: split-fp16 products, round 5)
 *   "st_chain"               fp32 mode: 0 runs everything behind a SpatialTransformer's self-attention as five launches (rounds 1-4); default -1 / 1:
 *                            one launch per block (stchain_kernel, round 5).  "st_chain_large" 0: only below the token-major threshold;
 *                            "st_chain_max_tiles" n: only while a launch has at most n (sample, 32-token tile) workgroups; "st_chain_dbg" 1: the fused
 *                            kernel also writes x1 / x2 / the cross-attention input to X1 / X2 / X3 (bring-up)
 *   "st_chain_slices"        fp32 mode, fused tail: 1 = always one workgroup per token tile; -1 / 3 (default) = launches of at most 85 (sample, tile) pairs run three
 *                            workgroups per tile, each with a third of the GEGLU / folded proj_out weight stream, partial sums met in memory in a fixed order (round 6)
 *   "st_chain_bf16"          bf16 mode, large batches: 0 = rgemm's six launches behind self-attention; -1 / 1 (default) = stchain_kernel<bf16>, one token tile per workgroup,
 *                            two workgroups per CU
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

#ifdef __cplusplus
}
#endif
#endif /* SAID_HIP_DEBUG_H */

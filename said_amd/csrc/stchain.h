// stchain.h — launch-level interface of the fused SpatialTransformer tail (stchain.hip); shared with engine.cpp, which packs its operands.
#pragma once
#include <hip/hip_runtime.h>

namespace said {

constexpr int CHAIN_KW = 56;                    // key rows of the cross-attention window tile kept in LDS per 32-token tile: the tile's windows must fit
                                                // (engine.cpp: set_band checks max(hi) - lo[t0] <= CHAIN_KW per tile, else the five-launch schedule runs)
constexpr size_t CHAIN_STREAM_UNITS = 4 * 168 + 2 * 138 + 2 * 102;
constexpr size_t CHAIN_STREAM_BYTES = CHAIN_STREAM_UNITS * 2048;   // 2 KB units (one k16 step: h + l fragments) of the eight waves' streams: 2,359,296 per transformer block
// round 6, small launches: three workgroups ("slices") per token tile, each with its own stream: [to_out1 | to_q | to_out2] as above, ONE GEGLU pair per wave
// (hidden tiles 8 c + w) and a third of the folded proj_out's K (k16 steps 16 c .. 16 c + 15 of the GEGLU product, steps 4 c .. 4 c + 3 of x2)
constexpr size_t CHAIN3_SLICE_UNITS = 4 * 80 + 2 * 70 + 2 * 34;             // 528 units = 1,081,344 bytes per slice
constexpr size_t CHAIN3_STREAM_BYTES = 3 * CHAIN3_SLICE_UNITS * 2048;
constexpr int CHAIN3_MAX_TILES = 85;                    // (sample, token tile) pairs a three-slice launch may have: 3 x 85 <= 256 workgroups, one round of the chip
// ... and two slices for launches of 86 .. 128 pairs: twelve GEGLU pairs per slice (waves 0-3 two each, waves 4-7 one), 24 + 6 k16 steps of the folded proj_out
constexpr size_t CHAIN2_SLICE_UNITS = 4 * 114 + 2 * 75 + 2 * 39;            // 684 units per slice
constexpr size_t CHAIN2_STREAM_BYTES = 2 * CHAIN2_SLICE_UNITS * 2048;
constexpr int CHAIN2_MAX_TILES = 128;
constexpr int CHAIN_VEC_FLOATS = 5 * 192 + 1536;        // global: b1, bq, bo2, c2, bffp (192 each), bff (1536: value rows, then gate rows)
constexpr int CHAIN_VEC_FLOATS_LDS = 4 * 192 + 1536;    // LDS: bo2 | c2 share a slot (conditional | unconditional sample)

struct ChainArgs {
    const float* wstream;    // this block's weight stream (fp16 h / l planes in each wave's consumption order: engine.cpp pack_chain_stream)
    const float* xin_part;   // GroupNorm partials of the block input [sample][tile][192][2] (mean, M2)
    const float* gn_coef;    // fp32 kernels (round 6): the block input's finalised GroupNorm (a, b) [sample][192][2] as the q/k/v GEMM or its operand preparation left them (coef_bs
                             // floats between samples) — 1 load per wave instead of 23 and no finalisation in front of the first barrier; the bf16 kernel finalises from xin_part
    long long coef_bs;
    const float* gn_gamma;   // SpatialTransformer.norm (eps 1e-6)
    const float* gn_beta;
    const float* vec;        // CHAIN_VEC_FLOATS
    const float* kvt;        // key-major copy of the cross-attention K / V: [sample][S][1536], this block's K at column koff, V at koff + 192
    const int* lo;           // [T] first visible key of query t; non-decreasing (ldm/attention.py:184-189)
    const int* hi;           // [T] one past the last visible key
    float* y;                // block output, channel-major [sample][192][pitch]
    float* stats_out;        // its GroupNorm partials [sample][tile][192][2], or null
    float* dbg_x1;           // debug (said_debug_option "st_chain_dbg"): x1 and x2 to channel-major buffers [sample][192][pitch]
    float* dbg_x2;
    float* dbg_o2;           // ... and the cross-attention output
    long long* clk;          // debug: shader-clock stamps [8 waves][16] of workgroup (8, last sample)
    long long part_bs, kvt_bs, y_bs, stats_bs;   // floats between samples
    int S;                   // keys
    int np;                  // partial tiles per channel = ceil(T / 32)
    int koff;                // = 384 * block index
    int wmax;                // max(hi - lo) <= 8
    float scale;             // dim_head ** -0.5
    int slices;              // 1, or 2 / 3: wstream is that slice count's stream (CHAIN2_ / CHAIN3_STREAM_BYTES) and the launch has 2 / 3 workgroups per token tile
    float* part;             // slices > 1: partial sums [sample][tile][slices][6 column tiles][16][64 lanes] fp32
    int* ticket;             // slices > 1: [sample][tile][6] arrival counters, zero before and after every launch
};

bool stchain_supports(const ChainArgs& a, int T, int pitch, long long o_bs, long long x_bs);
// o: attention output [.][192][pitch] (o_bs floats between samples), xin: block input (x_bs); sample s reads o / xin / partials of sample s % in_mod (in_mod > 0:
// guidance-shared prefix) and skips the cross-attention when s < n_uncond.
// bf16 = true (bf16 mode, large batches): o, xin and y are TOKEN-major bf16 [sample][row][192] (o_bs / x_bs / y_bs in elements, pitch unused), wstream the bf16 stream
// (1 KB units), every product one v_mfma_f32_32x32x16_bf16 on operands rounded to bf16; statistics, LayerNorm, softmax, residual sums stay fp32.
void launch_stchain(const ChainArgs& a, const float* o, const float* xin, int T, int pitch, long long o_bs, long long x_bs, int in_mod, int n_uncond, int nsamp, hipStream_t s,
                    bool bf16 = false);
void configure_stchain_kernel();

}  // namespace said

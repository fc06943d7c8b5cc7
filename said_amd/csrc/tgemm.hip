// tgemm.hip — bf16 GEMM on v_mfma_f32_32x32x16_bf16 for the audio encoder's bf16 mode (BASELINE.json configs[2]),
// plus the small token-major kernels around it.
//
//     Y[m][n] = epi( sum_k A[m][k] * W[n][k] )        A, W bf16 with k contiguous ("NT"), fp32 accumulation
//
// In bf16 mode the Wav2Vec2 encoder (wav2vec2.py:13-82 + the HF internals it inherits) keeps its activations TOKEN-major
// [t][c]: both MFMA operands then want 8 consecutive k per lane, i.e. one 16-byte load each, and a strided Conv1d over
// token-major data IS a GEMM whose A rows overlap — row m starts at element m * stride * C and spans taps * C
// contiguous elements — so the six 512-channel feature-extractor convolutions, the feature projection and the 48
// encoder-layer projections all run through this one kernel.  (The fp32 mode keeps the channel-major kernels.)
//
// Tile: 128 tokens x 128 outputs x 64 k per workgroup of 4 waves (2 x 2, each 64 x 64 = 2 x 2 MFMA tiles, 64 accumulator
// registers).  Operand tiles are staged through LDS with register double-buffering (global -> registers for tile k+1
// while tile k multiplies, one barrier per tile); LDS rows are padded to 72 halfs so that the 16-byte fragment reads of
// 8 consecutive lanes fall on distinct banks.  72 KB of LDS per workgroup: two workgroups share a CU.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gemm_common.h"
#include "tgemm.h"
#include "tgemm_dev.h"
#include "split_f16.h"

namespace said {


constexpr int TBM = 128, TBN = 128, TBK = 64, TLP = 72;   // tile, LDS row pitch in halfs


// BN = 128: wave grid 2 x 2, each wave 64 tokens x 64 outputs; BN = 64 (N = 192, 576): each wave 64 tokens x 32 outputs
// SB (single LDS buffer): one operand buffer and one register set instead of two of each — 36.9 KB of LDS and ~84 VGPRs, so FOUR
// workgroups share a CU (two with the double buffer); a k-tile then costs two barriers, which the other three workgroups fill.
template <int BN, bool SB = false>
__global__ __launch_bounds__(256) void tgemm_kernel(const TGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];   // [2 buffers][A 128 x 72 | W BN x 72]
    constexpr int NJ = BN / 64;                 // MFMA column tiles per wave
    constexpr int WCH = BN * 8 / 256;           // 16-byte W chunks per thread and tile
    constexpr int BUF = (TBM + BN) * TLP;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    // XCD-aware tile order.  The hardware places consecutive workgroup ids on consecutive XCDs (id % 8), each with its own L2.
    // The N / BN workgroups that share an A tile are given consecutive slots of ONE XCD (n tile fastest), so the A tile is
    // fetched from HBM / Infinity Cache once and hits that XCD's L2 for the other column tiles; the weights are small and
    // stay resident in every L2.  (The natural (m, n) grid re-fetched every A tile N / BN times from beyond L2.)
    // The 1-D grid enumerates (sample, M tile) pairs of the whole batch, so all eight XCDs stay busy whatever M is.
    const int NT = a.N / BN, MT = (a.M + TBM - 1) / TBM;
    const unsigned L = blockIdx.x, xcd = L & 7u, slot = L >> 3;
    const int nt = (int)(slot % (unsigned)NT);
    const int mg = (int)(slot / (unsigned)NT) * 8 + (int)xcd;   // global M-tile index over the batch
    const int b = mg / MT, mt_ = mg - b * MT;
    if (b >= a.batch) return;   // padding of the tile count to a multiple of 8 (the whole workgroup exits together)
    const int m0 = mt_ * TBM, n0 = nt * BN;
    // grouped launch (a.grp > 1: the positional convolution's 16 groups): the grid's batch axis is (sample, group); a group has its own
    // A columns / weights and writes columns [g col_gs, g col_gs + n_store) of the sample's output rows
    int bs = b, g = 0;
    if (a.grp > 1) { bs = b / a.grp; g = b - bs * a.grp; }
    const unsigned short* A = reinterpret_cast<const unsigned short*>(a.a) + (long long)bs * a.a_bs + (long long)g * a.a_gs;
    const unsigned short* A2 = reinterpret_cast<const unsigned short*>(a.a2) + (long long)bs * a.a2_bs;
    const unsigned short* W = reinterpret_cast<const unsigned short*>(a.w) + (long long)g * a.w_gs;
    const int nk = a.K / TBK;
    const int nk1 = (a.a2 ? a.K1 : a.K) / TBK;   // K tiles served by the first A segment

    // global -> register staging: chunk c of a tile: row c >> 3, k piece c & 7 (16 bytes).  TWO register sets: the tile two
    // steps ahead is requested while the current one multiplies, so every load has a whole k-step (plus the other
    // workgroup of the CU) to arrive — one set gave each load only the ~500 clocks of one step's MFMAs.
    struct RegTile { u32x4 a[4]; u32x4 w[WCH]; };
    RegTile S0, S1;
    long long aoff[4], a2off[4], woff[WCH];
    int loff[4], lwoff[WCH];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, row = c >> 3, kp = c & 7;
        const int m = min(m0 + row, a.M - 1);   // rows past M repeat the last row (never stored)
        aoff[i] = (long long)m * a.lda + kp * 8;
        a2off[i] = (long long)m * a.lda2 + kp * 8;
        loff[i] = row * TLP + kp * 8;
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int c = tid + 256 * i, row = c >> 3, kp = c & 7;
        woff[i] = (long long)(n0 + row) * a.K + kp * 8;
        lwoff[i] = row * TLP + kp * 8;
    }
    auto gload_tile = [&](RegTile& R, int kt) {
        const bool first = kt < nk1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned short* p = first ? A + aoff[i] + (long long)kt * TBK : A2 + a2off[i] + (long long)(kt - nk1) * TBK;
            R.a[i] = *reinterpret_cast<const u32x4*>(p);
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) R.w[i] = *reinterpret_cast<const u32x4*>(W + woff[i] + (long long)kt * TBK);
    };
    auto lds_store = [&](const RegTile& R, int buf) {
        unsigned short* pa = lds + buf * BUF;
        unsigned short* pw = pa + TBM * TLP;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(pa + loff[i]) = R.a[i];
#pragma unroll
        for (int i = 0; i < WCH; ++i) *reinterpret_cast<u32x4*>(pw + lwoff[i]) = R.w[i];
    };

    f32x16 acc0[NJ], acc1[NJ];   // two row tiles per wave (separate arrays: one [2][NJ] array of this size is not promoted to registers)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.f; acc1[j][r] = 0.f; }

    const int frow = l & 31, fk = 8 * (l >> 5);
    auto compute = [&](int buf) {
        const unsigned short* pa = lds + buf * BUF;
        const unsigned short* pw = pa + TBM * TLP;
#pragma unroll
        for (int ks = 0; ks < TBK / 16; ++ks) {
            bf16x8 fa[2], fb[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(pa + (wm * 64 + i * 32 + frow) * TLP + ks * 16 + fk);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(pw + (wn * (32 * NJ) + j * 32 + frow) * TLP + ks * 16 + fk);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j], acc0[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[j], acc1[j], 0, 0, 0);
            }
        }
    };

    // Every load and LDS store of the loop is UNCONDITIONAL (steps past the end re-request the last tile and park it in the
    // buffer nobody reads any more): with memory operations inside run-time branches the compiler's wait-count analysis
    // gives up at the joins and drains every outstanding load (s_waitcnt vmcnt(0)) before it issues the next tile's — which
    // is exactly the overlap the second register set exists for.
    if constexpr (SB) {
        gload_tile(S0, 0);
        lds_store(S0, 0);
        gload_tile(S0, min(1, nk - 1));
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {   // LDS holds tile kt, S0 tile kt + 1 (in flight)
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                 // every wave has read tile kt
            lds_store(S0, 0);
            gload_tile(S0, min(kt + 2, nk - 1));
            __syncthreads();
        }
    } else {
    gload_tile(S0, 0);
    gload_tile(S1, min(1, nk - 1));
    lds_store(S0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // even step: tile kt multiplies from buffer 0, tile kt + 1 sits in S1, tile kt + 2 is requested into S0
        gload_tile(S0, min(kt + 2, nk - 1));
        __builtin_amdgcn_sched_barrier(0);   // keep the order request -> multiply -> park: the scheduler otherwise hoists the
        compute(0);                          // LDS stores (and the wait for their loads) above the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        lds_store(S1, 1);
        __syncthreads();
        // odd step: tile kt + 1 from buffer 1, tile kt + 2 sits in S0, tile kt + 3 is requested into S1
        gload_tile(S1, min(kt + 3, nk - 1));
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) compute(1);
        __builtin_amdgcn_sched_barrier(0);
        lds_store(S0, 0);
        __syncthreads();
    }
    }

    // the K loop ended with a barrier: the operand buffers are free and serve as per-wave transposition scratch
    float* sc = reinterpret_cast<float*>(lds) + w * (32 * (32 * NJ + 4));
    const int ncol = n0 + g * a.col_gs + wn * (32 * NJ), nlim = a.grp > 1 ? g * a.col_gs + a.n_store : 0;
    tg_epilogue<NJ>(a, acc0, bs, m0 + wm * 64, ncol, l, sc, nullptr, 0, nlim);
    tg_epilogue<NJ>(a, acc1, bs, m0 + wm * 64 + 32, ncol, l, sc, nullptr, 0, nlim);
}

// ------------------------------------------------------------------------------------------------------------------
// 256-row tile (large M): 8 waves as 4 (rows) x 2 (columns), each 64 rows x BN / 2 columns, BN = 256 (N % 256 == 0: the audio
// encoder's 512 / 768 / 2304 / 3072-wide outputs, GEGLU) or 192 (the UNet's 192 / 576-wide outputs).  The 128-row kernel
// above was bound by operand bytes per FLOP, not by MFMA: 15.6-24 B per kFLOP from L2 with two tiles in flight per workgroup
// left the MFMA pipes 11-22 % busy and the waves 45 % parked on s_waitcnt (profiles/r02c_pmc_sq_b32_bf16.txt).  This shape moves
// 7.8 (256 x 256) / 9.1 (256 x 192) B per kFLOP and gives each wave 32 / 24 MFMAs per k-tile to hide the next tile's loads.
// With seg_rows > 0 the row axis is the whole batch (per-sample pitch seg_rows), so M = 600 does not cost tile padding.
// ------------------------------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(512) void tgemm256_kernel(const TGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];   // [2 buffers][A 256 x 72 | W BN x 72]
    constexpr int BM2 = 256;
    constexpr int NJ = BN / 64;                 // MFMA column tiles per wave (4 or 3)
    constexpr int WCH = BN * 8 / 512;           // 16-byte W chunks per thread and tile (4 or 3)
    constexpr int BUF = (BM2 + BN) * TLP;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int rows_tot = a.seg_rows > 0 ? a.batch * a.seg_rows : a.M;   // rows of the A operand per grid batch entry
    const int nbatch = a.seg_rows > 0 ? 1 : a.batch;
    const int NT = a.N / BN, MT = (rows_tot + BM2 - 1) / BM2;
    const unsigned L = blockIdx.x, xcd = L & 7u, slot = L >> 3;   // XCD-aware order, as in tgemm_kernel
    const int nt = (int)(slot % (unsigned)NT);
    const int mg = (int)(slot / (unsigned)NT) * 8 + (int)xcd;
    const int b = mg / MT, mt_ = mg - b * MT;
    if (b >= nbatch) return;
    const int m0 = mt_ * BM2, n0 = nt * BN;
    const unsigned short* A = reinterpret_cast<const unsigned short*>(a.a) + (long long)b * a.a_bs;
    const unsigned short* A2 = reinterpret_cast<const unsigned short*>(a.a2) + (long long)b * a.a2_bs;
    const unsigned short* W = reinterpret_cast<const unsigned short*>(a.w);
    const int nk = a.K / TBK;
    const int nk1 = (a.a2 ? a.K1 : a.K) / TBK;

    struct RegTile { u32x4 a[4]; u32x4 w[WCH]; };
    RegTile S0, S1;
    int aoff[4], a2off[4], woff[WCH];   // element offsets within one sample's operand / the weight matrix: < 2^31 (host-checked)
    int loff[4], lwoff[WCH];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 512 * i, row = c >> 3, kp = c & 7;
        const int m = min(m0 + row, rows_tot - 1);
        aoff[i] = m * a.lda + kp * 8;
        a2off[i] = m * a.lda2 + kp * 8;
        loff[i] = row * TLP + kp * 8;
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int c = tid + 512 * i, row = c >> 3, kp = c & 7;
        woff[i] = (n0 + row) * a.K + kp * 8;
        lwoff[i] = row * TLP + kp * 8;
    }
    auto gload_tile = [&](RegTile& R, int kt) {
        const bool first = kt < nk1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned short* p = first ? A + (aoff[i] + kt * TBK) : A2 + (a2off[i] + (kt - nk1) * TBK);
            R.a[i] = *reinterpret_cast<const u32x4*>(p);
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) R.w[i] = *reinterpret_cast<const u32x4*>(W + (woff[i] + kt * TBK));
    };
    auto lds_store = [&](const RegTile& R, int buf) {
        unsigned short* pa = lds + buf * BUF;
        unsigned short* pw = pa + BM2 * TLP;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(pa + loff[i]) = R.a[i];
#pragma unroll
        for (int i = 0; i < WCH; ++i) *reinterpret_cast<u32x4*>(pw + lwoff[i]) = R.w[i];
    };
    f32x16 acc0[NJ], acc1[NJ];   // two row tiles per wave (separate arrays: one [2][NJ] array of this size is not promoted to registers)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.f; acc1[j][r] = 0.f; }
    const int frow = l & 31, fk = 8 * (l >> 5);
    auto compute = [&](int buf) {
        const unsigned short* pa = lds + buf * BUF;
        const unsigned short* pw = pa + BM2 * TLP;
#pragma unroll
        for (int ks = 0; ks < TBK / 16; ++ks) {
            bf16x8 fa[2], fb[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(pa + (wm * 64 + i * 32 + frow) * TLP + ks * 16 + fk);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(pw + (wn * (32 * NJ) + j * 32 + frow) * TLP + ks * 16 + fk);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j], acc0[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[j], acc1[j], 0, 0, 0);
            }
        }
    };
    gload_tile(S0, 0);
    gload_tile(S1, min(1, nk - 1));
    lds_store(S0, 0);
    __syncthreads();
    for (int kt = 0; kt < ((a.dbg & 2) ? 0 : nk); kt += 2) {
        gload_tile(S0, min(kt + 2, nk - 1));
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        lds_store(S1, 1);
        __syncthreads();
        gload_tile(S1, min(kt + 3, nk - 1));
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) compute(1);
        __builtin_amdgcn_sched_barrier(0);
        lds_store(S0, 0);
        __syncthreads();
    }
    // the K loop ended with a barrier: the operand buffers are free and serve as per-wave transposition scratch
    float* sc = reinterpret_cast<float*>(lds) + w * (32 * (32 * NJ + 4));
    tg_epilogue<NJ>(a, acc0, b, m0 + wm * 64, n0 + wn * (32 * NJ), l, sc);
    tg_epilogue<NJ>(a, acc1, b, m0 + wm * 64 + 32, n0 + wn * (32 * NJ), l, sc);
}

// ------------------------------------------------------------------------------------------------------------------
// tgemm256d_kernel (round 6): the 256 x 256 x 64 tile with its operand tiles fetched global -> LDS DIRECTLY (buffer_load_dwordx4 ... lds): no staging registers, no
// LDS stores, one barrier per k-tile.  Direct loads write a wave's 64 x 16 bytes contiguously (8 rows x 128 bytes: no row padding), so the 16-byte chunks are
// XOR-swizzled instead — LDS chunk p of row r holds the row's k-chunk p ^ swz(r), and a fragment read of chunk c takes p = c ^ swz(r) (TG256D_SWZ below).  Bring-up and knock-outs: scripts/ubench/bgemm.hip (profiles/r06k_bgemm_bringup.txt): the audio encoder's four projection shapes at
// 32 clips take 77 / 28 / 85 / 83 us with a plain store epilogue where tgemm_kernel<128, SB> averages 124 and tgemm256_kernel 135; with every load, barrier and
// LDS read of the k16 steps knocked out the loop still takes 63-65 us: prologue, epilogue and three rounds of 256 workgroups are what is left above the MFMAs.
// Same operands, same k order per accumulator as the other bf16 tiles: bit-identical results.  Per-sample operands only (seg_rows == 0), one K segment, N % 256 == 0.
// ------------------------------------------------------------------------------------------------------------------
// The swizzle term of row r.  A 128-byte row is half of the 64 banks (its parity picks the half), and a ds_read_b128 is serviced in FOUR groups of 16 lanes that are NOT
// contiguous — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS table): a group holds eight rows of each parity, so the eight
// chunk positions must be told apart by (r >> 1) & 7.  The first version used r & 7 — right for sixteen CONSECUTIVE rows — and every fragment read was a 2-way conflict
// (SQ_LDS_BANK_CONFLICT 46 % of the LDS-active cycles: profiles/r06m_sq_lds_l2_counters.txt).
#ifndef TG256D_SWZ
#define TG256D_SWZ(r) (((r) >> 1) & 7)
#endif
constexpr int TG256D_TILE = (256 + 256) * 128;                                   // bytes of one buffer: A 256 rows + W 256 rows x 64 bf16
constexpr int TG256D_LDS = 2 * TG256D_TILE > 8 * 32 * (32 * 4 + 4) * 4 ? 2 * TG256D_TILE : 8 * 32 * (32 * 4 + 4) * 4;   // two buffers / the epilogue's per-wave scratch
__global__ __launch_bounds__(512) void tgemm256d_kernel(const TGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr int BM2 = 256, BN = 256, NJ = 4;
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int NT = a.N / BN, MT = (a.M + BM2 - 1) / BM2;
    const unsigned L = blockIdx.x, xcd = L & 7u, slot = L >> 3;   // XCD-aware order, as in tgemm_kernel
    const int nt = (int)(slot % (unsigned)NT);
    const int mg = (int)(slot / (unsigned)NT) * 8 + (int)xcd;
    const int b = mg / MT, mt_ = mg - b * MT;
    if (b >= a.batch) return;
    const int m0 = mt_ * BM2, n0 = nt * BN;
    const int nk = a.K / TBK;
    const rsrc_t ra = make_rsrc(reinterpret_cast<const unsigned short*>(a.a) + (long long)b * a.a_bs, (unsigned)(((long long)(a.M - 1) * a.lda + a.K) * 2));
    const rsrc_t rw = make_rsrc(a.w, (unsigned)((long long)a.N * a.K * 2));
    // wave-load j of an operand tile = rows 8 j .. 8 j + 7 (1 KB, contiguous in LDS); lane -> (row 8 j + (l >> 3), LDS chunk l & 7) = the row's k-chunk (l & 7) ^ (l >> 3)
    const int lrow = l >> 3;
    int aoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = w * 4 + i;
        const int lchunk = (l & 7) ^ TG256D_SWZ(8 * j + lrow);
        aoff[i] = (min(m0 + 8 * j + lrow, a.M - 1) * a.lda + lchunk * 8) * 2;   // rows past M repeat the last row (never stored)
        woff[i] = ((n0 + 8 * j + lrow) * a.K + lchunk * 8) * 2;
    }
    auto issue = [&](int kt, int buf) {
        char* base = ldsb + buf * TG256D_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = w * 4 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(base + j * 1024), 16, aoff[i], kt * (TBK * 2), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(base + BM2 * 128 + j * 1024), 16, woff[i], kt * (TBK * 2), 0, 0);
        }
    };
    f32x16 acc0[NJ], acc1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.f; acc1[j][r] = 0.f; }
    const int frow = l & 31, fh = l >> 5;
    auto compute = [&](int buf) {
        const char* pa = ldsb + buf * TG256D_TILE;
        const char* pw = pa + BM2 * 128;
#pragma unroll
        for (int ks = 0; ks < TBK / 16; ++ks) {
            const int c = ks * 2 + fh;
            bf16x8 fa[2], fb[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + frow;
                fa[i] = *reinterpret_cast<const bf16x8*>(pa + row * 128 + ((c ^ TG256D_SWZ(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = wn * (32 * NJ) + j * 32 + frow;
                fb[j] = *reinterpret_cast<const bf16x8*>(pw + row * 128 + ((c ^ TG256D_SWZ(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j], acc0[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[j], acc1[j], 0, 0, 0);
            }
        }
    };
    issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // (vmcnt(0): the tile is in LDS)
    __syncthreads();
    for (int kt = 0; kt < ((a.dbg & 2) ? 0 : nk); ++kt) {
        // tile kt + 1 goes into the buffer tile kt - 1 was read from: every wave passed the barrier that ended that step
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
    // the K loop ended with a barrier: the operand buffers are free and serve as per-wave transposition scratch
    float* sc = reinterpret_cast<float*>(lds) + w * (32 * (32 * NJ + 4));
    tg_epilogue<NJ>(a, acc0, b, m0 + wm * 64, n0 + wn * (32 * NJ), l, sc);
    tg_epilogue<NJ>(a, acc1, b, m0 + wm * 64 + 32, n0 + wn * (32 * NJ), l, sc);
}

// ------------------------------------------------------------------------------------------------------------------
// fp32 token-major GEMM (fp32 mode, large batches: BASELINE configs[3]'s per-GPU work) on v_mfma_f32_32x32x2_f32.
// Same operand geometry in BYTES as the bf16 kernels — a k-tile is 128 bytes per row (32 floats), LDS rows 144 bytes, 16-byte
// fragment reads — and the same epilogue.  A lane's 16-byte fragment holds k = 8 s + 4 (l >> 5) + {0..3}; MFMA i of step s
// multiplies element i of the A and W fragments, i.e. the k pair (8 s + i, 8 s + 4 + i) — a permutation of k both operands share.
//
// Workgroup: tile 64 rows x 32 NJ columns, ONE LDS operand buffer (23-28 KB), FOUR waves = 2 row halves x 2 K HALVES: wave
// (r, kh) multiplies rows 32 r .. 32 r + 31 by k = 16 kh .. 16 kh + 15 of every 32-k tile; the two K halves are added through
// LDS once, after the loop (12-16 KB per pair), and wave (r, 0) runs the epilogue.  3-4 workgroups share a CU, each in its own
// phase, so one's prologue / barriers / epilogue hide under the others' MFMAs.
//
// Why this shape (every step measured on the MI355X, scripts/gpu_r2_m.sh ... gpu_r2_u.sh):
//  * the channel-major ugemm family splits K over the 8 waves of a 32-token tile and pays a 64 KB LDS reduction per tile: 40 % of
//    the fp32 MFMA roof at Be = 64.  A first token-major shape (64 x 192 tile, 4 waves, each 32 x 96 over the whole K, double-
//    buffered) ran the 192-wide convolutions at 144 us where the MFMAs alone need 55.
//  * these GEMMs are MFMA-bound, and an MFMA-bound launch is a bin-packing of indivisible wave-tiles onto 1024 SIMDs: 38912 rows
//    x 192 columns in 32 x 96 wave-tiles over the whole K are 2432 units = 2.375 per SIMD -> 3 on the busiest, 79 % at best,
//    whatever the workgroup shape.  Halving K per wave halves the unit (4.75 -> 5 per SIMD: 95 %) WITHOUT extra operand traffic
//    — all four waves read the same LDS tiles.  (Smaller output tiles would also balance, but cost L2 bandwidth, see below.)
//  * knock-outs of this kernel's loop: no barriers -0 %, no LDS stores -3 %, no global loads -22 %.  The loads are not waited
//    for (average L2 latency seen by the L1 is 219 clocks, TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ; a second register set,
//    PF = 2, buys 5 %); what they cost is the MFMA RATE itself (power-managed clock, or register-file / issue contention — not
//    separated): scripts/ubench/mfma_with_loads.hip — pure fp32 MFMA loops on all CUs — sustains
//    150 TFLOP/s alone, 135 with 3.2 TB/s of independent L2 loads beside them, 114 with 5.4 TB/s, 112 with 10.5 TB/s.  A 64 x 96
//    tile needs 20 KB per 48 MFMA-times: ~5 TB/s at the rate it runs.  So ~115 TFLOP/s is the practical roof of an fp32 GEMM at
//    these tile sizes, and this kernel's 93-98 (convolutions), 87 (q/k/v, K = 192) and 93 (GEGLU) sit at 75-85 % of it.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FBK = 32;   // k per tile of the fp32 kernel (host-side checks)
template <int NJ>
__host__ __device__ constexpr int fgemm_lds_bytes() {
    const int tiles = (64 + 32 * NJ) * 144, exch = 2 * NJ * 16 * 64 * 4,
              scratch = NJ == 4 ? 4 * 32 * (32 * 2 + 4) * 4 : 2 * 32 * (32 * NJ + 4) * 4;
    return tiles > scratch ? (tiles > exch ? tiles : exch) : (scratch > exch ? scratch : exch);
}
// NJ = 3: 64 x 96 tile (N = 192 / 576), NJ = 4: 64 x 128 (GEGLU, value / gate column tiles interleaved as for the bf16 kernel).
// PF = 2: two register sets, the tile two k-steps ahead is in flight while the current one multiplies.
// BF: the same workgroup on bf16 operands (bf16 mode's UNet GEMMs): a k-tile is again 128 bytes per row (64 halfs), each K half
// two v_mfma_f32_32x32x16_bf16 per column tile.  There the point is not MFMA balance but spread: the 256-row bf16 tiles put a
// 192-wide convolution on 152 workgroups of a 256-CU chip, and its time is the fp32 epilogue traffic (§7.3).
constexpr int FGEMM_PK_LDS3 = 2 * (64 + 96) * 144 > fgemm_lds_bytes<3>() ? 2 * (64 + 96) * 144 : fgemm_lds_bytes<3>();   // packed mode: two operand buffers
constexpr int fgemm_occ(int NJ, int PF, bool BF) { return NJ == 3 ? 4 : 3; }   // workgroups per CU the registers are budgeted for
// SP (round 4, fp32 operands only): the products run on SPLIT-fp16 operands (split_f16.h: x = h + 2^-11 l, three v_mfma_f32_32x32x16_f16 per eight
// v_mfma_f32_32x32x2_f32, fp32 accumulation, the cross terms in a second accumulator set).  The LDS tiles stay fp32 — staging, K halves, exchange and
// epilogue are untouched; a wave's lane half takes the EIGHT consecutive k (16 kh + 8 lh ...) of the 32-k tile as two 16-byte reads per operand row and
// splits them in registers (A once, W once per column tile — ALL of a k-tile's operands in distinct registers: two workgroups per CU), then operand_fence(), then the
// 3 NJ MFMAs, then a second fence.  A variant that split one column tile at a time (three workgroups per CU) — its conversions rewriting the operand registers of
// MFMAs issued 16 idle slots earlier — was not bit-stable from one run to the next (profiles/r04i_attn_split_hazard.txt).
// PK (round 6, SP only): the operands ARRIVE split — every element of A / A2 / W is one dword h | l << 16 (prep_kernel's pack mode, engine.cpp's packed weight copies) — and a
// fragment is unpacked with eight v_perm_b32 instead of ~40 VALU instructions of conversion: with one k16 step (9 MFMAs of 8 passes) per k-tile and wave, the splits of A and of
// three W fragments were 2.4 x the matrix time.  Same planes, same products: bit-identical to the in-kernel split.
template <int NJ, int PF, bool BF, int OCC = fgemm_occ(NJ, PF, BF), bool SP = false, bool PK = false>
__global__ __launch_bounds__(256, OCC) void fgemm_kernel(const TGemmArgs a) {
    static_assert(!(SP && BF), "the split mode reads fp32 operands");
    static_assert(!PK || SP, "packed operands are split operands");
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];   // [A 64 rows | W BN rows] x 144 bytes
    float* const ldsf = reinterpret_cast<float*>(lds);
    typedef typename std::conditional<BF, unsigned short, float>::type elt_t;
    constexpr int EPC = BF ? 8 : 4;              // elements per 16-byte chunk
    constexpr int FBK = 8 * EPC, FLP = 9 * EPC;  // k per tile (128 bytes), LDS row pitch (144 bytes), in elements
    elt_t* const ldse = reinterpret_cast<elt_t*>(lds);
    constexpr int BM = 64, BN = 32 * NJ, NTH = 256;
    constexpr int ACH = BM * 8 / NTH, WCH = BN * 8 / NTH;   // 16-byte chunks per thread and tile: 2, NJ
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wr = w & 1, kh = w >> 1;
    const int rows_tot = a.batch * a.seg_rows;   // batch-as-rows addressing only (tgemm_supports)
    const int NT = a.N / BN, MT = (rows_tot + BM - 1) / BM;
    const unsigned L = blockIdx.x, xcd = L & 7u, slot = L >> 3;   // XCD-aware order, as in tgemm_kernel
    const int nt = (int)(slot % (unsigned)NT);
    const int mg = (int)(slot / (unsigned)NT) * 8 + (int)xcd;
    if (mg >= MT) return;   // padding of the tile count to a multiple of 8 (the whole workgroup exits together)
    const int m0 = mg * BM, n0 = nt * BN;
    const elt_t* A = reinterpret_cast<const elt_t*>(a.a);
    const elt_t* A2 = reinterpret_cast<const elt_t*>(a.a2);
    const elt_t* W = reinterpret_cast<const elt_t*>(a.w);
    const int nk = a.K / FBK;
    const int nk1 = (a.a2 ? a.K1 : a.K) / FBK;

    f32x4t ra[ACH], rw[WCH], ra1[PF == 2 ? ACH : 1], rw1[PF == 2 ? WCH : 1];
    int aoff[ACH], a2off[ACH], woff[WCH];   // element offsets: < 2^31 (host-checked)
    int loff[ACH], lwoff[WCH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int c = tid + NTH * i, row = c >> 3, kp = c & 7;
        const int m = min(m0 + row, rows_tot - 1);
        aoff[i] = m * a.lda + kp * EPC;
        a2off[i] = m * a.lda2 + kp * EPC;
        loff[i] = row * FLP + kp * EPC;
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int c = tid + NTH * i, row = c >> 3, kp = c & 7;
        woff[i] = (n0 + row) * a.K + kp * EPC;
        lwoff[i] = BM * FLP + row * FLP + kp * EPC;
    }
    auto gload_tile = [&](f32x4t* xa, f32x4t* xw, int kt) {
        const bool first = kt < nk1;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const elt_t* p = first ? A + (aoff[i] + kt * FBK) : A2 + (a2off[i] + (kt - nk1) * FBK);
            xa[i] = *reinterpret_cast<const f32x4t*>(p);
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) xw[i] = *reinterpret_cast<const f32x4t*>(W + (woff[i] + kt * FBK));
    };
    // PK: TWO operand buffers in LDS — the next tile is parked in the other buffer before the one barrier of a k-step (the single-buffer loop pays two barriers per
    // nine MFMAs of a wave); the other variants keep one buffer (their k loop is bound elsewhere, and their occupancy is budgeted on 23-28 KB)
    constexpr int BUFE = PK ? (BM + BN) * FLP : 0;   // elements between the two buffers
    auto lds_store = [&](const f32x4t* xa, const f32x4t* xw, int buf = 0) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) *reinterpret_cast<f32x4t*>(ldse + buf * BUFE + loff[i]) = xa[i];
#pragma unroll
        for (int i = 0; i < WCH; ++i) *reinterpret_cast<f32x4t*>(ldse + buf * BUFE + lwoff[i]) = xw[i];
    };
    f32x16 acc[NJ], accx[SP ? NJ : 1];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[j][r] = 0.f;
            if (SP) accx[SP ? j : 0][r] = 0.f;
        }
    // fragment of step ks: bytes 64 kh + 32 ks + 16 (l >> 5) of the row — the same byte offsets for both element types
    const int frow = l & 31, fk = EPC * (l >> 5) + 4 * EPC * kh;
    const elt_t* const pa = ldse + (wr * 32 + frow) * FLP + fk;
    const elt_t* const pw = ldse + BM * FLP + frow * FLP + fk;
    // (split mode: floats 16 kh + 8 lh .. + 7 of the row)
    const float* const paS = ldsf + (wr * 32 + frow) * 36 + 16 * kh + 8 * (l >> 5);
    const float* const pwS = ldsf + BM * 36 + frow * 36 + 16 * kh + 8 * (l >> 5);
    auto compute = [&](int buf = 0) {
        if constexpr (SP) {
            const float* const pa2 = paS + buf * BUFE;
            const float* const pw2 = pwS + buf * BUFE;
            const SplitH sa = PK ? unpack_f16x8(*reinterpret_cast<const f32x4s*>(pa2), *reinterpret_cast<const f32x4s*>(pa2 + 4))
                                 : split_f16x8(*reinterpret_cast<const f32x4s*>(pa2), *reinterpret_cast<const f32x4s*>(pa2 + 4));
            SplitH sb[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                sb[j] = PK ? unpack_f16x8(*reinterpret_cast<const f32x4s*>(pw2 + j * 32 * 36), *reinterpret_cast<const f32x4s*>(pw2 + j * 32 * 36 + 4))
                           : split_f16x8(*reinterpret_cast<const f32x4s*>(pw2 + j * 32 * 36), *reinterpret_cast<const f32x4s*>(pw2 + j * 32 * 36 + 4));
            operand_fence();
            // two MFMAs on the same accumulator are NJ - 1 or more apart (never back to back: attn.hip)
#pragma unroll
            for (int j = 0; j < NJ; ++j) accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa.l, sb[j].h, accx[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa.h, sb[j].h, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa.h, sb[j].l, accx[j], 0, 0, 0);
            operand_fence();   // (the next k-tile's split reuses these operand registers)
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (BF) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(pa + ks * 16);
                bf16x8 fb[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(pw + j * 32 * FLP + ks * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[j], acc[j], 0, 0, 0);
            } else {
                const f32x4t fa = *reinterpret_cast<const f32x4t*>(pa + ks * 8);
                f32x4t fb[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const f32x4t*>(pw + j * 32 * FLP + ks * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j][i], acc[j], 0, 0, 0);
            }
        }
    };
    // Every load and LDS store of the loop is unconditional, as in tgemm_kernel (steps past the end re-request the last tile).
    // One k-step: request a later tile -> multiply the tile in LDS -> barrier (all four waves have read it) -> park the next
    // tile -> barrier.
    const int nloop = (a.dbg & 2) ? 0 : nk;
    gload_tile(ra, rw, 0);
    if constexpr (PF == 2) gload_tile(ra1, rw1, min(1, nk - 1));
    lds_store(ra, rw);
    __syncthreads();
    if constexpr (PF == 2) {
        for (int kt = 0; kt < nloop; kt += 2) {
            gload_tile(ra, rw, min(kt + 2, nk - 1));
            __builtin_amdgcn_sched_barrier(0);
            compute();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            lds_store(ra1, rw1);
            __syncthreads();
            gload_tile(ra1, rw1, min(kt + 3, nk - 1));
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) compute();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            lds_store(ra, rw);
            __syncthreads();
        }
    } else if constexpr (PK) {
        // tile kt multiplies from buffer kt & 1 while tile kt + 1 (in registers since the previous step) is parked in the other one — free since every wave passed the
        // previous barrier behind its products on it — and tile kt + 2 is requested: ONE barrier per k-step
        gload_tile(ra, rw, min(1, nk - 1));
        for (int kt = 0; kt < nloop; ++kt) {
            __builtin_amdgcn_sched_barrier(0);
            compute(kt & 1);
            __builtin_amdgcn_sched_barrier(0);
            lds_store(ra, rw, (kt + 1) & 1);
            gload_tile(ra, rw, min(kt + 2, nk - 1));
            __syncthreads();
        }
    } else {
        for (int kt = 0; kt < nloop; ++kt) {
            gload_tile(ra, rw, min(kt + 1, nk - 1));
            __builtin_amdgcn_sched_barrier(0);
            compute();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            lds_store(ra, rw);
            __syncthreads();
        }
    }
    // ---- add the two K halves.  NJ = 4 (GEGLU: the epilogue is 18 % of the kernel, mostly erf) splits the epilogue as well: wave
    // (r, 0) finishes column tiles [0, 2), wave (r, 1) tiles [2, 4) — each parks the tiles the OTHER one finishes in the exchange
    // area (lane-linear, region r), a barrier, each adds its partner's half to its own; a second barrier frees the area, which
    // then serves as the waves' transposition scratch (245 -> 235 us).  NJ = 3: wave (r, 1) parks everything, wave (r, 0) finishes
    // all three tiles (the 2 : 1 split measured slower: 122.5 -> 126.7 us).
    if constexpr (SP) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = fmaf(accx[j][r], 0x1p-11f, acc[j][r]);
    }
    constexpr int NJ0 = NJ == 4 ? 2 : NJ, NJ1 = NJ - NJ0;
    // (unsplit: region r is also wave (r, 0)'s scratch, so the regions are spaced by the scratch size and never overlap)
    float* const xr = ldsf + wr * (NJ1 > 0 ? NJ * 16 * 64 : 32 * (32 * NJ + 4));
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if ((kh == 1) == (j < NJ0)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xr[(j * 16 + r) * 64 + l] = acc[j][r];
        }
    }
    __syncthreads();
    if (NJ1 > 0 || kh == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if ((kh == 0) == (j < NJ0)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += xr[(j * 16 + r) * 64 + l];
            }
        }
    }
    if constexpr (NJ1 > 0) {
        __syncthreads();
        float* sc = ldsf + w * (32 * (32 * NJ0 + 4));
        if (kh == 0) tg_epilogue<NJ, 0, NJ0>(a, acc, 0, m0 + wr * 32, n0, l, sc);
        else tg_epilogue<NJ, NJ0, (NJ1 > 0 ? NJ1 : 1)>(a, acc, 0, m0 + wr * 32, n0 + 32 * NJ0, l, sc);
    } else {
        // region r holds only wave (r, 0)'s partner data, which it has just consumed: the row half's transposition scratch (in-order
        // LDS).  Round 3: wave (r, 0) runs phase 1 alone, then BOTH waves of the row half share phase 2 — half the channels (channel-
        // major results) or half the rows (token-major ones) each; before, wave (r, 1) had exited and two of the workgroup's four
        // waves carried the whole memory-facing half of the kernel.
        if (kh == 0) {
            __builtin_amdgcn_wave_barrier();
            tg_epilogue<NJ, 0, NJ, -1, 1>(a, acc, 0, m0 + wr * 32, n0, l, xr);
        }
        __syncthreads();
        tg_epilogue<NJ, 0, NJ, -1, 2>(a, acc, 0, m0 + wr * 32, n0, l, xr, nullptr, kh);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// xgemm_kernel (round 3) — GEMMs on TOKEN-MAJOR ACTIVATIONS with the operand transform inside.
//
// Round 2's large-batch path kept fp32 channel-major activations between kernels and ran a preparation kernel in front of every
// GEMM (20 launches, 17 % of the bf16 step, and the GEMM epilogues wrote 59 MB of fp32 per 192-wide convolution against 14.8 MB
// of bf16 operand).  Here the activations between the UNet kernels ARE token-major [sample][token][192] in the element type
// (bf16 / fp32), and the consuming GEMM does the normalisation itself:
//   * RS (resident source): the workgroup's source tile — its 64 tokens (+ the two Conv1d halo tokens) x 192 channels — is
//     loaded once, transformed once per element (silu(GroupNorm) / LayerNorm / LayerNorm(GroupNorm); the GroupNorm coefficients
//     are finalised from the producer's partials in the prologue, the LayerNorm statistics taken over the row's four threads)
//     and parked in LDS [66][192 + pad]; the three taps of a convolution are three row offsets into it, a concatenated input
//     (384 channels) is two passes over the same 26 / 52 KB buffer.  Only the weights stream through the k loop.
//   * SS (streamed sources): raw operands (attention output, GEGLU product, x2, the 1x1 skip over the concatenated input) go
//     through the k-tile pipeline of fgemm_kernel; up to three sources are chained along K.
//   * TR: the product is formed transposed (D[channel][token]: operand roles swapped) for the banded cross-attention epilogue,
//     where a lane owns one query token and a head's 32 channels sit in 16 registers of the two lane halves.
// Workgroup, wave roles (2 row halves x 2 K halves), k-tile geometry and the K-half exchange are fgemm_kernel's.
// ------------------------------------------------------------------------------------------------------------------
constexpr int XRR = 66;                       // resident rows: 64 tokens + 2 halo
// Workgroups per CU the registers are budgeted for.  The resident-source variants at THREE per CU (168 VGPRs) spilled 56-128 bytes per lane
// — outside the k loop, and still enough to turn the token-major-activation schedule from 3 % faster than the default into 3 % slower
// (1.768 vs 1.892 ms per step, bf16, 32 clips): two per CU (196-212 VGPRs, scratch 0).
#ifndef XG_OCC_RS
#define XG_OCC_RS 2
#endif
#ifndef XG_OCC_SS
#define XG_OCC_SS 3
#endif
template <bool BF> __host__ __device__ constexpr int x_rp_bytes() { return 192 * (BF ? 2 : 4) + 16; }   // resident row pitch: 400 / 784 bytes (conflict-free 16-byte fragment reads)
constexpr int X_COEF_BYTES = 2 * 192 * 4;     // GroupNorm (a, b) of ONE 192-channel source at a time; after the prologue the region carries the
                                              // epilogue's statistics exchange (a kernel never needs both at once)
template <int NJ, bool BF>
__host__ __device__ constexpr int xgemm_lds_bytes(bool resident) {
    return fgemm_lds_bytes<NJ>() + X_COEF_BYTES + (resident ? XRR * x_rp_bytes<BF>() : 0);
}


template <int NJ, bool BF, bool RS, bool SS, bool TR, int EK, int OCC>
__global__ __launch_bounds__(256, OCC) void xgemm_kernel(const TGemmArgs a) {
    static_assert(RS || SS, "a GEMM needs an operand");
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];   // [stream A 64 rows | W BN rows] x 144 bytes | coefficients | resident tile
    float* const ldsf = reinterpret_cast<float*>(lds);
    typedef typename std::conditional<BF, unsigned short, float>::type elt_t;
    constexpr int EPC = BF ? 8 : 4;              // elements per 16-byte chunk
    constexpr int FBK = 8 * EPC, FLP = 9 * EPC;  // k per tile (128 bytes), LDS row pitch of the streamed tiles (144 bytes), in elements
    constexpr int CT = 192 / FBK;                // k-tiles per source and tap of the resident segment (3 bf16, 6 fp32)
    constexpr int RP = x_rp_bytes<BF>() / (int)sizeof(elt_t);   // resident row pitch in elements
    elt_t* const ldse = reinterpret_cast<elt_t*>(lds);
    float* const coefS = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + fgemm_lds_bytes<NJ>());
    elt_t* const ares = reinterpret_cast<elt_t*>(reinterpret_cast<char*>(lds) + fgemm_lds_bytes<NJ>() + X_COEF_BYTES);
    constexpr int BM = 64, BN = 32 * NJ, NTH = 256;
    constexpr int ACH = BM * 8 / NTH, WCH = BN * 8 / NTH;   // 16-byte chunks per thread and tile: 2, NJ
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wr = w & 1, kh = w >> 1;
    const int rows_tot = a.batch * a.seg_rows;
    const int NT = a.N / (BN * (a.ntw > 0 ? a.ntw : 1)), MT = rows_tot / BM;   // seg_rows % 64 == 0 (host-checked): a tile never straddles samples
    const unsigned L = blockIdx.x, xcd = L & 7u, slot = L >> 3;   // XCD-aware order, as in tgemm_kernel
    const int nt = (int)(slot % (unsigned)NT);
    const int mg = (int)(slot / (unsigned)NT) * 8 + (int)xcd;
    if (mg >= MT) return;
    const int m0 = mg * BM, n0 = nt * BN * (a.ntw > 0 ? a.ntw : 1);
    const int b = m0 / a.seg_rows, t0 = m0 - b * a.seg_rows;
    if (t0 >= a.M) return;                                    // a tile of padding tokens only
    const elt_t* W = reinterpret_cast<const elt_t*>(a.w);
    const int nsrc = RS ? (a.ra[1] ? 2 : 1) : 0;
    const int ntap = RS ? a.rtaps : 0;
    const int nkr = ntap * nsrc * CT;                          // resident k-tiles
    const int nk = a.K / FBK;
    const int nst = nk - nkr;                                  // streamed k-tiles
    const int sk0 = a.sk[0] / FBK, sk1 = a.sk[1] / FBK;

    // ---- GroupNorm coefficients from a producer's partials -> coefS (4 waves x 48 channels; per-wave scratch inside the tile area,
    // which is idle at both call sites: kernel entry, and the source switch of a concatenated input)
    auto gn_coefs = [&](const float* part, float eps, const float* gamma, const float* beta) {
        // (the lane / wave ids pass through an opaque asm: this lambda is inlined at several call sites, and without it the compiler
        // shares the dozens of lane-derived offsets and masks between them — i.e. keeps them alive in registers across the whole k loop)
        int lo_ = l, wo_ = w;
        asm volatile("" : "+v"(lo_), "+v"(wo_));
        const GnP gp = {a.gn_cpg, a.gn_nparts, a.M, eps, gamma, beta, 192};
        const rsrc_t rp = make_rsrc(part + (long long)b * a.gn_part_bs, 192u * (unsigned)a.gn_nparts * 8u);
        GnLoads gl;
        // per-wave scratch: in the resident tile's region when there is one (empty at kernel entry, dead at the source switch of a
        // concatenated input — the weight buffers in the tile area are live there), else in the idle tile area
        float* const gsc = (RS ? reinterpret_cast<float*>(ares) : ldsf) + wo_ * GN_SCRATCH;
        gn_issue(gp, rp, wo_ * 48, 48, lo_, gl);
        gn_finish(gp, rp, wo_ * 48, 48, lo_, gl, gsc, coefS);
        __syncthreads();
    };
    if (!RS && a.res_gn) gn_coefs(a.res_part, a.res_eps, a.res_gamma, a.res_beta);
    // ---- resident tile of source `ph`: 8 threads per row (24 channels each), 32 rows per pass; a convolution's two halo rows
    // (resident rows 64, 65) by the first 16 threads in a third pass.  Few live registers on purpose: this prologue must not
    // cost the k loop its occupancy.
    auto load_resident = [&](int ph) {
        if (a.rmode == 1 || a.rmode == 3) gn_coefs(a.gn_part[ph], a.gn_eps, a.gn_gamma + ph * 192, a.gn_beta + ph * 192);
        const elt_t* src = reinterpret_cast<const elt_t*>(a.ra[ph]);
        const int halo = a.rtaps == 3 ? 1 : 0;
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));   // (as in gn_coefs: nothing of this prologue is to stay alive through the k loop)
        const int q8 = tid_ & 7;
        const float* cf = coefS + 48 * q8;
#pragma unroll 1
        for (int pass = 0; pass < 2 + halo; ++pass) {
            const int r = pass * 32 + (tid_ >> 3);
            if (pass == 2 && tid_ >= 16) break;
            const int tt = t0 + r - halo;
            const bool valid = tt >= 0 && tt < a.M;
            const elt_t* p = src + ((long long)b * a.seg_rows + min(max(tt, 0), a.M - 1)) * 192 + 24 * q8;
            float x[24];
            if constexpr (BF) {
                u32x4 raw[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) raw[i] = *reinterpret_cast<const u32x4*>(p + 8 * i);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x[8 * i + 2 * e] = __builtin_bit_cast(float, raw[i][e] << 16);
                        x[8 * i + 2 * e + 1] = __builtin_bit_cast(float, raw[i][e] & 0xffff0000u);
                    }
            } else {
                f32x4t raw[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) raw[i] = *reinterpret_cast<const f32x4t*>(p + 4 * i);
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[4 * i + e] = raw[i][e];
            }
            // (sched_barriers: without them the scheduler hoists all 48 coefficient reads / 48 LayerNorm parameter loads of a row in
            // front of the arithmetic — 120+ live registers in a prologue, which then set the whole kernel's occupancy)
            if (a.rmode == 1 || a.rmode == 3) {
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    x[i] = fmaf(x[i], cf[2 * i], cf[2 * i + 1]);
                    if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (a.rmode == 1) {
#pragma unroll
                for (int i = 0; i < 24; ++i) x[i] = silu_f(x[i]);
            }
            if (a.rmode >= 2) {   // LayerNorm over the row's 192 channels: sums over this thread's 24, then over the row's eight threads
                const float ref = __shfl(x[0], (tid_ & 63) & ~7);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 24; ++i) { const float d = x[i] - ref; s1 += d; s2 = fmaf(d, d, s2); }
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
            elt_t* d = ares + r * RP + 24 * q8;
            if constexpr (BF) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const bf16x8 o = {(__bf16)x[8 * i], (__bf16)x[8 * i + 1], (__bf16)x[8 * i + 2], (__bf16)x[8 * i + 3],
                                      (__bf16)x[8 * i + 4], (__bf16)x[8 * i + 5], (__bf16)x[8 * i + 6], (__bf16)x[8 * i + 7]};
                    *reinterpret_cast<bf16x8*>(d + 8 * i) = o;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const f32x4t o = {x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]};
                    *reinterpret_cast<f32x4t*>(d + 4 * i) = o;
                }
            }
        }
    };

    // ---- k-tile pipeline: weights always, streamed A tiles when SS
    f32x4t ra_[ACH], rw[WCH];
    int woff[WCH], lwoff[WCH], loff[ACH], arow[ACH], akp[ACH];
    auto setup_offsets = [&]() {   // (called AFTER the prologue: these are live through the whole k loop, the prologue's registers are not)
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = tid + NTH * i, row = c >> 3, kp = c & 7;
            arow[i] = min(m0 + row, rows_tot - 1);
            akp[i] = kp * EPC;
            loff[i] = row * FLP + kp * EPC;
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + NTH * i, row = c >> 3, kp = c & 7;
            woff[i] = (n0 + row) * a.K + kp * EPC;
            lwoff[i] = BM * FLP + row * FLP + kp * EPC;
        }
    };
    // W k-offset of tile kt: the resident segment runs source-major ([source][tap][channel tile]) over a tap-major weight
    auto wk_of = [&](int kt) -> int {
        if (RS && kt < nkr) {
            const int per = ntap * CT;
            const int ph = kt / per, rem = kt - ph * per;
            const int tap = rem / CT, ct = rem - tap * CT;
            return tap * (nsrc * 192) + ph * 192 + ct * FBK;
        }
        return kt * FBK;
    };
    auto lds_store = [&](const f32x4t* xa, const f32x4t* xw) {
        if constexpr (SS) {
#pragma unroll
            for (int i = 0; i < ACH; ++i) *reinterpret_cast<f32x4t*>(ldse + loff[i]) = xa[i];
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) *reinterpret_cast<f32x4t*>(ldse + lwoff[i]) = xw[i];
    };
    f32x16 acc[NJ];
    const int frow = l & 31, fk = EPC * (l >> 5) + 4 * EPC * kh;
    const elt_t* const pa_s = ldse + (wr * 32 + frow) * FLP + fk;
    const elt_t* const pa_r = ares + (wr * 32 + frow) * RP + fk;
    const elt_t* const pw = ldse + BM * FLP + frow * FLP + fk;
    auto compute = [&](int kt) {
        const elt_t* pa = pa_s;
        if (RS && kt < nkr) {
            const int rem = kt % (ntap * CT);
            const int tap = rem / CT, ct = rem - tap * CT;
            pa = pa_r + tap * RP + ct * FBK;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (BF) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(pa + ks * 16);
                bf16x8 fb[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(pw + j * 32 * FLP + ks * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (TR) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa, acc[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[j], acc[j], 0, 0, 0);
                }
            } else {
                const f32x4t fa = *reinterpret_cast<const f32x4t*>(pa + ks * 8);
                f32x4t fb[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const f32x4t*>(pw + j * 32 * FLP + ks * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if constexpr (TR) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][i], fa[i], acc[j], 0, 0, 0);
                        else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j][i], acc[j], 0, 0, 0);
                    }
            }
        }
    };
    // The workgroup walks over `ntw` consecutive column tiles with its resident source tile (the prologue — GroupNorm finalisation,
    // tile load, transform, LayerNorm — is paid once per ntw x 32 NJ output columns instead of once per column tile: with one
    // column tile per workgroup the 12 workgroups of a GEGLU row tile each repeated it, 163 us per launch against 71 for round
    // 2's GEMM + 19 for its preparation kernel).  (column tile, k-tile) pairs form ONE sequence of steps through the k-tile
    // pipeline: request the next step's tile -> multiply the tile in LDS -> barrier -> [last k-tile of a column tile: add the K
    // halves, epilogue] -> park the next tile -> barrier.  The next column tile's first weights are in flight during the epilogue.
    const int ntw = RS ? (a.ntw > 0 ? a.ntw : 1) : 1;   // (streamed-only GEMMs have no prologue to amortise)
    const int nsteps = ntw * nk;
    const int e0 = (RS && nsrc == 2) ? ntap * CT : -1;   // a concatenated input (ntw == 1): the second source takes the resident buffer over
    constexpr int NJ0 = (NJ == 4 && !TR) ? 2 : NJ, NJ1 = NJ - NJ0;
    float* const xr = ldsf + wr * (NJ1 > 0 ? NJ * 16 * 64 : 32 * (32 * NJ + 4));
    auto gload_step = [&](int step) {
        const int jn = step / nk, kt = step - jn * nk;
        const int wk = wk_of(kt) + jn * BN * a.K;
#pragma unroll
        for (int i = 0; i < WCH; ++i) rw[i] = *reinterpret_cast<const f32x4t*>(W + (woff[i] + wk));
        if constexpr (SS) {
            const int st = min(max(kt - nkr, 0), nst - 1);        // (resident steps re-request the first streamed tile: an L1 hit)
            const elt_t* base; int ld, off;
            if (st < sk0) { base = reinterpret_cast<const elt_t*>(a.sa[0]); ld = a.sld[0]; off = st * FBK; }
            else if (st < sk0 + sk1) { base = reinterpret_cast<const elt_t*>(a.sa[1]); ld = a.sld[1]; off = (st - sk0) * FBK; }
            else { base = reinterpret_cast<const elt_t*>(a.sa[2]); ld = a.sld[2]; off = (st - sk0 - sk1) * FBK; }
#pragma unroll
            for (int i = 0; i < ACH; ++i) ra_[i] = *reinterpret_cast<const f32x4t*>(base + ((long long)arow[i] * ld + off + akp[i]));
        }
    };
    clk_stamp_p(a.clk, w, l, 0);
    if constexpr (!SS) {
        // ---- resident source only (convolutions without a skip, q/k/v, banded cross-attention, GEGLU): just the weights stream, so
        // the tile area holds TWO weight tiles (128-byte rows, 16-byte chunks XOR-swizzled by the row instead of padded) and a k-step
        // costs ONE barrier: park the next tile in the other buffer, request the one after, multiply the current one.
        load_resident(0);
        clk_stamp_p(a.clk, w, l, 1);
        int wo[WCH], wl[WCH];
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + NTH * i, row = c >> 3, kp = c & 7;
            wo[i] = (n0 + row) * a.K + kp * EPC;
            wl[i] = row * FBK + ((kp ^ (row & 7)) * EPC);
        }
        auto wload = [&](int jn, int kt) {
            const int wk = wk_of(kt) + jn * BN * a.K;
#pragma unroll
            for (int i = 0; i < WCH; ++i) rw[i] = *reinterpret_cast<const f32x4t*>(W + (wo[i] + wk));
        };
        auto wstore = [&](int buf) {
#pragma unroll
            for (int i = 0; i < WCH; ++i) *reinterpret_cast<f32x4t*>(ldse + buf * (BN * FBK) + wl[i]) = rw[i];
        };
        const int fr = l & 31;
        const int sw0 = (((l >> 5) + 4 * kh) ^ (fr & 7)) * EPC, sw1 = (((l >> 5) + 4 * kh + 2) ^ (fr & 7)) * EPC;   // the K half's two operand steps
        const elt_t* const par = ares + (wr * 32 + fr) * RP + EPC * (l >> 5) + 4 * EPC * kh;
        auto wcompute = [&](int buf, int kt) {
            const int rem = (nsrc == 2 && kt >= ntap * CT) ? kt - ntap * CT : kt;
            const int tap = CT == 3 ? (rem * 43) >> 7 : (rem * 43) >> 8;   // rem / CT for rem < 64
            const elt_t* pa = par + tap * RP + (rem - tap * CT) * FBK;
            const elt_t* pwb = ldse + buf * (BN * FBK) + fr * FBK;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int sw = ks ? sw1 : sw0;
                if constexpr (BF) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(pa + ks * 16);
                    bf16x8 fb[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(pwb + j * 32 * FBK + sw);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if constexpr (TR) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa, acc[j], 0, 0, 0);
                        else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[j], acc[j], 0, 0, 0);
                    }
                } else {
                    const f32x4t fa = *reinterpret_cast<const f32x4t*>(pa + ks * 8);
                    f32x4t fb[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const f32x4t*>(pwb + j * 32 * FBK + sw);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            if constexpr (TR) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][i], fa[i], acc[j], 0, 0, 0);
                            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j][i], acc[j], 0, 0, 0);
                        }
                }
            }
        };
        const int ntw = a.ntw > 0 ? a.ntw : 1;
        const int e0 = nsrc == 2 ? ntap * CT : -1;
        constexpr int NJ0 = (NJ == 4 && !TR) ? 2 : NJ, NJ1 = NJ - NJ0;
        float* const xr = ldsf + wr * (NJ1 > 0 ? NJ * 16 * 64 : 32 * (32 * NJ + 4));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        wload(0, 0);
        wstore(0);
        wload(0, 1);
        __syncthreads();
        clk_stamp_p(a.clk, w, l, 2);
        for (int jn = 0; jn < ntw; ++jn) {
            const int jnx = min(jn + 1, ntw - 1);
            // invariant at step kt: buffer kt & 1 holds tile kt, the registers tile kt + 1 (behind the last tile: the next column tile's first)
            for (int kt = 0; kt < nk - 1; ++kt) {
                wstore((kt + 1) & 1);
                { const bool nx = kt + 2 < nk; wload(nx ? jn : jnx, nx ? kt + 2 : 0); }   // (one unconditional request: exact wait counts)
                __builtin_amdgcn_sched_barrier(0);
                wcompute(kt & 1, kt);
                __builtin_amdgcn_sched_barrier(0);
                if (kt == e0 - 1) { __syncthreads(); load_resident(1); }   // every wave is done with the first source's tile
                __syncthreads();
            }
            wcompute((nk - 1) & 1, nk - 1);
            clk_stamp_p(a.clk, w, l, 3 + 2 * min(jn, 5));
            __syncthreads();   // the tile area becomes exchange / transposition scratch
            const int n0j = n0 + jn * BN;
            // ---- add the two K halves (fgemm_kernel's exchange), then the epilogue of column tile jn
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if ((kh == 1) == (j < NJ0)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) xr[(j * 16 + r) * 64 + l] = acc[j][r];
                }
            }
            __syncthreads();
            if (NJ1 > 0 || kh == 0) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if ((kh == 0) == (j < NJ0)) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[j][r] += xr[(j * 16 + r) * 64 + l];
                    }
                }
            }
            if constexpr (TR) {
                if (kh == 0) {   // banded cross-attention: lane -> query token, one head per column tile
                    const int t = t0 + wr * 32 + (l & 31);
                    const bool tv = t < a.M;
                    const int tc = min(t, a.M - 1);
                    const int lo = a.band_lo[tc], hi = a.band_hi[tc];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) band_head<BF>(a, acc[j], b, t, tv, lo, hi, n0j / 32 + j, l);
                }
            } else if constexpr (NJ1 > 0) {
                __syncthreads();
                float* sc = ldsf + w * (32 * (32 * NJ0 + 4));
                if (kh == 0) tg_epilogue<NJ, 0, NJ0, EK>(a, acc, 0, m0 + wr * 32, n0j, l, sc, coefS);
                else tg_epilogue<NJ, NJ0, (NJ1 > 0 ? NJ1 : 1), EK>(a, acc, 0, m0 + wr * 32, n0j + 32 * NJ0, l, sc, coefS);
            } else if constexpr (EK == 0) {
                // ---- token-major activation epilogue on ALL FOUR waves.  (Run by the two K-half-0 waves alone, with the residual gathered
                // in the MFMA layout — 2-byte loads, lane == column — it was 40-60 % of these kernels: knock-outs, profiles/r03_*.)
                // phase 1 (K-half-0 waves, lane == column): acc + bias + timestep-embedding term -> scratch [32 rows][CW + 4] of this row half
                constexpr int CW = 32 * NJ, CP = CW + 4;
                float* const sc = xr;
                int le = l;
                asm volatile("" : "+v"(le));   // (keeps this epilogue's lane-derived values out of the k loop: the column-tile loop around both
                                               // would otherwise have them computed once, up front, and held in registers)
                const int mt = t0 + wr * 32;                       // first token of this row half
                const int nrows = min(32, a.M - mt);               // <= 0: padding rows only
                if (kh == 0 && nrows > 0) {
                    const int lc = le & 31, lh = le >> 5;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int n = n0j + j * 32 + lc;
                        float add = a.bias ? a.bias[n] : 0.f;
                        if (a.emb) add += a.emb[(long long)n * a.emb_pitch + (a.step_ptr ? *a.step_ptr : 0) + b * a.emb_b_stride];
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * lh) * CP + j * 32 + lc] = acc[j][r] + add;
                    }
                }
                __syncthreads();
                // phase 2 (all waves): wave (wr, kh) takes rows 16 kh .. 16 kh + 15; 16 lanes per row (CW / 8 of them active), each 8
                // consecutive columns: residual (16-byte loads, optionally GroupNorm'ed), rounding to the element type, GroupNorm partial
                // sums of the stored values, 16-byte stores
                const int rr = le >> 4, cq = le & 15;
                const bool lane_on = cq < CW / 8;
                const int n = n0j + 8 * min(cq, CW / 8 - 1);
                float ref[8], s1[8], s2[8], rca[8], rcb[8], add2[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ref[e] = sc[8 * min(cq, CW / 8 - 1) + e];   // any common shift will do: row 0 of the row half, before the residual
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
                    if (a.res_tm && !(a.dbg & 4)) {
                        if constexpr (BF) {
                            const u32x4 rv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(a.res_tm) + (R0 + row) * a.ldr_tm + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[2 * e] += fmaf(__builtin_bit_cast(float, rv[e] << 16), rca[2 * e], rcb[2 * e]);
                                v[2 * e + 1] += fmaf(__builtin_bit_cast(float, rv[e] & 0xffff0000u), rca[2 * e + 1], rcb[2 * e + 1]);
                            }
                        } else {
                            const float* rp = reinterpret_cast<const float*>(a.res_tm) + (R0 + row) * a.ldr_tm + n;
                            const f32x4t r0 = *reinterpret_cast<const f32x4t*>(rp), r1 = *reinterpret_cast<const f32x4t*>(rp + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] += fmaf(r0[e], rca[e], rcb[e]); v[4 + e] += fmaf(r1[e], rca[4 + e], rcb[4 + e]); }
                        }
                    }
                    const long long o = (R0 + row) * a.ldy + n;
                    if constexpr (BF) {
                        const bf16x8 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.y_tm) + o) = ov;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (float)ov[e];   // the statistics are those of the stored values
                        if (a.y2_tm) {
                            const bf16x8 o2 = {(__bf16)(v[0] + add2[0]), (__bf16)(v[1] + add2[1]), (__bf16)(v[2] + add2[2]), (__bf16)(v[3] + add2[3]),
                                               (__bf16)(v[4] + add2[4]), (__bf16)(v[5] + add2[5]), (__bf16)(v[6] + add2[6]), (__bf16)(v[7] + add2[7])};
                            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.y2_tm) + o + a.y2_row_off * a.ldy) = o2;
                        }
                    } else {
                        float* y = reinterpret_cast<float*>(a.y_tm) + o;
                        const f32x4t w0 = {v[0], v[1], v[2], v[3]}, w1 = {v[4], v[5], v[6], v[7]};
                        *reinterpret_cast<f32x4t*>(y) = w0;
                        *reinterpret_cast<f32x4t*>(y + 4) = w1;
                        if (a.y2_tm) {
                            float* y2 = reinterpret_cast<float*>(a.y2_tm) + o + a.y2_row_off * a.ldy;
                            const f32x4t u0 = {v[0] + add2[0], v[1] + add2[1], v[2] + add2[2], v[3] + add2[3]};
                            const f32x4t u1 = {v[4] + add2[4], v[5] + add2[5], v[6] + add2[6], v[7] + add2[7]};
                            *reinterpret_cast<f32x4t*>(y2) = u0;
                            *reinterpret_cast<f32x4t*>(y2 + 4) = u1;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = v[e] - ref[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
                }
                if (a.stats) {
                    // sums over the wave's 16 rows (lanes l, l ^ 16, l ^ 32, l ^ 48 share their columns), then over the two waves of the row
                    // half through the statistics exchange [2 row halves][CW columns][2] (the coefficient region: free after the prologue)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s1[e] += __shfl_xor(s1[e], 16); s2[e] += __shfl_xor(s2[e], 16);
                        s1[e] += __shfl_xor(s1[e], 32); s2[e] += __shfl_xor(s2[e], 32);
                    }
                    float* const ex = coefS + wr * (2 * CW);
                    if (kh == 1 && le < CW / 8) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { ex[2 * (8 * le + e)] = s1[e]; ex[2 * (8 * le + e) + 1] = s2[e]; }
                    }
                    __syncthreads();
                    if (kh == 0 && le < CW / 8 && nrows > 0) {
                        float* so = a.stats + (long long)b * a.stats_bs + ((long long)(mt >> 5) * a.N + n) * 2;   // [tile][channel][2]
                        const float cnt = (float)nrows, inv = 1.0f / cnt;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float S1 = s1[e] + ex[2 * (8 * le + e)], S2 = s2[e] + ex[2 * (8 * le + e) + 1];
                            const float md = S1 * inv;
                            so[2 * e] = ref[e] + md;                          // mean
                            so[2 * e + 1] = fmaxf(S2 - cnt * md * md, 0.f);   // M2 = sum (x - mean)^2
                        }
                    }
                }
            } else {
                if (kh == 0) {
                    __builtin_amdgcn_wave_barrier();
                    tg_epilogue<NJ, 0, NJ, EK>(a, acc, 0, m0 + wr * 32, n0j, l, xr, coefS);
                }
            }
            clk_stamp_p(a.clk, w, l, 4 + 2 * min(jn, 5));
            if (jn + 1 < ntw) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
                __syncthreads();   // the tile area is free again
                wstore(0);
                wload(jn + 1, 1);
                __syncthreads();
            }
        }
        return;
    }
    if constexpr (RS) { if (!(a.dbg & 8)) load_resident(0); }   // (before anything of the k loop is live in registers)
    setup_offsets();
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    gload_step(0);
    lds_store(ra_, rw);
    __syncthreads();
    const int nk_loop = (a.dbg & 2) ? 1 : nk;   // (timing experiment: no k loop)
    for (int jn = 0; jn < ntw; ++jn) {
        const int s0 = jn * nk;
        // all k-tiles but the last: request the next tile -> multiply -> barrier -> park the next tile -> barrier
        for (int kt = 0; kt < nk_loop - 1; ++kt) {
            gload_step(s0 + kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(kt);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (kt == e0 - 1) load_resident(1);   // all waves are done with the first source's tile (concatenated input)
            lds_store(ra_, rw);
            __syncthreads();
        }
        // last k-tile: the NEXT column tile's first weights are requested and stay in registers through the epilogue
        if constexpr (RS) gload_step(min(s0 + nk, nsteps - 1));
        __builtin_amdgcn_sched_barrier(0);
        compute(nk - 1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // ---- add the two K halves (fgemm_kernel's exchange), then the epilogue of column tile jn
        const int n0j = n0 + jn * BN;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if ((kh == 1) == (j < NJ0)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xr[(j * 16 + r) * 64 + l] = acc[j][r];
            }
        }
        __syncthreads();
        if (NJ1 > 0 || kh == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if ((kh == 0) == (j < NJ0)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] += xr[(j * 16 + r) * 64 + l];
                }
            }
        }
        if constexpr (TR) {
            if (kh == 0) {   // banded cross-attention: lane -> query token, one head per column tile
                const int t = t0 + wr * 32 + (l & 31);
                const bool tv = t < a.M;
                const int tc = min(t, a.M - 1);
                const int lo = a.band_lo[tc], hi = a.band_hi[tc];
#pragma unroll
                for (int j = 0; j < NJ; ++j) band_head<BF>(a, acc[j], b, t, tv, lo, hi, n0j / 32 + j, l);
            }
        } else if constexpr (NJ1 > 0) {
            __syncthreads();
            float* sc = ldsf + w * (32 * (32 * NJ0 + 4));
            if (kh == 0) tg_epilogue<NJ, 0, NJ0, EK>(a, acc, 0, m0 + wr * 32, n0j, l, sc, coefS);
            else tg_epilogue<NJ, NJ0, (NJ1 > 0 ? NJ1 : 1), EK>(a, acc, 0, m0 + wr * 32, n0j + 32 * NJ0, l, sc, coefS);
        } else if constexpr (EK == 0) {
            // ---- token-major activation epilogue on ALL FOUR waves.  (Run by the two K-half-0 waves alone, with the residual gathered
            // in the MFMA layout — 2-byte loads, lane == column — it was 40-60 % of these kernels: knock-outs, profiles/r03_*.)
            // phase 1 (K-half-0 waves, lane == column): acc + bias + timestep-embedding term -> scratch [32 rows][CW + 4] of this row half
            constexpr int CW = 32 * NJ, CP = CW + 4;
            float* const sc = xr;
            int le = l;
            asm volatile("" : "+v"(le));   // (keeps this epilogue's lane-derived values out of the k loop: the column-tile loop around both
                                           // would otherwise have them computed once, up front, and held in registers)
            const int mt = t0 + wr * 32;                       // first token of this row half
            const int nrows = min(32, a.M - mt);               // <= 0: padding rows only
            if (kh == 0 && nrows > 0) {
                const int lc = le & 31, lh = le >> 5;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = n0j + j * 32 + lc;
                    float add = a.bias ? a.bias[n] : 0.f;
                    if (a.emb) add += a.emb[(long long)n * a.emb_pitch + (a.step_ptr ? *a.step_ptr : 0) + b * a.emb_b_stride];
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * lh) * CP + j * 32 + lc] = acc[j][r] + add;
                }
            }
            __syncthreads();
            // phase 2 (all waves): wave (wr, kh) takes rows 16 kh .. 16 kh + 15; 16 lanes per row (CW / 8 of them active), each 8
            // consecutive columns: residual (16-byte loads, optionally GroupNorm'ed), rounding to the element type, GroupNorm partial
            // sums of the stored values, 16-byte stores
            const int rr = le >> 4, cq = le & 15;
            const bool lane_on = cq < CW / 8;
            const int n = n0j + 8 * min(cq, CW / 8 - 1);
            float ref[8], s1[8], s2[8], rca[8], rcb[8], add2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ref[e] = sc[8 * min(cq, CW / 8 - 1) + e];   // any common shift will do: row 0 of the row half, before the residual
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
                if (a.res_tm && !(a.dbg & 4)) {
                    if constexpr (BF) {
                        const u32x4 rv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(a.res_tm) + (R0 + row) * a.ldr_tm + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] += fmaf(__builtin_bit_cast(float, rv[e] << 16), rca[2 * e], rcb[2 * e]);
                            v[2 * e + 1] += fmaf(__builtin_bit_cast(float, rv[e] & 0xffff0000u), rca[2 * e + 1], rcb[2 * e + 1]);
                        }
                    } else {
                        const float* rp = reinterpret_cast<const float*>(a.res_tm) + (R0 + row) * a.ldr_tm + n;
                        const f32x4t r0 = *reinterpret_cast<const f32x4t*>(rp), r1 = *reinterpret_cast<const f32x4t*>(rp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += fmaf(r0[e], rca[e], rcb[e]); v[4 + e] += fmaf(r1[e], rca[4 + e], rcb[4 + e]); }
                    }
                }
                const long long o = (R0 + row) * a.ldy + n;
                if constexpr (BF) {
                    const bf16x8 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.y_tm) + o) = ov;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)ov[e];   // the statistics are those of the stored values
                    if (a.y2_tm) {
                        const bf16x8 o2 = {(__bf16)(v[0] + add2[0]), (__bf16)(v[1] + add2[1]), (__bf16)(v[2] + add2[2]), (__bf16)(v[3] + add2[3]),
                                           (__bf16)(v[4] + add2[4]), (__bf16)(v[5] + add2[5]), (__bf16)(v[6] + add2[6]), (__bf16)(v[7] + add2[7])};
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.y2_tm) + o + a.y2_row_off * a.ldy) = o2;
                    }
                } else {
                    float* y = reinterpret_cast<float*>(a.y_tm) + o;
                    const f32x4t w0 = {v[0], v[1], v[2], v[3]}, w1 = {v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f32x4t*>(y) = w0;
                    *reinterpret_cast<f32x4t*>(y + 4) = w1;
                    if (a.y2_tm) {
                        float* y2 = reinterpret_cast<float*>(a.y2_tm) + o + a.y2_row_off * a.ldy;
                        const f32x4t u0 = {v[0] + add2[0], v[1] + add2[1], v[2] + add2[2], v[3] + add2[3]};
                        const f32x4t u1 = {v[4] + add2[4], v[5] + add2[5], v[6] + add2[6], v[7] + add2[7]};
                        *reinterpret_cast<f32x4t*>(y2) = u0;
                        *reinterpret_cast<f32x4t*>(y2 + 4) = u1;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[e] - ref[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
            }
            if (a.stats) {
                // sums over the wave's 16 rows (lanes l, l ^ 16, l ^ 32, l ^ 48 share their columns), then over the two waves of the row
                // half through the statistics exchange [2 row halves][CW columns][2] (the coefficient region: free after the prologue)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += __shfl_xor(s1[e], 16); s2[e] += __shfl_xor(s2[e], 16);
                    s1[e] += __shfl_xor(s1[e], 32); s2[e] += __shfl_xor(s2[e], 32);
                }
                float* const ex = coefS + wr * (2 * CW);
                if (kh == 1 && le < CW / 8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ex[2 * (8 * le + e)] = s1[e]; ex[2 * (8 * le + e) + 1] = s2[e]; }
                }
                __syncthreads();
                if (kh == 0 && le < CW / 8 && nrows > 0) {
                    float* so = a.stats + (long long)b * a.stats_bs + ((long long)(mt >> 5) * a.N + n) * 2;   // [tile][channel][2]
                    const float cnt = (float)nrows, inv = 1.0f / cnt;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float S1 = s1[e] + ex[2 * (8 * le + e)], S2 = s2[e] + ex[2 * (8 * le + e) + 1];
                        const float md = S1 * inv;
                        so[2 * e] = ref[e] + md;                          // mean
                        so[2 * e + 1] = fmaxf(S2 - cnt * md * md, 0.f);   // M2 = sum (x - mean)^2
                    }
                }
            }
        } else {
            if (kh == 0) {
                __builtin_amdgcn_wave_barrier();
                tg_epilogue<NJ, 0, NJ, EK>(a, acc, 0, m0 + wr * 32, n0j, l, xr, coefS);
            }
        }
        if (jn + 1 < ntw) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            __syncthreads();   // the tile area served as exchange / transposition scratch
            lds_store(ra_, rw);
            __syncthreads();
        }
    }
}

bool tgemm_supports(const TGemmArgs& a) {
    if (a.f32) {   // fp32 kernel: 4-float chunks, 32-float k-tiles; 96- or 128-wide column tiles; batch-as-rows addressing only
        if (!(a.M >= 1 && (a.N % 96 == 0 || a.N % 128 == 0) && a.K >= FBK && a.K % FBK == 0 && a.lda % 4 == 0 && a.seg_rows > 0)) return false;
        if (a.geglu && a.N % 128) return false;
        if (a.a2 && (a.K1 % FBK || a.K1 <= 0 || a.K1 >= a.K || a.lda2 % 4)) return false;
        if (a.geglu && (a.N % 256 || !a.yf)) return false;
        if (a.yb) return false;
    } else
    if (!(a.M >= 1 && a.N >= 64 && a.N % 64 == 0 && a.K >= TBK && a.K % TBK == 0 && a.lda % 8 == 0 && a.a_bs % 8 == 0)) return false;
    if (a.qk && (a.qk_n % 32 || a.head_dim % 32)) return false;
    if (a.a2 && (a.K1 % TBK || a.K1 <= 0 || a.K1 >= a.K || a.lda2 % 8 || a.a2_bs % 8)) return false;
    if (!a.f32 && a.geglu && (a.N % 256 || !a.yb)) return false;   // the GEGLU row interleaving is the 256-wide tile's (tgemm_geglu_src_row)
    if (a.y_cm && (a.cm_pitch % 4 || a.cm_pitch < ((a.M + 3) & ~3))) return false;
    if (a.seg_rows && (a.seg_rows % 32 || a.seg_rows < a.M)) return false;
    if (a.yb && (a.ldy % 8 || a.y_bs % 8)) return false;   // 16-byte bf16 stores
    if ((long long)a.N * a.K > 0x7fffffffLL) return false;   // 32-bit element offsets in the 256-row kernel
    return true;
}
// GEGLU weight-row interleaving for the 256-wide tile: tile-local column tile pairs (2p, 2p + 1) of each wave are (value, gate)
// of the same 32 channels.  Returns the source row (value rows [0, N/2), gate rows [N/2, N)) of permuted row n.
int tgemm_geglu_src_row(int n, int N) {
    const int tile = n / 256, wn = (n % 256) / 128, j = (n % 128) / 32, i = n % 32;
    const int c = tile * 128 + wn * 64 + (j >> 1) * 32 + i;
    return (j & 1) ? N / 2 + c : c;
}
void configure_tgemm_kernel() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm256d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TG256D_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (TBM + 128) * TLP * 2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TBM + 128) * TLP * 2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TBM + 64) * TLP * 2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm256_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * TLP * 2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm256_kernel<192>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 192) * TLP * 2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm_kernel<3, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, fgemm_lds_bytes<3>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm_kernel<4, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, fgemm_lds_bytes<4>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm_kernel<3, 1, false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, fgemm_lds_bytes<3>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm_kernel<3, 1, false, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FGEMM_PK_LDS3);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm_kernel<4, 1, false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, fgemm_lds_bytes<4>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm_kernel<3, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, fgemm_lds_bytes<3>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm_kernel<3, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, fgemm_lds_bytes<3>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm_kernel<4, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, fgemm_lds_bytes<4>());
}
bool launch_tgemm(const TGemmArgs& a, int batch, hipStream_t s) {
    if (!tgemm_supports(a)) return false;
    TGemmArgs a2 = a;
    a2.batch = batch;
    static const int dbg = dev_env("SAID_TG_DBG") ? atoi(dev_env("SAID_TG_DBG")) : 0;
    a2.dbg = dbg;
    static const bool no256 = dev_env("SAID_NO_TGEMM256") != nullptr;
    const long long rows_tot = a.seg_rows > 0 ? (long long)batch * a.seg_rows : a.M;
    const int nb = a.seg_rows > 0 ? 1 : batch;
    if (a.f32) {
        if (a.grp > 1) return false;
        if ((rows_tot + 2) * (long long)std::max(a.lda, a.lda2) >= 0x7fffffffLL) return false;   // 32-bit operand offsets
        const long long mt8 = ((rows_tot + 63) / 64 + 7) / 8 * 8;   // 64-row tiles, padded to the 8 XCDs
        constexpr int LDS3 = fgemm_lds_bytes<3>(), LDS4 = fgemm_lds_bytes<4>();
        if (a.f32_split) {   // products on split-fp16 operands (TGemmArgs::f32_split)
            if (a.f32_packed) {   // ... which arrive split (NJ = 3 shapes: the ResBlock convolutions and q / k / v)
                if (a.N % 96 || a.geglu) return false;
                hipLaunchKernelGGL((fgemm_kernel<3, 1, false, 2, true, true>), dim3((unsigned)(mt8 * (a.N / 96))), dim3(256), FGEMM_PK_LDS3, s, a2);
                return true;
            }
            if (a.N % 128 == 0 && (a.geglu || a.N % 96)) hipLaunchKernelGGL((fgemm_kernel<4, 1, false, 2, true>), dim3((unsigned)(mt8 * (a.N / 128))), dim3(256), LDS4, s, a2);
            else hipLaunchKernelGGL((fgemm_kernel<3, 1, false, 2, true>), dim3((unsigned)(mt8 * (a.N / 96))), dim3(256), LDS3, s, a2);
            return true;
        }
        if (a.N % 128 == 0 && (a.geglu || a.N % 96)) hipLaunchKernelGGL((fgemm_kernel<4, 1, false>), dim3((unsigned)(mt8 * (a.N / 128))), dim3(256), LDS4, s, a2);
        else hipLaunchKernelGGL((fgemm_kernel<3, 2, false>), dim3((unsigned)(mt8 * (a.N / 96))), dim3(256), LDS3, s, a2);   // (one register
        // set at five workgroups per CU — the bf16 variant's choice — spills and measured 344 vs 328 ms here)
        return true;
    }
    // bf16, batch-as-rows (UNet): the 64-row K-split tile of the fp32 path on bf16 operands.  The kernels' time is their fp32
    // epilogue traffic, and the 256-row tiles give a 192-wide output 152 workgroups on 256 CUs (42.0 -> 35.8 us on the small tile).
    // For q/k/v and GEGLU (456 / 912 big workgroups) the isolated replays of said_profile_unet favour the big tile (29.8 vs 32.7,
    // 69.7 vs 75.0 us) but the real step does not: 122.1 vs 120.3 ms per 32 clips x 50 steps, three alternating runs on one box
    // (scripts/gpu_r2_ak.sh) — one 147 KB-LDS workgroup per CU starts and drains badly between neighbours of other shapes.  So
    // the small tile is the rule; SAID_TGEMM_SMALL=0 restores the 256-row tiles, =1 uses them only where they fill the chip.
    static const int small_bf = dev_env("SAID_TGEMM_SMALL") ? atoi(dev_env("SAID_TGEMM_SMALL")) : -1;
    if (a.seg_rows > 0 && small_bf != 0 && a.K % 64 == 0 && (!a.a2 || a.K1 % 64 == 0) && (a.N % 96 == 0 || a.N % 128 == 0)) {
        const bool wide_n = a.N % 128 == 0 && (a.geglu || a.N % 96);
        const long long big_grid = ((rows_tot + 255) / 256) * (a.N / (wide_n ? 256 : 192));
        const bool can_big = !no256 && (a.N % 256 == 0 || a.N % 192 == 0) && rows_tot >= 4096;
        if (small_bf != 1 || big_grid < 256 || !can_big) {
            const long long mt8 = ((rows_tot + 63) / 64 + 7) / 8 * 8;
            constexpr int LDS3 = fgemm_lds_bytes<3>(), LDS4 = fgemm_lds_bytes<4>();
            // (the GEGLU tile squeezed to 128 VGPRs for four per CU spills five registers and measured no better: 121.7 vs 120.1 ms)
            if (wide_n) hipLaunchKernelGGL((fgemm_kernel<4, 1, true>), dim3((unsigned)(mt8 * (a.N / 128))), dim3(256), LDS4, s, a2);
            else {
                // one register set at FIVE workgroups per CU (96 VGPRs): the 1216 workgroups of a 192-wide launch at Be = 64 are all
                // resident at once instead of 1024 + a tail of 192 — 124.5 -> 121.0 ms per 32 clips x 50 steps, three alternating
                // runs on one box (scripts/gpu_r2_ar.sh).  SAID_BF_OCC5=0: two register sets at four per CU.
                static const int occ5 = dev_env("SAID_BF_OCC5") ? atoi(dev_env("SAID_BF_OCC5")) : 1;
                if (occ5) hipLaunchKernelGGL((fgemm_kernel<3, 1, true>), dim3((unsigned)(mt8 * (a.N / 96))), dim3(256), LDS3, s, a2);
                else hipLaunchKernelGGL((fgemm_kernel<3, 2, true>), dim3((unsigned)(mt8 * (a.N / 96))), dim3(256), LDS3, s, a2);
            }
            return true;
        }
    }
    if (a.grp > 1 && (a.seg_rows > 0 || a.a2 || a.n_store < 1 || a.col_gs < a.n_store)) return false;   // grouped launches: tgemm_kernel only
    const bool big = a.grp <= 1 && !no256 && (a.N % 256 == 0 || a.N % 192 == 0) && rows_tot * nb >= 4096 &&
                     (rows_tot + 2) * (long long)std::max(a.lda, a.lda2) < 0x7fffffffLL;
    if (a.geglu && !big) return false;   // the GEGLU epilogue needs the 256-wide tile
    // Per-sample operands (audio encoder): a 256-row tile holds one workgroup per CU, so its grid runs in rounds of 256 — the
    // encoder's 768-wide GEMMs at 32 clips x 600 frames are 288 workgroups = two rounds, the second 12 % full.  Where the
    // 128 x 128 tile (two per CU, rounds of 512) fills its rounds clearly better, it is used instead (SAID_TGEMM_BALANCE=0: never).
    static const int balance = dev_env("SAID_TGEMM_BALANCE") ? atoi(dev_env("SAID_TGEMM_BALANCE")) : 15;   // margin in percent; 0: never
    bool use_big = big;
    if (big && balance > 0 && a.seg_rows == 0 && !a.geglu && a.N % 128 == 0) {
        const long long mt_big = (rows_tot + 255) / 256, mt_128 = (a.M + TBM - 1) / TBM;
        const long long g_big = (long long)nb * mt_big * (a.N / (a.N % 256 == 0 ? 256 : 192));
        const long long g_128 = (long long)batch * mt_128 * (a.N / 128);
        // efficiency = how full the rounds are x how full the row tiles are (rows past M repeat the last row: wasted work)
        const double e_big = (double)g_big / (double)(((g_big + 255) / 256) * 256) * (double)rows_tot / (double)(mt_big * 256);
        const double e_128 = (double)g_128 / (double)(((g_128 + 511) / 512) * 512) * (double)a.M / (double)(mt_128 * TBM);
        if (e_128 > e_big + 0.01 * balance) use_big = false;
    }
    // round 6: per-sample operands with 256-wide outputs (the audio encoder's projections): the direct-to-LDS 256 x 256 tile (a.direct; said_debug_option "tgemm_direct")
    if (a.direct && a.grp <= 1 && a.seg_rows == 0 && !a.geglu && !a.a2 && a.N % 256 == 0 && a.K % TBK == 0 && (long long)a.M * batch >= 4096 &&
        ((long long)(a.M - 1) * a.lda + a.K) * 2 < 0x7fffffffLL && (long long)a.N * a.K * 2 < 0x7fffffffLL) {
        const long long mt8 = ((long long)batch * ((a.M + 255) / 256) + 7) / 8 * 8;
        hipLaunchKernelGGL(tgemm256d_kernel, dim3((unsigned)(mt8 * (a.N / 256))), dim3(512), TG256D_LDS, s, a2);
        return true;
    }
    if (a.sb && a.seg_rows == 0 && !a.geglu && a.N % 128 == 0) use_big = false;   // the single-buffer 128 x 128 variant was asked for
    if (use_big) {
        const long long mt8 = ((long long)nb * ((rows_tot + 255) / 256) + 7) / 8 * 8;
        if (a.N % 256 == 0) {
            dim3 grid((unsigned)(mt8 * (a.N / 256)));
            hipLaunchKernelGGL(tgemm256_kernel<256>, grid, dim3(512), 2 * (256 + 256) * TLP * 2, s, a2);
        } else {
            dim3 grid((unsigned)(mt8 * (a.N / 192)));
            hipLaunchKernelGGL(tgemm256_kernel<192>, grid, dim3(512), 2 * (256 + 192) * TLP * 2, s, a2);
        }
        return true;
    }
    if (a.seg_rows > 0) return false;   // batch-as-rows addressing needs the 256-row tile
    const long long mtiles8 = ((long long)batch * ((a.M + TBM - 1) / TBM) + 7) / 8 * 8;   // (sample, M tile) pairs padded to the 8 XCDs
    if (a.N % 128 == 0) {
        dim3 grid((unsigned)(mtiles8 * (a.N / 128)));
        if (a.sb) hipLaunchKernelGGL((tgemm_kernel<128, true>), grid, dim3(256), (TBM + 128) * TLP * 2, s, a2);
        else hipLaunchKernelGGL(tgemm_kernel<128>, grid, dim3(256), 2 * (TBM + 128) * TLP * 2, s, a2);
    } else {
        dim3 grid((unsigned)(mtiles8 * (a.N / 64)));
        hipLaunchKernelGGL(tgemm_kernel<64>, grid, dim3(256), 2 * (TBM + 64) * TLP * 2, s, a2);
    }
    return true;
}

// ---- host side of xgemm_kernel ---------------------------------------------------------------------------------------
bool xgemm_supports(const TGemmArgs& a) {
    const int fbk = a.f32 ? 32 : 64;
    if (a.seg_rows <= 0 || a.seg_rows % 64 || a.M < 1 || a.M > a.seg_rows) return false;
    if (!(a.N % 96 == 0 || a.N % 128 == 0)) return false;
    if (a.geglu && a.N % 256) return false;
    int kres = 0;
    if (a.ra[0]) {
        if (a.rtaps != 1 && a.rtaps != 3) return false;
        if (a.rmode < 0 || a.rmode > 3) return false;
        if ((a.rmode == 1 || a.rmode == 3) && (!a.gn_part[0] || !a.gn_gamma || !a.gn_beta || (a.ra[1] && !a.gn_part[1]))) return false;
        if (a.rmode >= 2 && (a.ra[1] || !a.ln_gamma || !a.ln_beta)) return false;
        kres = a.rtaps * (a.ra[1] ? 384 : 192);
    }
    int kst = 0;
    for (int i = 0; i < 3; ++i) {
        if (a.sk[i] < 0 || a.sk[i] % fbk) return false;
        if (a.sk[i] > 0 && (!a.sa[i] || a.sld[i] % (a.f32 ? 4 : 8))) return false;
        if (i > 0 && a.sk[i] > 0 && a.sk[i - 1] == 0) return false;
        kst += a.sk[i];
    }
    if (kres + kst != a.K || a.K < fbk) return false;
    if (a.band_k && (a.N % 96 || !a.ra[0] || kst || !a.y_tm || !a.band_lo || !a.band_hi || a.band_wmax < 1 || a.band_wmax > 8)) return false;
    if (a.res_gn && (!a.res_tm || !a.res_part || !a.res_gamma || !a.res_beta || (a.ra[0] && (a.rmode == 1 || a.rmode == 3)))) return false;
    if ((long long)a.N * a.K > 0x7fffffffLL) return false;
    if ((long long)a.batch * a.seg_rows > 0x7fffffffLL / 768) return false;   // 32-bit row arithmetic in the epilogue helpers
    if (a.y_cm && (a.cm_pitch % 4 || a.cm_pitch < ((a.M + 3) & ~3))) return false;
    return true;
}
template <int NJ, bool BF, bool RS, bool SS, bool TR, int EK, int OCC>
static void launch_xgemm_one(const TGemmArgs& a, hipStream_t s) {
    const long long mt8 = ((long long)a.batch * a.seg_rows / 64 + 7) / 8 * 8;
    const int smem = xgemm_lds_bytes<NJ, BF>(RS);   // (the dynamic-LDS limit of every instantiation is raised by configure_xgemm_kernels)
    hipLaunchKernelGGL((xgemm_kernel<NJ, BF, RS, SS, TR, EK, OCC>), dim3((unsigned)(mt8 * (a.N / (32 * NJ * a.ntw)))), dim3(256), smem, s, a);
}
template <bool BF>
static bool launch_xgemm_p(const TGemmArgs& a, hipStream_t s) {
    const bool rs = a.ra[0] != nullptr, ss = a.sk[0] > 0;
    constexpr int O_RS = BF ? XG_OCC_RS : 2, O_SS = XG_OCC_SS;   // (fp32: the 52 KB resident tile allows two workgroups per CU anyway)
    if (a.band_k) { launch_xgemm_one<3, BF, true, false, true, 4, 1>(a, s); return true; }
    if (a.geglu) { if (rs && !ss) { launch_xgemm_one<4, BF, true, false, false, 2, 2>(a, s); return true; } return false; }
    if (a.qk) { if (rs && !ss) { launch_xgemm_one<3, BF, true, false, false, 1, O_RS>(a, s); return true; } return false; }
    if (a.y_cm) { if (!rs && ss) { launch_xgemm_one<3, BF, false, true, false, 3, O_SS>(a, s); return true; } return false; }
    if (!a.y_tm || a.N % 96) return false;
    if (rs && ss) launch_xgemm_one<3, BF, true, true, false, 0, O_RS>(a, s);
    else if (rs) launch_xgemm_one<3, BF, true, false, false, 0, O_RS>(a, s);
    else launch_xgemm_one<3, BF, false, true, false, 0, O_SS>(a, s);
    return true;
}
// Once per context and device (said_create), like every other kernel family: a process-wide "configured" flag inside the launch helper
// (round 3) left an Engine on a second GPU without the attribute and was written from the clip-group host threads.
template <int NJ, bool BF, bool RS, bool SS, bool TR, int EK, int OCC>
static void config_xgemm_one() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xgemm_kernel<NJ, BF, RS, SS, TR, EK, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              xgemm_lds_bytes<NJ, BF>(RS));
}
template <bool BF>
static void config_xgemm_p() {
    constexpr int O_RS = BF ? XG_OCC_RS : 2, O_SS = XG_OCC_SS;
    config_xgemm_one<3, BF, true, false, true, 4, 1>();
    config_xgemm_one<4, BF, true, false, false, 2, 2>();
    config_xgemm_one<3, BF, true, false, false, 1, O_RS>();
    config_xgemm_one<3, BF, false, true, false, 3, O_SS>();
    config_xgemm_one<3, BF, true, true, false, 0, O_RS>();
    config_xgemm_one<3, BF, true, false, false, 0, O_RS>();
    config_xgemm_one<3, BF, false, true, false, 0, O_SS>();
}
void configure_xgemm_kernels() { config_xgemm_p<true>(); }   // (bf16 only since round 6: the fp32 token-major-activation schedule was measured slower and removed)
bool launch_xgemm(const TGemmArgs& a_in, int batch, hipStream_t s) {
    TGemmArgs a = a_in;
    a.batch = batch;
    if (!xgemm_supports(a)) return false;
    const bool rs = a.ra[0] != nullptr;
    const bool nj4 = a.N % 128 == 0 && (a.geglu || a.N % 96);
    {   // column tiles per workgroup: with a resident source all of them (the prologue is paid once per row tile) unless the caller
        // chose; a concatenated input re-uses the resident buffer for its second source, so it stays at one
        const int ntiles = a.N / (nj4 ? 128 : 96);
        // (at most six: a GEGLU row tile's twelve column tiles in ONE workgroup leave 640 heavy workgroups on 512-768 slots — in situ
        // 2.135 ms per step against 2.037 with six, 32 clips x 50 steps bf16)
        int ntw = a.ntw > 0 ? a.ntw : ((rs && !a.ra[1]) ? (ntiles > 6 ? 6 : ntiles) : 1);
        if (a.ra[1]) ntw = 1;
        if (ntw > ntiles) ntw = ntiles;
        while (ntiles % ntw) --ntw;
        a.ntw = ntw;
    }
    if (a.f32) return false;   // (fp32 instantiations removed in round 6)
    return launch_xgemm_p<true>(a, s);
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm coefficients (a, b) per (sample, channel) from the producer's Welford partials, once per tensor: the same
// combination code as inside the GEMM kernels (gemm_common.h), one workgroup per sample, 48 channels per wave.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_coef_kernel(const float* __restrict__ part, long long part_bs, int cpg, int nparts, int T, float eps,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ coef_out,
                                                      long long coef_bs) {
    __shared__ float coef[2 * 192];
    __shared__ float gns[4 * GN_SCRATCH];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, b = blockIdx.x;
    const GnP gp = {cpg, nparts, T, eps, gamma, beta, 192};
    const rsrc_t rp = make_rsrc(part + (long long)b * part_bs, 192u * (unsigned)nparts * 8u);
    GnLoads gl;
    gn_issue(gp, rp, w * 48, 48, l, gl);
    gn_finish(gp, rp, w * 48, 48, l, gl, gns + w * GN_SCRATCH, coef);
    __syncthreads();
    for (int i = tid; i < 2 * 192; i += 256) coef_out[(long long)b * coef_bs + i] = coef[i];
}
void launch_gn_coef(const float* part, long long part_bs, int cpg, int nparts, int T, float eps, const float* gamma, const float* beta,
                    float* coef_out, long long coef_bs, int batch, hipStream_t s) {
    hipLaunchKernelGGL(gn_coef_kernel, dim3(batch), dim3(256), 0, s, part, part_bs, cpg, nparts, T, eps, gamma, beta, coef_out, coef_bs);
}

// ------------------------------------------------------------------------------------------------------------------
// UNet operand preparation: one workgroup = 32 tokens x 192 channels of one sample.  The tile is read with six 16-byte
// loads per thread (all in flight at once), transformed once (GroupNorm affine from the precomputed coefficients, SiLU,
// LayerNorm over channels), transposed through LDS and written token-major in bf16 with 16-byte stores (a token's 384
// bytes are contiguous).  HBM-bound by construction: 24.6 KB in, 12.3 KB out per workgroup.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 5) void prep_kernel(const PrepArgs a) {   // five workgroups per CU: 84 VGPRs, 31 KB LDS
    __shared__ float tile[192][33];     // RAW values [channel][token]
    __shared__ float coefS[192][2];     // GroupNorm (a, b) per channel (modes 0, 1)
    // one scratch area: first the GroupNorm finalisation's per-wave scratch, then (after the tile barrier) the LayerNorm partials —
    // 31.4 KB of LDS in all, so FIVE workgroups share a CU and the 1216 workgroups of a Be = 64 launch are resident at once
    // (with the two areas separate it was four: a second round of 0.75 workgroups per CU, 12 -> 21 us)
    __shared__ float gns[4 * GN_SCRATCH];
    float (*lnp)[32][2] = reinterpret_cast<float (*)[32][2]>(gns);          // [8][32][2]
    float (*lnst)[2] = reinterpret_cast<float (*)[2]>(gns + 8 * 32 * 2);     // [32][2]
    static_assert(4 * GN_SCRATCH >= 8 * 32 * 2 + 32 * 2, "LayerNorm partials alias the GroupNorm scratch");
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 32, b = blockIdx.y;
    const int T = a.T;
    const bool gn = a.mode <= 1, ln = a.mode == 1 || a.mode == 2;
    const float* xb = a.x + (long long)b * a.x_bs;
    // ---- load the raw tile -> LDS [channel][token]; GroupNorm coefficients -> LDS
    float4 v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int idx = tid + 256 * i, row = idx >> 3, q = idx & 7;
        v[i] = (t0 + 4 * q < a.pitch) ? *reinterpret_cast<const float4*>(xb + (long long)row * a.pitch + t0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (gn && a.part) {   // finalise the GroupNorm coefficients here: 4 waves x 48 channels, as gn_coef_kernel
        const int l = tid & 63, w = tid >> 6;
        const GnP gp = {a.gn_cpg, a.gn_nparts, T, a.gn_eps, a.gn_gamma, a.gn_beta, 192};
        const rsrc_t rp = make_rsrc(a.part + (long long)b * a.part_bs, 192u * (unsigned)a.gn_nparts * 8u);
        GnLoads gl;   // (20 loads up front — one round trip instead of two at T = 600 — cost 96 VGPRs + spills: 118.5 vs 116.4 ms in situ)
        gn_issue(gp, rp, w * 48, 48, l, gl);
        gn_finish(gp, rp, w * 48, 48, l, gl, gns + w * GN_SCRATCH, &coefS[0][0]);
        if (a.coef_out && blockIdx.x == 0) {   // the tensor's coefficients for a later consumer (the GroupNorm'ed residual of attn1.to_out)
            __syncthreads();
            float* co = a.coef_out + (long long)b * a.coef_out_bs;
            for (int i = tid; i < 2 * 192; i += 256) co[i] = (&coefS[0][0])[i];
        }
    } else if (gn) {
        const float* cf = a.coef + (long long)b * a.coef_bs;
        for (int i = tid; i < 2 * 192; i += 256) (&coefS[0][0])[i] = cf[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int idx = tid + 256 * i, row = idx >> 3, q = idx & 7;
        const float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[row][4 * q + k] = (t0 + 4 * q + k < T) ? e[k] : 0.f;
    }
    __syncthreads();
    // ---- LayerNorm statistics per token (modes 1, 2), over the GroupNorm'ed values in mode 1
    float mu = 0.f, rs = 1.f;
    if (ln) {
        const int tt = tid & 31, part = tid >> 5;   // 8 parts x 24 channels
        // (mean, M2) of this part's 24 channels in two passes, merged over the eight parts with Chan's update (round 6: the shifted one-pass sums of rounds 2-5 —
        // d = x - x[channel 0] — lose digits when channel 0 is an outlier channel: gains of 10 on trained-like weights)
        float xs[24];
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const int c = part * 24 + i;
            float x = tile[c][tt];
            if (gn) x = fmaf(x, coefS[c][0], coefS[c][1]);
            xs[i] = x;
            sm += x;
        }
        const float mp = sm * (1.0f / 24.0f);
        float qp = 0.f;
#pragma unroll
        for (int i = 0; i < 24; ++i) { const float d = xs[i] - mp; qp = fmaf(d, d, qp); }
        lnp[part][tt][0] = mp;
        lnp[part][tt][1] = qp;
        __syncthreads();
        if (tid < 32) {
            float mean = lnp[0][tid][0], M2 = lnp[0][tid][1];
#pragma unroll
            for (int p = 1; p < 8; ++p) {
                const float d = lnp[p][tid][0] - mean;
                const float n = 24.f * (float)p, nn = n + 24.f;
                mean = fmaf(d, 24.f / nn, mean);
                M2 += lnp[p][tid][1] + d * d * (n * 24.f / nn);
            }
            lnst[tid][0] = mean;
            lnst[tid][1] = 1.0f / sqrtf(M2 * (1.0f / 192.0f) + 1e-5f);
        }
        __syncthreads();
    }
    // ---- transform + write token-major.  A token's 192 channels are one contiguous row of the destination (768 B in fp32, 384 B in
    // bf16) and the tile's 32 rows are 16-byte chunk g = token * (chunks per row) + chunk: thread tid takes chunks tid, tid + 256, ...,
    // so the 64 lanes of every store instruction write 1 KB of consecutive bytes (with 24 channels per thread each instruction
    // scattered 64 16-byte pieces at a 96-byte stride: six partial writes per cache line).
    const int row_off = a.mode == 0 ? 1 : 0;   // conv operand: row 0 is the left padding
    if (a.f32) {   // fp32 operands (fgemm_kernel): 48 chunks of 4 channels per token
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int g = i * 256 + tid, tt = g / 48, c0 = 4 * (g - tt * 48);
            const int t = t0 + tt;
            const bool tv = t < T;
            if (!(tv || (a.mode == 0 && t == T))) continue;   // the conv operand's right padding row (token T) is written as zeros
            if (ln) { mu = lnst[tt][0]; rs = lnst[tt][1]; }
            f32x4t o, r;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;
                const float raw = tile[c][tt];
                float x = raw;
                if (gn) x = fmaf(x, coefS[c][0], coefS[c][1]);
                if (a.mode == 0) x = silu_f(x);
                if (ln) x = fmaf((x - mu) * rs, a.ln_gamma[c], a.ln_beta[c]);
                o[k] = tv ? x : 0.f;
                r[k] = raw;
                if (a.pack) { o[k] = pack_split_f16(o[k]); r[k] = pack_split_f16(r[k]); }   // (0 packs to 0: the padding rows stay all-zero bits)
            }
            *reinterpret_cast<f32x4t*>(reinterpret_cast<float*>(a.dst) + (long long)b * a.dst_bs + (long long)(t + row_off) * a.ldd + a.coff + c0) = o;
            if (a.dst2 && tv)   // raw copy (1x1 skip conv over the ResBlock input; x2 for the folded proj_out)
                *reinterpret_cast<f32x4t*>(reinterpret_cast<float*>(a.dst2) + (long long)b * a.dst2_bs + (long long)t * a.ldd2 + a.coff2 + c0) = r;
        }
        if (a.mode == 0 && t0 == 0 && tid < 48) {   // left padding row
            const f32x4t zero = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4t*>(reinterpret_cast<float*>(a.dst) + (long long)b * a.dst_bs + a.coff + 4 * tid) = zero;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // bf16 operands: 24 chunks of 8 channels per token
        const int g = i * 256 + tid, tt = g / 24, c0 = 8 * (g - tt * 24);
        const int t = t0 + tt;
        const bool tv = t < T;
        if (!(tv || (a.mode == 0 && t == T))) continue;
        if (ln) { mu = lnst[tt][0]; rs = lnst[tt][1]; }
        __attribute__((aligned(16))) __bf16 o[8];
        __attribute__((aligned(16))) __bf16 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k;
            const float raw = tile[c][tt];
            float x = raw;
            if (gn) x = fmaf(x, coefS[c][0], coefS[c][1]);
            if (a.mode == 0) x = silu_f(x);
            if (ln) x = fmaf((x - mu) * rs, a.ln_gamma[c], a.ln_beta[c]);
            o[k] = (__bf16)(tv ? x : 0.f);
            r[k] = (__bf16)raw;
        }
        *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(a.dst) + (long long)b * a.dst_bs + (long long)(t + row_off) * a.ldd + a.coff + c0) = *reinterpret_cast<const u32x4*>(o);
        if (a.dst2 && tv)
            *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(a.dst2) + (long long)b * a.dst2_bs + (long long)t * a.ldd2 + a.coff2 + c0) = *reinterpret_cast<const u32x4*>(r);
    }
    if (a.mode == 0 && t0 == 0 && tid < 24) {   // left padding row
        const u32x4 zero = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(a.dst) + (long long)b * a.dst_bs + a.coff + 8 * tid) = zero;
    }
}
bool launch_prep(const PrepArgs& a, int batch, hipStream_t s) {
    if (a.C != 192 || a.T < 1 || a.ldd % 8 || a.coff % 8 || a.dst_bs % 8 || (a.dst2 && (a.ldd2 % 8 || a.coff2 % 8 || a.dst2_bs % 8)) || a.pitch % 4) return false;
    dim3 grid(a.T / 32 + 1, batch);   // one tile past ceil(T / 32) when T % 32 == 0: the conv operand's right padding row
    static const int pad = dev_env("SAID_PREP_PAD_LDS") ? atoi(dev_env("SAID_PREP_PAD_LDS")) : 0;   // occupancy experiment: unused dynamic LDS
    hipLaunchKernelGGL(prep_kernel, grid, dim3(256), pad, s, a);
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// channel-major fp32 [b][C][pitch] -> token-major bf16 [b][T][C]   (conv0 activation, attention output)
// ------------------------------------------------------------------------------------------------------------------
__global__ void cm_to_tm_bf16_kernel(const float* __restrict__ src, long long src_bs, int pitch, unsigned short* __restrict__ dst, long long dst_bs,
                                     int T, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        tile[r][tx] = (t < T && c < C) ? src[(long long)b * src_bs + (long long)c * pitch + t] : 0.f;
    }
    __syncthreads();
    __bf16* d = reinterpret_cast<__bf16*>(dst) + (long long)b * dst_bs;
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        if (c < C && t < T) d[(long long)t * C + c] = (__bf16)tile[tx][r];
    }
}
__global__ void tm_to_group_bf16_kernel(const float* __restrict__ src, long long src_bs, unsigned short* __restrict__ dst, int T, int G, int CG, int R,
                                        int lpad) {
    const int r = blockIdx.x, b = blockIdx.y, C = G * CG;
    const int t = r - lpad;
    const bool live = t >= 0 && t < T;
    __bf16* d = reinterpret_cast<__bf16*>(dst);
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        const int g = i / CG, c = i - g * CG;
        const float v = live ? src[(long long)b * src_bs + (long long)t * C + i] : 0.f;
        d[(((long long)b * G + g) * R + r) * CG + c] = (__bf16)v;
    }
}
void launch_tm_to_group_bf16(const float* src, long long src_bs, void* dst, int B, int T, int G, int CG, int R, int lpad, hipStream_t s) {
    hipLaunchKernelGGL(tm_to_group_bf16_kernel, dim3(R, B), dim3(256), 0, s, src, src_bs, reinterpret_cast<unsigned short*>(dst), T, G, CG, R, lpad);
}
void launch_cm_to_tm_bf16(const float* src, long long src_bs, int pitch, void* dst, long long dst_bs, int B, int T, int C, hipStream_t s) {
    dim3 grid((T + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(cm_to_tm_bf16_kernel, grid, dim3(256), 0, s, src, src_bs, pitch, reinterpret_cast<unsigned short*>(dst), dst_bs, T, C);
}

// ------------------------------------------------------------------------------------------------------------------
// token-major LayerNorm over C channels, one wave per token: y = LN(x [+ add]) -> fp32 and/or bf16 copies
// ------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void ln_tm_kernel(const float* __restrict__ x, const float* __restrict__ add, float* __restrict__ yf,
                                                    unsigned short* __restrict__ yb, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, long long ntok, float eps) {
    constexpr int PER = C / 64;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= ntok) return;
    const int l = threadIdx.x & 63;
    float v[PER];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        v[i] = x[tok * C + l + 64 * i];
        if (add) v[i] += add[tok * C + l + 64 * i];
        s1 += v[i];
    }
    const float mean = wave_sum(s1) * (1.0f / C);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = v[i] - mean; s2 = fmaf(d, d, s2); }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) * (1.0f / C) + eps);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = l + 64 * i;
        const float o = fmaf((v[i] - mean) * rstd, gamma[c], beta[c]);
        if (yf) yf[tok * C + c] = o;
        if (yb) reinterpret_cast<__bf16*>(yb)[tok * C + c] = (__bf16)o;
    }
}
void launch_ln_tm(const float* x, const float* add, float* yf, void* yb, const float* gamma, const float* beta, long long ntok, int C, float eps,
                  hipStream_t s) {
    const dim3 grid((unsigned)((ntok + 3) / 4));
    if (C == 768) hipLaunchKernelGGL(ln_tm_kernel<768>, grid, dim3(256), 0, s, x, add, yf, reinterpret_cast<unsigned short*>(yb), gamma, beta, ntok, eps);
    else if (C == 512) hipLaunchKernelGGL(ln_tm_kernel<512>, grid, dim3(256), 0, s, x, add, yf, reinterpret_cast<unsigned short*>(yb), gamma, beta, ntok, eps);
    else launch_fault("ln_tm for C=%d not instantiated", C);
}

// ------------------------------------------------------------------------------------------------------------------
// F.interpolate(linear, align_corners=True) along t of token-major bf16 features (wav2vec2.py:41-44), then the feature
// projection's LayerNorm(512) — one wave per output frame -> bf16 token-major [b][Tout][C]
// ------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void interp_ln_tm_kernel(const unsigned short* __restrict__ src, long long src_bs, int Tin,
                                                           unsigned short* __restrict__ dst, long long dst_bs, int Tout, float scale,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps) {
    constexpr int PER = C / 64;
    const int b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= Tout) return;
    const int l = threadIdx.x & 63;
    const float pos = __fmul_rn(scale, (float)i);
    int i0 = min((int)pos, Tin - 1);
    const int i1 = i0 + ((i0 < Tin - 1) ? 1 : 0);
    const float l1 = __fsub_rn(pos, (float)i0), l0 = __fsub_rn(1.0f, l1);
    const __bf16* s0 = reinterpret_cast<const __bf16*>(src) + (long long)b * src_bs + (long long)i0 * C;
    const __bf16* s1p = reinterpret_cast<const __bf16*>(src) + (long long)b * src_bs + (long long)i1 * C;
    float v[PER];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        v[k] = __fadd_rn(__fmul_rn(l0, (float)s0[l + 64 * k]), __fmul_rn(l1, (float)s1p[l + 64 * k]));
        sum += v[k];
    }
    const float mean = wave_sum(sum) * (1.0f / C);
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const float d = v[k] - mean; s2 = fmaf(d, d, s2); }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) * (1.0f / C) + eps);
    __bf16* d = reinterpret_cast<__bf16*>(dst) + (long long)b * dst_bs + (long long)i * C;
#pragma unroll
    for (int k = 0; k < PER; ++k) d[l + 64 * k] = (__bf16)fmaf((v[k] - mean) * rstd, gamma[l + 64 * k], beta[l + 64 * k]);
}
void launch_interp_ln_tm(const void* src, long long src_bs, int Tin, void* dst, long long dst_bs, int Tout, int B, int C, const float* gamma,
                         const float* beta, float eps, hipStream_t s) {
    if (C != 512) { launch_fault("interp_ln_tm for C=%d not instantiated", C); return; }
    const float scale = (Tout > 1) ? (float)(Tin - 1) / (float)(Tout - 1) : 0.f;
    dim3 grid((Tout + 3) / 4, B);
    hipLaunchKernelGGL(interp_ln_tm_kernel<512>, grid, dim3(256), 0, s, reinterpret_cast<const unsigned short*>(src), src_bs, Tin,
                       reinterpret_cast<unsigned short*>(dst), dst_bs, Tout, scale, gamma, beta, eps);
}

}  // namespace said

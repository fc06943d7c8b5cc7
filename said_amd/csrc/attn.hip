// attn.hip — fp32 self-attention on gfx950.
//
// One workgroup = one (batch, head, 32-query tile); its KS waves split the key tiles
// (flash-decoding style) and merge their online-softmax states through LDS in a fixed order.
// Everything stays in registers between the two MFMAs:
//   S^T[j][i] = sum_d K[j][d] Q[i][d]      A = K fragment, B = Q fragment
//   O^T[d][i] = sum_j V[d][j] P^T[j][i]    A = V fragment, B = p[r] as-is
// Computing S transposed leaves each lane with one query column (i = lane & 31), so the row max /
// row sum are in-lane reductions plus one exchange with lane^32, the rescale factor is a per-lane
// scalar, and — because v_mfma_f32_32x32x2_f32 consumes exactly the two keys (j, j+4) that lanes
// l and l+32 already hold in accumulator register r — P feeds the second MFMA without any
// cross-lane movement or LDS round trip.
//
// Operand layouts are chosen so that every fragment is fetched with dwordx4 loads (a CU issues ~13 clocks
// per vector-memory instruction whatever its width: 112 dword loads per wave were the kernel's bottleneck):
//   * q and k arrive token-major [row][d] (the q/k/v projection writes them that way).  The contraction
//     index of MFMA number dp is d = lh * D/2 + dp for lane half lh — any pairing works as long as A and B
//     agree — so a lane needs D/2 CONSECUTIVE floats of its row: D/8 dwordx4 loads;
//   * v arrives channel-major [d][j]: MFMA r of the second product needs keys j0 + 8*(r>>2) + 4*lh + (r&3),
//     i.e. register quadruple r>>2 is one dwordx4 at column j0 + 8*(r>>2) + 4*lh of row d.
// Columns j >= T of v are never written by the projection and stay at their (finite) initial value; their
// probabilities are exactly 0.
// Reference semantics: ldm/attention.py:86-128 (scale after QK^T, softmax over all keys).
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "split_f16.h"

#ifndef SAID_ATTN_SP_ORDER
#define SAID_ATTN_SP_ORDER 0
#endif

namespace said {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4a __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8a __attribute__((ext_vector_type(8)));
// eight floats (two loaded quads) -> eight bf16 (round to nearest even): the v_mfma_f32_32x32x16_bf16 operand of one lane
static __device__ __forceinline__ bf16x8a pk_bf16x8(const f32x4a a, const f32x4a b) {
    const bf16x8a v = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    return v;
}

// The arguments are 14 scalar kernel parameters — exactly what the hardware preloads into SGPRs (build.py compiles this
// file with -amdgpu-kernarg-preload-count=14) — so the first operand request needs no scalar-memory round trip.
struct AttnView {
    const float* qk; const float* v; float* o;
    int v_bstride, o_bstride, pitch, T, heads, rows;
    float scale;
    int b0;
    int o_mode;   // 0: o channel-major fp32 [b][h * D + d][pitch]; 1 / 2: TOKEN-major fp32 / bf16 [b * o_bstride + i][h * D + d] (o_bstride =
                  // sample pitch in tokens) — the layout of the round-3 token-major activation path (tgemm.hip: xgemm_kernel)
};
// BF: both products run on v_mfma_f32_32x32x16_bf16 (said_set_precision; round 2: CDNA4's 16-deep opcode, half the MFMA count of
// the 32x32x8 form).  The operand registers are the same ones: MFMA m of S^T contracts d = lh * D/2 + 8m + (0..7), i.e. the
// loaded quads 2m, 2m+1 of each lane half (a permutation of d that K and Q share); MFMA m of O^T contracts keys
// j0 + 16m + {0..3, 8..11} + 4lh, i.e. accumulator registers 8m..8m+7 and V quads 2m, 2m+1.  Scores, softmax statistics and
// accumulation stay fp32.
// QW > 1 (large batches, KS == 1): the workgroup's QW waves take QW CONSECUTIVE query tiles of the same (batch, head), each over
// all keys, instead of splitting the keys of one query tile.  They walk the same K / V tiles in step, so a tile is fetched
// from L2 once per workgroup and the other waves hit in the CU's L1: at large batch this kernel is bound by the bytes a CU can
// keep in flight (every 32-query workgroup re-reads all K and V of its head: 62 B per kFLOP), not by MFMA or VALU work.
// PM: product mode — 0: v_mfma_f32_32x32x2_f32 on the fp32 operands, 1: bf16 operands (said_set_precision), 2: split-fp16 operands (above).
// (launch bounds: a workgroup of <= 256 threads may be alone on its SIMDs with 512 registers per wave, and for such kernels hipcc selects the MFMAs' ACCUMULATION-register
//  form — every score tile then travels to the vector registers through 32 v_accvgpr_read, the output accumulators back and forth on every rescale: 96 of ~370 VALU
//  instructions per key tile (ISA, round 6).  Asking for two waves per SIMD caps the budget at 256 registers, where the vector-register form is selected and the moves
//  disappear; head_dim 32 shapes need 126-166 registers, head_dim 64 (the audio encoder) 136-240.)
template <int ND, int KS, int PM, int QW = 1>
__global__ __launch_bounds__(64 * KS * QW, (64 * KS * QW <= 256) ? 2 : 1) void attn_kernel(const float* pqk, const float* pv, float* po, int v_bstride, int o_bstride, int ppitch,
                                                       int pT, int pheads, int prows, float pscale, int pb0, int po_mode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef SAID_AB_FLOOR
    if (pT > 0) return;
#endif
    const AttnView a = {pqk, pv, po, v_bstride, o_bstride, ppitch, pT, pheads, prows, pscale, pb0, po_mode};
    constexpr int D = 32 * ND, NQ = D / 8;   // NQ dwordx4 per lane and operand row
    constexpr bool BF = PM == 1, SP = PM == 2 || PM == 3, PS = PM == 3;   // PS: K and V arrive split (packed h | l dwords: split_f16.h pack_split_f16)
    const int tid = threadIdx.x, l = tid & 63, lt = l & 31, lh = l >> 5;
    static_assert(QW == 1 || KS == 1, "query-tile waves do not split keys");
    const int w_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = QW > 1 ? 0 : w_all;               // key-split index
    const int wq = QW > 1 ? w_all : 0;              // query tile of this wave within the workgroup
    float* const sm_w = smem + wq * (KS * 64 + KS * ND * 16 * 64);   // private merge scratch per query-tile wave
    // (round 5, measured neutral and not kept: an XCD-affine block order — every XCD a contiguous run of (sample, head, query tile) so that a head's K / V is
    //  fetched into one or two L2s instead of eight: cfg4 24.15k vs 24.14k frames/s, headline -0.2 %: profiles/r05g_attn_xcd_affine_ab.txt.  The kernel is not bound by L2 fills.)
    const int i0 = (blockIdx.x * QW + wq) * 32, h = blockIdx.y, b = blockIdx.z + a.b0;
    const int T = a.T, pitch = a.pitch, H = a.heads, rows = a.rows;
    const float* qb = a.qk + (((long long)b * 2 * H + h) * rows) * D + lh * (D / 2);
    const float* kb = a.qk + (((long long)b * 2 * H + H + h) * rows) * D + lh * (D / 2);
    const float* vb = a.v + (long long)b * a.v_bstride + (long long)(h * D + lt) * pitch + 4 * lh;

    f32x4a qf[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) qf[q] = *reinterpret_cast<const f32x4a*>(qb + (long long)min(i0 + lt, rows - 1) * D + 4 * q);

    bf16x8a qh[NQ / 2];   // bf16 mode: the query fragments are converted once, not once per key tile
#pragma unroll
    for (int q = 0; q < NQ / 2; ++q) qh[q] = pk_bf16x8(qf[2 * q], qf[2 * q + 1]);
    SplitH qs[NQ / 2];    // split mode: likewise
#pragma unroll
    for (int q = 0; q < NQ / 2; ++q) qs[q] = split_f16x8(qf[2 * q], qf[2 * q + 1]);
    float m = -1.0e30f, lsum = 0.f;
    constexpr bool ROT = SP && SAID_ATTN_SP_ORDER == 3;   // (round 4's three-accumulator rotation: race-hunt builds only)
    f32x16 o[ND], ox[SP ? ND : 1], oy[ROT ? ND : 1];   // ox (, oy): the split mode's cross terms v.l p.h + v.h p.l (x 2^11)
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[nd][r] = 0.f;
            if (SP) ox[SP ? nd : 0][r] = 0.f;
            if (ROT) oy[ROT ? nd : 0][r] = 0.f;
        }

    const int nkt = (T + 31) >> 5;
    // K and V fragments of a key tile are fetched together and one tile ahead of the MFMAs that use them
    // (register double buffer).  Every load is unconditional — tiles past the end re-read the last tile —
    // so that the compiler can count outstanding loads exactly instead of draining them all.
    auto load_kv = [&](int kt, f32x4a (&kf)[NQ], f32x4a (&vf)[ND][4]) {
        const int j0 = min(kt, nkt - 1) * 32;
#pragma unroll
        for (int q = 0; q < NQ; ++q) kf[q] = *reinterpret_cast<const f32x4a*>(kb + (long long)(j0 + lt) * D + 4 * q);
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                vf[nd][q] = *reinterpret_cast<const f32x4a*>(vb + (long long)(nd * 32) * pitch + j0 + 8 * q);
    };
    auto compute = [&](int kt, const f32x4a (&kf)[NQ], const f32x4a (&vf)[ND][4]) {
        const int j0 = kt * 32;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        if constexpr (SP) {
            SplitH ks[NQ / 2];
#pragma unroll
            for (int q = 0; q < NQ / 2; ++q) ks[q] = PS ? unpack_f16x8(kf[2 * q], kf[2 * q + 1]) : split_f16x8(kf[2 * q], kf[2 * q + 1]);
            operand_fence();
            // Two accumulators: main (k.h q.h) and cross (k.l q.h + k.h q.l, scaled by 2^-11 at the end).  Round 4 believed same-accumulator MFMAs had to be
            // "in rotation over three accumulators" to be bit-stable beside other streams; round 5 found the actual mechanism (a packed-fp32 operand misread in
            // OTHER kernels' waves while fp16 / bf16 MFMAs run beside them: DESIGN.md 8.4, said_amd/build.py NO_SLP) — the issue order here never mattered.
            // SAID_ATTN_SP_ORDER (race-hunt builds, scripts/race_localise.py): 2 = idle slots between the MFMAs, the densest aggressor pattern round 4 had
            // (every concurrent run wrong before the fix), 3 = round 4's three-accumulator rotation.
            f32x16 sxa, sxb;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sxa[r] = 0.f; sxb[r] = 0.f; }   // (sxb: the rotation build's third accumulator)
#if SAID_ATTN_SP_ORDER == 3
#pragma unroll
            for (int q = 0; q < NQ / 2; ++q) {
                sxa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[q].l, qs[q].h, sxa, 0, 0, 0);
                sxb = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[q].h, qs[q].l, sxb, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[q].h, qs[q].h, s, 0, 0, 0);
            }
#else
#pragma unroll
            for (int q = 0; q < NQ / 2; ++q) {
                sxa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[q].l, qs[q].h, sxa, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[q].h, qs[q].h, s, 0, 0, 0);
                if (SAID_ATTN_SP_ORDER == 2) idle_slots16();
                sxa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[q].h, qs[q].l, sxa, 0, 0, 0);
                if (SAID_ATTN_SP_ORDER == 2) idle_slots16();
            }
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = fmaf(ROT ? sxa[r] + sxb[r] : sxa[r], 0x1p-11f, s[r]);
        }
        if constexpr (BF) {   // (converted operands first, then the fence, then the MFMAs: operand_fence)
            bf16x8a kh[NQ / 2];
#pragma unroll
            for (int q = 0; q < NQ / 2; ++q) kh[q] = pk_bf16x8(kf[2 * q], kf[2 * q + 1]);
            operand_fence();
#pragma unroll
            for (int q = 0; q < NQ / 2; ++q) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[q], qh[q], s, 0, 0, 0);
        }
#pragma unroll
        for (int dp = 0; dp < ((SP || BF) ? 0 : ND * 16); ++dp) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[dp >> 2][dp & 3], qf[dp >> 2][dp & 3], s, 0, 0, 0);
        if constexpr (BF) {
            // bf16 mode: the VALU work per score bounds this kernel once the products are 8x cheaper, so it is trimmed: the
            // running maximum is kept in RAW score units and the scale is folded into the exponent (one fma + one exp2 per
            // score), keys past T are masked only in the tile that contains T, and the accumulator rescale is skipped while
            // no lane's maximum moved.  (The fp32 path below keeps the reference's op order: scale, subtract, exp.)
            const float c2 = a.scale * 1.4426950408889634f;
            if (j0 + 32 > T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    s[r] = (j < T) ? s[r] : -1.0e30f;
                }
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f((m - mn) * c2);   // v_exp_f32: arguments are <= 0, underflow to 0 is the wanted result
            m = mn;
            const float off = -mn * c2;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, off));
                ps += s[r];
            }
            lsum = lsum * alpha + ps;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int nd = 0; nd < ND; ++nd)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[nd][r] *= alpha;
            }
        } else {
        // fp32 path: the reference's op order (scale, subtract the maximum, exp).  Two value-preserving trims: keys past T are
        // masked only in the tile that contains T, and the accumulator rescale is skipped while no lane's maximum moved
        // (alpha == 1 exactly: multiplying by it changes nothing).
        float mx = -1.0e30f;
        if (j0 + 32 > T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                s[r] = (j < T) ? s[r] * a.scale : -1.0e30f;
                mx = fmaxf(mx, s[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = s[r] * a.scale;
                mx = fmaxf(mx, s[r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);
        const float alpha = __expf(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - mn);
            ps += s[r];
        }
        lsum = lsum * alpha + ps;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[nd][r] *= alpha;
                    if (SP) ox[SP ? nd : 0][r] *= alpha;
                    if (ROT) oy[ROT ? nd : 0][r] *= alpha;
                }
        }
        }
        // keys past T: p is exactly 0, but the never-written columns of v hold whatever the workspace held (0 x NaN is NaN; in split mode 0 x inf
        // once a finite value leaves fp16's range): zeroed in the one tile that has such keys, so no result depends on memory nobody wrote
        // (scripts/poison_ws.py: every workspace byte 0xFF before the run)
        f32x4a vz[ND][4];
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int q = 0; q < 4; ++q) vz[nd][q] = vf[nd][q];
        if (j0 + 32 > T) {
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) vz[nd][q][e] = (j0 + 8 * q + 4 * lh + e < T) ? vz[nd][q][e] : 0.f;
        }
        if constexpr (SP) {
            SplitH psa[2], vsa[2][ND];
#pragma unroll
            for (int m8 = 0; m8 < 2; ++m8) {
                const f32x4a p0 = {s[8 * m8], s[8 * m8 + 1], s[8 * m8 + 2], s[8 * m8 + 3]}, p1 = {s[8 * m8 + 4], s[8 * m8 + 5], s[8 * m8 + 6], s[8 * m8 + 7]};
                psa[m8] = split_f16x8(p0, p1);
#pragma unroll
                for (int nd = 0; nd < ND; ++nd) vsa[m8][nd] = PS ? unpack_f16x8(vz[nd][2 * m8], vz[nd][2 * m8 + 1]) : split_f16x8(vz[nd][2 * m8], vz[nd][2 * m8 + 1]);
            }
            operand_fence();
#pragma unroll
            for (int m8 = 0; m8 < 2; ++m8)
#pragma unroll
                for (int nd = 0; nd < ND; ++nd) {   // (two accumulators, as for the scores)
#if SAID_ATTN_SP_ORDER == 3
                    ox[nd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vsa[m8][nd].l, psa[m8].h, ox[nd], 0, 0, 0);
                    oy[ROT ? nd : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vsa[m8][nd].h, psa[m8].l, oy[ROT ? nd : 0], 0, 0, 0);
                    o[nd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vsa[m8][nd].h, psa[m8].h, o[nd], 0, 0, 0);
#else
                    ox[nd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vsa[m8][nd].l, psa[m8].h, ox[nd], 0, 0, 0);
                    o[nd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vsa[m8][nd].h, psa[m8].h, o[nd], 0, 0, 0);
                    if (SAID_ATTN_SP_ORDER == 2) idle_slots16();
                    ox[nd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vsa[m8][nd].h, psa[m8].l, ox[nd], 0, 0, 0);
                    if (SAID_ATTN_SP_ORDER == 2) idle_slots16();
#endif
                }
        }
        if constexpr (BF) {
            bf16x8a pb[2], vb8[2][ND];
#pragma unroll
            for (int m8 = 0; m8 < 2; ++m8) {
                const f32x4a p0 = {s[8 * m8], s[8 * m8 + 1], s[8 * m8 + 2], s[8 * m8 + 3]}, p1 = {s[8 * m8 + 4], s[8 * m8 + 5], s[8 * m8 + 6], s[8 * m8 + 7]};
                pb[m8] = pk_bf16x8(p0, p1);
#pragma unroll
                for (int nd = 0; nd < ND; ++nd) vb8[m8][nd] = pk_bf16x8(vz[nd][2 * m8], vz[nd][2 * m8 + 1]);
            }
            operand_fence();
#pragma unroll
            for (int m8 = 0; m8 < 2; ++m8)
#pragma unroll
                for (int nd = 0; nd < ND; ++nd) o[nd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb8[m8][nd], pb[m8], o[nd], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < ((SP || BF) ? 0 : 16); ++r)
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) o[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(vz[nd][r >> 2][r & 3], s[r], o[nd], 0, 0, 0);
    };
    if constexpr (QW > 1) {
        // K / V tiles staged ONCE per workgroup through LDS and shared by its QW query-tile waves.  Fetching fragments straight
        // from global memory costs one cache line per lane and instruction (a token-major K row / a channel-major V row is a
        // line of its own, read in four 16-byte pieces): 256 line transactions per key tile and wave for 8 KB of data, times
        // QW waves — the kernel sat on the CU's L1 path (73 % issue stall, 13 % MFMA busy: profiles/r02c_pmc_sq_b32_bf16.txt).
        // Here 256 threads move the 2 x (32 x D) floats of a tile with one or two coalesced 16-byte loads each, one tile ahead
        // (registers -> the other LDS buffer after the current tile's products), one barrier per tile.
        constexpr int KP = D + 4, VP = 36;                      // LDS row pitches (floats): conflict-free 16-byte fragment reads
        constexpr int TILE_F = 32 * KP + D * VP;
        constexpr int NLD = (32 * D / 4) / (64 * QW);           // float4 per thread and operand (1 for D = 32, 2 for D = 64)
        float* kv = smem + QW * (KS * 64 + KS * ND * 16 * 64);  // [2][K 32 x KP | V D x VP] behind the merge scratch
        const float* kg = a.qk + (((long long)b * 2 * H + H + h) * rows) * D;
        const float* vg = a.v + (long long)b * a.v_bstride + (long long)(h * D) * pitch;
        f32x4a rk[NLD], rv[NLD];
        auto gload = [&](int kt) {
            const int j0 = min(kt, nkt - 1) * 32;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx = tid + 64 * QW * i;
                const int krow = idx / (D / 4), kq = idx - krow * (D / 4);     // K: 32 rows of D floats
                rk[i] = *reinterpret_cast<const f32x4a*>(kg + (long long)(j0 + krow) * D + 4 * kq);
                const int vrow = idx >> 3, vq = idx & 7;                      // V: D rows of 32 floats
                rv[i] = *reinterpret_cast<const f32x4a*>(vg + (long long)vrow * pitch + j0 + 4 * vq);
            }
        };
        auto lstore = [&](int buf) {
            float* ks = kv + buf * TILE_F;
            float* vs = ks + 32 * KP;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int idx = tid + 64 * QW * i;
                const int krow = idx / (D / 4), kq = idx - krow * (D / 4);
                *reinterpret_cast<f32x4a*>(ks + krow * KP + 4 * kq) = rk[i];
                const int vrow = idx >> 3, vq = idx & 7;
                *reinterpret_cast<f32x4a*>(vs + vrow * VP + 4 * vq) = rv[i];
            }
        };
        gload(0);
        lstore(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            gload(kt + 1);
            const float* ks = kv + (kt & 1) * TILE_F;
            const float* vs = ks + 32 * KP;
            f32x4a kA[NQ], vA[ND][4];
#pragma unroll
            for (int q = 0; q < NQ; ++q) kA[q] = *reinterpret_cast<const f32x4a*>(ks + lt * KP + lh * (D / 2) + 4 * q);
#pragma unroll
            for (int nd = 0; nd < ND; ++nd)
#pragma unroll
                for (int q = 0; q < 4; ++q) vA[nd][q] = *reinterpret_cast<const f32x4a*>(vs + (nd * 32 + lt) * VP + 8 * q + 4 * lh);
            __builtin_amdgcn_sched_barrier(0);
            compute(kt, kA, vA);
            __builtin_amdgcn_sched_barrier(0);
            lstore((kt + 1) & 1);
            __syncthreads();
        }
    } else if constexpr (ND == 1) {
        f32x4a kA[NQ], vA[ND][4], kB[NQ], vB[ND][4];
        // sched_barrier: without it the machine scheduler sinks the prefetch loads down between the MFMAs that consume
        // them (s_waitcnt vmcnt(0) in front of every fourth MFMA) and the double buffer hides nothing
        load_kv(w, kA, vA);
        __builtin_amdgcn_sched_barrier(0);
        for (int kt = w; kt < nkt; kt += 2 * KS) {
            load_kv(kt + KS, kB, vB);
            __builtin_amdgcn_sched_barrier(0);
            compute(kt, kA, vA);
            __builtin_amdgcn_sched_barrier(0);
            load_kv(kt + 2 * KS, kA, vA);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + KS < nkt) compute(kt + KS, kB, vB);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {  // head_dim 64: one register buffer (a second one would spill); loads of a tile still go out together
        f32x4a kA[NQ], vA[ND][4];
        for (int kt = w; kt < nkt; kt += KS) {
            load_kv(kt, kA, vA);
            compute(kt, kA, vA);
        }
    }
    lsum += __shfl_xor(lsum, 32);
    if constexpr (SP) {
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[nd][r] = fmaf(ROT ? ox[nd][r] + oy[ROT ? nd : 0][r] : ox[nd][r], 0x1p-11f, o[nd][r]);
    }

    // ---- merge the KS partial states ----
    float* ml = sm_w;                 // [KS][2][32]
    float* ob = sm_w + KS * 64;       // [KS][ND][16][64]
    if (lh == 0) {
        ml[(w * 2 + 0) * 32 + lt] = m;
        ml[(w * 2 + 1) * 32 + lt] = lsum;
    }
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[((w * ND + nd) * 16 + r) * 64 + l] = o[nd][r];
    __syncthreads();
    float M = -1.0e30f;
#pragma unroll
    for (int w2 = 0; w2 < KS; ++w2) M = fmaxf(M, ml[(w2 * 2) * 32 + lt]);
    float f[KS];
    float L = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < KS; ++w2) {
        f[w2] = BF ? __builtin_amdgcn_exp2f((ml[(w2 * 2) * 32 + lt] - M) * (a.scale * 1.4426950408889634f)) : __expf(ml[(w2 * 2) * 32 + lt] - M);   // BF: raw-score maxima
        L += ml[(w2 * 2 + 1) * 32 + lt] * f[w2];
    }
    const float invL = 1.0f / L;
    constexpr int NV = ND * 16;
    static_assert(NV % KS == 0, "");
    if (a.o_mode != 0) {
        // token-major destination (KS == 1 only: every register of this wave is final).  Registers 4 q .. 4 q + 3 are channels
        // 8 q + 4 lh + (0 .. 3) of the head: four consecutive elements of the query's row
        if constexpr (KS == 1) {
            const int i = i0 + lt;
            if (i < T) {
                const long long row = ((long long)b * a.o_bstride + i) * (H * D) + h * D;
#pragma unroll
                for (int nd = 0; nd < ND; ++nd)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v0 = o[nd][4 * q] * invL, v1 = o[nd][4 * q + 1] * invL, v2 = o[nd][4 * q + 2] * invL, v3 = o[nd][4 * q + 3] * invL;
                        const long long off = row + nd * 32 + 8 * q + 4 * lh;
                        if (a.o_mode == 2) {
                            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                            const bf16x4 ov = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
                            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.o) + off) = ov;
                        } else {
                            const f32x4a ov = {v0, v1, v2, v3};
                            *reinterpret_cast<f32x4a*>(a.o + off) = ov;
                        }
                    }
            }
        }
        return;
    }
    float* ob_out = a.o + (long long)b * a.o_bstride + (long long)(h * D) * pitch;
#pragma unroll
    for (int jv = 0; jv < NV / KS; ++jv) {
        const int v = w + jv * KS;
        const int nd = v >> 4, r = v & 15;
        float acc = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < KS; ++w2) acc += ob[((w2 * ND + nd) * 16 + r) * 64 + l] * f[w2];
        const int d = nd * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int i = i0 + lt;
        if (i < T) ob_out[(long long)d * pitch + i] = acc * invL;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// battn_kernel (round 4) — self-attention of the bf16 large-batch schedule on bf16 OPERANDS, a head's K and V resident in LDS.
//
// attn_kernel<1, 1, true, 4> took fp32 q / k / v, staged one 32-key tile at a time through LDS behind a workgroup barrier per tile and
// converted every fragment to bf16 in each of its four waves: 57 us per launch at 64 x 600 tokens against ~15 us of exp / max / sum
// work (the MFMAs are a quarter of that).  Here the q/k/v projection (rgemm.hip, TGemmArgs::qkv_bf16) writes bf16 in this kernel's
// operand layout, and a workgroup (four waves = four 32-query tiles of one (sample, head)) copies the head's WHOLE K [T][32] and
// V^T [32][T] — 2 x 40 KB for T <= 640 — into LDS once, with plain 16-byte pieces, then runs its key loop WITHOUT any further barrier:
//   S^T[j][i] = sum_d K[j][d] Q[i][d]   (A = K rows, B = Q rows: two v_mfma_f32_32x32x16_bf16)
//   O^T[d][i] = sum_j V^T[d][j] P^T[j][i]   (A = V^T rows, B = P^T straight from the S^T accumulators: registers 8 m .. 8 m + 7 of a lane
//   half are keys 16 m + {0-3, 8-11} + 4 lh — the order V^T's tokens are stored in, so a lane's eight keys are one 16-byte piece)
// LDS images: K unpadded with the 16-byte piece index XOR-swizzled by the row (piece ^ ((row >> 2) & 3)), V^T rows padded by one piece:
// conflict-free ds_read_b128 for both; 80 KB: two workgroups per CU, one copying while the other multiplies.  Online softmax as in attn_kernel's bf16 path (raw-score running maximum, scale folded into the exp2 argument, rescale
// skipped while no maximum moved).  Reference semantics: ldm/attention.py:86-128.
// QT = query tiles (= waves) per workgroup.  The key loop is one dependent chain per wave (LDS read -> MFMA -> max -> exp2 -> MFMA): with
// four waves per workgroup and two workgroups per CU a SIMD holds two such chains and the kernel ran latency-bound (52 us); eight waves
// put four on every SIMD and halve the K / V copies per query tile.
typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
template <int QT>
__global__ __launch_bounds__(64 * QT, (QT == 10 ? 5 : (QT == 8 ? 4 : 2))) void battn_kernel(const unsigned short* __restrict__ qk, const unsigned short* __restrict__ vt, unsigned short* __restrict__ outp,
                                                       int v_bstride, int o_bstride, int pitch, int T, int heads, int rows, float scale) {
    extern __shared__ __attribute__((aligned(16))) u32x4b bsm[];
    const int nkt = (T + 31) >> 5, nkr = nkt * 32;
    u32x4b* const Ks = bsm;                 // [nkr keys][4 pieces]
    u32x4b* const Vs = bsm + nkr * 4;       // [32 channels][81 pieces]: rows one piece longer than 80 -> conflict-free 16-byte reads of 16 consecutive rows
    const int tid = threadIdx.x, l = tid & 63, lt = l & 31, lh = l >> 5, w = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z, H = heads;
    constexpr int NTH = 64 * QT;
    const int i0 = (blockIdx.x * QT + w) * 32;
    const unsigned short* kg = qk + (((long long)b * 2 * H + H + h) * rows) * 32;
    const unsigned short* vg = vt + (long long)b * v_bstride + (long long)(h * 32) * pitch;
    // ---- the head's K and V^T -> LDS (rows past T: the last row again / the zeros the projection wrote)
    {
        const int npc = nkr >> 3;           // 16-byte pieces per V^T row
        const float rnpc = __builtin_amdgcn_rcpf((float)npc);   // idx / npc through one float multiply (exact here: idx < 2^12, npc <= 80, the + 0.5 keeps the quotient clear of
                                                                // rounding at multiples of npc; an integer division is ~25 dependent vector instructions, four of them per lane and pass)
        for (int i = tid; i < nkr * 4; i += NTH * 4) {
            u32x4b kv[4], vv[4];
            int ki[4], vi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = min(i + NTH * u, nkr * 4 - 1);
                const int row = idx >> 2, pc = idx & 3;
                kv[u] = *reinterpret_cast<const u32x4b*>(kg + (long long)min(row, rows - 1) * 32 + 8 * pc);
                ki[u] = row * 4 + (pc ^ ((row >> 2) & 3));
                const int d = (int)(((float)idx + 0.5f) * rnpc), p = idx - d * npc;        // (32 npc == 4 nkr: the same index range)
                vv[u] = *reinterpret_cast<const u32x4b*>(vg + (long long)d * pitch + 8 * p);
                vi[u] = d * 81 + p;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { Ks[ki[u]] = kv[u]; Vs[vi[u]] = vv[u]; }
        }
    }
    const unsigned short* qb = qk + (((long long)b * 2 * H + h) * rows + min(i0 + lt, rows - 1)) * 32 + 8 * lh;
    const bf16x8a q0 = __builtin_bit_cast(bf16x8a, *reinterpret_cast<const u32x4b*>(qb));
    const bf16x8a q1 = __builtin_bit_cast(bf16x8a, *reinterpret_cast<const u32x4b*>(qb + 16));
    __syncthreads();
    if (i0 >= T) return;                    // (a workgroup's spare wave: it only helped with the copy)
    // The loop below is bound by VALU ISSUE — knock-outs on the box (profiles/r04e_battn_knockouts.txt): copy 13 us, loop 30 us = 420 clocks per
    // key tile and SIMD, of which the 16 v_exp_f32 are 256 (a wave64 VALU instruction holds the SIMD for 4 clocks, a transcendental for 16;
    // the six MFMAs hide behind them) — so it is written for few instructions per score:
    //  * q arrives pre-multiplied by scale * log2(e) (the projection's epilogue, before the rounding to bf16): scores are exponents;
    //  * the score accumulator starts at -ref (a per-query reference, lane == query), so p = exp2(acc) with no subtraction; ref is the
    //    first tile's maximum and moves only when a later score exceeds it by more than 16 (then o and the sums are rescaled: rare);
    //  * the row sums come from two extra MFMAs against a fragment of ones (the matrix pipe has room: 6 of ~60 instructions);
    //  * V^T rows are padded by one 16-byte piece instead of swizzled: the fragment address advances by a constant.
    f32x16 o, ls, nref;                     // nref: every register = -(the lane's reference): the score MFMAs accumulate onto it (no 16 moves per tile)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[r] = 0.f; ls[r] = 0.f; nref[r] = 0.f; }
    const bf16x8a ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    const u32x4b* kp = Ks + lt * 4;
    const int ksw0 = lh ^ ((lt >> 2) & 3), ksw1 = (2 + lh) ^ ((lt >> 2) & 3);   // (key rows advance by 32: the swizzle term is loop-invariant)
    const u32x4b* vp = Vs + lt * 81 + lh;
    // One key tile.  MASK: the tile holds keys past T (only the last tile of a T % 32 != 0 sequence) — a separate instantiation, because the
    // compiler turns a run-time `if (j0 + 32 > T)` around the 16 selects into 32 compare / select instructions executed for EVERY tile (ISA, round 4).
    auto key_tile = [&](int kt, auto mask_c) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(mask_c)::value;
        const int j0 = kt * 32;
        const bf16x8a k0 = __builtin_bit_cast(bf16x8a, kp[kt * 128 + ksw0]);
        const bf16x8a k1 = __builtin_bit_cast(bf16x8a, kp[kt * 128 + ksw1]);
        const bf16x8a v0 = __builtin_bit_cast(bf16x8a, vp[kt * 4]);
        const bf16x8a v1 = __builtin_bit_cast(bf16x8a, vp[kt * 4 + 2]);
        f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0, nref, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q1, s, 0, 0, 0);
        if constexpr (MASK) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                s[r] = (j < T) ? s[r] : -1.0e30f;
            }
        }
        float mx = __builtin_fmaxf(__builtin_fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[r]), s[r + 1]);   // (v_max3_f32)
        mx = fmaxf(mx, s[15]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const bool rebase = kt == 0 || mx > 16.0f;
        if (__builtin_amdgcn_ballot_w64(rebase)) {
            asm volatile("; rebase (rare)");               // (a statement the compiler may not speculate: without it the 56 instructions below were
                                                           //  if-converted into EVERY tile's path with d = 0, f = 1 — ISA, round 4)
            const float d = rebase ? mx : 0.f;             // this lane's reference moves up (or, first tile, to) by d
            const float f = kt == 0 ? 0.f : __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] -= d; nref[r] -= d; o[r] *= f; ls[r] *= f; }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
        const f32x4a p0 = {s[0], s[1], s[2], s[3]}, p1 = {s[4], s[5], s[6], s[7]}, p2 = {s[8], s[9], s[10], s[11]}, p3 = {s[12], s[13], s[14], s[15]};
        const bf16x8a pa = pk_bf16x8(p0, p1), pb = pk_bf16x8(p2, p3);
        operand_fence();   // (the converted probabilities are MFMA operands: see operand_fence; never seen failing here, costs nothing measurable)
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pa, o, 0, 0, 0);
        ls = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pa, ls, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb, o, 0, 0, 0);
        ls = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pb, ls, 0, 0, 0);
    };
    const int nfull = (T & 31) ? nkt - 1 : nkt;
    for (int kt = 0; kt < nfull; ++kt) key_tile(kt, std::false_type{});
    if (nfull < nkt) key_tile(nkt - 1, std::true_type{});
    const float invL = 1.0f / ls[0];        // (every row of the ones-product is the query's sum over all keys)
    const int i = i0 + lt;
    if (i < T) {   // registers 4 q .. 4 q + 3 are channels 8 q + 4 lh + (0 .. 3) of the head: four consecutive bf16 of the query's token-major row
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        __bf16* orow = reinterpret_cast<__bf16*>(outp) + ((long long)b * o_bstride + i) * (H * 32) + h * 32 + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bf16x4 ov = {(__bf16)(o[4 * q] * invL), (__bf16)(o[4 * q + 1] * invL), (__bf16)(o[4 * q + 2] * invL), (__bf16)(o[4 * q + 3] * invL)};
            *reinterpret_cast<bf16x4*>(orow + 8 * q) = ov;
        }
    }
}
bool battn_supports(const AttnArgs& a, int head_dim) {
    return head_dim == 32 && a.T >= 1 && a.T <= 640 && a.pitch >= ((a.T + 31) & ~31) && a.rows >= ((a.T + 31) & ~31) && a.v_bstride <= 0x7fffffffLL && a.o_bstride <= 0x7fffffffLL;
}
// q / k: bf16 [b][2 heads][rows][32] at a.qk, v: bf16 [b][heads * 32][pitch] at a.v (tokens permuted per 16: TGemmArgs::qkv_bf16), o: bf16 token-major
void launch_battn(const AttnArgs& a, int batch, hipStream_t s, int qt) {
    // K: the key rows actually used; V^T: 81 pieces per row.  T = 600: 80,384 bytes — two workgroups per CU with room to spare (2 x 81,920 is the
    // whole LDS of a CU to the byte, and measured like ONE workgroup per CU)
    const int lds_bytes = ((((a.T + 31) / 32) * 32) * 4 + 32 * 81) * 16;
    if (qt == 4) {
        dim3 grid((((a.T + 31) / 32) + 3) / 4, a.heads, batch);
        hipLaunchKernelGGL(battn_kernel<4>, grid, dim3(256), lds_bytes, s, reinterpret_cast<const unsigned short*>(a.qk), reinterpret_cast<const unsigned short*>(a.v),
                           reinterpret_cast<unsigned short*>(a.o), (int)a.v_bstride, (int)a.o_bstride, a.pitch, a.T, a.heads, a.rows, a.scale);
        return;
    }
    if (qt == 10) {   // 19 query tiles (T = 600) as two workgroups of ten waves: two K / V copies per head instead of three, one idle wave instead of five
        dim3 grid((((a.T + 31) / 32) + 9) / 10, a.heads, batch);
        hipLaunchKernelGGL(battn_kernel<10>, grid, dim3(640), lds_bytes, s, reinterpret_cast<const unsigned short*>(a.qk), reinterpret_cast<const unsigned short*>(a.v),
                           reinterpret_cast<unsigned short*>(a.o), (int)a.v_bstride, (int)a.o_bstride, a.pitch, a.T, a.heads, a.rows, a.scale);
        return;
    }
    dim3 grid((((a.T + 31) / 32) + 7) / 8, a.heads, batch);
    hipLaunchKernelGGL(battn_kernel<8>, grid, dim3(512), lds_bytes, s, reinterpret_cast<const unsigned short*>(a.qk), reinterpret_cast<const unsigned short*>(a.v),
                       reinterpret_cast<unsigned short*>(a.o), (int)a.v_bstride, (int)a.o_bstride, a.pitch, a.T, a.heads, a.rows, a.scale);
}

template <int ND, int KS, int BF, int QW = 1>
static void launch_attn_one(const AttnArgs& a, int batch, hipStream_t s) {
    constexpr int D_ = 32 * ND;
    const int smem = (QW * (KS * 64 + KS * ND * 16 * 64) + (QW > 1 ? 2 * (32 * (D_ + 4) + D_ * 36) : 0)) * (int)sizeof(float);
    dim3 grid(((a.T + 31) / 32 + QW - 1) / QW, a.heads, batch);
    if (a.v_bstride > 0x7fffffffLL || a.o_bstride > 0x7fffffffLL) { launch_fault("attention batch stride exceeds 31 bits"); return; }
    hipLaunchKernelGGL((attn_kernel<ND, KS, BF, QW>), grid, dim3(64 * KS * QW), smem, s, a.qk, a.v, a.o, (int)a.v_bstride, (int)a.o_bstride, a.pitch, a.T,
                       a.heads, a.rows, a.scale, a.b0, a.o_mode);
}
template <int ND, int KS, int BF, int QW = 1>
static void configure_attn_one() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<ND, KS, BF, QW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
}
template <int PM>
static void configure_attn_modes() {
    configure_attn_one<1, 8, PM>(); configure_attn_one<1, 4, PM>(); configure_attn_one<1, 1, PM>(); configure_attn_one<1, 1, PM, 4>();
    configure_attn_one<2, 8, PM>(); configure_attn_one<2, 4, PM>(); configure_attn_one<2, 1, PM>(); configure_attn_one<2, 1, PM, 4>();
}
void configure_attn_kernels() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&battn_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&battn_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&battn_kernel<10>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    configure_attn_modes<0>(); configure_attn_modes<1>(); configure_attn_modes<2>();
    configure_attn_one<1, 8, 3>(); configure_attn_one<1, 4, 3>(); configure_attn_one<1, 1, 3, 4>();
    configure_attn2q_kernel();
}

template <int PM>
static bool launch_attn_mode(const AttnArgs& a, int batch, int head_dim, int KS, hipStream_t s) {
    if (KS == -4) {   // four query tiles per workgroup, no key split (large batches)
        if (head_dim == 32) return launch_attn_one<1, 1, PM, 4>(a, batch, s), true;
        if (head_dim == 64) return launch_attn_one<2, 1, PM, 4>(a, batch, s), true;
    }
    if (head_dim == 32 && KS == 8) return launch_attn_one<1, 8, PM>(a, batch, s), true;
    if (head_dim == 32 && KS == 4) return launch_attn_one<1, 4, PM>(a, batch, s), true;
    if (head_dim == 32 && KS == 1) return launch_attn_one<1, 1, PM>(a, batch, s), true;
    if (head_dim == 64 && KS == 8) return launch_attn_one<2, 8, PM>(a, batch, s), true;
    if (head_dim == 64 && KS == 4) return launch_attn_one<2, 4, PM>(a, batch, s), true;
    if (head_dim == 64 && KS == 1) return launch_attn_one<2, 1, PM>(a, batch, s), true;
    return false;
}
// mode: 0 fp32 MFMA, 1 bf16 operands, 2 split-fp16 operands, 3 split-fp16 operands with K and V stored pre-split by the q/k/v GEMM (head_dim 32: key-split shapes and the four-query-tile shape)
void launch_attn(const AttnArgs& a, int batch, int head_dim, int KS, hipStream_t s, int mode) {
    if (mode == 3 && head_dim == 32 && KS == 34) { launch_attn2q(a, batch, s); return; }   // three query tiles per wave (attn2q.hip)
    if (mode == 3) {
        if (head_dim == 32 && KS == 8) { launch_attn_one<1, 8, 3>(a, batch, s); return; }
        if (head_dim == 32 && KS == 4) { launch_attn_one<1, 4, 3>(a, batch, s); return; }
        if (head_dim == 32 && KS == -4) { launch_attn_one<1, 1, 3, 4>(a, batch, s); return; }   // (round 6: large batches — the token-major q/k/v GEMM packs k and v too)
        launch_fault("pre-split attention operands: unsupported config D=%d KS=%d", head_dim, KS);
        return;
    }
    const bool ok = mode == 1 ? launch_attn_mode<1>(a, batch, head_dim, KS, s) : (mode == 2 ? launch_attn_mode<2>(a, batch, head_dim, KS, s) : launch_attn_mode<0>(a, batch, head_dim, KS, s));
    if (!ok) launch_fault("unsupported attention config D=%d KS=%d", head_dim, KS);
}

}  // namespace said

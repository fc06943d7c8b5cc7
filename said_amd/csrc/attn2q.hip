// attn2q.hip — self-attention of long sequences at small batch on pre-split operands: several query tiles per wave (round 6).
// Reference semantics: ldm/attention.py:86-128 (scale after QK^T, softmax over all keys).  Operand layouts, product order and merge: attn.hip.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "split_f16.h"

namespace said {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4a __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------------
// attn2q_kernel (round 6) — long sequences at small batch (configs[4]: T = 1800): QT = 3 query tiles per wave.
//
// attn_kernel<1, 4, 3, 1> gives every 32-query tile of a (sample, head) its own workgroup, and each of them pulls the head's whole K and V (1800 x 32 packed pairs x 2 =
// 460 KB) through its CU's L1: 684 workgroups x 460 KB = 315 MB of L2 -> L1 traffic per launch for 9.7 MB of operands (PMC: 44 MB even from HBM, every XCD's L2 fetching
// every head) — 42-44 us per launch, four launches = 27 % of the step.  Here a wave keeps the states of QT consecutive query tiles and runs all of them against each K / V
// fragment it fetches: a third of the fragment traffic and of the unpacking per score; 57 tiles in threes are 228 workgroups of four waves: one wave per SIMD, one round.
// Same operands (pre-split K / V: split_f16.h pack_split_f16), same key tiles per wave (kt = w, w + KS, ...), same per-tile arithmetic and the same merge as
// attn_kernel<1, KS, 3, 1>: bit-identical results (tests/test_gpu_round6.py).  The raw fragments of the NEXT tile are requested into the registers the current tile's were
// unpacked from.  This file is compiled with -mllvm -amdgpu-mfma-vgpr-form (said_amd/build.py): a 256-thread workgroup may have 512 registers per wave, and hipcc then
// selects the MFMAs' accumulation-register form — every score tile travelled to the vector registers through 32 v_accvgpr_read, 288 such moves per loop pass
// (first version, 391 registers: 39.3 against 44 us per launch, profiles/r06h_attn2q_ab.txt).  With the vector-register form the three tiles' state has to fit 256
// registers: the query fragments (48) live in LDS and are read per product.  Two tiles per wave (342 workgroups in two rounds) measured slower than one and is not built.
// ------------------------------------------------------------------------------------------------------------------
template <int KS, int QT>
__global__ __launch_bounds__(64 * KS, 1) void attn2q_kernel(const float* pqk, const float* pv, float* po, int v_bstride, int o_bstride, int ppitch,
                                                         int pT, int pheads, int prows, float pscale, int pb0, int po_mode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    (void)po_mode;
    constexpr int D = 32, NQ = 4;
    const int tid = threadIdx.x, l = tid & 63, lt = l & 31, lh = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z + pb0;
    const int T = pT, pitch = ppitch, H = pheads, rows = prows;
    const float* qb = pqk + (((long long)b * 2 * H + h) * rows) * D + lh * (D / 2);
    const float* kb = pqk + (((long long)b * 2 * H + H + h) * rows) * D + lh * (D / 2);
    const float* vb = pv + (long long)b * v_bstride + (long long)(h * D + lt) * pitch + 4 * lh;
    int i0[QT];
    // the query fragments live in LDS (a private copy per wave: no barrier): 16 registers per tile that the vector-register form of the MFMAs needs for accumulators
    constexpr int SCR = KS * 64 + KS * 16 * 64;
    f16x8a* const qL = reinterpret_cast<f16x8a*>(smem + QT * SCR) + (w * QT * 4) * 64 + l;   // [wave][tile][q0.h, q0.l, q1.h, q1.l][64 lanes]
    float m[QT], lsum[QT];
    f32x16 o[QT], ox[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        i0[t] = (blockIdx.x * QT + t) * 32;
        f32x4a qf[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) qf[q] = *reinterpret_cast<const f32x4a*>(qb + (long long)min(i0[t] + lt, rows - 1) * D + 4 * q);
#pragma unroll
        for (int q = 0; q < NQ / 2; ++q) {
            const SplitH sq = split_f16x8(qf[2 * q], qf[2 * q + 1]);
            qL[((t * 2 + q) * 2 + 0) * 64] = sq.h;
            qL[((t * 2 + q) * 2 + 1) * 64] = sq.l;
        }
        m[t] = -1.0e30f; lsum[t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[t][r] = 0.f; ox[t][r] = 0.f; }
    }
    const int nkt = (T + 31) >> 5;
    f32x4a kR[NQ], vR[4];
    auto load_kv = [&](int kt) {   // (unconditional: tiles past the end re-read the last one)
        const int j0 = min(kt, nkt - 1) * 32;
#pragma unroll
        for (int q = 0; q < NQ; ++q) kR[q] = *reinterpret_cast<const f32x4a*>(kb + (long long)(j0 + lt) * D + 4 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) vR[q] = *reinterpret_cast<const f32x4a*>(vb + j0 + 8 * q);
    };
    load_kv(w);
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = w; kt < nkt; kt += KS) {
        const int j0 = kt * 32;
        // keys past T: p is exactly 0, but the never-written columns of v hold whatever the workspace held: zeroed in the one tile that has such keys (attn_kernel)
        if (j0 + 32 > T) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) vR[q][e] = (j0 + 8 * q + 4 * lh + e < T) ? vR[q][e] : 0.f;
        }
        SplitH ks[NQ / 2], vs[2];
#pragma unroll
        for (int q = 0; q < NQ / 2; ++q) ks[q] = unpack_f16x8(kR[2 * q], kR[2 * q + 1]);
#pragma unroll
        for (int m8 = 0; m8 < 2; ++m8) vs[m8] = unpack_f16x8(vR[2 * m8], vR[2 * m8 + 1]);
        __builtin_amdgcn_sched_barrier(0);
        load_kv(kt + KS);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            f32x16 s, sxa;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; sxa[r] = 0.f; }
#pragma unroll
            for (int q = 0; q < NQ / 2; ++q) {
                const f16x8a qh = qL[((t * 2 + q) * 2 + 0) * 64], ql = qL[((t * 2 + q) * 2 + 1) * 64];
                sxa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[q].l, qh, sxa, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[q].h, qh, s, 0, 0, 0);
                sxa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[q].h, ql, sxa, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = fmaf(sxa[r], 0x1p-11f, s[r]);
            // the reference's op order (scale, subtract the maximum, exp): attn_kernel's fp32 path
            float mx = -1.0e30f;
            if (j0 + 32 > T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    s[r] = (j < T) ? s[r] * pscale : -1.0e30f;
                    mx = fmaxf(mx, s[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = s[r] * pscale;
                    mx = fmaxf(mx, s[r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m[t], mx);
            const float alpha = __expf(m[t] - mn);
            m[t] = mn;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __expf(s[r] - mn);
                ps += s[r];
            }
            lsum[t] = lsum[t] * alpha + ps;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[t][r] *= alpha; ox[t][r] *= alpha; }
            }
            SplitH psa[2];
#pragma unroll
            for (int m8 = 0; m8 < 2; ++m8) {
                const f32x4a p0 = {s[8 * m8], s[8 * m8 + 1], s[8 * m8 + 2], s[8 * m8 + 3]}, p1 = {s[8 * m8 + 4], s[8 * m8 + 5], s[8 * m8 + 6], s[8 * m8 + 7]};
                psa[m8] = split_f16x8(p0, p1);
            }
#pragma unroll
            for (int m8 = 0; m8 < 2; ++m8) {
                ox[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vs[m8].l, psa[m8].h, ox[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vs[m8].h, psa[m8].h, o[t], 0, 0, 0);
                ox[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vs[m8].h, psa[m8].l, ox[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- merge the KS partial states of each query tile (attn_kernel's merge, per tile) ----
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        lsum[t] += __shfl_xor(lsum[t], 32);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = fmaf(ox[t][r], 0x1p-11f, o[t][r]);
        float* ml = smem + t * SCR;
        float* ob = ml + KS * 64;
        if (lh == 0) {
            ml[(w * 2 + 0) * 32 + lt] = m[t];
            ml[(w * 2 + 1) * 32 + lt] = lsum[t];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[(w * 16 + r) * 64 + l] = o[t][r];
    }
    __syncthreads();
    float* ob_out = po + (long long)b * o_bstride + (long long)(h * D) * pitch;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float* ml = smem + t * SCR;
        const float* ob = ml + KS * 64;
        float M = -1.0e30f;
#pragma unroll
        for (int w2 = 0; w2 < KS; ++w2) M = fmaxf(M, ml[(w2 * 2) * 32 + lt]);
        float f[KS];
        float L = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < KS; ++w2) {
            f[w2] = __expf(ml[(w2 * 2) * 32 + lt] - M);
            L += ml[(w2 * 2 + 1) * 32 + lt] * f[w2];
        }
        const float invL = 1.0f / L;
#pragma unroll
        for (int jv = 0; jv < 16 / KS; ++jv) {
            const int r = w + jv * KS;
            float acc = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < KS; ++w2) acc += ob[(w2 * 16 + r) * 64 + l] * f[w2];
            const int d = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int i = i0[t] + lt;
            if (i < T) ob_out[(long long)d * pitch + i] = acc * invL;
        }
    }
}


constexpr int kA2QT = 3, kA2KS = 4;
void launch_attn2q(const AttnArgs& a, int batch, hipStream_t s) {
    if (a.o_mode != 0 || a.v_bstride > 0x7fffffffLL || a.o_bstride > 0x7fffffffLL) { launch_fault("attn2q: channel-major output, 31-bit strides"); return; }
    const int smem = kA2QT * (kA2KS * 64 + kA2KS * 16 * 64) * (int)sizeof(float) + kA2KS * kA2QT * 4 * 64 * 16;   // merge scratch + the waves' query fragments
    dim3 grid(((a.T + 31) / 32 + kA2QT - 1) / kA2QT, a.heads, batch);
    hipLaunchKernelGGL((attn2q_kernel<kA2KS, kA2QT>), grid, dim3(64 * kA2KS), smem, s, a.qk, a.v, a.o, (int)a.v_bstride, (int)a.o_bstride, a.pitch, a.T, a.heads, a.rows, a.scale, a.b0, a.o_mode);
}
void configure_attn2q_kernel() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2q_kernel<kA2KS, kA2QT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace said

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
#include <cstdio>
#include <cstdlib>

#include "gemm_common.h"
#include "tgemm.h"

namespace said {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TBM = 128, TBN = 128, TBK = 64, TLP = 72;   // tile, LDS row pitch in halfs

__global__ __launch_bounds__(256) void tgemm_kernel(const TGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];   // [2 buffers][A 128 x 72 | W 128 x 72]
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.x * TBM, n0 = blockIdx.y * TBN, b = blockIdx.z;
    const unsigned short* A = reinterpret_cast<const unsigned short*>(a.a) + (long long)b * a.a_bs;
    const unsigned short* W = reinterpret_cast<const unsigned short*>(a.w);
    const int nk = a.K / TBK;

    // global -> register staging: thread owns 4 chunks of 16 bytes of each operand tile (chunk c: row c >> 3, k piece c & 7)
    u32x4 ra[4], rw[4];
    long long aoff[4], woff[4];
    int loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, row = c >> 3, kp = c & 7;
        const int m = min(m0 + row, a.M - 1);   // rows past M repeat the last row (never stored)
        aoff[i] = (long long)m * a.lda + kp * 8;
        woff[i] = (long long)(n0 + row) * a.K + kp * 8;
        loff[i] = row * TLP + kp * 8;
    }
    auto gload_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const u32x4*>(A + aoff[i] + (long long)kt * TBK);
            rw[i] = *reinterpret_cast<const u32x4*>(W + woff[i] + (long long)kt * TBK);
        }
    };
    auto lds_store = [&](int buf) {
        unsigned short* pa = lds + buf * (2 * TBM * TLP);
        unsigned short* pw = pa + TBM * TLP;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(pa + loff[i]) = ra[i];
            *reinterpret_cast<u32x4*>(pw + loff[i]) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload_tile(0);
    lds_store(0);
    __syncthreads();
    const int frow = l & 31, fk = 8 * (l >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload_tile(kt + 1);
        const unsigned short* pa = lds + (kt & 1) * (2 * TBM * TLP);
        const unsigned short* pw = pa + TBM * TLP;
#pragma unroll
        for (int ks = 0; ks < TBK / 16; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(pa + (wm * 64 + i * 32 + frow) * TLP + ks * 16 + fk);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(pw + (wn * 64 + j * 32 + frow) * TLP + ks * 16 + fk);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lds_store((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: D[i = token][j = output]: lane -> output n (l & 31), register r -> token (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    const int lh = l >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (l & 31);
        const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mt = m0 + wm * 64 + i * 32;
            if (a.qk && n >= a.qk_n) {
                // v rows channel-major [c][t] for the attention kernel: 4 consecutive tokens per register quadruple
                const int c = n - a.qk_n;
                float* vp = a.vt + (long long)b * a.v_bs + (long long)c * a.v_pitch;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mt + 8 * q + 4 * lh;
                    if (m + 3 < a.M) {
                        float4 v4 = make_float4(acc[i][j][4 * q] + bias, acc[i][j][4 * q + 1] + bias, acc[i][j][4 * q + 2] + bias, acc[i][j][4 * q + 3] + bias);
                        *reinterpret_cast<float4*>(vp + m) = v4;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (m + e < a.M) vp[m + e] = acc[i][j][4 * q + e] + bias;
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                float v = acc[i][j][r] + bias;
                if (a.act == 1) v = gelu_f(v);
                if (a.res) v += a.res[(long long)b * a.res_bs + (long long)m * a.ldr + n];
                if (a.qk) {   // q / k heads token-major [b][2 heads][rows][head_dim]
                    const int h = n / a.head_dim, d = n - h * a.head_dim;
                    a.qk[(((long long)b * a.heads2 + h) * a.rows + m) * a.head_dim + d] = v;
                } else {
                    if (a.yf) a.yf[(long long)b * a.y_bs + (long long)m * a.ldy + n] = v;
                    if (a.yb) reinterpret_cast<__bf16*>(a.yb)[(long long)b * a.y_bs + (long long)m * a.ldy + n] = (__bf16)v;
                }
            }
        }
    }
}

bool tgemm_supports(const TGemmArgs& a) {
    return a.M >= 1 && a.N >= TBN && a.N % TBN == 0 && a.K >= TBK && a.K % TBK == 0 && a.lda % 8 == 0 && a.a_bs % 8 == 0 &&
           (!a.qk || (a.qk_n % 32 == 0 && a.head_dim % 32 == 0));
}
void configure_tgemm_kernel() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * TBM * TLP * 2);
}
void launch_tgemm(const TGemmArgs& a, int batch, hipStream_t s) {
    if (!tgemm_supports(a)) { fprintf(stderr, "said: tgemm shape M=%d N=%d K=%d unsupported\n", a.M, a.N, a.K); abort(); }
    dim3 grid((a.M + TBM - 1) / TBM, a.N / TBN, batch);
    hipLaunchKernelGGL(tgemm_kernel, grid, dim3(256), 2 * 2 * TBM * TLP * 2, s, a);
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
    else { fprintf(stderr, "said: ln_tm for C=%d not instantiated\n", C); abort(); }
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
    if (C != 512) { fprintf(stderr, "said: interp_ln_tm for C=%d not instantiated\n", C); abort(); }
    const float scale = (Tout > 1) ? (float)(Tin - 1) / (float)(Tout - 1) : 0.f;
    dim3 grid((Tout + 3) / 4, B);
    hipLaunchKernelGGL(interp_ln_tm_kernel<512>, grid, dim3(256), 0, s, reinterpret_cast<const unsigned short*>(src), src_bs, Tin,
                       reinterpret_cast<unsigned short*>(dst), dst_bs, Tout, scale, gamma, beta, eps);
}

}  // namespace said

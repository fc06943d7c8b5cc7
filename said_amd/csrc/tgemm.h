// tgemm.h — launch interface of the bf16 token-major GEMM and its companion kernels (tgemm.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace said {

struct TGemmArgs {
    const void* a;         // bf16 A: row m of batch b starts at a + b * a_bs + m * lda (elements), K contiguous.  A strided
    long long a_bs;        // Conv1d over token-major data is lda = stride * C, K = taps * C (overlapping rows).
    int lda;
    const void* w;         // bf16 W [N][K], K contiguous
    const float* bias;     // [N] or null
    const float* res;      // fp32 residual [b][m][ldr] or null, added after the activation
    long long res_bs;
    int ldr;
    float* yf;             // fp32 output [b][m][ldy] (nullable)
    void* yb;              // bf16 output, same indexing (nullable)
    long long y_bs;
    int ldy;
    // q/k/v split for attn.hip's operand layout (used when qk != null): outputs n < qk_n go to qk token-major per head
    // [b][heads2][rows][head_dim] (fp32), the rest to vt channel-major [b][N - qk_n][v_pitch] (fp32)
    float* qk;
    float* vt;
    long long v_bs;
    int qk_n, head_dim, rows, heads2, v_pitch;
    int M, N, K;
    int act;               // 0 none, 1 GELU (erf)
};
bool tgemm_supports(const TGemmArgs& a);
void launch_tgemm(const TGemmArgs& a, int batch, hipStream_t s);
void configure_tgemm_kernel();
void launch_cm_to_tm_bf16(const float* src, long long src_bs, int pitch, void* dst, long long dst_bs, int B, int T, int C, hipStream_t s);
void launch_ln_tm(const float* x, const float* add, float* yf, void* yb, const float* gamma, const float* beta, long long ntok, int C, float eps,
                  hipStream_t s);
void launch_interp_ln_tm(const void* src, long long src_bs, int Tin, void* dst, long long dst_bs, int Tout, int B, int C, const float* gamma,
                         const float* beta, float eps, hipStream_t s);

}  // namespace said

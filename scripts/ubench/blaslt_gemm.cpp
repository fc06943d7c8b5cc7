// micro-benchmark: what hipBLASLt delivers on the audio encoder's bf16 projections (32 clips x 499 frames), against tgemm_kernel's 124 us average
// D[M][N] (row-major) = A[M][K] (row-major bf16) . W[N][K]^T (row-major bf16) + bias  ==  column-major  D^T (N x M) = W^T-op . A
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define CB(x) do { hipblasStatus_t s_ = (x); if (s_ != HIPBLAS_STATUS_SUCCESS) { printf("hipBLASLt error %d at line %d\n", (int)s_, __LINE__); exit(1); } } while (0)

static void run(hipblasLtHandle_t h, int M, int N, int K, hipDataType dtype_d, hipblasLtEpilogue_t epi, void* ws, size_t wsz) {
    void *A, *W, *D, *bias;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&D, (size_t)M * N * 4)); CK(hipMalloc(&bias, (size_t)N * 4));
    CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(W, 0x3c, (size_t)N * K * 2)); CK(hipMemset(bias, 0, (size_t)N * 4));
    hipblasLtMatmulDesc_t desc; hipblasLtMatrixLayout_t la, lb, ld;
    CB(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof ta));
    CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof tb));
    CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof epi));
    if (epi != HIPBLASLT_EPILOGUE_DEFAULT) {
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias));
        hipDataType bt = HIP_R_32F;
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt));
    }
    CB(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, N, K));   // W stored [N][K] = column-major K x N, used transposed
    CB(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, K, M, K));   // A stored [M][K] = column-major K x M
    CB(hipblasLtMatrixLayoutCreate(&ld, dtype_d, N, M, N));      // D stored [M][N] = column-major N x M
    hipblasLtMatmulPreference_t pref; CB(hipblasLtMatmulPreferenceCreate(&pref));
    CB(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof wsz));
    hipblasLtMatmulHeuristicResult_t res[8]; int nres = 0;
    CB(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, ld, ld, pref, 8, res, &nres));
    if (nres == 0) { printf("M %d N %d K %d: no algorithm\n", M, N, K); return; }
    float alpha = 1.f, beta = 0.f;
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int a = 0; a < nres && a < 4; ++a) {
        for (int i = 0; i < 3; ++i) CB(hipblasLtMatmul(h, desc, &alpha, W, la, A, lb, &beta, D, ld, D, ld, &res[a].algo, ws, wsz, s));
        CK(hipEventRecord(e0, s));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) CB(hipblasLtMatmul(h, desc, &alpha, W, la, A, lb, &beta, D, ld, D, ld, &res[a].algo, ws, wsz, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        if (us < best) best = us;
    }
    printf("M %6d N %5d K %5d  D %s  epilogue %3d: %8.1f us  %7.1f TFLOP/s (%d algorithms offered)\n", M, N, K, dtype_d == HIP_R_32F ? "f32 " : "bf16", (int)epi, best,
           2.0 * M * N * K / best * 1e-6, nres);
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(D)); CK(hipFree(bias));
}
int main() {
    hipblasLtHandle_t h; CB(hipblasLtCreate(&h));
    size_t wsz = 64u << 20; void* ws; CK(hipMalloc(&ws, wsz));
    const int M = 32 * 499;
    for (hipDataType dt : {HIP_R_16BF, HIP_R_32F}) {
        run(h, M, 2304, 768, dt, HIPBLASLT_EPILOGUE_BIAS, ws, wsz);
        run(h, M, 768, 768, dt, HIPBLASLT_EPILOGUE_BIAS, ws, wsz);
        run(h, M, 3072, 768, dt, HIPBLASLT_EPILOGUE_GELU_BIAS, ws, wsz);
        run(h, M, 3072, 768, dt, HIPBLASLT_EPILOGUE_BIAS, ws, wsz);
        run(h, M, 768, 3072, dt, HIPBLASLT_EPILOGUE_BIAS, ws, wsz);
    }
    run(h, 499, 2304, 768, HIP_R_16BF, HIPBLASLT_EPILOGUE_BIAS, ws, wsz);
    run(h, 499, 768, 3072, HIP_R_32F, HIPBLASLT_EPILOGUE_BIAS, ws, wsz);
    return 0;
}

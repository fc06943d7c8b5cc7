// micro-benchmark / bring-up of a bf16 NT GEMM for the audio encoder's projections (Y[m][n] = sum_k A[m][k] W[n][k], bf16 in, fp32 accumulation, bf16 out):
// 256 x 256 x 64 tile, 8 waves (4 x 2: 64 rows x 128 columns each), operand tiles fetched global -> LDS DIRECTLY (buffer_load_dwordx4 ... lds: no staging registers,
// no ds_write), 16-byte chunks XOR-swizzled by the row so that the unpadded 128-byte LDS rows read conflict-free, two LDS buffers, ONE barrier per k-tile.
// Reference points (profiles/r05n_hipblaslt_encoder_shapes.txt, same shapes): hipBLASLt 20.2 / 53.9 / 62.8 / 52.7 us; tgemm_kernel<128, SB> averages 124 us over the four.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bgemm scripts/ubench/bgemm.hip && /tmp/bgemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = (BM + BN) * BK * 2;   // 64 KB per buffer

#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef KO
#define KO 0   // timing knock-outs (wrong results): 1 no global loads in the loop, 2 no barrier, 4 fragments read once per tile (no LDS reads in the k16 steps)
#endif

__global__ __launch_bounds__(512) void bgemm_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, unsigned short* __restrict__ Y, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int NT = N / BN;
    const unsigned L = blockIdx.x, xcd = L & 7u, slot = L >> 3;   // n tile fastest inside an XCD: an A tile is fetched into one L2
    const int nt = (int)(slot % (unsigned)NT), mt = (int)(slot / (unsigned)NT) * 8 + (int)xcd;
    if (mt * BM >= M) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = K / BK;
    const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, (int)((long long)M * K * 2 > 0x7fffffffLL ? 0x7fffffff : (long long)M * K * 2), 0x00020000);
    const rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, N * K * 2, 0x00020000);
    // wave-load j covers rows 8 j .. 8 j + 7 of an operand tile (8 rows x 128 bytes = 1 KB, contiguous in LDS); lane -> (row 8 j + (l >> 3), LDS chunk l & 7), which
    // holds the row's global chunk (l & 7) ^ (row & 7)
    const int lrow = l >> 3, lchunk = (l & 7) ^ lrow;   // (row & 7) == l >> 3
    int aoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = w * 4 + i;
        aoff[i] = (min(m0 + 8 * j + lrow, M - 1) * K + lchunk * 8) * 2;
        woff[i] = ((n0 + 8 * j + lrow) * K + lchunk * 8) * 2;
    }
    auto issue = [&](int kt, int buf) {
        char* base = lds + buf * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = w * 4 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(base + j * 1024), 16, aoff[i], kt * (BK * 2), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(base + BM * 128 + j * 1024), 16, woff[i], kt * (BK * 2), 0, 0);
        }
    };
    f32x16 acc0[4], acc1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.f; acc1[j][r] = 0.f; }
    const int frow = l & 31, fh = l >> 5;
    auto compute = [&](int buf) {
        const char* pa = lds + buf * TILE_BYTES;
        const char* pw = pa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int c = ((KO & 4) ? 0 : ks) * 2 + fh;
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + frow;
                fa[i] = *reinterpret_cast<const bf16x8*>(pa + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 128 + j * 32 + frow;
                fb[j] = *reinterpret_cast<const bf16x8*>(pw + row * 128 + ((c ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j], acc0[j], 0, 0, 0);
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[j], acc1[j], 0, 0, 0);
            }
        }
    };
    issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // (vmcnt(0): the tile is in LDS)
    __syncthreads();
#if VARIANT == 0
    for (int kt = 0; kt < nk; ++kt) {
#if !(KO & 1)
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);   // KO 1: no global loads (stale tiles)
#endif
        __builtin_amdgcn_sched_barrier(0);
        compute(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
#if !(KO & 2)
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();                                   // KO 2: no barrier
#endif
    }
#else
    // VARIANT 1: the fragments of k16 step s + 1 are requested BEFORE the MFMAs of step s (two register sets, the order pinned by scheduling fences): the matrix pipe
    // no longer waits out an LDS round trip per step; the loads of tile kt + 1 are unconditional (past the end: the last tile again, into the buffer nobody reads)
    bf16x8 fa[2][2], fb[2][4];
    auto rd = [&](int buf, int ks, bf16x8 (&xa)[2], bf16x8 (&xb)[4]) {
        const char* pa = lds + buf * TILE_BYTES;
        const char* pw = pa + BM * 128;
        const int c = ks * 2 + fh;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + frow;
            xa[i] = *reinterpret_cast<const bf16x8*>(pa + row * 128 + ((c ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wn * 128 + j * 32 + frow;
            xb[j] = *reinterpret_cast<const bf16x8*>(pw + row * 128 + ((c ^ (row & 7)) << 4));
        }
    };
    auto mm = [&](const bf16x8 (&xa)[2], const bf16x8 (&xb)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], xb[j], acc0[j], 0, 0, 0);
            acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1], xb[j], acc1[j], 0, 0, 0);
        }
    };
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        issue(min(kt + 1, nk - 1), buf ^ 1);
        rd(buf, 0, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        rd(buf, 1, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        rd(buf, 2, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        rd(buf, 3, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
#endif
    // plain epilogue of the bring-up: bf16 stores straight from the accumulators (lane -> column, register -> row)
    auto store = [&](const f32x16 (&acc)[4], int mrow0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * fh, n = n0 + wn * 128 + j * 32 + frow;
                if (m < M) { const __bf16 v = (__bf16)acc[j][r]; Y[(long long)m * N + n] = __builtin_bit_cast(unsigned short, v); }
            }
    };
    store(acc0, m0 + wm * 64);
    store(acc1, m0 + wm * 64 + 32);
}

__global__ void ref_kernel(const unsigned short* A, const unsigned short* W, float* Y, int M, int N, int K, int mstep) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y * mstep;
    if (n >= N || m >= M) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)__builtin_bit_cast(__bf16, A[(long long)m * K + k]) * (float)__builtin_bit_cast(__bf16, W[(long long)n * K + k]);
    Y[(long long)blockIdx.y * N + n] = s;
}

int main() {
    const int shapes[4][3] = {{15968, 2304, 768}, {15968, 768, 768}, {15968, 3072, 768}, {15968, 768, 3072}};
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bgemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_BYTES));
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; const float f = ((s >> 9) & 0x7fff) / 32768.0f - 0.5f; const __bf16 b = (__bf16)f; return __builtin_bit_cast(unsigned short, b); };
        for (auto& v : ha) v = rnd();
        for (auto& v : hw) v = rnd();
        unsigned short *da, *dw, *dy; float* dr;
        const int mstep = 97, nref = (M + mstep - 1) / mstep;
        CK(hipMalloc(&da, ha.size() * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dy, (size_t)M * N * 2)); CK(hipMalloc(&dr, (size_t)nref * N * 4));
        CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        const int MT8 = ((M + BM - 1) / BM + 7) / 8 * 8;
        dim3 grid(MT8 * (N / BN));
        hipLaunchKernelGGL(bgemm_kernel, grid, dim3(512), 2 * TILE_BYTES, 0, da, dw, dy, M, N, K);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, nref), dim3(256), 0, 0, da, dw, dr, M, N, K, mstep);
        CK(hipDeviceSynchronize());
        std::vector<unsigned short> hy((size_t)M * N); std::vector<float> hr((size_t)nref * N);
        CK(hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), dr, hr.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int i = 0; i < nref; ++i)
            for (int n = 0; n < N; ++n) {
                const float y = (float)__builtin_bit_cast(__bf16, hy[(size_t)(i * mstep) * N + n]), r = hr[(size_t)i * N + n];
                maxerr = std::max(maxerr, (double)fabsf(y - r)); maxref = std::max(maxref, (double)fabsf(r));
            }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 20;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(bgemm_kernel, grid, dim3(512), 2 * TILE_BYTES, 0, da, dw, dy, M, N, K);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(bgemm_kernel, grid, dim3(512), 2 * TILE_BYTES, 0, da, dw, dy, M, N, K);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("M %6d N %5d K %5d: %7.1f us  %7.1f TFLOP/s  (%4.1f %% of 2.5 PFLOP/s)   max |err| %.3e of max |ref| %.2f  grid %u\n", M, N, K, us, 2.0 * M * N * K / us / 1e6,
               2.0 * M * N * K / us / 1e6 / 25.0, maxerr, maxref, grid.x);
        CK(hipFree(da)); CK(hipFree(dw)); CK(hipFree(dy)); CK(hipFree(dr));
    }
    return 0;
}

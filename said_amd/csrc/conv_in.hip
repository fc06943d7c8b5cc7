// conv_in.hip — the first kernel of a denoise step: input_blocks.0 of the UNet, Conv1d(32 -> 192, k=3, pad 1)
// (openaimodel.py:458-462, 700-703) on the channel-major latents, plus the GroupNorm partial statistics of its output
// and the step-counter increment.
//
// Under classifier-free guidance the unconditional and the conditional sample of a clip are the SAME latents
// (diffusion.py:421-426 duplicates them) and this layer sees no conditioning, so its output is computed once per clip
// and written to both halves of the UNet batch.
//
// Workgroup = one 32-channel output tile x 32 tokens of one clip; 4 waves, each owning 8 of the 32 input channels for
// all three taps (12 MFMAs): its 8-channel x 32-token slice is one dwordx4 load (+ the two halo columns), parked in a
// wave-private LDS tile; weight fragments are three dwordx4 loads; the four partial tiles are summed through LDS in a
// fixed order.  The 14 scalar parameters arrive preloaded in SGPRs (no argument fetch at all).
#include "gemm_common.h"

namespace said {

typedef float f32x4c __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ f32x4c ci_bload4(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4c, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

constexpr int CI_W = 4, CI_XP = 40;

// TM (round 4, large batches in bf16 mode): the result goes out token-major in bf16, y = [sample][seg rows][192] (dims = seg | copies << 16, Cout = 192) —
// what the persistent GEMMs read — instead of channel-major fp32 followed by a transposing launch; the partial statistics are the same fp32 ones.
template <bool TM>
__global__ __launch_bounds__(64 * CI_W) void conv_in_kernel(const float* x, const float* w4, const float* bias, float* y, float* stats,
                                                            int* step_inc, int T_pitch, int dims) {
    // T_pitch = T | pitch << 16; dims = Cout | copies << 16 (copies: batch halves that receive the result)
    __shared__ __attribute__((aligned(16))) float smem[CI_W * 8 * CI_XP + CI_W * 16 * 64];
    const int tid = threadIdx.x, l = tid & 63, lt = l & 31, lh = l >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = T_pitch & 0xffff, pitch = (int)((unsigned)T_pitch >> 16);
    const int Cout = TM ? 192 : (dims & 0xffff), copies = (int)((unsigned)dims >> 16), seg = dims & 0xffff;
    const int t0 = blockIdx.x * 32, tile = blockIdx.y, b = blockIdx.z, B = gridDim.z;
    const int np = (T + 31) >> 5;
    if (step_inc && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) *step_inc += 1;

    const int sr = l >> 3, sq = l & 7;
    const rsrc_t rx = make_rsrc(x + (long long)b * 32 * pitch, 32u * (unsigned)pitch * 4u);
    const rsrc_t rw = make_rsrc(w4, (unsigned)((Cout + 31) >> 5) * 3u * 4u * 1024u);
    const f32x4c xv = ci_bload4(rx, (sr * pitch + t0 + 4 * sq) * 4, (w * 8) * pitch * 4);
    const int hrow = l >> 1, htin = (l & 1) ? (t0 + 32) : (t0 - 1);
    const bool hok = (hrow < 8) && ((unsigned)htin < (unsigned)T);
    const float halo = bload(rx, hok ? (hrow * pitch + htin) * 4 : (int)0x80000000, (w * 8) * pitch * 4);
    f32x4c wv[3];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) wv[tap] = ci_bload4(rw, l * 16, ((tile * 3 + tap) * 4 + w) * 1024);

    float* xt = smem + w * (8 * CI_XP);
    {
        f32x4c o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (t0 + 4 * sq + e < T) ? xv[e] : 0.f;
        *reinterpret_cast<f32x4c*>(xt + sr * CI_XP + 4 + 4 * sq) = o;
        if (hrow < 8) xt[hrow * CI_XP + ((l & 1) ? 36 : 3)] = hok ? halo : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* xrow = xt + lh * CI_XP + lt + 3;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[tap][j], xrow[2 * j * CI_XP + tap], acc, 0, 0, 0);

    float* red = smem + CI_W * 8 * CI_XP;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + l] = acc[r];
    __syncthreads();
    const int t = t0 + lt;
    const float cnt = (float)min(32, T - t0);
    float tmv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = w * 4 + j;
        float val = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < CI_W; ++w2) val += red[(w2 * 16 + r) * 64 + l];
        const int n = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool nok = n < Cout;
        val += nok ? bias[n] : 0.f;
        const float vv = (t < T) ? val : 0.f;
        const float mean = half32_sum(vv) * __builtin_amdgcn_rcpf(cnt);
        const float d = (t < T) ? (val - mean) : 0.f;
        const float m2 = half32_sum(d * d);
        tmv[j] = val;
        for (int k = 0; k < copies; ++k) {
            const long long bo = (long long)(b + k * B);
            if (!TM && nok && t < T) y[(bo * Cout + n) * pitch + t] = val;
            if (stats && nok && lt == 0) {
                float* so = stats + ((bo * np + blockIdx.x) * Cout + n) * 2;   // [sample][tile][channel][2]
                so[0] = mean;
                so[1] = m2;
            }
        }
    }
    if constexpr (TM) {   // this thread's four values are channels tile * 32 + 8 w + 4 lh + (0 .. 3) of token t: one 8-byte store per copy
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x4 ov = {(__bf16)tmv[0], (__bf16)tmv[1], (__bf16)tmv[2], (__bf16)tmv[3]};
        if (t < T)
            for (int k = 0; k < copies; ++k)
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(y) + ((long long)(b + k * B) * seg + t) * 192 + tile * 32 + 8 * w + 4 * lh) = ov;
    }
}

bool conv_in_supports(int Cin, int Cout, int taps, int T, int pitch, int copies) {
    return Cin == 32 && taps == 3 && Cout <= 0xffff && T <= 0xffff && pitch <= 0xffff && copies >= 1 && copies <= 0xffff;
}
void launch_conv_in(const float* x, const float* w4, const float* bias, float* y, float* stats, int* step_inc, int B, int copies, int T,
                    int pitch, int Cout, hipStream_t s) {
    dim3 grid((T + 31) / 32, (Cout + 31) / 32, B);
    hipLaunchKernelGGL(conv_in_kernel<false>, grid, dim3(64 * CI_W), 0, s, x, w4, bias, y, stats, step_inc, T | (pitch << 16), Cout | (copies << 16));
}
// token-major bf16 result [sample][seg][192] (Cout = 192, seg <= 0xffff)
void launch_conv_in_tm(const float* x, const float* w4, const float* bias, void* y_tm, int seg, float* stats, int* step_inc, int B, int copies, int T,
                       int pitch, hipStream_t s) {
    dim3 grid((T + 31) / 32, 6, B);
    hipLaunchKernelGGL(conv_in_kernel<true>, grid, dim3(64 * CI_W), 0, s, x, w4, bias, reinterpret_cast<float*>(y_tm), stats, step_inc, T | (pitch << 16),
                       seg | (copies << 16));
}

}  // namespace said

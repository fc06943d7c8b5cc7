// misc.hip — layout changes, timestep embedding, the scheduler update, and the small
// audio-encoder kernels (conv0, per-row GroupNorm+GELU, linear interpolation, LayerNorm).
// All are HBM/latency-bound elementwise or row-reduction kernels: coalesced along t.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "sched_math.h"

namespace said {

// ------------------------------------------------------------------------------------------
// spin_kernel: one wave busy for `ticks` of the 100 MHz wall clock — the probe behind the clip groups' stream pool (engine.cpp:
// two streams run side by side only if they sit on different hardware queues, and that can only be found out by timing)
// ------------------------------------------------------------------------------------------
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void launch_spin(long long ticks, hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks); }

// ------------------------------------------------------------------------------------------
// launch_fault (kernels.h)
// ------------------------------------------------------------------------------------------
static thread_local char g_fault[256];
static thread_local bool g_fault_set = false;
void launch_fault(const char* fmt, ...) {
    if (g_fault_set) return;   // keep the first one
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_fault, sizeof g_fault, fmt, ap);
    va_end(ap);
    g_fault_set = true;
}
const char* launch_fault_peek() { return g_fault_set ? g_fault : nullptr; }
void launch_fault_clear() { g_fault_set = false; }


// ------------------------------------------------------------------------------------------
// token-major (B,T,C) <-> channel-major [B][C][pitch]
// ------------------------------------------------------------------------------------------
__global__ void tm_to_cm_kernel(const float* __restrict__ src, float* __restrict__ dst, int T, int C, int pitch,
                                long long dst_bstride) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        tile[r][tx] = (t < T && c < C) ? src[((long long)b * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        if (c < C && t < T) dst[(long long)b * dst_bstride + (long long)c * pitch + t] = tile[tx][r];
    }
}
__global__ void cm_to_tm_kernel(const float* __restrict__ src, float* __restrict__ dst, int T, int C, int pitch,
                                long long src_bstride) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        tile[r][tx] = (t < T && c < C) ? src[(long long)b * src_bstride + (long long)c * pitch + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        if (c < C && t < T) dst[((long long)b * T + t) * C + c] = tile[tx][r];
    }
}
void launch_tm_to_cm(const float* src, float* dst, int B, int T, int C, int pitch, long long dst_bstride, hipStream_t s) {
    dim3 grid((T + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(tm_to_cm_kernel, grid, dim3(256), 0, s, src, dst, T, C, pitch, dst_bstride);
}
void launch_cm_to_tm(const float* src, float* dst, int B, int T, int C, int pitch, long long src_bstride, hipStream_t s) {
    dim3 grid((T + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(cm_to_tm_kernel, grid, dim3(256), 0, s, src, dst, T, C, pitch, src_bstride);
}

// VAE encoder input: window w = (L, C) token-major rows starting at src + w * win_stride -> dst[w][c][t]
__global__ void windows_to_cm_kernel(const float* __restrict__ src, long long win_stride, float* __restrict__ dst, int L, int C, int pitch,
                                     long long dst_bstride) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* sw = src + (long long)b * win_stride;
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        tile[r][tx] = (t < L && c < C) ? sw[(long long)t * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        if (c < C && t < L) dst[(long long)b * dst_bstride + (long long)c * pitch + t] = tile[tx][r];
    }
}
void launch_windows_to_cm(const float* src, long long win_stride, float* dst, int n, int L, int C, int pitch, long long dst_bstride, hipStream_t s) {
    dim3 grid((L + 31) / 32, (C + 31) / 32, n);
    hipLaunchKernelGGL(windows_to_cm_kernel, grid, dim3(256), 0, s, src, win_stride, dst, L, C, pitch, dst_bstride);
}
// nn.Flatten((n, C, T)) as the feature-major FC operand: dst[c * T + t][w] = src[w][c][t]; w is the contiguous index
__global__ void flatten_cm_kernel(const float* __restrict__ src, long long src_bstride, int src_pitch, float* __restrict__ dst, int dst_pitch,
                                  int n, int C, int T) {
    __shared__ float tile[32][33];
    const int c = blockIdx.z, t0 = blockIdx.y * 32, w0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {   // rows = windows, columns = t (contiguous in src)
        const int w = w0 + r, t = t0 + tx;
        tile[r][tx] = (w < n && t < T) ? src[(long long)w * src_bstride + (long long)c * src_pitch + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {   // rows = t, columns = windows (contiguous in dst)
        const int t = t0 + r, w = w0 + tx;
        if (t < T && w < n) dst[((long long)c * T + t) * dst_pitch + w] = tile[tx][r];
    }
}
void launch_flatten_cm(const float* src, long long src_bstride, int src_pitch, float* dst, int dst_pitch, int n, int C, int T, hipStream_t s) {
    dim3 grid((n + 31) / 32, (T + 31) / 32, C);
    hipLaunchKernelGGL(flatten_cm_kernel, grid, dim3(256), 0, s, src, src_bstride, src_pitch, dst, dst_pitch, n, C, T);
}

__global__ void fill_cm_vec_kernel(const float* __restrict__ vec, float* __restrict__ dst, int T, int pitch,
                                   long long dst_bstride) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) dst[(long long)b * dst_bstride + (long long)c * pitch + t] = vec[c];
}
void launch_fill_cm_vec(const float* vec, float* dst, int B, int T, int C, int pitch, long long dst_bstride, hipStream_t s) {
    dim3 grid((T + 255) / 256, C, B);
    hipLaunchKernelGGL(fill_cm_vec_kernel, grid, dim3(256), 0, s, vec, dst, T, pitch, dst_bstride);
}

// ------------------------------------------------------------------------------------------
// timestep_embedding (ldm/util.py:66-90): dst[k][r] = cos(t_r f_k), dst[half+k][r] = sin(t_r f_k)
// freqs are supplied by the host (computed with the reference's own fp32 op order).
// ------------------------------------------------------------------------------------------
__global__ void temb_kernel(const long long* __restrict__ ts, const float* __restrict__ freqs, float* __restrict__ dst,
                            int n, int half, int pitch) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (r >= n) return;
    const float arg = __fmul_rn((float)ts[r], freqs[k]);
    dst[(long long)k * pitch + r] = cosf(arg);
    dst[(long long)(k + half) * pitch + r] = sinf(arg);
}
void launch_timestep_embedding(const long long* timesteps_dev, const float* freqs_dev, float* dst, int n, int dim, int pitch, hipStream_t s) {
    dim3 grid((n + 63) / 64, dim / 2);
    hipLaunchKernelGGL(temb_kernel, grid, dim3(64), 0, s, timesteps_dev, freqs_dev, dst, n, dim / 2, pitch);
}

__global__ void step_advance_kernel(int* p) { *p = *p + 1; }
void launch_step_advance(int* step_ptr, hipStream_t s) { hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, step_ptr); }

// Welford partials of e_c and eps_cfg for rescale_noise_cfg: part[b][which][blk] = (n, mean, M2)
__global__ void rescale_partials_kernel(const SchedArgs a, float* __restrict__ part) {
    __shared__ float red[2][4][3];
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int total = a.C * a.T;
    const int per = (total + nblk - 1) / nblk;
    const int lo = blk * per, hi = min(total, lo + per);
    const float* ec = a.eps + (long long)(a.B + b) * a.eps_bstride;
    const float* eu = a.eps + (long long)b * a.eps_bstride;
    float n = 0.f, m0 = 0.f, q0 = 0.f, m1 = 0.f, q1 = 0.f;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int c = i / a.T, t = i % a.T;
        const float vc = ec[(long long)c * a.pitch + t];
        const float vu = eu[(long long)c * a.pitch + t];
        const float vg = cfg_combine(vc, vu, a.guidance_scale);
        n += 1.f;
        float d = vc - m0; m0 += d / n; q0 += d * (vc - m0);
        d = vg - m1; m1 += d / n; q1 += d * (vg - m1);
    }
    // merge lanes then waves (Chan)
    auto merge = [](float& na, float& ma, float& qa, float nb, float mb, float qb) {
        const float nt = na + nb;
        if (nt > 0.f) {
            const float d = mb - ma;
            ma += d * nb / nt;
            qa += qb + d * d * na * nb / nt;
        }
        na = nt;
    };
    float n1 = n;
    for (int o = 32; o >= 1; o >>= 1) {
        const float nb_ = __shfl_xor(n, o), mb0 = __shfl_xor(m0, o), qb0 = __shfl_xor(q0, o);
        const float mb1 = __shfl_xor(m1, o), qb1 = __shfl_xor(q1, o);
        float na = n;
        merge(na, m0, q0, nb_, mb0, qb0);
        merge(n1, m1, q1, nb_, mb1, qb1);
        n = na;
        n1 = n;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][w][0] = n; red[0][w][1] = m0; red[0][w][2] = q0;
        red[1][w][0] = n; red[1][w][1] = m1; red[1][w][2] = q1;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int k = threadIdx.x;
        float na = red[k][0][0], ma = red[k][0][1], qa = red[k][0][2];
        for (int w2 = 1; w2 < (int)(blockDim.x >> 6); ++w2) merge(na, ma, qa, red[k][w2][0], red[k][w2][1], red[k][w2][2]);
        float* o = part + (((long long)b * 2 + k) * nblk + blk) * 3;
        o[0] = na; o[1] = ma; o[2] = qa;
    }
}
void launch_rescale_partials(const SchedArgs& a, float* part_out, hipStream_t s) {
    dim3 grid(a.rescale_nblk, a.B);
    hipLaunchKernelGGL(rescale_partials_kernel, grid, dim3(256), 0, s, a, part_out);
}

__global__ void sched_step_kernel(const SchedArgs a) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int step = *a.step_ptr;
    const float* cf = a.coef + step * 8;
    float ratio = 1.f;
    if (a.guidance_rescale > 0.f) {  // uniform branch
        // combine partials of sample b (tiny: rescale_nblk entries each)
        float nn[2], mm[2], qq[2];
        for (int k = 0; k < 2; ++k) {
            const float* p = a.rescale_part + ((long long)b * 2 + k) * a.rescale_nblk * 3;
            float na = p[0], ma = p[1], qa = p[2];
            for (int i = 1; i < a.rescale_nblk; ++i) {
                const float nb_ = p[3 * i], mb = p[3 * i + 1], qb = p[3 * i + 2];
                const float nt = na + nb_;
                const float d = mb - ma;
                ma += d * nb_ / nt;
                qa += qb + d * d * na * nb_ / nt;
                na = nt;
            }
            nn[k] = na; mm[k] = ma; qq[k] = qa;
        }
        const float std_text = sqrtf(qq[0] / (nn[0] - 1.f));
        const float std_cfg = sqrtf(qq[1] / (nn[1] - 1.f));
        ratio = std_text / std_cfg;
        (void)mm;
    }
    if (t >= a.T) return;
    const long long off = (long long)c * a.pitch + t;
    float e;
    if (a.cfg) {
        const float eu = a.eps[(long long)b * a.eps_bstride + off];
        const float ec = a.eps[(long long)(a.B + b) * a.eps_bstride + off];
        e = cfg_combine(ec, eu, a.guidance_scale);
        if (a.guidance_rescale > 0.f) {
            // rescale_noise_cfg: phi * (cfg * std_text/std_cfg) + (1 - phi) * cfg
            const float resc = __fmul_rn(e, ratio);
            e = __fadd_rn(__fmul_rn(a.guidance_rescale, resc), __fmul_rn(__fsub_rn(1.0f, a.guidance_rescale), e));
        }
    } else {
        e = a.eps[(long long)b * a.eps_bstride + off];
    }
    float* xp = a.x + (long long)b * a.x_bstride + off;
    const float x = *xp;
    note_nonfinite(e, a.status, step);
    if (a.inter) a.inter[(((long long)step * a.B + b) * a.T + t) * a.C + c] = x / a.latent_scale;
    float prev = ddim_prev(e, x, cf, a.prediction_type);
    if (a.step_noise) {
        const float nz = a.step_noise[((long long)step * a.B + b) * a.x_bstride + off];
        prev = __fadd_rn(prev, __fmul_rn(cf[4], nz));
    } else if (a.noise_seed) {
        prev = __fadd_rn(prev, __fmul_rn(cf[4], philox_normal(a.noise_seed[0], a.noise_seed[1], (unsigned)step, a.noise_elem0 + (unsigned)((b * a.T + t) * a.C + c))));
    }
    if (a.mask) {
        const long long o2 = (long long)b * a.x_bstride + off;
        prev = mask_blend(prev, a.init[o2], a.edit_noise[o2], a.mask[o2], cf);
    }
    *xp = prev;
}
void launch_sched_step(const SchedArgs& a, hipStream_t s) {
    dim3 grid((a.T + 63) / 64, a.C, a.B);
    hipLaunchKernelGGL(sched_step_kernel, grid, dim3(64), 0, s, a);
}

__global__ void philox_normal_kernel(const unsigned* __restrict__ seed, int step0, long long n_per_step, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = philox_normal(seed[0], seed[1], (unsigned)(step0 + (int)(i / n_per_step)), (unsigned)(i % n_per_step));
}
void launch_philox_normal(const unsigned* seed_dev, int step0, int nsteps, long long n_per_step, float* out, hipStream_t s) {
    const long long n = n_per_step * nsteps;
    if (n > 0) hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed_dev, step0, n_per_step, n, out);
}

__global__ void ddim_flat_kernel(const float* __restrict__ eps, const float* __restrict__ eps_u, float gs,
                                 const float* __restrict__ x, const float* __restrict__ cf, int pred,
                                 const float* __restrict__ noise, const float* __restrict__ init,
                                 const float* __restrict__ enoise, const float* __restrict__ mask, float* __restrict__ out,
                                 long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float e = eps[i];
    if (eps_u) e = cfg_combine(e, eps_u[i], gs);
    float prev = ddim_prev(e, x[i], cf, pred);
    if (noise) prev = __fadd_rn(prev, __fmul_rn(cf[4], noise[i]));
    if (mask) prev = mask_blend(prev, init[i], enoise[i], mask[i], cf);
    out[i] = prev;
}
void launch_ddim_flat(const float* eps, const float* eps_u, float gs, const float* x, const float* coef_dev, int pred,
                      const float* noise, const float* init, const float* edit_noise, const float* mask, float* out,
                      long long n, hipStream_t s) {
    hipLaunchKernelGGL(ddim_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, eps, eps_u, gs, x, coef_dev, pred,
                       noise, init, edit_noise, mask, out, n);
}

__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ c,
                             const float* __restrict__ y, float* __restrict__ out, long long n) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long o = (long long)b * n + i;
    float v = __fmul_rn(a[b], x[o]);
    if (y) v = __fadd_rn(v, __fmul_rn(c[b], y[o]));
    out[o] = v;
}
void launch_axpby(const float* a_dev, const float* x, const float* c_dev, const float* y, float* out, int B, long long n,
                  hipStream_t s) {
    dim3 grid((unsigned)((n + 255) / 256), B);
    hipLaunchKernelGGL(axpby_kernel, grid, dim3(256), 0, s, a_dev, x, c_dev, y, out, n);
}

__global__ void finish_kernel(const float* __restrict__ x, long long x_bstride, int pitch, int T, int C, float latent_scale,
                              float* __restrict__ latents_tm, float* __restrict__ result_tm, int* __restrict__ status) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        tile[r][tx] = (t < T && c < C) ? x[(long long)b * x_bstride + (long long)c * pitch + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        if (c < C && t < T) {
            const float v = tile[tx][r];
            const long long o = ((long long)b * T + t) * C + c;
            if (status && !(fabsf(v) <= 3.4028234663852886e38f)) status[1] = 1;
            if (latents_tm) latents_tm[o] = v;
            const float q = v / latent_scale;
            if (result_tm) result_tm[o] = (q != q) ? q : fminf(fmaxf(q, 0.f), 1.f);  // diffusion.py:470 (torch.clamp keeps a NaN)
        }
    }
}
void launch_finish(const float* x_cm, long long x_bstride, int pitch, int B, int T, int C, float latent_scale,
                   float* latents_tm, float* result_tm, hipStream_t s, int* status) {
    dim3 grid((T + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(finish_kernel, grid, dim3(256), 0, s, x_cm, x_bstride, pitch, T, C, latent_scale, latents_tm, result_tm, status);
}
__global__ void nonfinite_check_kernel(const float* __restrict__ x, long long x_bstride, int pitch, int T, int* __restrict__ status) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T && !(fabsf(x[(long long)blockIdx.z * x_bstride + (long long)blockIdx.y * pitch + t]) <= 3.4028234663852886e38f)) status[1] = 1;
}
void launch_nonfinite_check(const float* x_cm, long long x_bstride, int pitch, int B, int T, int C, int* status, hipStream_t s) {
    hipLaunchKernelGGL(nonfinite_check_kernel, dim3((T + 255) / 256, C, B), dim3(256), 0, s, x_cm, x_bstride, pitch, T, status);
}

// ------------------------------------------------------------------------------------------
// Audio encoder helpers
// ------------------------------------------------------------------------------------------
// conv0: y[b][c][t] = sum_k w[c][k] * wav[b][t*S + k]   (K <= 16; 8 channels per thread)
__global__ void conv0_kernel(const float* __restrict__ wav, const float* __restrict__ w, float* __restrict__ y, int Ta, int C,
                             int K, int S, int Tout, int pitch, long long y_bstride) {
    const int b = blockIdx.z, c0 = blockIdx.y * 8;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tout) return;
    float xv[16];
    const float* xp = wav + (long long)b * Ta + (long long)t * S;
#pragma unroll
    for (int k = 0; k < 16; ++k) xv[k] = (k < K) ? xp[k] : 0.f;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        const int c = c0 + cc;
        if (c < C) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < K) acc = fmaf(w[c * K + k], xv[k], acc);
            y[(long long)b * y_bstride + (long long)c * pitch + t] = acc;
        }
    }
}
void launch_conv0(const float* wav, const float* w, float* y, int B, int Ta, int C, int K, int S, int Tout, int pitch,
                  long long y_bstride, hipStream_t s) {
    if (K > 16) { launch_fault("conv0 kernel size %d > 16 unsupported", K); return; }
    dim3 grid((Tout + 255) / 256, (C + 7) / 8, B);
    hipLaunchKernelGGL(conv0_kernel, grid, dim3(256), 0, s, wav, w, y, Ta, C, K, S, Tout, pitch, y_bstride);
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// GroupNorm with one channel per group (Wav2Vec2GroupNormConvLayer) + exact GELU, in place.
__global__ void rownorm_gelu_kernel(float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    int T, int pitch, long long bstride, float eps) {
    __shared__ float sh[4];
    const int c = blockIdx.x, b = blockIdx.y;
    float* row = y + (long long)b * bstride + (long long)c * pitch;
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) s += row[t];
    const float mean = block_sum_256(s, sh) / (float)T;
    float q = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) {
        const float d = row[t] - mean;
        q = fmaf(d, d, q);
    }
    const float var = block_sum_256(q, sh) / (float)T;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float g = gamma[c] * rstd, bb = beta[c] - mean * gamma[c] * rstd;
    for (int t = threadIdx.x; t < T; t += 256) row[t] = gelu_exact(fmaf(row[t], g, bb));
}
void launch_rownorm_gelu(float* y, const float* gamma, const float* beta, int rows_per_batch, int B, int T, int pitch,
                         long long bstride, float eps, hipStream_t s) {
    hipLaunchKernelGGL(rownorm_gelu_kernel, dim3(rows_per_batch, B), dim3(256), 0, s, y, gamma, beta, T, pitch, bstride, eps);
}

// ------------------------------------------------------------------------------------------
// bf16 encoder front end: conv0 + per-channel GroupNorm + GELU -> token-major bf16, WITHOUT storing the fp32 conv0 activation
// (65 MB per 10 s clip: written, read three times by the normalisation, written, read and transposed = 9.4 GB per 32 clips).
// conv0 is 10 multiply-adds per output, so it is computed twice instead: once for the per-channel statistics (nothing stored
// but (sum, sum of squares) per 256-frame tile), once more to normalise, GELU and write the 1 KB bf16 row of every frame.
//   stats: workgroup = (256-frame tile, clip), thread = channels tid and tid + 256
//   coef : workgroup = clip, thread = channel: Chan's merge of the tile moments -> (gamma * rstd, beta - mean * gamma * rstd)
//   apply: workgroup = (256-frame tile, clip), wave = every fourth frame, lane = 8 consecutive channels = one 16-byte store
// Same multiply-add order per output as conv0_kernel (k ascending); K = 10, S = 5 (wav2vec2-base).
// ------------------------------------------------------------------------------------------
constexpr int C0_FT = 256, C0_K = 10, C0_S = 5, C0_C = 512;

__global__ __launch_bounds__(256) void conv0_stats_kernel(const float* __restrict__ wav, const float* __restrict__ w, float* __restrict__ part,
                                                          int Ta, int Tout, int ntile) {
    __shared__ float seg[C0_FT * C0_S + C0_K];
    const int tid = threadIdx.x, tile = blockIdx.x, b = blockIdx.y;
    const int f0 = tile * C0_FT, nf = min(C0_FT, Tout - f0);
    const float* xp = wav + (long long)b * Ta + (long long)f0 * C0_S;
    const int nsamp = min(nf * C0_S + C0_K - C0_S, Ta - f0 * C0_S);
    for (int i = tid; i < C0_FT * C0_S + C0_K; i += 256) seg[i] = i < nsamp ? xp[i] : 0.f;
    float w0[C0_K], w1[C0_K];
#pragma unroll
    for (int k = 0; k < C0_K; ++k) { w0[k] = w[tid * C0_K + k]; w1[k] = w[(tid + 256) * C0_K + k]; }
    __syncthreads();
    float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
    float x[C0_K];
#pragma unroll
    for (int k = 0; k < C0_S; ++k) x[C0_S + k] = seg[k];
    for (int f = 0; f < nf; ++f) {
#pragma unroll
        for (int k = 0; k < C0_S; ++k) { x[k] = x[C0_S + k]; x[C0_S + k] = seg[f * C0_S + C0_S + k]; }
        float ya = 0.f, yb = 0.f;
#pragma unroll
        for (int k = 0; k < C0_K; ++k) { ya = fmaf(w0[k], x[k], ya); yb = fmaf(w1[k], x[k], yb); }
        a1 += ya; a2 = fmaf(ya, ya, a2);
        b1 += yb; b2 = fmaf(yb, yb, b2);
    }
    float2* pp = reinterpret_cast<float2*>(part) + ((long long)b * ntile + tile) * C0_C;
    pp[tid] = make_float2(a1, a2);
    pp[tid + 256] = make_float2(b1, b2);
}

__global__ __launch_bounds__(512) void conv0_coef_kernel(const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ coef, int Tout, int ntile, float eps) {
    const int c = threadIdx.x, b = blockIdx.x;
    const float2* pp = reinterpret_cast<const float2*>(part) + (long long)b * ntile * C0_C + c;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int t = 0; t < ntile; ++t) {
        const float2 v = pp[(long long)t * C0_C];
        const float nb = (float)min(C0_FT, Tout - t * C0_FT);
        const float mb = v.x / nb, m2b = fmaxf(v.y - v.x * mb, 0.f);
        const float nn = n + nb, d = mb - mean;
        mean += d * (nb / nn);
        m2 += m2b + d * d * (n * nb / nn);
        n = nn;
    }
    const float rstd = 1.0f / sqrtf(m2 / n + eps);
    coef[((long long)b * C0_C + c) * 2] = gamma[c] * rstd;
    coef[((long long)b * C0_C + c) * 2 + 1] = beta[c] - mean * gamma[c] * rstd;
}

__global__ __launch_bounds__(256) void conv0_apply_tm_kernel(const float* __restrict__ wav, const float* __restrict__ w, const float* __restrict__ coef,
                                                             unsigned short* __restrict__ dst, int Ta, int Tout) {
    __shared__ float seg[C0_FT * C0_S + C0_K];
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, tile = blockIdx.x, b = blockIdx.y;
    const int f0 = tile * C0_FT, nf = min(C0_FT, Tout - f0);
    const float* xp = wav + (long long)b * Ta + (long long)f0 * C0_S;
    const int nsamp = min(nf * C0_S + C0_K - C0_S, Ta - f0 * C0_S);
    for (int i = tid; i < C0_FT * C0_S + C0_K; i += 256) seg[i] = i < nsamp ? xp[i] : 0.f;
    float wr[8][C0_K], g[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int k = 0; k < C0_K; ++k) wr[j][k] = w[(8 * l + j) * C0_K + k];
        g[j] = coef[((long long)b * C0_C + 8 * l + j) * 2];
        bb[j] = coef[((long long)b * C0_C + 8 * l + j) * 2 + 1];
    }
    __syncthreads();
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    for (int f = wv; f < nf; f += 4) {
        float x[C0_K];
#pragma unroll
        for (int k = 0; k < C0_K; ++k) x[k] = seg[f * C0_S + k];
        unsigned int o[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y = 0.f;
#pragma unroll
            for (int k = 0; k < C0_K; ++k) y = fmaf(wr[j][k], x[k], y);
            const float v = gelu_exact(fmaf(y, g[j], bb[j]));
            const unsigned int u = __builtin_bit_cast(unsigned int, v);
            const unsigned int r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;     // round to nearest even (finite values)
            if (j & 1) o[j >> 1] |= r << 16; else o[j >> 1] = r;
        }
        u32x4 ov = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<u32x4*>(dst + ((long long)b * Tout + f0 + f) * C0_C + 8 * l) = ov;
    }
}

bool launch_conv0_gn_gelu_tm_bf16(const float* wav, const float* w, const float* gamma, const float* beta, float* scratch, void* dst, int B, int Ta,
                                  int C, int K, int S, int Tout, float eps, hipStream_t s) {
    if (C != C0_C || K != C0_K || S != C0_S || Tout < 1) return false;
    const int ntile = (Tout + C0_FT - 1) / C0_FT;
    float* part = scratch;                                   // [B][ntile][512][2]
    float* coef = scratch + (size_t)B * ntile * C0_C * 2;    // [B][512][2]
    hipLaunchKernelGGL(conv0_stats_kernel, dim3(ntile, B), dim3(256), 0, s, wav, w, part, Ta, Tout, ntile);
    hipLaunchKernelGGL(conv0_coef_kernel, dim3(B), dim3(512), 0, s, part, gamma, beta, coef, Tout, ntile, eps);
    hipLaunchKernelGGL(conv0_apply_tm_kernel, dim3(ntile, B), dim3(256), 0, s, wav, w, coef, reinterpret_cast<unsigned short*>(dst), Ta, Tout);
    return true;
}

// F.interpolate(mode="linear", align_corners=True) along t (wav2vec2.py:41-44)
__global__ void interp_linear_kernel(const float* __restrict__ src, float* __restrict__ dst, int Tin, int Tout, int src_pitch,
                                     int dst_pitch, long long src_bstride, long long dst_bstride, float scale) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Tout) return;
    const float pos = __fmul_rn(scale, (float)i);
    int i0 = (int)pos;
    i0 = min(i0, Tin - 1);
    const int i1 = i0 + ((i0 < Tin - 1) ? 1 : 0);
    const float l1 = __fsub_rn(pos, (float)i0), l0 = __fsub_rn(1.0f, l1);
    const float* sp = src + (long long)b * src_bstride + (long long)c * src_pitch;
    dst[(long long)b * dst_bstride + (long long)c * dst_pitch + i] = __fadd_rn(__fmul_rn(l0, sp[i0]), __fmul_rn(l1, sp[i1]));
}
void launch_interp_linear(const float* src, float* dst, int B, int C, int Tin, int Tout, int src_pitch, int dst_pitch,
                          long long src_bstride, long long dst_bstride, hipStream_t s) {
    const float scale = (Tout > 1) ? (float)(Tin - 1) / (float)(Tout - 1) : 0.f;
    dim3 grid((Tout + 63) / 64, C, B);
    hipLaunchKernelGGL(interp_linear_kernel, grid, dim3(64), 0, s, src, dst, Tin, Tout, src_pitch, dst_pitch, src_bstride,
                       dst_bstride, scale);
}

// y[b][c][t] = LN_c(x[b][c][t] + add[b][c][t]) for a channel-major tensor (Wav2Vec2 post-LN encoder layers).
// One workgroup owns 16 tokens x all C channels: the tile is read ONCE (64-byte row segments, 4 rows per wave
// instruction), kept in LDS, reduced per token with a shifted two-pass variance (exact mean first), and written once.
// The first version walked the tensor three times with one 128-byte row per thread step and 19 workgroups at T = 600:
// 59 us per call, 25 calls per clip (rocprofv3, profiles/r01c_kernel_trace_variants.txt).
constexpr int LN_TT = 16;
__global__ __launch_bounds__(256) void layernorm_cm_kernel(const float* __restrict__ x, const float* __restrict__ add, float* __restrict__ y,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int C, int T, int pitch,
                                                           long long bstride, float eps) {
    extern __shared__ float tile[];            // [C][LN_TT + 1]
    __shared__ float red[16][LN_TT];
    __shared__ float stat[2][LN_TT];
    const int b = blockIdx.y, t0 = blockIdx.x * LN_TT;
    const int tx = threadIdx.x & (LN_TT - 1), cy = threadIdx.x / LN_TT;   // 16 channel rows per pass
    const int t = min(t0 + tx, T - 1);
    const float* xb = x + (long long)b * bstride + t;
    const float* ab = add ? add + (long long)b * bstride + t : nullptr;
    float s = 0.f;
    for (int c = cy; c < C; c += 16) {
        float v = xb[(long long)c * pitch];
        if (ab) v += ab[(long long)c * pitch];
        tile[c * (LN_TT + 1) + tx] = v;
        s += v;
    }
    red[cy][tx] = s;
    __syncthreads();
    if (threadIdx.x < LN_TT) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) tot += red[g][threadIdx.x];
        stat[0][threadIdx.x] = tot / (float)C;
    }
    __syncthreads();
    const float mean = stat[0][tx];
    float q = 0.f;
    for (int c = cy; c < C; c += 16) {
        const float d = tile[c * (LN_TT + 1) + tx] - mean;
        q = fmaf(d, d, q);
    }
    red[cy][tx] = q;
    __syncthreads();
    if (threadIdx.x < LN_TT) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) tot += red[g][threadIdx.x];
        stat[1][threadIdx.x] = 1.0f / sqrtf(tot / (float)C + eps);
    }
    __syncthreads();
    const float rstd = stat[1][tx];
    if (t0 + tx < T) {
        float* yb = y + (long long)b * bstride + t0 + tx;
        for (int c = cy; c < C; c += 16) yb[(long long)c * pitch] = fmaf((tile[c * (LN_TT + 1) + tx] - mean) * rstd, gamma[c], beta[c]);
    }
}
void launch_layernorm_cm(const float* x, const float* add, float* y, const float* gamma, const float* beta, int B, int C,
                         int T, int pitch, long long bstride, float eps, hipStream_t s) {
    dim3 grid((T + LN_TT - 1) / LN_TT, B);
    const size_t smem = (size_t)C * (LN_TT + 1) * sizeof(float);   // 52 KB at C = 768
    hipLaunchKernelGGL(layernorm_cm_kernel, grid, dim3(256), smem, s, x, add, y, gamma, beta, C, T, pitch, bstride, eps);
}


// ---- banded cross-attention for alignment windows wider than the fused epilogue's eight keys (S >> T through SAID.forward; ldm/attention.py:170-191
// handles any (T, S)).  Never reached by SAID.inference (audio features are interpolated to one token per frame: windows of <= 3 keys), so
// this path is written for correctness, not speed: one lane = one query of one head, q / o in place in a channel-major [C][pitch] tile,
// keys lo[t] .. hi[t] - 1 walked with an online softmax (scale after q k^T as attention.py:101, masked keys simply absent: their
// softmax weight in the reference is exp(-finfo.max - max) = 0).
__global__ __launch_bounds__(64) void band_wide_kernel(float* qo, long long qo_bstride, int pitch, const float* kk, const float* vv, long long kv_bstride,
                                                       int kv_pitch, const int* lo, const int* hi, int T, float scale) {
    const int t = blockIdx.x * 64 + threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    float* q = qo + (long long)b * qo_bstride + (long long)(h * 32) * pitch + t;
    const float* kb = kk + (long long)b * kv_bstride + (long long)(h * 32) * kv_pitch;
    const float* vb = vv + (long long)b * kv_bstride + (long long)(h * 32) * kv_pitch;
    float qv[32], o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) { qv[d] = q[(long long)d * pitch]; o[d] = 0.f; }
    float m = -3.0e38f, lsum = 0.f;
    for (int j = lo[t]; j < hi[t]; ++j) {
        float sdot = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) sdot = fmaf(qv[d], kb[(long long)d * kv_pitch + j], sdot);
        sdot *= scale;
        const float mn = fmaxf(m, sdot);
        const float f = __expf(m - mn), pj = __expf(sdot - mn);
        lsum = lsum * f + pj;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = fmaf(pj, vb[(long long)d * kv_pitch + j], o[d] * f);
        m = mn;
    }
    const float inv = 1.0f / lsum;
#pragma unroll
    for (int d = 0; d < 32; ++d) q[(long long)d * pitch] = o[d] * inv;
}
void launch_band_wide(float* qo, long long qo_bstride, int pitch, const float* k, const float* v, long long kv_bstride, int kv_pitch, const int* lo,
                      const int* hi, int T, int heads, int batch, float scale, hipStream_t s) {
    hipLaunchKernelGGL(band_wide_kernel, dim3((T + 63) / 64, heads, batch), dim3(64), 0, s, qo, qo_bstride, pitch, k, v, kv_bstride, kv_pitch, lo, hi, T, scale);
}

}  // namespace said

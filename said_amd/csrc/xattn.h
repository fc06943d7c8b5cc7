// xattn.h — launch interface of the fused token-local chain kernel (xattn.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace said {

constexpr int XA_KW = 24;   // key/value window columns staged per 16-query tile (the host checks that this covers the band)

struct XAttnArgs {
    const float* o;          // attn1 output, channel-major [n][192][pitch]
    long long o_bs;
    const float* res;        // SpatialTransformer input x_in (before its GroupNorm) [n][192][pitch]
    long long res_bs;
    const float* gn_part;    // GroupNorm partial statistics of x_in [n][192][gn_nparts][2]
    long long gn_part_bs;
    const float* gn_gamma;   // SpatialTransformer.norm (eps 1e-6)
    const float* gn_beta;
    const float* w1;         // attn1.to_out.0 weight, pack16
    const float* b1;
    const float* ln_g;       // norm2
    const float* ln_b;
    const float* wq;         // attn2.to_q weight, pack16 (no bias)
    const float* k;          // this block's cross-attention keys / values of sample 0, channel-major [n][192][kv_pitch]
    const float* v;
    long long kv_bs;
    const int* lo;           // alignment window [lo[t], hi[t]) of query t (ldm/attention.py:184-189)
    const int* hi;
    const float* w2;         // attn2.to_out.0 weight, pack16
    const float* b2;
    const float* c2;         // attn2 output of the unconditional half (per-channel constant), modes 1 and 2
    float* x2;               // result x2, channel-major [n][192][pitch]
    long long x2_bs;
    int gn_nparts;
    float gn_eps;
    int kv_pitch;
    int wmax;                // max(hi - lo) <= 8
    float scale;             // dim_head ** -0.5
    int pitch, T;
    int mode;                // 0: every sample runs the whole chain
                             // 1: guidance, samples [0, Bc) unconditional (x2 = x1 + c2), [Bc, 2 Bc) conditional (whole chain)
                             // 2: guidance-shared block, grid = Bc clips: x1 once per clip; x2[s] = x1 + c2, x2[Bc + s] = chain with
                             //    the conditional sample's keys / values
    int Bc;
};

bool xattn_supports(const XAttnArgs& a, int n_samples);
void launch_xattn(const XAttnArgs& a, int n_samples, hipStream_t s);
void configure_xattn_kernel();

}  // namespace said

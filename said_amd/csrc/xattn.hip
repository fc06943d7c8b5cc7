// xattn.hip — the token-local chain after self-attention of a BasicTransformerBlock in ONE kernel (small batches):
//
//     x1 = attn1.to_out(o) + GroupNorm(x_in)                    (ldm/attention.py:127, 168, 227)
//     x2 = attn2.to_out(band_softmax(to_q(norm2(x1)) K^T) V) + x1   (ldm/attention.py:170-191)
//
// i.e. what were three launches (to_out + GroupNorm'ed residual | LayerNorm -> to_q -> banded cross-attention | to_out +
// residual: 5.8 + 6.8 + 5.1 us at B = 1, profiles/r01c_kernel_trace_variants.txt).  Every step of the chain is local to a
// token (the cross-attention keys/values come from the audio only and are precomputed per clip), so a workgroup that owns
// ALL 192 channels of a token tile needs no grid-wide dependency inside.  At B = 1 a launch of this kind is bounded by
// its dependent phases, not by throughput: the tile is 16 tokens (v_mfma_f32_16x16x4_f32), which gives 38-76 workgroups
// per launch and 72 MFMAs (2.3k clocks) per wave and GEMM.
//
// Work split of one 192x192 GEMM on a [192][16] LDS operand tile: 12 row tiles x 2 K halves = 24 units, three per wave
// (wave w: K half w & 1, row tiles w/2, w/2 + 4, w/2 + 8), so the three accumulators of a wave share every B fragment
// (one ds_read_b32 per MFMA triple); the two K halves of a row tile are summed through LDS in a fixed order.  Weight
// fragments (18 dwordx4 per wave and GEMM, host-packed, `pack16`) are requested one GEMM ahead.
//
// Guidance (diffusion.py:397-400, 421-423): the unconditional half's cross-attention output is the per-channel constant
// c2 (all keys equal), so those workgroups stop after x1 and store x2 = x1 + c2; in the guidance-shared first block the
// chain is computed once per clip and feeds both halves.
#include "gemm_common.h"
#include "xattn.h"

namespace said {

typedef float f32x4x __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ f32x4x xa_bload4(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4x, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

constexpr int XC = 192;          // channels
constexpr int XT = 16;           // tokens per workgroup
constexpr int XKW = XA_KW;       // key window columns staged per tile
constexpr int XW = 8;            // waves

// LDS carve (floats)
constexpr int O_TA = 0;                       // [192][16] GEMM operand: o, then norm2(x1), then the attn2 output
constexpr int O_TR = O_TA + XC * XT;          // GroupNorm(x_in) tile
constexpr int O_X1 = O_TR + XC * XT;          // x1
constexpr int O_TQ = O_X1 + XC * XT;          // q
constexpr int O_TK = O_TQ + XC * XT;          // [192][XKW] keys of the tile's window
constexpr int O_TV = O_TK + XC * XKW;
constexpr int O_RED = O_TV + XC * XKW;        // K-half hand-over [4 wave pairs][3 tiles][4 regs][64 lanes]
constexpr int O_COEF = O_RED + 4 * 3 * 4 * 64;   // GroupNorm (a, b) per channel
constexpr int O_GNS = O_COEF + 2 * XC;        // per-wave GroupNorm scratch
constexpr int O_PART = O_GNS + XW * GN_SCRATCH;  // band partial dots [24 chunks][8][16]
constexpr int O_PROB = O_PART + 24 * 8 * XT;  // [6 heads][8][16]
constexpr int O_LN = O_PROB + 6 * 8 * XT;     // LayerNorm partials [32][16][2] + stats [16][2]
constexpr int O_VEC = O_LN + 32 * XT * 2 + XT * 2;   // b1, c2, ln gamma, ln beta, b2 (5 x 192): fetched once at kernel entry
constexpr int O_END = O_VEC + 5 * XC;
static_assert(O_END * 4 <= 160 * 1024, "LDS carve exceeds a CU's 160 KB");

struct WFrag { f32x4x v[18]; };   // 3 units x 6 dwordx4 (24 k-steps each)

static __device__ __forceinline__ void load_w(const float* wp, int w, int l, WFrag& f) {
    // pack16: Wp[row tile][kq 0..11][lane][4]; unit u of wave w: row tile (w >> 1) + 4 u, K half w & 1 -> kq = 6 (w & 1) + i
    const rsrc_t r = make_rsrc(wp, 12u * 12u * 1024u);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int i = 0; i < 6; ++i) f.v[u * 6 + i] = xa_bload4(r, l * 16, ((((w >> 1) + 4 * u) * 12) + 6 * (w & 1) + i) * 1024);
}

// acc[u] += W[unit u] . tile[K half of this wave]   (72 MFMAs, 24 ds_read_b32 requested ahead of the MFMA stream in three
// groups of eight so that no MFMA triple waits for its own LDS read)
static __device__ __forceinline__ void gemm16(const WFrag& f, const float* tile, int w, int l, f32x4x (&acc)[3]) {
    const float* bp = tile + ((w & 1) * 96 + (l >> 4)) * XT + (l & 15);
#pragma unroll
    for (int u = 0; u < 3; ++u) acc[u] = {0.f, 0.f, 0.f, 0.f};
    float b[24];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) b[ks] = bp[ks * 4 * XT];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        if (g < 2) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) b[(g + 1) * 8 + ks] = bp[((g + 1) * 8 + ks) * 4 * XT];
        }
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            const int ks = g * 8 + k8;
#pragma unroll
            for (int u = 0; u < 3; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[u * 6 + (ks >> 2)][ks & 3], b[ks], acc[u], 0, 0, 0);
        }
    }
}

// sum the two K halves of each row tile: odd waves hand their accumulators to the even wave of the pair
static __device__ __forceinline__ void pair_reduce(float* red, int w, int l, f32x4x (&acc)[3]) {
    float* rp = red + (w >> 1) * (3 * 4 * 64);
    if (w & 1) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) rp[(u * 4 + r) * 64 + l] = acc[u][r];
    }
    __syncthreads();
    if (!(w & 1)) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[u][r] += rp[(u * 4 + r) * 64 + l];
    }
}

__global__ __launch_bounds__(64 * XW) void xattn_kernel(const XAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * XT, s = blockIdx.y;
    const int T = a.T, pitch = a.pitch;
    // which samples this workgroup reads / writes (file header)
    bool full = true, wr_unc = false;
    int s_kv = s, s_full = s;
    if (a.mode == 1) { full = s >= a.Bc; wr_unc = !full; }
    else if (a.mode == 2) { wr_unc = true; s_kv = a.Bc + s; s_full = a.Bc + s; }

    // ---------------- phase 0: every request that does not depend on another phase ----------------
    WFrag wf, wn;
    load_w(a.w1, w, l, wf);
    GnLoads gl;
    const GnP gp = {6, a.gn_nparts, T, a.gn_eps, a.gn_gamma, a.gn_beta, XC};
    const rsrc_t rgn = make_rsrc(a.gn_part + (long long)s * a.gn_part_bs, (unsigned)XC * (unsigned)a.gn_nparts * 8u);
    gn_issue(gp, rgn, w * 24, 24, l, gl);
    // o and x_in tiles: thread -> (row = tid >> 2 [+ 128], token quad tid & 3)
    const int row0 = tid >> 2, q4 = tid & 3;
    const rsrc_t ro = make_rsrc(a.o + (long long)s * a.o_bs, (unsigned)XC * (unsigned)pitch * 4u);
    const rsrc_t rr = make_rsrc(a.res + (long long)s * a.res_bs, (unsigned)XC * (unsigned)pitch * 4u);
    const int voff0 = (row0 * pitch + t0 + 4 * q4) * 4;
    const bool second = row0 + 128 < XC;
    const int voff1 = second ? ((row0 + 128) * pitch + t0 + 4 * q4) * 4 : (int)0x80000000;
    const f32x4x o0 = xa_bload4(ro, voff0, 0), o1 = xa_bload4(ro, voff1, 0);
    const f32x4x r0 = xa_bload4(rr, voff0, 0), r1 = xa_bload4(rr, voff1, 0);
    // the five per-channel vectors of the chain (each epilogue would otherwise pay its own memory round trip) and this
    // thread's query window: requested now, parked in LDS / registers
    float vec0 = 0.f, vec1 = 0.f;
    {
        const int i0 = tid, i1 = tid + 512;   // 960 entries over 512 threads
        const float* const vp[5] = {a.b1, a.c2, a.ln_g, a.ln_b, a.b2};
        const int k0 = i0 / XC, k1 = i1 / XC;
        const float* p0 = vp[0];
        const float* p1 = vp[2];
#pragma unroll
        for (int k = 0; k < 5; ++k) { if (k0 == k) p0 = vp[k]; if (k1 == k) p1 = vp[k]; }
        if (p0) vec0 = gload(p0, i0 - k0 * XC);
        if (i1 < 5 * XC && p1) vec1 = gload(p1, i1 - k1 * XC);
    }
    const int tq_ = min(t0 + (tid & 15), T - 1);
    const int my_lo = cload(a.lo, tq_), my_hi = cload(a.hi, tq_);
    // key / value window of the tile: columns kb .. kb + XKW - 1 of every channel row (host checks that it covers the
    // alignment windows of all 16 queries); loads past the row block read 0 through the descriptor's range check
    const int tl = min(t0 + XT - 1, T - 1);
    const int kb = full ? cload(a.lo, t0) : 0;
    (void)tl;
    float* tK = sm + O_TK;
    float* tV = sm + O_TV;
    f32x4x kq[3], vq[3];
    if (full) {
        const rsrc_t rk = make_rsrc(a.k + (long long)s_kv * a.kv_bs, (unsigned)XC * (unsigned)a.kv_pitch * 4u);
        const rsrc_t rv = make_rsrc(a.v + (long long)s_kv * a.kv_bs, (unsigned)XC * (unsigned)a.kv_pitch * 4u);
#pragma unroll
        for (int i = 0; i < 3; ++i) {   // 192 rows x 6 quads = 1152 pieces over 512 threads
            const int p = tid + i * 512;
            const int row = p / 6, qq = p - row * 6;
            const bool ok = p < XC * 6 && (kb + 4 * qq) < a.kv_pitch;
            const int vo = ok ? (row * a.kv_pitch + kb + 4 * qq) * 4 : (int)0x80000000;
            kq[i] = xa_bload4(rk, vo, 0);
            vq[i] = xa_bload4(rv, vo, 0);
        }
    }

    // ---------------- phase 1: GroupNorm coefficients of x_in, operand tiles into LDS ----------------
    float* coef = sm + O_COEF;
    gn_finish(gp, rgn, w * 24, 24, l, gl, sm + O_GNS + w * GN_SCRATCH, coef);
    float* tA = sm + O_TA;
    float* tR = sm + O_TR;
    float* tX1 = sm + O_X1;
    {
        f32x4x z0, z1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool tv = t0 + 4 * q4 + e < T;
            z0[e] = tv ? o0[e] : 0.f;
            z1[e] = tv ? o1[e] : 0.f;
        }
        *reinterpret_cast<f32x4x*>(tA + row0 * XT + 4 * q4) = z0;
        if (second) *reinterpret_cast<f32x4x*>(tA + (row0 + 128) * XT + 4 * q4) = z1;
    }
    if (full) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int p = tid + i * 512;
            const int row = p / 6, qq = p - row * 6;
            if (p < XC * 6) {
                *reinterpret_cast<f32x4x*>(tK + row * XKW + 4 * qq) = kq[i];
                *reinterpret_cast<f32x4x*>(tV + row * XKW + 4 * qq) = vq[i];
            }
        }
    }
    float* vecs = sm + O_VEC;
    vecs[tid] = vec0;
    if (tid + 512 < 5 * XC) vecs[tid + 512] = vec1;
    __syncthreads();   // coefficient table (written per wave slice), the o tile and the vector table are complete
    {
        const float2 c0 = *reinterpret_cast<const float2*>(coef + 2 * row0);
        f32x4x z0, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) z0[e] = (t0 + 4 * q4 + e < T) ? fmaf(r0[e], c0.x, c0.y) : 0.f;
        *reinterpret_cast<f32x4x*>(tR + row0 * XT + 4 * q4) = z0;
        if (second) {
            const float2 c1 = *reinterpret_cast<const float2*>(coef + 2 * (row0 + 128));
#pragma unroll
            for (int e = 0; e < 4; ++e) z1[e] = (t0 + 4 * q4 + e < T) ? fmaf(r1[e], c1.x, c1.y) : 0.f;
            *reinterpret_cast<f32x4x*>(tR + (row0 + 128) * XT + 4 * q4) = z1;
        }
    }

    // ---------------- phase 2: x1 = to_out(o) + b + GroupNorm(x_in) ----------------
    load_w(a.wq, w, l, wn);   // next GEMM's fragments travel while this one multiplies (unconditional: exact wait counts)
    f32x4x acc[3];
    gemm16(wf, tA, w, l, acc);
    pair_reduce(sm + O_RED, w, l, acc);   // (its barrier also orders the tR writes above)
    const int col = l & 15;
    const int t = t0 + col;
    if (!(w & 1)) {
        float* const yu = a.x2 + (long long)s * a.x2_bs;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ((w >> 1) + 4 * u) * 16 + 4 * (l >> 4) + r;
                const float v = acc[u][r] + vecs[row] + tR[row * XT + col];
                tX1[row * XT + col] = v;
                if (wr_unc && t < T) gstore(yu, (long long)row * pitch + t, v + vecs[XC + row]);
            }
    }
    if (!full) return;
    __syncthreads();

    // ---------------- phase 3: norm2 (LayerNorm over channels, eps 1e-5) -> operand tile ----------------
    float* lnp = sm + O_LN;
    {
        const int tt = tid & 15, part = tid >> 4;   // 32 parts x 6 channels
        const float ref = tX1[tt];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float d = tX1[(part * 6 + i) * XT + tt] - ref;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
        lnp[(part * XT + tt) * 2] = s1;
        lnp[(part * XT + tt) * 2 + 1] = s2;
        __syncthreads();
        if (tid < XT) {
            float S1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int p = 0; p < 32; ++p) { S1 += lnp[(p * XT + tid) * 2]; S2 += lnp[(p * XT + tid) * 2 + 1]; }
            const float md = S1 * (1.0f / XC);
            const float var = fmaxf(S2 * (1.0f / XC) - md * md, 0.f);
            lnp[32 * XT * 2 + tid * 2] = tX1[tid] + md;
            lnp[32 * XT * 2 + tid * 2 + 1] = __builtin_amdgcn_rsqf(var + 1e-5f);
        }
        __syncthreads();
        const float mu = lnp[32 * XT * 2 + tt * 2], rs = lnp[32 * XT * 2 + tt * 2 + 1];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = part * 6 + i;
            tA[c * XT + tt] = fmaf((tX1[c * XT + tt] - mu) * rs, vecs[2 * XC + c], vecs[3 * XC + c]);
        }
    }
    __syncthreads();

    // ---------------- phase 4: q = to_q(norm2(x1)) ----------------
    load_w(a.w2, w, l, wf);
    gemm16(wn, tA, w, l, acc);
    pair_reduce(sm + O_RED, w, l, acc);
    float* tQ = sm + O_TQ;
    if (!(w & 1)) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ((w >> 1) + 4 * u) * 16 + 4 * (l >> 4) + r;
                tQ[row * XT + col] = acc[u][r];
            }
    }
    __syncthreads();

    // ---------------- phase 5: banded softmax over the audio keys (alignment window of each query) ----------------
    {
        const int tt = tid & 15, ch = tid >> 4;          // 24 chunks of 8 channels (4 per head) x 16 queries
        const int lo = my_lo, hi = my_hi;   // tt = tid & 15: the window requested at kernel entry
        const int rel = lo - kb;
        float* part = sm + O_PART;
        float* prob = sm + O_PROB;
        const int wmax = a.wmax;
        if (ch < 24) {
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = ch * 8 + i;
                const float qv = tQ[c * XT + tt];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < wmax) p[j] = fmaf(qv, tK[c * XKW + min(rel + j, XKW - 1)], p[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) part[(ch * 8 + j) * XT + tt] = p[j];
        }
        __syncthreads();
        if (tid < 6 * XT) {
            const int h = tid >> 4;   // tt = tid & 15 as above
            float sc[8], mx = -3.0e38f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float sum = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) sum += part[((h * 4 + g) * 8 + j) * XT + tt];
                const bool vis = (j < wmax) && (lo + j < hi);
                sc[j] = vis ? sum * a.scale : -3.0e38f;
                mx = fmaxf(mx, sc[j]);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool vis = (j < wmax) && (lo + j < hi);
                sc[j] = vis ? __expf(sc[j] - mx) : 0.f;
                den += sc[j];
            }
            const float inv = 1.0f / den;
#pragma unroll
            for (int j = 0; j < 8; ++j) prob[(h * 8 + j) * XT + tt] = sc[j] * inv;
        }
        __syncthreads();
        if (ch < 24) {
            const int h = ch >> 2;
            float pj[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pj[j] = prob[(h * 8 + j) * XT + tt];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = ch * 8 + i;
                float o = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < wmax) o = fmaf(pj[j], tV[c * XKW + min(rel + j, XKW - 1)], o);   // invisible slots have p = 0
                tA[c * XT + tt] = o;
            }
        }
    }
    __syncthreads();

    // ---------------- phase 6: x2 = to_out(attn2) + b + x1 ----------------
    gemm16(wf, tA, w, l, acc);
    pair_reduce(sm + O_RED, w, l, acc);
    if (!(w & 1)) {
        float* const yf = a.x2 + (long long)s_full * a.x2_bs;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ((w >> 1) + 4 * u) * 16 + 4 * (l >> 4) + r;
                const float v = acc[u][r] + vecs[4 * XC + row] + tX1[row * XT + col];
                if (t < T) gstore(yf, (long long)row * pitch + t, v);
            }
    }
}

bool xattn_supports(const XAttnArgs& a, int n_samples) {
    if (a.T < 1 || a.T > 0xffff || n_samples < 1) return false;
    if (a.wmax < 1 || a.wmax > 8) return false;
    if (!a.o || !a.res || !a.gn_part || !a.w1 || !a.wq || !a.w2 || !a.k || !a.v || !a.x2) return false;
    if (a.mode != 0 && !a.c2) return false;
    return true;
}

void configure_xattn_kernel() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, O_END * 4);
}

void launch_xattn(const XAttnArgs& a, int n_samples, hipStream_t s) {
    dim3 grid((a.T + XT - 1) / XT, n_samples);
    hipLaunchKernelGGL(xattn_kernel, grid, dim3(64 * XW), O_END * 4, s, a);
}

}  // namespace said

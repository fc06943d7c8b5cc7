// micro-benchmark: what does a dependent launch cost, and can a flag-chained pair of streams hide it?
//   A  one stream (barrier-bit chain), kernels whose workgroups return at once          -> the floor per node
//   B  one stream, every workgroup spins W us, reads its predecessor's output, writes its own
//   C  two / three streams taking the kernels in turn; kernel q waits on a counter its predecessor's workgroups bump after an agent-scope
//      release (and acquires before reading): the hardware launch of q + 1 overlaps the execution of q.  Four counters in turn; kernel q, once past its own wait,
//      clears the one kernel q + S - 1 will bump (S streams <= 3: its previous waiter, q - 1, is complete, and its next waiter, q + S, is not dispatched before q completes)
//   D  two streams, no flags, no work (are the two chains of a captured graph really concurrent?)
// Every run checks the data (kernel q must see q - 1 in a region written by a workgroup of another XCD; the same lines held q - 3 two kernels
// earlier, so a stale L2 line is caught) and counts spin time-outs (bounded spins: a broken protocol reports, it does not hang).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <bool FLAG>
__global__ __launch_bounds__(512) void k_work(unsigned* ctr, int q, int wait_n, int reset_slot, float* buf, int work_ticks, unsigned* err, int check, int nelem) {
    const int tid = threadIdx.x, b = blockIdx.x, n = gridDim.x;
    if (work_ticks < 0) return;   // (A, D: leave at once)
    if (FLAG && wait_n) {
        if (tid == 0) {
            int it = 0;
            while (__hip_atomic_load(&ctr[(q + 3) & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)wait_n) {
                __builtin_amdgcn_s_sleep(2);
                if (++it > (1 << 16)) { atomicAdd(&err[1], 1u); break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (FLAG && b == 0 && tid == 0) __hip_atomic_store(&ctr[reset_slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    float v = 0.f;
    if (check && q > 0) {
        const int src = (b * 37 + 11) % n;
        v = buf[(size_t)((q + 1) & 1) * nelem + (size_t)src * 512 + tid];
        if (v != (float)(q - 1)) atomicAdd(&err[0], 1u);
    }
    while (wall_clock64() - t0 < work_ticks) __builtin_amdgcn_s_sleep(4);
    buf[(size_t)(q & 1) * nelem + (size_t)b * 512 + tid] = (float)q;
    if (FLAG) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(&ctr[q & 3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct Res { double us_per_node; unsigned bad, timeouts; };

static Res run(int nstreams, bool flag, int n, int work_us, int K, int lds) {
    std::vector<hipStream_t> st(nstreams);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *ctr, *err; float* buf;
    const int nelem = n * 512;
    CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&err, 64)); CK(hipMalloc(&buf, sizeof(float) * 2 * nelem));
    CK(hipMemset(ctr, 0, 64)); CK(hipMemset(err, 0, 64)); CK(hipMemset(buf, 0, sizeof(float) * 2 * nelem));
    const int ticks = work_us < 0 ? -1 : work_us * 100;   // wall_clock64: 100 MHz
    const int check = work_us >= 0;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_work<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_work<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t fork, join[8];
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (auto& j : join) CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeRelaxed));
    if (nstreams > 1) { CK(hipEventRecord(fork, st[0])); for (int i = 1; i < nstreams; ++i) CK(hipStreamWaitEvent(st[i], fork, 0)); }
    for (int q = 0; q < K; ++q) {
        hipStream_t s = st[q % nstreams];
        if (flag) hipLaunchKernelGGL(k_work<true>, dim3(n), dim3(512), lds, s, ctr, q, q ? n : 0, (q + nstreams - 1) & 3, buf, ticks, err, check, nelem);
        else hipLaunchKernelGGL(k_work<false>, dim3(n), dim3(512), lds, s, ctr, q, 0, 0, buf, ticks, err, check, nelem);
    }
    for (int i = 1; i < nstreams; ++i) { CK(hipEventRecord(join[i], st[i])); CK(hipStreamWaitEvent(st[0], join[i], 0)); }
    CK(hipStreamEndCapture(st[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st[0])); CK(hipStreamSynchronize(st[0]));
    const int reps = 10;
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st[0]));
    CK(hipStreamSynchronize(st[0]));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps / K;
    unsigned h[2]; CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipFree(ctr)); CK(hipFree(err)); CK(hipFree(buf));
    for (auto& s : st) CK(hipStreamDestroy(s));
    return {us, h[0], h[1]};
}

int main() {
    const int K = 240;
    for (int n : {38, 228}) {
        for (int lds : {40 * 1024, 150 * 1024}) {
            printf("== %d workgroups x 512 threads, %d KB of LDS, graph of %d kernels\n", n, lds / 1024, K);
            Res a = run(1, false, n, -1, K, lds);
            printf("A one stream, workgroups leave at once            %6.2f us per node\n", a.us_per_node);
            Res d = run(2, false, n, -1, K, lds);
            printf("D two streams, no flags, leave at once            %6.2f us per node\n", d.us_per_node);
            for (int w : {0, 2, 4, 8, 16}) {
                Res b = run(1, false, n, w, K, lds);
                Res c2 = run(2, true, n, w, K, lds);
                Res c3 = run(3, true, n, w, K, lds);
                printf("W = %2d us: B one stream %6.2f (bad %u) | C two streams + flags %6.2f (bad %u, time-outs %u) | three streams %6.2f (bad %u, time-outs %u)\n", w,
                       b.us_per_node, b.bad, c2.us_per_node, c2.bad, c2.timeouts, c3.us_per_node, c3.bad, c3.timeouts);
                fflush(stdout);
            }
        }
    }
    return 0;
}

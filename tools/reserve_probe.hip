// Can a persistent MFMA-saturating kernel keep OFF a set of compute units by itself (workgroups that find themselves on a
// reserved CU exit at once, the others pull tasks from a queue), so that a latency-bound kernel chain on another stream -- also
// inside a captured graph, where CU-masked streams are not available -- has CUs of its own?
//   1. census: where do the 3 x 256 workgroups of kP land (HW_REG_HW_ID / HW_REG_XCC_ID), how many stay per CU
//   2. eager, two streams: chain of small kernels beside kP with r reserved CUs per XCD
//   3. the same as ONE captured graph with three branches: kP | gate (bounded spin on a flag kP sets) + chain | second chain
// usage: reserve_probe <reserved CUs per XCD> <tasks> <mfma iters per task>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

// persistent "Gram": 512 threads, 32 KiB LDS, <= 80 VGPRs -> three per CU.  ctl[0] = task queue head, ctl[1] = tasks done,
// ctl[2] = workgroups that stayed, ctl[3] = workgroups that left
__global__ __launch_bounds__(512, 6) void kP(int ntask, int iters, int reserve, unsigned *ctl, unsigned *where, unsigned long long *t, double *sink)
{
    __shared__ double lds[4096];
    __shared__ int s_task;
    const unsigned hw = hw_id(), xcc = xcc_id();
    const unsigned cu = (hw >> 8) & 15u, se = (hw >> 13) & 7u;
    if (threadIdx.x == 0) where[blockIdx.x] = (hw & 0xffffu) | (xcc << 16);
    if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = __builtin_amdgcn_s_memrealtime();
    // reserved: the CUs with the highest ids of every shader engine, `reserve` per XCD in all (round-robin over the 4 SEs)
    const unsigned per_se = (reserve + 3 - se) / 4;          // SE 0 gets the remainder first
    const bool reserved = cu >= 8u - per_se && cu < 8u ? true : false;
    if (reserved) { if (threadIdx.x == 0) atomicAdd(&ctl[3], 1u); return; }
    if (threadIdx.x == 0) atomicAdd(&ctl[2], 1u);
    d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0}, c4 = {0, 0, 0, 0}, c5 = {0, 0, 0, 0}, c6 = {0, 0, 0, 0}, c7 = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3;
    lds[threadIdx.x] = a;
    for (;;) {
        if (threadIdx.x == 0) s_task = (int)atomicAdd(&ctl[0], 1u);
        __syncthreads();
        const int task = s_task;
        __syncthreads();
        if (task >= ntask) break;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c7, 0, 0, 0);
        }
        __syncthreads();
        if (threadIdx.x == 0) { __threadfence(); atomicAdd(&ctl[1], 1u); }
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] + lds[(threadIdx.x * 7) & 4095] == 12345.678) sink[0] = 1.0;
    if (threadIdx.x == 0) atomicMax(&t[1], __builtin_amdgcn_s_memrealtime());
}
// latency chain link: 68 workgroups x 256 threads with many VGPRs (cannot squeeze in beside kP), wave 0 runs a dependent chain
__global__ __launch_bounds__(256, 1) void kB(int iters, unsigned long long *t, int slot, double *sink, unsigned *where)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) t[slot] = t0;
    double keep[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) keep[i] = threadIdx.x * 1e-3 + i;
    if (threadIdx.x < 64) {
        double x = 1.0 + threadIdx.x * 1e-6, acc = 0.0;
        for (int k = 0; k < iters; ++k) {
            double y = __builtin_amdgcn_rsq(x); x = fma(y, y, x) * 0.5 + 0.7; acc = fma(x, y, acc);
#pragma unroll
            for (int i = 0; i < 96; i += 8) keep[i] = fma(keep[i], y, x);
        }
        if (acc == 12345.678) sink[1] = acc;
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += keep[i];
    if (s == 12345.678) sink[2] = s;
    if (threadIdx.x == 0) {
        if (where) where[blockIdx.x] = (hw_id() & 0xffffu) | (xcc_id() << 16);
        atomicMax(&t[slot + 1], __builtin_amdgcn_s_memrealtime());
    }
}
// gate: one wave waits (bounded) until kP has finished `need` tasks; out[0] = 1 passed / 2 timed out, out[1] = polls
__global__ void kGate(const unsigned *ctl, unsigned need, unsigned *out, unsigned long long *t, int slot)
{
    if (threadIdx.x != 0) return;
    unsigned polls = 0, ok = 2;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 300000ull) {          // 3 ms at 100 MHz
        if (__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(20);
        ++polls;
    }
    __threadfence();
    out[0] = ok; out[1] = polls;
    t[slot] = t0; t[slot + 1] = __builtin_amdgcn_s_memrealtime();
}

int main(int argc, char **argv)
{
    const int reserve = argc > 1 ? atoi(argv[1]) : 4, ntask = argc > 2 ? atoi(argv[2]) : 2016, iters = argc > 3 ? atoi(argv[3]) : 200, variant = argc > 4 ? atoi(argv[4]) : 0, NB = 12;
    unsigned *ctl, *where, *whereB, *gate; unsigned long long *t; double *sink;
    (void)hipMalloc(&ctl, 64); (void)hipMalloc(&where, 4 * 4096); (void)hipMalloc(&whereB, 4 * 4096); (void)hipMalloc(&gate, 64);
    (void)hipMalloc(&t, 8 * 256); (void)hipMalloc(&sink, 64);
    (void)hipFuncSetAttribute((const void *)kB, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipStream_t s0, s1, s2, s3;
    (void)hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int G = 768;
    // ---- 1. census
    for (int rep = 0; rep < (variant ? 0 : 2); ++rep) {
        (void)hipMemset(ctl, 0, 64); (void)hipMemset(t, 0, 8 * 256);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(kP, dim3(G), dim3(512), 0, s0, ntask, iters, reserve, ctl, where, t, sink);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> w(G), c(16); std::vector<unsigned long long> h(256);
        (void)hipMemcpy(w.data(), where, 4 * G, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), ctl, 64, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h.data(), t, 8 * 256, hipMemcpyDeviceToHost);
        std::map<unsigned, int> percu; std::map<unsigned, int> cuids, seids;
        for (int i = 0; i < G; ++i) { unsigned key = ((w[i] >> 16) << 8) | (((w[i] >> 13) & 7) << 4) | ((w[i] >> 8) & 15); percu[key]++; cuids[(w[i] >> 8) & 15]++; seids[(w[i] >> 13) & 7]++; }
        std::map<int, int> hist; for (auto &kv : percu) hist[kv.second]++;
        printf("census rep %d: kP alone %.1f us; stayed %u left %u tasks done %u; distinct CUs seen %zu; workgroups per CU histogram:", rep, (h[1] - h[0]) / 100.0, c[2], c[3], c[1], percu.size());
        for (auto &kv : hist) printf(" %dx%d", kv.second, kv.first);
        printf("\n  cu ids:"); for (auto &kv : cuids) printf(" %u:%d", kv.first, kv.second);
        printf("  se ids:"); for (auto &kv : seids) printf(" %u:%d", kv.first, kv.second);
        printf("\n  first 16 workgroups (xcc:se:cu):"); for (int i = 0; i < 16; ++i) printf(" %u:%u:%u", w[i] >> 16, (w[i] >> 13) & 7, (w[i] >> 8) & 15);
        printf("\n");
    }
    // ---- 2. eager: chain beside kP
    for (int withP = 0; withP < (variant ? 0 : 2); ++withP) {
        (void)hipMemset(ctl, 0, 64); (void)hipMemset(t, 0, 8 * 256);
        (void)hipDeviceSynchronize();
        if (withP) hipLaunchKernelGGL(kP, dim3(G), dim3(512), 0, s0, ntask, iters, reserve, ctl, where, t, sink);
        if (withP) hipLaunchKernelGGL(kGate, dim3(1), dim3(64), 0, s1, ctl, 64u, gate, t, 100);
        for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(kB, dim3(68), dim3(256), 72 * 1024, s1, 300, t, 2 + 2 * i, sink, i == 3 ? whereB : (unsigned *)nullptr);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(256); std::vector<unsigned> w(68), gt(16);
        (void)hipMemcpy(h.data(), t, 8 * 256, hipMemcpyDeviceToHost); (void)hipMemcpy(w.data(), whereB, 4 * 68, hipMemcpyDeviceToHost); (void)hipMemcpy(gt.data(), gate, 64, hipMemcpyDeviceToHost);
        printf("eager %s kP: chain of %d kernels %.1f us (%.2f each; first %.2f last %.2f)", withP ? "with   " : "without", NB, (h[3 + 2 * (NB - 1)] - h[2]) / 100.0, (h[3 + 2 * (NB - 1)] - h[2]) / 100.0 / NB,
               (h[3] - h[2]) / 100.0, (h[3 + 2 * (NB - 1)] - h[2 + 2 * (NB - 1)]) / 100.0);
        if (withP) printf("; kP %.1f us; gate %s after %.1f us (%u polls); chain began %.1f us after kP", (h[1] - h[0]) / 100.0, gt[0] == 1 ? "passed" : "TIMED OUT", (h[101] - h[100]) / 100.0, gt[1], ((double)h[2] - (double)h[0]) / 100.0);
        printf("\n");
        if (withP) { printf("  chain link 3 landed on (xcc:se:cu):"); for (int i = 0; i < 68; i += 3) printf(" %u:%u:%u", w[i] >> 16, (w[i] >> 13) & 7, (w[i] >> 8) & 15); printf("\n"); }
    }
    // ---- 3. one captured graph, three branches forked from s0
    {
        hipEvent_t ef, e1, e2, ev3, ef2; (void)hipEventCreateWithFlags(&ev3, hipEventDisableTiming); (void)hipEventCreateWithFlags(&ef2, hipEventDisableTiming); (void)hipEventCreateWithFlags(&ef, hipEventDisableTiming); (void)hipEventCreateWithFlags(&e1, hipEventDisableTiming); (void)hipEventCreateWithFlags(&e2, hipEventDisableTiming);
        hipGraph_t graph; hipGraphExec_t gexec;
        hipError_t e = hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal);
        (void)hipMemsetAsync(ctl, 0, 64, s0);
        (void)hipEventRecord(ef, s0);
        (void)hipStreamWaitEvent(s1, ef, 0); (void)hipStreamWaitEvent(s2, ef, 0);
        hipStream_t sP = variant == 1 ? s0 : s1, sF = variant == 1 ? s1 : s0;     // variant 1: kP on the origin stream, the free chain on a fork
        if (variant >= 7) {
            // 7: a trivial first-captured branch (one tiny kernel on s3), then kP | gate + chain | free chain as in variant 0;  8: the same, two fork/join rounds
            for (int round = 0; round < (variant == 8 ? 2 : 1); ++round) {
                if (round) { (void)hipEventRecord(e1, s1); (void)hipEventRecord(e2, s2); (void)hipEventRecord(ev3, s3); (void)hipStreamWaitEvent(s0, e1, 0); (void)hipStreamWaitEvent(s0, e2, 0); (void)hipStreamWaitEvent(s0, ev3, 0);
                    (void)hipMemsetAsync(ctl, 0, 64, s0); (void)hipEventRecord(ef2, s0); (void)hipStreamWaitEvent(s1, ef2, 0); (void)hipStreamWaitEvent(s2, ef2, 0); (void)hipStreamWaitEvent(s3, ef2, 0); }
                else (void)hipStreamWaitEvent(s3, ef, 0);
                hipLaunchKernelGGL(kB, dim3(1), dim3(256), 72 * 1024, s3, 1, t, 90, sink, (unsigned *)nullptr);
                hipLaunchKernelGGL(kP, dim3(G), dim3(512), 0, s1, ntask, iters, reserve, ctl, where, t + 120 * round, sink);
                hipLaunchKernelGGL(kGate, dim3(1), dim3(64), 0, s2, ctl, 64u, gate, t + 120 * round, 100);
                for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(kB, dim3(68), dim3(256), 72 * 1024, s2, 300, t + 120 * round, 2 + 2 * i, sink, (unsigned *)nullptr);
                for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(kB, dim3(40), dim3(256), 72 * 1024, s0, 300, t + 120 * round, 40 + 2 * i, sink, (unsigned *)nullptr);
            }
            (void)hipEventRecord(ev3, s3); (void)hipStreamWaitEvent(s0, ev3, 0);
        } else if (variant >= 3) {
            // two branches only.  3: kP on the fork, gate + chain on the origin; 4: kP on the origin, gate + chain on the fork; 5/6: the same with the chain captured first
            hipStream_t sk = variant == 3 || variant == 5 ? s1 : s0, sc = variant == 3 || variant == 5 ? s0 : s1;
            if (variant <= 4) hipLaunchKernelGGL(kP, dim3(G), dim3(512), 0, sk, ntask, iters, reserve, ctl, where, t, sink);
            hipLaunchKernelGGL(kGate, dim3(1), dim3(64), 0, sc, ctl, 64u, gate, t, 100);
            for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(kB, dim3(68), dim3(256), 72 * 1024, sc, 300, t, 2 + 2 * i, sink, (unsigned *)nullptr);
            if (variant > 4) hipLaunchKernelGGL(kP, dim3(G), dim3(512), 0, sk, ntask, iters, reserve, ctl, where, t, sink);
        } else {
        if (variant != 2) hipLaunchKernelGGL(kP, dim3(G), dim3(512), 0, sP, ntask, iters, reserve, ctl, where, t, sink);
        hipLaunchKernelGGL(kGate, dim3(1), dim3(64), 0, s2, ctl, 64u, gate, t, 100);
        for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(kB, dim3(68), dim3(256), 72 * 1024, s2, 300, t, 2 + 2 * i, sink, (unsigned *)nullptr);
        for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(kB, dim3(40), dim3(256), 72 * 1024, sF, 300, t, 40 + 2 * i, sink, (unsigned *)nullptr);
        if (variant == 2) hipLaunchKernelGGL(kP, dim3(G), dim3(512), 0, sP, ntask, iters, reserve, ctl, where, t, sink);   // variant 2: kP captured last
        }
        (void)hipEventRecord(e1, s1); (void)hipEventRecord(e2, s2);
        (void)hipStreamWaitEvent(s0, e1, 0); (void)hipStreamWaitEvent(s0, e2, 0);
        hipLaunchKernelGGL(kB, dim3(8), dim3(256), 72 * 1024, s0, 10, t, 80, sink, (unsigned *)nullptr);
        hipError_t e3 = hipStreamEndCapture(s0, &graph);
        hipError_t e4 = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
        printf("graph capture: %s / %s / %s\n", hipGetErrorString(e), hipGetErrorString(e3), hipGetErrorString(e4));
        if (e3 == hipSuccess && e4 == hipSuccess) {
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(t, 0, 8 * 256);
                (void)hipDeviceSynchronize();
                (void)hipGraphLaunch(gexec, s0);
                (void)hipDeviceSynchronize();
                std::vector<unsigned long long> h(256); std::vector<unsigned> gt(16);
                (void)hipMemcpy(h.data(), t, 8 * 256, hipMemcpyDeviceToHost); (void)hipMemcpy(gt.data(), gate, 64, hipMemcpyDeviceToHost);
                printf("graph rep %d: kP %.1f us; gate %s after %.1f us; gated chain %.2f us each, began %.1f after kP began; free chain %.2f us each, began %.1f after kP began; join node at %.1f\n", rep,
                       (h[1] - h[0]) / 100.0, gt[0] == 1 ? "passed" : "TIMED OUT", (h[101] - h[100]) / 100.0,
                       (h[3 + 2 * (NB - 1)] - h[2]) / 100.0 / NB, ((double)h[2] - (double)h[0]) / 100.0,
                       (h[41 + 2 * (NB - 1)] - h[40]) / 100.0 / NB, ((double)h[40] - (double)h[0]) / 100.0, ((double)h[80] - (double)h[0]) / 100.0);
                if (variant == 8) printf("   round 2: kP %.1f us began %.1f after round 1's; gate after %.1f us; gated chain %.2f us each, began %.1f after kP began; free chain %.2f each, began %.1f after kP began\n", (h[121] - h[120]) / 100.0, ((double)h[120] - (double)h[0]) / 100.0,
                       (h[221] - h[220]) / 100.0, (h[123 + 2 * (NB - 1)] - h[122]) / 100.0 / NB, ((double)h[122] - (double)h[120]) / 100.0, (h[161 + 2 * (NB - 1)] - h[160]) / 100.0 / NB, ((double)h[160] - (double)h[120]) / 100.0);
            }
        }
    }
    return 0;
}

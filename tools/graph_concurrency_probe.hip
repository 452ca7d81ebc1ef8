// Do two captured graphs launched on two streams from ONE host thread run side by side?  Each graph: a linear chain of 40 kernels (8 workgroups,
// 10 us busy each).  Alone: 40 x (10 + launch gap).  Two at once: the same time if they overlap, twice if the runtime serialises graph launches.
// hipcc --offload-arch=gfx950 -O3 -o graph_concurrency_probe tools/graph_concurrency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_busy(unsigned long long ticks, int *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    if (ticks == 0) sink[0] = 1;
}
static int make(hipStream_t s, int links, int branches, hipStream_t s2, int *sink, hipGraphExec_t *ge)
{
    hipGraph_t g;
    hipEvent_t e0, e1;
    CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    if (branches == 2) { CK(hipEventRecord(e0, s)); CK(hipStreamWaitEvent(s2, e0, 0)); }
    for (int l = 0; l < links; ++l) {
        hipLaunchKernelGGL(k_busy, dim3(8), dim3(256), 0, s, 1000ull, sink);
        if (branches == 2) hipLaunchKernelGGL(k_busy, dim3(8), dim3(256), 0, s2, 1000ull, sink);
    }
    if (branches == 2) { CK(hipEventRecord(e1, s2)); CK(hipStreamWaitEvent(s, e1, 0)); }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(ge, g, nullptr, nullptr, 0));
    return 0;
}
int main()
{
    hipStream_t sa, sb, sa2, sb2;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sa2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb2, hipStreamNonBlocking));
    int *sink; CK(hipMalloc(&sink, 4));
    hipEvent_t t0, t1, t2;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreate(&t2));
    for (int branches = 1; branches <= 2; ++branches) {
        hipGraphExec_t ga, gb;
        if (make(sa, 40, branches, sa2, sink, &ga) || make(sb, 40, branches, sb2, sink, &gb)) return 1;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, sa)); CK(hipGraphLaunch(ga, sa)); CK(hipEventRecord(t1, sa));
            CK(hipDeviceSynchronize());
            float one; CK(hipEventElapsedTime(&one, t0, t1));
            CK(hipEventRecord(t0, sa)); CK(hipEventRecord(t2, sb));
            CK(hipGraphLaunch(ga, sa)); CK(hipGraphLaunch(gb, sb));
            CK(hipEventRecord(t1, sa));
            CK(hipDeviceSynchronize());
            float a; CK(hipEventElapsedTime(&a, t0, t1));
            CK(hipEventRecord(t1, sb)); CK(hipDeviceSynchronize());
            float both; CK(hipEventElapsedTime(&both, t0, t1));
            printf("%d branch(es) per graph, 40 links of 10 us: one graph %.0f us; two graphs on two streams: the first done after %.0f us, both after %.0f us\n", branches, one * 1e3, a * 1e3, both * 1e3);
        }
    }
    return 0;
}

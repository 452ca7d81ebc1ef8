// What slows a chain of short dependent kernels when ANOTHER queue has a kernel running?  (round 3: the panel steps of the factorization take
// 7.0 us alone and 9.5-10 us while k_tail / k_node of the scalar branch run beside them, 21 us beside k_xpass.)
//   chain: 16 dependent launches of `wgs` workgroups x 256 threads, each workgroup busy for `busy_us` (ALU spin, or a pointer chase through
//          memory when mem = 1), replayed from a captured graph;
//   side kernel on a second stream: nB workgroups x tB threads spinning for the whole time -- kind 0: s_sleep loop, 1: VALU loop, 2: streaming loads.
// hipcc --offload-arch=gfx950 -O3 -o interfere_probe tools/interfere_probe.hip && ./interfere_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_link(unsigned long long ticks, const int *chase, int mem, int *sink, int prio)
{
    if (prio) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int v = threadIdx.x + blockIdx.x * 256;
    if (mem) { for (int i = 0; i < 6; ++i) v = chase[v & 0xfffff]; }     // six dependent loads from a 4 MB table
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    if (v == -1) sink[0] = v;
}
__global__ void k_side(unsigned long long ticks, int kind, const double *buf, size_t nbuf, double *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double acc = threadIdx.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (kind == 0) __builtin_amdgcn_s_sleep(32);
        else if (kind == 1) { for (int k = 0; k < 64; ++k) acc = fma(acc, 1.0000001, 0.5); }
        else { for (int k = 0; k < 16; ++k) { acc += buf[i % nbuf]; i += (size_t)gridDim.x * blockDim.x; } }
    }
    if (acc == -1.0) out[0] = acc;
}
int main()
{
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    int *chase, *sink; double *buf, *out;
    const size_t nbuf = 32u << 20;
    CK(hipMalloc(&chase, (1 << 20) * sizeof(int))); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&buf, nbuf * 8)); CK(hipMalloc(&out, 8));
    std::vector<int> h(1 << 20);
    for (int i = 0; i < (1 << 20); ++i) h[i] = (int)(((long long)i * 2654435761ll + 12345) & 0xfffff);
    CK(hipMemcpy(chase, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemset(buf, 0, nbuf * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int cfgs[][4] = {{136, 1, 0, 0}, {136, 1, 1, 0}, {1096, 1, 0, 0}, {1096, 1, 1, 0}};     // chain: workgroups, mem, s_setprio 3 at the start of every wave
    for (auto &c : cfgs) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < 16; ++l) hipLaunchKernelGGL(k_link, dim3(c[0]), dim3(256), 0, sa, 500ull, chase, c[1], sink, c[2]);   // 5 us busy
        CK(hipStreamEndCapture(sa, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        struct { int nB, tB, kind; const char *what; } sides[] = {{0, 0, 0, "nothing beside"}, {1, 64, 0, "1 x 64 sleeping"}, {8, 1024, 0, "8 x 1024 sleeping"},
            {8, 1024, 1, "8 x 1024 VALU"}, {100, 64, 1, "100 x 64 VALU"}, {1280, 256, 2, "1280 x 256 streaming 256 MB"}};
        for (auto &sd : sides) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                if (sd.nB) hipLaunchKernelGGL(k_side, dim3(sd.nB), dim3(sd.tB), 0, sb, 40000ull, sd.kind, buf, nbuf, out);   // 400 us
                // let the side kernel get going
                hipLaunchKernelGGL(k_link, dim3(1), dim3(64), 0, sa, 2000ull, chase, 0, sink, 0);
                CK(hipEventRecord(e0, sa));
                CK(hipGraphLaunch(ge, sa));
                CK(hipEventRecord(e1, sa));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("chain %4d workgroups%s%s, 16 links of 5 us busy: %6.1f us  (%5.2f us per link)  beside: %s\n", c[0], c[1] ? " + 6 dependent loads" : "", c[2] ? ", s_setprio 3" : "", best * 1e3, best * 1e3 / 16, sd.what);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}

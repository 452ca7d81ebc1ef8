// Which resource keeps a small kernel from being placed beside a RESIDENT one-workgroup-per-CU kernel?  (notes r4 F: k_node does not start beside k_chol_df)
// hog: 256 workgroups x 256 threads (one per CU by its LDS), VG vector registers (forced by an asm clobber), spins for `dur` us.
// small: 100 workgroups x 64 threads on a second stream, VS vector registers, its own LDS; every workgroup records when it started.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/coreside_probe tools/coreside_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int VG, int AG = 0>
__global__ __launch_bounds__(256) void hog(unsigned long long *t, int dur_us, int prio)
{
    extern __shared__ double lds[];
    if (prio & 1) __builtin_amdgcn_s_setprio(3);
    if (AG == 72) asm volatile("v_accvgpr_write_b32 a71, 0" ::: "a71");
    if (AG == 40) asm volatile("v_accvgpr_write_b32 a39, 0" ::: "a39");
    if (VG > 200) asm volatile("v_mov_b32 v239, 0" ::: "v239");
    else if (VG > 150) asm volatile("v_mov_b32 v179, 0" ::: "v179");
    else if (VG > 100) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    else asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if ((prio & 2) && (blockIdx.x & 7) != 0) return;           // like k_chol_df with one chain: only the workgroups of XCD 0 stay
    lds[threadIdx.x] = (double)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
    if (prio & 4) {                                            // busy: dependent f64 arithmetic + LDS traffic + barriers, no sleep
        double a = lds[threadIdx.x];
        while (__builtin_amdgcn_s_memrealtime() - t0 < 100ull * (unsigned)dur_us) {
            for (int i = 0; i < 64; ++i) a = fma(a, 1.0000001, lds[(threadIdx.x + i) & 255]);
            __syncthreads();
        }
        if (a == 12345.0) t[3] = 1;
    } else
    while (__builtin_amdgcn_s_memrealtime() - t0 < 100ull * (unsigned)dur_us) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = __builtin_amdgcn_s_memrealtime();
    if (lds[threadIdx.x] < 0) t[2] = 1;
}
template <int VS>
__global__ __launch_bounds__(64) void small_k(unsigned long long *out)
{
    extern __shared__ double lds[];
    if (VS > 200) asm volatile("v_mov_b32 v239, 0" ::: "v239");
    else if (VS > 150) asm volatile("v_mov_b32 v179, 0" ::: "v179");
    else if (VS > 100) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    else asm volatile("v_mov_b32 v63, 0" ::: "v63");
    lds[threadIdx.x] = 1.0;
    if (threadIdx.x == 0) {
        unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        out[2 * blockIdx.x + 1] = v & 7u;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 100ull * 10) __builtin_amdgcn_s_sleep(2);     // 10 us of "work"
    if (lds[threadIdx.x] < 0) out[0] = 0;
}

template <int VG, int VS, int AG = 0>
static int run(int hog_lds, int small_lds, int prio, int hog_wg)
{
    unsigned long long *t, *out;
    CHK(hipMalloc(&t, 64)); CHK(hipMalloc(&out, 2 * 128 * 8));
    CHK(hipMemset(t, 0, 64)); CHK(hipMemset(out, 0, 2 * 128 * 8));
    hipStream_t s1, s2;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CHK(hipFuncSetAttribute((const void *)&hog<VG, AG>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CHK(hipFuncSetAttribute((const void *)&small_k<VS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CHK(hipDeviceSynchronize());
    const int N = 100;
    if (prio & 16) {                                 // two branches of CHAINS: branch 1 = nop -> hog -> nop, branch 2 = small -> small -> small (the last one's stamps are read)
        hipGraph_t graph; hipGraphExec_t gexec; hipEvent_t ef, ej;
        unsigned long long *out2; CHK(hipMalloc(&out2, 2 * 128 * 8));
        CHK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        CHK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(small_k<64>, dim3(1), dim3(64), 1024, s1, out2);                       // a common predecessor (the "back-projection")
        CHK(hipEventRecord(ef, s1)); CHK(hipStreamWaitEvent(s2, ef, 0));
        hipLaunchKernelGGL(small_k<64>, dim3(8), dim3(64), 1024, s1, out2);                       // branch 1: a short kernel in front of the hog (the "Gram")
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hog<VG, AG>), dim3(hog_wg), dim3(256), hog_lds, s1, t, 300, prio);
        hipLaunchKernelGGL(small_k<64>, dim3(8), dim3(64), 1024, s1, out2);                       // ... and one behind it (the "solve")
        hipLaunchKernelGGL(small_k<VS>, dim3(N), dim3(64), small_lds, s2, out2);                  // branch 2: "k_tail" (10 us)
        hipLaunchKernelGGL(small_k<VS>, dim3(N), dim3(64), small_lds, s2, out2);                  //           another 10 us
        hipLaunchKernelGGL(small_k<VS>, dim3(N), dim3(64), small_lds, s2, out);                   //           "k_node": when do its workgroups start?
        CHK(hipEventRecord(ej, s2)); CHK(hipStreamWaitEvent(s1, ej, 0));
        CHK(hipStreamEndCapture(s1, &graph));
        CHK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) { CHK(hipGraphLaunch(gexec, s1)); CHK(hipDeviceSynchronize()); }
        CHK(hipGraphExecDestroy(gexec)); CHK(hipGraphDestroy(graph)); (void)hipFree(out2);
    } else if (prio & 8) {                                  // the same pair as two branches of a captured graph
        hipGraph_t graph; hipGraphExec_t gexec; hipEvent_t ef, ej;
        CHK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        CHK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        CHK(hipEventRecord(ef, s1)); CHK(hipStreamWaitEvent(s2, ef, 0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hog<VG, AG>), dim3(hog_wg), dim3(256), hog_lds, s1, t, 300, prio);
        hipLaunchKernelGGL(small_k<VS>, dim3(N), dim3(64), small_lds, s2, out);
        CHK(hipEventRecord(ej, s2)); CHK(hipStreamWaitEvent(s1, ej, 0));
        CHK(hipStreamEndCapture(s1, &graph));
        CHK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) { CHK(hipGraphLaunch(gexec, s1)); CHK(hipDeviceSynchronize()); }
        CHK(hipGraphExecDestroy(gexec)); CHK(hipGraphDestroy(graph));
    } else
    for (int rep = 0; rep < 2; ++rep) {              // (the first repetition warms up)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(hog<VG, AG>), dim3(hog_wg), dim3(256), hog_lds, s1, t, 300, prio);
        hipLaunchKernelGGL(small_k<VS>, dim3(N), dim3(64), small_lds, s2, out);
        CHK(hipDeviceSynchronize());
    }
    unsigned long long ht[2], ho[2 * 128];
    CHK(hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost)); CHK(hipMemcpy(ho, out, 2 * N * 8, hipMemcpyDeviceToHost));
    int during = 0, after = 0;
    double latest = 0, latest0 = -1e9, latestx = -1e9;
    for (int i = 0; i < N; ++i) {
        const double rel = ((double)ho[2 * i] - (double)ht[0]) / 100.0;
        if (ho[2 * i] < ht[1]) ++during; else ++after;
        latest = std::max(latest, rel);
        if (ho[2 * i + 1] == 0) latest0 = std::max(latest0, rel); else latestx = std::max(latestx, rel);
    }
    if (prio & 16) printf("   chains of kernels: last small kernel's workgroups on XCD 0 start at %+.1f us at the latest, on the other XCDs at %+.1f us (hog: 0 .. 300 us)\n", latest0, latestx);
    printf("hog %3d wg x %3d VGPR + %2d AGPR x %6d B LDS prio %d | small %3d VGPR x %6d B LDS : %3d of %d small workgroups started while the hog ran (300 us), %3d after; latest start at %+.1f us\n",
           hog_wg, VG, AG, hog_lds, prio, VS, small_lds, during, N, after, latest);
    (void)hipFree(t); (void)hipFree(out); (void)hipStreamDestroy(s1); (void)hipStreamDestroy(s2);
    return 0;
}
int main()
{
    run<180, 240>(95 * 1024, 41 * 1024, 0, 256);
    run<180, 240>(95 * 1024, 41 * 1024, 1, 256);
    run<180, 240>(64 * 1024, 41 * 1024, 0, 256);
    run<180, 240>(32 * 1024, 41 * 1024, 0, 256);
    run<180, 128>(95 * 1024, 41 * 1024, 0, 256);
    run<128, 240>(95 * 1024, 41 * 1024, 0, 256);
    run<128, 128>(95 * 1024, 41 * 1024, 0, 256);
    run<64, 64>(95 * 1024, 41 * 1024, 0, 256);
    run<64, 64>(95 * 1024, 8 * 1024, 0, 256);
    run<180, 240>(95 * 1024, 8 * 1024, 0, 256);
    run<180, 240>(95 * 1024, 41 * 1024, 0, 32);
    run<240, 240>(95 * 1024, 41 * 1024, 0, 256);
    run<180, 240, 72>(95 * 1024, 41 * 1024, 0, 256);      // k_chol_df's allocation: 180 + 72 = 252 -> 256 registers
    run<180, 240, 40>(95 * 1024, 41 * 1024, 0, 256);
    run<180, 128, 72>(95 * 1024, 41 * 1024, 0, 256);
    printf("prio bit 1: only the hog workgroups of XCD 0 stay (32, one per CU of that XCD); bit 2: the hog is busy (f64 + LDS + barriers) instead of sleeping\n");
    run<180, 240, 72>(95 * 1024, 41 * 1024, 2, 256);
    run<180, 240, 72>(95 * 1024, 41 * 1024, 6, 256);
    run<180, 240, 72>(95 * 1024, 41 * 1024, 4, 256);
    run<180, 240, 72>(95 * 1024, 41 * 1024, 7, 256);
    printf("prio bit 3: the pair as two branches of a captured hipGraph\n");
    run<180, 240, 72>(95 * 1024, 41 * 1024, 8, 256);
    run<180, 240, 72>(95 * 1024, 41 * 1024, 8 + 2, 256);
    run<180, 240, 72>(64 * 1024, 41 * 1024, 8, 256);
    run<180, 240, 72>(32 * 1024, 41 * 1024, 8, 256);
    run<180, 240, 72>(95 * 1024, 8 * 1024, 8, 256);
    run<64, 64>(32 * 1024, 8 * 1024, 8, 256);
    run<128, 128>(95 * 1024, 41 * 1024, 8, 256);
    run<180, 128, 72>(95 * 1024, 41 * 1024, 8, 256);
    printf("prio bit 4: two branches of kernel CHAINS in a captured graph (bit 1: the hog's workgroups stay on XCD 0 only)\n");
    run<180, 240, 72>(95 * 1024, 41 * 1024, 16, 256);
    run<180, 240, 72>(95 * 1024, 41 * 1024, 16 + 2, 256);
    run<64, 64>(8 * 1024, 8 * 1024, 16 + 2, 256);
    run<64, 64>(8 * 1024, 8 * 1024, 16 + 2, 8);
    return 0;
}

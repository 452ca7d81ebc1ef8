// Which compute units does a bit of hipExtStreamCreateWithCUMask's mask enable?  For a few masks: a census kernel on the masked stream
// reports the (XCC, SE, cu id) set it ran on.  Also: does a captured graph launched ON a masked stream keep the mask for the kernels of
// its origin branch, and where do the kernels of a forked branch run?
// usage: mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
__device__ __forceinline__ unsigned hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }
__global__ void census(unsigned *out, int spin)
{
    if (threadIdx.x == 0) {
        const unsigned hw = hw_id(), xcc = xcc_id();
        atomicOr(&out[xcc * 4 + ((hw >> 13) & 3u)], 1u << ((hw >> 8) & 15u));
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(4);
    }
}
static void show(const char *what, unsigned *d)
{
    unsigned h[32];
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int total = 0;
    printf("%-44s", what);
    for (int x = 0; x < 8; ++x) { printf(" |"); for (int s = 0; s < 4; ++s) { printf(" %03x", h[x * 4 + s]); total += __builtin_popcount(h[x * 4 + s]); } }
    printf(" | CUs %d\n", total);
}
int main()
{
    unsigned *d; (void)hipMalloc(&d, 32 * 4);
    hipStream_t plain; (void)hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    (void)hipMemset(d, 0, 128); hipLaunchKernelGGL(census, dim3(4096), dim3(1024), 0, plain, d, 2000); (void)hipDeviceSynchronize();
    show("no mask (cu-id bits per XCC | SE0 SE1 SE2 SE3)", d);
    struct { const char *name; uint32_t word; } masks[] = {{"0x000000ff", 0x000000ffu}, {"0x0000ff00", 0x0000ff00u}, {"0x01010101", 0x01010101u}, {"0x0000000f", 0x0000000fu},
                                                            {"0x7f7f7f7f", 0x7f7f7f7fu}, {"0x0fffffff", 0x0fffffffu}, {"0x77777777", 0x77777777u}, {"0xeeeeeeee", 0xeeeeeeeeu}};
    for (auto &m : masks) {
        std::vector<uint32_t> w(8, m.word);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, w.data()) != hipSuccess) { printf("mask %s: create failed\n", m.name); continue; }
        (void)hipMemset(d, 0, 128); hipLaunchKernelGGL(census, dim3(4096), dim3(1024), 0, s, d, 2000); (void)hipDeviceSynchronize();
        show((std::string("mask ") + m.name + " in every word").c_str(), d);
        (void)hipStreamDestroy(s);
    }
    {
        // per-word masks: drop CU index 7 / indices 6 and 7 of every shader engine
        uint32_t m1[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0u}, m2[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0u, 0u}, m3[8] = {0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
        const char *names[3] = {"words 0..6 ones, word 7 zero", "words 0..5 ones, words 6, 7 zero", "word 0 zero, words 1..7 ones"};
        uint32_t *ms[3] = {m1, m2, m3};
        for (int k = 0; k < 3; ++k) {
            hipStream_t s;
            if (hipExtStreamCreateWithCUMask(&s, 8, ms[k]) != hipSuccess) { printf("%s: create failed\n", names[k]); continue; }
            (void)hipMemset(d, 0, 128); hipLaunchKernelGGL(census, dim3(4096), dim3(1024), 0, s, d, 2000); (void)hipDeviceSynchronize();
            show(names[k], d);
            (void)hipStreamDestroy(s);
        }
    }
    // graph launched on a masked stream: origin-branch kernel vs forked-branch kernel
    {
        std::vector<uint32_t> w(8, 0x0000ffffu);
        hipStream_t sm, sf; (void)hipExtStreamCreateWithCUMask(&sm, 8, w.data()); (void)hipStreamCreateWithFlags(&sf, hipStreamNonBlocking);
        unsigned *d2; (void)hipMalloc(&d2, 128);
        hipEvent_t ef, ej; (void)hipEventCreateWithFlags(&ef, hipEventDisableTiming); (void)hipEventCreateWithFlags(&ej, hipEventDisableTiming);
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(sm, hipStreamCaptureModeThreadLocal);
        (void)hipEventRecord(ef, sm); (void)hipStreamWaitEvent(sf, ef, 0);
        hipLaunchKernelGGL(census, dim3(4096), dim3(1024), 0, sf, d2, 2000);            // forked branch
        hipLaunchKernelGGL(census, dim3(4096), dim3(1024), 0, sm, d, 2000);             // origin branch
        (void)hipEventRecord(ej, sf); (void)hipStreamWaitEvent(sm, ej, 0);
        hipError_t e1 = hipStreamEndCapture(sm, &g), e2 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        printf("capture on a masked stream: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
        (void)hipMemset(d, 0, 128); (void)hipMemset(d2, 0, 128);
        (void)hipGraphLaunch(ge, sm); (void)hipDeviceSynchronize();
        show("graph on stream masked 0x0000ffff: origin", d); show("graph on stream masked 0x0000ffff: fork", d2);
        (void)hipMemset(d, 0, 128); (void)hipMemset(d2, 0, 128);
        (void)hipGraphLaunch(ge, plain); (void)hipDeviceSynchronize();
        show("the same graph launched on a plain stream: origin", d); show("the same graph launched on a plain stream: fork", d2);
    }
    return 0;
}

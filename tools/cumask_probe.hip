// Does a latency-bound kernel chain keep its speed beside an MFMA-saturating kernel when the two run on DISJOINT CU sets
// (hipExtStreamCreateWithCUMask)?  A: G workgroups x 512 threads of back-to-back f64 MFMAs (~30 us each), two per CU.
// B: chain of NB kernels, 68 workgroups x 256 threads, wave 0 runs a dependent f64 FMA/rsq chain (~5 us).
// usage: cumask_probe <lat_cus_per_xcd> <mask 0|1>   -- with mask 1, A's stream gets CUs [0, 32-l) of every XCD, B's stream the last l
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 2) void kA(int iters, unsigned long long *t, double *sink)
{
    d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3;
    if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678) sink[0] = 1.0;
    if (threadIdx.x == 0) atomicMax(&t[1], __builtin_amdgcn_s_memrealtime());
}
__global__ __launch_bounds__(256) void kB(int iters, unsigned long long *t, int i, double *sink, unsigned int *where)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) t[2 + 2 * i] = t0;
    if (threadIdx.x < 64) {
        double x = 1.0 + threadIdx.x * 1e-6, acc = 0.0;
        for (int k = 0; k < iters; ++k) { double y = __builtin_amdgcn_rsq(x); x = fma(y, y, x) * 0.5 + 0.7; acc = fma(x, y, acc); }
        if (acc == 12345.678) sink[1] = acc;
    }
    if (threadIdx.x == 0) {
        unsigned int hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (i == 0) where[blockIdx.x] = (hw & 0xffffu) | (xcc << 16);
        atomicMax(&t[3 + 2 * i], __builtin_amdgcn_s_memrealtime());
    }
}
int main(int argc, char **argv)
{
    const int l = argc > 1 ? atoi(argv[1]) : 8, use_mask = argc > 2 ? atoi(argv[2]) : 1, NB = 12;
    hipStream_t s1, s2;
    if (use_mask) {
        // 256 CUs = 8 XCDs x 32: mask words cover CUs in device order; try "the last l CUs of every group of 32" for B
        std::vector<uint32_t> ma(8, 0), mb(8, 0);
        for (int x = 0; x < 8; ++x) { mb[x] = l >= 32 ? 0xffffffffu : (((1u << l) - 1u) << (32 - l)); ma[x] = ~mb[x]; }
        hipError_t e1 = hipExtStreamCreateWithCUMask(&s1, 8, ma.data()), e2 = hipExtStreamCreateWithCUMask(&s2, 8, mb.data());
        printf("masked streams: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
        if (e1 != hipSuccess || e2 != hipSuccess) return 1;
    } else {
        (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
        (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    }
    unsigned long long *t; double *sink; unsigned int *where;
    (void)hipMalloc(&t, 64 * 8); (void)hipMalloc(&sink, 64); (void)hipMalloc(&where, 4 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        for (int withA = 0; withA < 2; ++withA) {
            (void)hipMemset(t, 0, 64 * 8);
            (void)hipDeviceSynchronize();
            if (withA) hipLaunchKernelGGL(kA, dim3(2048), dim3(512), 0, s1, 1000, t, sink);
            for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(kB, dim3(68), dim3(256), 0, s2, 300, t, i, sink, where);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h(64);
            (void)hipMemcpy(h.data(), t, 64 * 8, hipMemcpyDeviceToHost);
            double first = (double)h[2], last = (double)h[3 + 2 * (NB - 1)];
            printf("rep %d %s A: B chain of %d kernels took %.1f us (%.2f us each)", rep, withA ? "with   " : "without", NB, (last - first) / 100.0, (last - first) / 100.0 / NB);
            if (withA) printf("; A ran %.1f us", (h[1] - h[0]) / 100.0);
            printf("\n");
        }
    }
    std::vector<unsigned int> w(68);
    (void)hipMemcpy(w.data(), where, 68 * 4, hipMemcpyDeviceToHost);
    printf("B workgroups landed on (xcc:se:cu):");
    for (int i = 0; i < 68; i += 4) printf(" %u:%u:%u", w[i] >> 16, (w[i] >> 13) & 7, (w[i] >> 8) & 15);
    printf("\n");
    return 0;
}

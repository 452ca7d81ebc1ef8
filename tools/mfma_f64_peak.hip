// Issue-rate microbenchmark of v_mfma_f64_16x16x4_f64 on gfx950: the measured denominator quoted beside AMD's
// 78.6 TFLOP/s FP64-matrix spec.  Sweeps waves per SIMD and independent accumulators per wave; reports wall TFLOP/s,
// the in-kernel shader clock (s_memtime / s_memrealtime) and shader cycles per MFMA per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak tools/mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *clk, int iters, double a0, double b0)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC>
void run(int waves_per_simd, int iters)
{
    int blocks = 256 * waves_per_simd;
    double *out; unsigned long long *clk;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    (void)hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 1.0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0, 1.0);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> ghz(blocks), cyc(blocks);
    for (int b = 0; b < blocks; ++b) { ghz[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1; cyc[b] = (double)h[2 * b]; }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    double cyc_per_mfma = cyc[blocks / 2] / ((double)iters * NACC * waves_per_simd);
    printf("waves/SIMD=%d NACC=%2d : %8.3f ms  %6.2f TFLOP/s  clock %.2f GHz  %.1f shader cycles per MFMA per SIMD\n", waves_per_simd, NACC, ms,
           flops / ms / 1e9, ghz[blocks / 2], cyc_per_mfma);
    (void)hipFree(out); (void)hipFree(clk);
}
int main()
{
    const int it = 20000;
    run<1>(1, it); run<2>(1, it); run<4>(1, it); run<8>(1, it); run<16>(1, it);
    run<1>(2, it); run<2>(2, it); run<4>(2, it); run<8>(2, it);
    run<1>(4, it); run<2>(4, it); run<4>(4, it); run<8>(4, it);
    run<1>(8, it); run<2>(8, it); run<4>(8, it);
    return 0;
}

// Issue-rate microbenchmark of v_mfma_f64_16x16x4_f64 on gfx950: the denominator used beside AMD's 78.6 TF/s spec.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak tools/mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters)
{
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("NACC=%2d blocks=%4d iters=%d : %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz, 1 wave/SIMD-equivalent)\n", NACC, blocks, iters, ms,
           flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * (blocks / 256.0)));
    hipFree(out);
}
int main()
{
    run<1>(256, 20000);
    run<4>(256, 20000);
    run<16>(256, 20000);
    run<16>(512, 20000);
    run<4>(1024, 20000);
    return 0;
}

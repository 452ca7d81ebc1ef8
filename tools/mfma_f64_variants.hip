// Which ingredient of the Gram inner loop slows the f64 MFMA pipe?  4 waves/SIMD, 4 accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, const double *in, int iters)
{
    __shared__ double lds[4096];
    d4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = in[i];
    __syncthreads();
    double a0 = in[lane], a1 = in[64 + lane], b0 = in[128 + lane], b1 = in[192 + lane];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) { a0 = a0 * 1.0000001; a1 = a1 * 1.0000001; b0 = b0 * 0.9999999; b1 = b1 * 0.9999999; }
        if (MODE == 2) { int o = ((it & 7) * 256 + lane) ; a0 = lds[o]; a1 = lds[o + 64]; b0 = lds[o + 128]; b1 = lds[o + 192]; }
        if (MODE == 3) { int o = ((it & 7) * 256 + lane) ; a0 = in[o]; a1 = in[o + 64]; b0 = in[o + 128]; b1 = in[o + 192]; }
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c3, 0, 0, 0);
        if (MODE == 4 && (it & 1)) __syncthreads();
    }
    out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int MODE> void run(const char *name, double *out, double *in, int iters)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, in, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, in, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double nm = (double)iters * 4 * 4;   // MFMAs per SIMD
    printf("%-34s %8.3f ms  %6.2f TFLOP/s  %.1f cycles/MFMA/SIMD @2.4GHz\n", name, ms, 256.0 * 16 * iters * 4 * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / nm);
}
int main()
{
    double *out, *in; (void)hipMalloc(&out, 8 * 256 * 1024); (void)hipMalloc(&in, 8 * 4096);
    double h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 0.5 + (i % 97) * 0.01;
    (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    run<0>("constant operands", out, in, 20000);
    run<1>("operands updated by VALU", out, in, 20000);
    run<2>("operands from LDS", out, in, 20000);
    run<3>("operands from global (L1/L2)", out, in, 20000);
    run<4>("constant + barrier per 8 MFMA", out, in, 20000);
    return 0;
}

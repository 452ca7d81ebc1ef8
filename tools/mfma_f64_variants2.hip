// Step-by-step replica of the k_gram inner loop to find what slows the f64 MFMA pipe (4 waves/SIMD, 4 accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define CS 80
#define PANEL (8 * CS)
template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, const double *in, int iters)
{
    __shared__ double sred[4 * 64 * 64];
    d4 c00 = {0,0,0,0}, c01 = c00, c10 = c00, c11 = c00;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kg = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int tg = threadIdx.x & 255, c = tg >> 5, rp = tg & 31;
    double *stg = sred + kg * (4 * PANEL);
    const int woff = c * CS + 2 * rp;
    for (int i = threadIdx.x; i < 4 * 64 * 64; i += 1024) sred[i] = in[i & 4095];
    __syncthreads();
    d2 r0 = *(const d2 *)(in + 2 * tg), r1 = *(const d2 *)(in + 512 + 2 * tg);
    for (int b = 0; b < iters; ++b) {
        const double *bufI = stg + (b & 1) * (2 * PANEL), *bufJ = bufI + PANEL;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int kk = (4 * k2 + lk) * CS;
            double a0 = bufJ[kk + wj * 32 + li], a1 = bufJ[kk + wj * 32 + 16 + li];
            double b0 = bufI[kk + wi * 32 + li], b1 = bufI[kk + wi * 32 + 16 + li];
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
        }
        if (MODE >= 1) {
            double *nxt = stg + ((b + 1) & 1) * (2 * PANEL);
            *(d2 *)(nxt + woff) = r0 * 0.999;
            *(d2 *)(nxt + PANEL + woff) = r1;
        }
        if (MODE >= 2) { r0 = *(const d2 *)(in + ((b * 1024 + 2 * tg) & 4094)); r1 = *(const d2 *)(in + ((b * 1024 + 512 + 2 * tg) & 4094)); }
        __syncthreads();
    }
    out[blockIdx.x * 1024 + threadIdx.x] = c00[0] + c01[1] + c10[2] + c11[3];
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
    double nm = (double)iters * 8 * 4;
    printf("%-44s %8.3f ms  %6.2f TFLOP/s  %.1f cycles/MFMA/SIMD @2.4GHz\n", name, ms, 256.0 * 16 * iters * 8 * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / nm);
}
int main()
{
    double *out, *in; (void)hipMalloc(&out, 8 * 256 * 1024); (void)hipMalloc(&in, 8 * 4096);
    double h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (i % 3 == 0) ? 0.0 : 0.5 + (i % 97) * 0.01;
    (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    run<0>("LDS fragments (stride 80) + barrier/8 MFMA", out, in, 10000);
    run<1>(" + LDS staging writes", out, in, 10000);
    run<2>(" + global loads", out, in, 10000);
    run<0>("same, 23 iterations (kernel length)", out, in, 23);
    run<2>("full, 23 iterations", out, in, 23);
    return 0;
}

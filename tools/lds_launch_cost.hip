// Launch cost of a near-empty 1024-thread kernel as a function of its static LDS size and of its VGPR allocation.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB, int NREG>
__global__ __launch_bounds__(1024) void k(double *out, int n)
{
    __shared__ double lds[KB * 128];
    double r[NREG];
    for (int i = 0; i < NREG; ++i) r[i] = out[(threadIdx.x + i) & 1023];
    lds[threadIdx.x] = r[0];
    __syncthreads();
    double s = lds[(threadIdx.x * 7) & 1023];
    for (int i = 0; i < NREG; ++i) s += r[i] * (i + 1);
    if (n == 12345) out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int KB, int NREG> void run(double *out, int blocks)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<KB, NREG>), dim3(blocks), dim3(1024), 0, 0, out, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 100; ++r) hipLaunchKernelGGL((k<KB, NREG>), dim3(blocks), dim3(1024), 0, 0, out, 0);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("LDS %3d KiB, ~%2d live doubles/thread, %d blocks: %.2f us per launch\n", KB, NREG, blocks, ms * 10.0);
}
int main()
{
    double *out; (void)hipMalloc(&out, 8 * 256 * 1024);
    run<8, 4>(out, 252); run<32, 4>(out, 252); run<64, 4>(out, 252); run<65, 4>(out, 252); run<96, 4>(out, 252); run<128, 4>(out, 252); run<160, 4>(out, 252);
    run<8, 40>(out, 252); run<128, 40>(out, 252); run<64, 40>(out, 504); run<128, 4>(out, 64);
    return 0;
}

// Can small kernels on a second stream run WHILE a many-round kernel that leaves half of every CU free is executing?
// A: G workgroups x 512 threads, LDS 81 KiB (one per CU), each spinning `a_us`; B: chain of NB kernels of 64 x 256 threads, each `b_us`,
// launched on stream 2 right after A.  Prints when each B kernel started/ended relative to A's start (100 MHz realtime ticks -> us).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void kA(unsigned long long ticks, unsigned long long *t)
{
    extern __shared__ double lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = t0;
    lds[threadIdx.x] = (double)t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(&t[1], __builtin_amdgcn_s_memrealtime());
}
__global__ void kB(unsigned long long ticks, unsigned long long *t, int i)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) t[2 + 2 * i] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(&t[3 + 2 * i], __builtin_amdgcn_s_memrealtime());
}
int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 2048, lds = argc > 2 ? atoi(argv[2]) : 81 * 1024, NB = 12;
    const int bthreads = argc > 3 ? atoi(argv[3]) : 256;
    hipFuncSetAttribute((const void *)kA, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    unsigned long long *t;
    hipMalloc(&t, 64 * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(t, 0, 64 * 8);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(kA, dim3(G), dim3(512), lds, s1, 3000ull, t);             // 30 us per workgroup
        for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(kB, dim3(64), dim3(bthreads), 0, s2, 500ull, t, i);   // 5 us each
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(64);
        hipMemcpy(h.data(), t, 64 * 8, hipMemcpyDeviceToHost);
        printf("rep %d: A ran %.1f us (G=%d, lds=%d);  B kernels (start..end us after A's start):", rep, (h[1] - h[0]) / 100.0, G, lds);
        for (int i = 0; i < NB; ++i) printf(" %.0f..%.0f", ((double)h[2 + 2 * i] - (double)h[0]) / 100.0, ((double)h[3 + 2 * i] - (double)h[0]) / 100.0);
        printf("\n");
    }
    return 0;
}

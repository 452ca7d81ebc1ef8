// LDS round-trip times seen by ONE wave (shader cycles, s_memtime): what a poll / a batch of broadcast reads costs the panel pipeline (round 6, notes B).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_rtt tools/lds_rtt.hip && tools/bin/lds_rtt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void *)p; }
__global__ __launch_bounds__(256) void k(unsigned long long *out, int busy)
{
    __shared__ double s[32 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) s[i] = 1.0 + i;
    __syncthreads();
    if (wave != 0) {
        if (!busy) return;
        // the other three waves poll like the pipeline's consumers do
        double acc = 0;
        const unsigned a = lds_addr(&s[lane]), m = lds_addr(&s[8 * wave]);
        for (int it = 0; it < 3000; ++it) {
            double c0, c1; d2 q[8];
            asm volatile("ds_read_b64 %0, %10\n\tds_read_b64 %1, %10 offset:512\n\tds_read_b128 %2, %11\n\tds_read_b128 %3, %11 offset:16\n\tds_read_b128 %4, %11 offset:32\n\tds_read_b128 %5, %11 offset:48\n\t"
                         "ds_read_b128 %6, %11 offset:512\n\tds_read_b128 %7, %11 offset:528\n\tds_read_b128 %8, %11 offset:544\n\tds_read_b128 %9, %11 offset:560\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(c0), "=&v"(c1), "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]) : "v"(a), "v"(m) : "memory");
            acc += c0 + q[7][1];
        }
        if (acc == 123.0) out[100] = 1;
        return;
    }
    const unsigned a = lds_addr(&s[lane]), m = lds_addr(&s[8]);
    unsigned long long t[8];
    double sink = 0;
    // 0: one b64 per-lane read
    t[0] = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) { double c; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c) : "v"(a) : "memory"); sink += c; }
    t[1] = __builtin_amdgcn_s_memtime();
    // 1: one b64 uniform-address read
    for (int it = 0; it < 64; ++it) { double c; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c) : "v"(m) : "memory"); sink += c; }
    t[2] = __builtin_amdgcn_s_memtime();
    // 2: 2 x b64 per lane + 8 x b128 uniform (the pipeline's trip)
    for (int it = 0; it < 64; ++it) {
        double c0, c1; d2 q[8];
        asm volatile("ds_read_b64 %0, %10\n\tds_read_b64 %1, %10 offset:512\n\tds_read_b128 %2, %11\n\tds_read_b128 %3, %11 offset:16\n\tds_read_b128 %4, %11 offset:32\n\tds_read_b128 %5, %11 offset:48\n\t"
                     "ds_read_b128 %6, %11 offset:512\n\tds_read_b128 %7, %11 offset:528\n\tds_read_b128 %8, %11 offset:544\n\tds_read_b128 %9, %11 offset:560\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(c0), "=&v"(c1), "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]) : "v"(a), "v"(m) : "memory");
        sink += c0 + q[3][0];
    }
    t[3] = __builtin_amdgcn_s_memtime();
    // 3: 2 x b64 per lane + 16 x b64 uniform
    for (int it = 0; it < 64; ++it) {
        double c0, c1, q[16];
        asm volatile("ds_read_b64 %0, %18\n\tds_read_b64 %1, %18 offset:512\n\t"
                     "ds_read_b64 %2, %19\n\tds_read_b64 %3, %19 offset:8\n\tds_read_b64 %4, %19 offset:16\n\tds_read_b64 %5, %19 offset:24\n\tds_read_b64 %6, %19 offset:32\n\tds_read_b64 %7, %19 offset:40\n\tds_read_b64 %8, %19 offset:48\n\tds_read_b64 %9, %19 offset:56\n\t"
                     "ds_read_b64 %10, %19 offset:512\n\tds_read_b64 %11, %19 offset:520\n\tds_read_b64 %12, %19 offset:528\n\tds_read_b64 %13, %19 offset:536\n\tds_read_b64 %14, %19 offset:544\n\tds_read_b64 %15, %19 offset:552\n\tds_read_b64 %16, %19 offset:560\n\tds_read_b64 %17, %19 offset:568\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(c0), "=&v"(c1), "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]), "=&v"(q[8]), "=&v"(q[9]), "=&v"(q[10]), "=&v"(q[11]), "=&v"(q[12]), "=&v"(q[13]), "=&v"(q[14]), "=&v"(q[15])
                     : "v"(a), "v"(m) : "memory");
        sink += c0 + q[15];
    }
    t[4] = __builtin_amdgcn_s_memtime();
    // 4: write then read back the own value (write -> visible round trip)
    for (int it = 0; it < 64; ++it) { double c; asm volatile("ds_write_b64 %1, %2\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c) : "v"(a), "v"(sink) : "memory"); sink += c; }
    t[5] = __builtin_amdgcn_s_memtime();
    // 5: 32 v_readlane_b32 + 16 v_fma_f64 (the other way to get the multipliers)
    double acc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int it = 0; it < 64; ++it) {
        double mm[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { int lo = __builtin_amdgcn_readlane(__double2loint(sink), 8 + c), hi = __builtin_amdgcn_readlane(__double2hiint(sink), 8 + c); mm[c] = __hiloint2double(hi, lo); }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = fma(-sink, mm[c], acc[c]);
    }
    t[6] = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < 8; ++c) sink += acc[c];
    if (lane == 0) { for (int i = 0; i < 6; ++i) out[i] = (t[i + 1] - t[i]) / 64; out[50] = (unsigned long long)sink; }
}
int main()
{
    unsigned long long *d, h[8];
    hipMalloc(&d, 1024);
    for (int busy = 0; busy < 2; ++busy) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, busy);
        hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%s: b64 per-lane %llu | b64 uniform %llu | 2 b64 + 8 b128 uniform %llu | 2 b64 + 16 b64 uniform %llu | write + read back %llu | 16 readlane pairs + 8 fma %llu   (cycles per round trip)\n",
               busy ? "three waves polling beside" : "alone", h[0], h[1], h[2], h[3], h[4], h[5]);
    }
    return 0;
}

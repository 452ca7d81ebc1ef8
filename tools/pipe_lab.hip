// Stand-alone timing + check of the panel sweep of the factorization (csrc/bnr_kernels.h: bnr_panel_sweep_pipe, bnr_panel_sweep): ONE workgroup alone on the chip sweeps a random
// 64 x 32 panel [D ; B] (D SPD) `reps` times; cycles from the function's entry to the LAST wave's end (s_memtime), result checked against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form [-DBNR_PANEL_PIPE=0 | -D<variant switches>] -I bayesiannetworkregression.jl_amd/csrc -o tools/bin/pipe_lab tools/pipe_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "bnr_kernels.h"
#if BNR_PANEL_PIPE
typedef bnr_panelp_lds bnr_step_lds;
#else
typedef bnr_panel_lds bnr_step_lds;
#endif
__global__ __launch_bounds__(256, 1) void k_lab(const double *D, const double *B, double *out, unsigned long long *cyc, int reps)
{
    __shared__ bnr_step_lds sh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, mt = wave >> 1, nt = wave & 1, ln = lane & 15, lq = lane >> 4;
    bnr_d4 cD, cB;
    for (int r = 0; r < 4; ++r) { cD[r] = D[(nt * 16 + ln) + 32 * (mt * 16 + lq + 4 * r)]; cB[r] = B[(nt * 16 + ln) + 32 * (mt * 16 + lq + 4 * r)]; }
    unsigned long long tot = 0, worst = 0, totw[4] = {0, 0, 0, 0}, tots[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    __shared__ unsigned long long s_end, s_endw[4];
    for (int rep = 0; rep < reps; ++rep) {
        if (tid == 0) s_end = 0;
#ifdef BNR_LAB_COLD_ICACHE       // every repetition starts with an empty instruction cache, as every launch of the product kernel does (the code of a panel step runs once per launch)
        asm volatile("s_icache_inv\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" ::: "memory");
#endif
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#if BNR_PANEL_PIPE
        int bad = bnr_panel_sweep_pipe(sh, cD, cB, tid, out, 32);
#else
        int bad = bnr_panel_sweep(sh, cD, cB, tid, out, 32);
#endif
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { atomicMax(&s_end, t1 - t0); s_endw[wave] = t1 - t0; }
        __syncthreads();
        if (tid == 0) { tot += s_end; worst = s_end > worst ? s_end : worst; if (bad) cyc[3] = 1; for (int w = 0; w < 4; ++w) totw[w] += s_endw[w];
#ifdef BNR_LAB_STAMPS
            for (int i = 0; i < 8; ++i) tots[i] += bnr_lab_stamp[i] - t0;
#endif
        }
        __syncthreads();
    }
    if (tid == 0) { cyc[0] = tot / reps; cyc[1] = worst; for (int w = 0; w < 4; ++w) cyc[4 + w] = totw[w] / reps; for (int i = 0; i < 8; ++i) cyc[8 + i] = tots[i] / reps; }
}
int main()
{
    std::mt19937_64 g(5);
    std::normal_distribution<double> N(0, 1);
    std::vector<double> A(32 * 40), D(32 * 32), B(32 * 32), L(32 * 32, 0.0), X(32 * 32);
    for (auto &v : A) v = N(g);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = (i == j) ? 1.0 : 0.0; for (int k = 0; k < 40; ++k) s += A[i * 40 + k] * A[j * 40 + k]; D[i + 32 * j] = s; }
    for (auto &v : B) v = N(g);
    for (int j = 0; j < 32; ++j) {                                                       // host Cholesky (column-major), then X = B L^-T
        double d = D[j + 32 * j]; for (int k = 0; k < j; ++k) d -= L[j + 32 * k] * L[j + 32 * k];
        L[j + 32 * j] = std::sqrt(d);
        for (int i = j + 1; i < 32; ++i) { double s = D[i + 32 * j]; for (int k = 0; k < j; ++k) s -= L[i + 32 * k] * L[j + 32 * k]; L[i + 32 * j] = s / L[j + 32 * j]; }
    }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = B[i + 32 * j]; for (int k = 0; k < j; ++k) s -= X[i + 32 * k] * L[j + 32 * k]; X[i + 32 * j] = s / L[j + 32 * j]; }
    double *dD, *dB, *dO; unsigned long long *dC, hC[16] = {0};
    hipMalloc(&dD, 8 * 1024); hipMalloc(&dB, 8 * 1024); hipMalloc(&dO, 8 * 1024); hipMalloc(&dC, 256);
    hipMemcpy(dD, D.data(), 8 * 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 8 * 1024, hipMemcpyHostToDevice); hipMemset(dC, 0, 256); hipMemset(dO, 0, 8 * 1024);
    for (int it = 0; it < 3; ++it) { hipLaunchKernelGGL(k_lab, dim3(1), dim3(256), 0, 0, dD, dB, dO, dC, 200); hipDeviceSynchronize(); }
    std::vector<double> O(32 * 32);
    hipMemcpy(O.data(), dO, 8 * 1024, hipMemcpyDeviceToHost); hipMemcpy(hC, dC, 128, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 1024; ++i) err = std::fmax(err, std::fabs(O[i] - X[i]) / (1e-12 + std::fabs(X[i])));
    printf("panel sweep: %llu cycles on average from entry to the last wave's end (worst %llu) over 200 sweeps; waves 0..3 end at %llu %llu %llu %llu; max relative error against the host %.2e; bad pivot flag %llu\n", hC[0], hC[1], hC[4], hC[5], hC[6], hC[7], err, hC[3]);
    // the same sweep ONCE per launch (as the product runs it: a panel step's code runs once per launch on a CU that was idle before), 1 and 48 workgroups, 40 launches each
    for (int wgs : {1, 48}) {
        unsigned long long acc[16] = {0};
        for (int it = 0; it < 40; ++it) {
            hipLaunchKernelGGL(k_lab, dim3(wgs), dim3(256), 0, 0, dD, dB, dO, dC, 1); hipDeviceSynchronize();
            unsigned long long h2[16]; hipMemcpy(h2, dC, 128, hipMemcpyDeviceToHost);
            if (it >= 8) for (int i = 0; i < 16; ++i) acc[i] += h2[i];
        }
        printf("one sweep per launch, %2d workgroup(s) (the last one to write its stamps is reported): %llu cycles; waves 0..3 end at %llu %llu %llu %llu; own pivots %llu..%llu  %llu..%llu  %llu..%llu  %llu..%llu\n", wgs, acc[0] / 32, acc[4] / 32, acc[5] / 32,
               acc[6] / 32, acc[7] / 32, acc[8] / 32, acc[9] / 32, acc[10] / 32, acc[11] / 32, acc[12] / 32, acc[13] / 32, acc[14] / 32, acc[15] / 32);
    }
#ifdef BNR_LAB_STAMPS
    printf("  own pivots of waves 0..3 start..end: %llu..%llu  %llu..%llu  %llu..%llu  %llu..%llu\n", hC[8], hC[9], hC[10], hC[11], hC[12], hC[13], hC[14], hC[15]);
#endif
    return err < 1e-9 ? 0 : 1;
}

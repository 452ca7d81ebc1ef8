// Issue-rate microbenchmark of v_mfma_f64_16x16x4_f64 on gfx950 (round 5 rewrite; VERDICT r4 "next 1a").
//
// What was wrong with the round-1 version: it assumed that a grid of 256 x W blocks of 256 threads puts exactly W waves
// on every SIMD.  It never checked, so a line whose kernel was not fully resident (8 waves per SIMD asked, fewer admitted)
// divided one wave's cycles by the wrong wave count: "55.4 cycles per MFMA" (below the architectural 64) and a wall
// figure that contained a tail round.  This version takes a census instead: every wave records HW_REG_HW_ID / HW_REG_XCC_ID
// and its start / end on both clocks (s_memtime = shader cycles, s_memrealtime = 100 MHz), and the host derives per SIMD
//   * the peak number of waves that were resident together,
//   * cycles per MFMA = (last end - first start of the SIMD's waves, in shader cycles) / MFMAs issued on that SIMD,
// and prints the wall-clock figure (HIP events) beside it.  A line is flagged when the two disagree by more than 5 % or
// when any SIMD reports less than 64 cycles per MFMA (16 passes x 4 cycles: 2048 flops / 64 cycles x 1024 SIMDs x 2.4 GHz
// = 78.6 TFLOP/s, the spec).
//
// Forms (what feeds the pipe):
//   0  builtin, one A / B register pair for all accumulators, accumulators in VGPRs
//   1  inline asm, accumulators in named AGPRs (a[0:7], a[8:15], ...), a different A / B pair per accumulator
//   2  inline asm, accumulators in VGPRs ("+v"), a different A / B pair per accumulator
//   3  builtin, a different A / B pair per accumulator, accumulators in VGPRs
// The second thing wrong with the round-1 version: built without -amdgpu-mfma-vgpr-form, hipcc keeps the loop-carried
// accumulators in VGPRs and copies all of them into AGPRs in front of the MFMAs and back behind them on EVERY iteration
// (8 v_accvgpr_write + 8 v_accvgpr_read + an s_nop 13 per MFMA): its "139 cycles per MFMA for one wave" was that copy loop,
// not the matrix pipe.  (k_gram / k_gram8 never had the copies: their accumulators live in VGPRs, checked in the .s.)
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o build_tools/mfma_peak tools/mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
struct rec { unsigned hw, xcc; unsigned long long c0, c1, r0, r1; };

__device__ __forceinline__ unsigned hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

template <int NACC, int FORM, int THREADS>
__global__ __launch_bounds__(THREADS) void k(double *out, rec *recs, int iters, const double *in)
{
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { a[i] = in[(threadIdx.x + 64 * i) & 1023]; b[i] = in[(threadIdx.x + 64 * i + 512) & 1023]; }
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (FORM == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[i], 0, 0, 0);
            else if (FORM == 3) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
            else if (FORM == 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]));
        }
        if (FORM == 1) {
#define MF(I, LO, HI) if (NACC > I) asm volatile("v_mfma_f64_16x16x4_f64 a[" #LO ":" #HI "], %0, %1, a[" #LO ":" #HI "]" :: "v"(a[I < NACC ? I : 0]), "v"(b[I < NACC ? I : 0]) : AGPRS)
#define AGPRS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
              "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
            MF(0, 0, 7); MF(1, 8, 15); MF(2, 16, 23); MF(3, 24, 31); MF(4, 32, 39); MF(5, 40, 47); MF(6, 48, 55); MF(7, 56, 63);
        }
    }
    if (FORM == 1 || FORM == 2) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // results of the last MFMAs (inline asm: no hazard padding)
    if (FORM == 1) {                 // the AGPR accumulators start from whatever the registers held: the sum is not a result, only the rate is measured
        int t; asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(t) :: AGPRS); acc[0][0] = (double)t;
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * THREADS + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        rec r; r.hw = hw_id(); r.xcc = xcc_id(); r.c0 = c0; r.c1 = c1; r.r0 = r0; r.r1 = r1;
        recs[(size_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = r;
    }
}

static double *g_out, *g_in; static rec *g_recs;
static int g_flagged = 0;

template <int NACC, int FORM, int THREADS>
void run(int blocks_per_cu_x, int iters, const char *note = "")
{
    const int blocks = 256 * blocks_per_cu_x, wpb = THREADS / 64, nw = blocks * wpb;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NACC, FORM, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, g_out, g_recs, iters, g_in);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, FORM, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, g_out, g_recs, iters, g_in);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<rec> h(nw);
    (void)hipMemcpy(h.data(), g_recs, sizeof(rec) * nw, hipMemcpyDeviceToHost);
    // clock: shader cycles per 10 ns tick
    std::vector<double> ghz(nw);
    for (int i = 0; i < nw; ++i) ghz[i] = (double)(h[i].c1 - h[i].c0) / (double)(h[i].r1 - h[i].r0) * 0.1;
    std::sort(ghz.begin(), ghz.end());
    const double clock = ghz[nw / 2];
    // per SIMD: key = xcc | se/sh/cu (bits 8..15 of HW_ID) | simd (bits 4..5)
    struct simd_t { std::vector<std::pair<unsigned long long, int>> ev; unsigned long long r0 = ~0ull, r1 = 0; long n = 0; };
    std::map<unsigned, simd_t> simds;
    for (int i = 0; i < nw; ++i) {
        const unsigned key = (h[i].xcc << 16) | (((h[i].hw >> 8) & 0xFF) << 4) | ((h[i].hw >> 4) & 3);
        simd_t &s = simds[key];
        s.ev.push_back({h[i].r0, +1}); s.ev.push_back({h[i].r1, -1});
        s.r0 = std::min(s.r0, h[i].r0); s.r1 = std::max(s.r1, h[i].r1); s.n += 1;
    }
    std::vector<double> cpm; std::vector<int> peak;
    for (auto &kv : simds) {
        simd_t &s = kv.second;
        std::sort(s.ev.begin(), s.ev.end());
        int cur = 0, pk = 0;
        for (auto &e : s.ev) { cur += e.second; pk = std::max(pk, cur); }
        peak.push_back(pk);
        const double cycles = (double)(s.r1 - s.r0) * 10.0 * clock;          // 10 ns ticks -> shader cycles
        cpm.push_back(cycles / ((double)s.n * iters * NACC));
    }
    std::sort(cpm.begin(), cpm.end()); std::sort(peak.begin(), peak.end());
    const size_t ns = simds.size();
    const double flops = (double)nw * iters * NACC * 2048.0;
    const double tf = flops / ms / 1e9;
    const double wall_cpm = (double)ms * 1e-3 * clock * 1e9 / ((double)nw / 1024.0 * iters * NACC);   // if all 1024 SIMDs shared the waves evenly
    const bool bad = cpm[0] < 62.0 ||   /* (the census converts 10 ns ticks with the MEDIAN clock: +-2 % per SIMD) */ fabs(wall_cpm / cpm[ns / 2] - 1.0) > 0.05 || ns != 1024;
    if (bad) ++g_flagged;
    printf("form %d  NACC %2d  %4d-thr blocks x %d/CU | SIMDs seen %4zu  resident waves/SIMD min %d med %d max %d | %8.3f ms  %6.2f TFLOP/s wall | clock %.3f GHz | "
           "cycles/MFMA/SIMD: census min %.1f med %.1f max %.1f, from wall %.1f %s%s\n",
           FORM, NACC, THREADS, blocks_per_cu_x, ns, peak[0], peak[ns / 2], peak[ns - 1], ms, tf, clock, cpm[0], cpm[ns / 2], cpm[ns - 1], wall_cpm,
           bad ? " <-- INCONSISTENT (not every wave resident, uneven placement, or a tail round)" : "", note);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const bool longrun = argc > 1 && !strcmp(argv[1], "long");
    (void)hipMalloc(&g_out, sizeof(double) * 256 * 8 * 1024);
    (void)hipMalloc(&g_recs, sizeof(rec) * 256 * 8 * 16);
    (void)hipMalloc(&g_in, sizeof(double) * 1024);
    std::vector<double> hin(1024);
    for (int i = 0; i < 1024; ++i) hin[i] = 0.5 + (double)((i * 2654435761u) % 1000u) * 1e-3;
    (void)hipMemcpy(g_in, hin.data(), sizeof(double) * 1024, hipMemcpyHostToDevice);
    const int it = 20000;
    printf("# v_mfma_f64_16x16x4_f64 issue rate; spec = 64 cycles per MFMA per SIMD = 78.6 TFLOP/s at 2.4 GHz on 1024 SIMDs\n");
    printf("# one wave per SIMD (256-thread blocks, one per CU): accumulators and operand form\n");
    run<1, 0, 256>(1, it); run<2, 0, 256>(1, it); run<4, 0, 256>(1, it); run<8, 0, 256>(1, it); run<16, 0, 256>(1, it);
    run<4, 3, 256>(1, it); run<8, 3, 256>(1, it);
    run<4, 1, 256>(1, it); run<8, 1, 256>(1, it);
    run<4, 2, 256>(1, it); run<8, 2, 256>(1, it);
    printf("# waves per SIMD (256-thread blocks, k per CU), 4 accumulators\n");
    run<4, 0, 256>(2, it); run<4, 0, 256>(3, it); run<4, 0, 256>(4, it); run<4, 0, 256>(5, it); run<4, 0, 256>(6, it); run<4, 0, 256>(7, it); run<4, 0, 256>(8, it);
    run<4, 1, 256>(2, it); run<4, 1, 256>(4, it); run<4, 1, 256>(6, it); run<4, 1, 256>(8, it);
    run<4, 3, 256>(2, it); run<4, 3, 256>(4, it); run<4, 3, 256>(6, it); run<4, 3, 256>(8, it);
    printf("# 512-thread blocks (two waves per SIMD each), 4 accumulators: k_gram8's geometry is 3 of them per CU\n");
    run<4, 0, 512>(1, it); run<4, 0, 512>(2, it); run<4, 0, 512>(3, it); run<4, 0, 512>(4, it);
    run<4, 1, 512>(3, it); run<4, 3, 512>(3, it);
    printf("# 1, 2 accumulators at 6 and 8 waves per SIMD\n");
    run<1, 0, 256>(6, it); run<2, 0, 256>(6, it); run<1, 0, 256>(8, it); run<2, 0, 256>(8, it);
    if (longrun) {
        printf("# sustained: the best geometry repeated for ~2 s (does the clock hold?)\n");
        for (int rep = 0; rep < 40; ++rep) run<4, 0, 256>(8, 4 * it, rep % 10 == 9 ? " (sustained)" : "");
    }
    printf("# flagged lines: %d\n", g_flagged);
    return 0;
}

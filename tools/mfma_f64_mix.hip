// What does an instruction cost when it is issued between v_mfma_f64_16x16x4_f64 on gfx950?  (round 5)
// One loop iteration = 16 independent MFMAs (16 accumulators in VGPRs, the Gram's per-wave tile for four chains) + NX extra
// instructions of one kind spread evenly between them, everything in volatile inline asm so that the order is the source order.
// 256-thread blocks, W per CU (= W waves per SIMD).  Reported: shader cycles per MFMA per SIMD (64 = the pipe's rate) and the
// cycles every extra instruction adds.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_mix tools/mfma_f64_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
enum { X_NONE, X_MUL64, X_FMA64, X_MOV32, X_DSREAD, X_DSWRITE, X_BARRIER, X_GLOAD, X_MUL64_DEP, X_ADD64, X_MUL32, X_SALU };

template <int KIND>
__device__ __forceinline__ void extra(double &t0, double &t1, double x, double y, unsigned ldsaddr, d2 &w, const double *gp, d2 &gl, double &dep)
{
    if (KIND == X_MUL64) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t0) : "v"(x), "v"(y));
    if (KIND == X_ADD64) asm volatile("v_add_f64 %0, %1, %2" : "=v"(t0) : "v"(x), "v"(y));
    if (KIND == X_FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(t0) : "v"(x), "v"(y));
    if (KIND == X_MOV32) { int r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"((int)ldsaddr)); (void)r; }
    if (KIND == X_MUL32) { float r; asm volatile("v_mul_f32 %0, %1, %1" : "=v"(r) : "v"((float)ldsaddr)); (void)r; }
    if (KIND == X_SALU) { int r; asm volatile("s_add_u32 %0, %1, 1" : "=s"(r) : "s"(7) : "scc"); (void)r; }
    if (KIND == X_DSREAD) asm volatile("ds_read_b64 %0, %1" : "=v"(t1) : "v"(ldsaddr));
    if (KIND == X_DSWRITE) asm volatile("ds_write_b128 %0, %1" :: "v"(ldsaddr), "v"(w) : "memory");
    if (KIND == X_BARRIER) asm volatile("s_barrier" ::: "memory");
    if (KIND == X_GLOAD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gl) : "v"(gp));
    if (KIND == X_MUL64_DEP) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(dep) : "v"(x), "v"(y));
}

template <int KIND, int NX>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *cyc, int iters, const double *in)
{
    __shared__ double lds[4096];
    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i & 1023];
    double a = in[threadIdx.x & 1023], b = in[(threadIdx.x + 512) & 1023];
    double t0 = 1.0, t1 = 0.0, dep = b;
    d2 w = {a, b}, gl = {0, 0};
    const unsigned ldsaddr = (unsigned)(size_t)(lds) + (threadIdx.x & 255) * 16;
    const double *gp = in + 2 * (threadIdx.x & 255);
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == X_MUL64_DEP) {
                // the Gram's form: the B operand of the next MFMAs is the product computed just before them
                if (NX > 0 && (i % (16 / (NX > 16 ? 16 : NX))) == 0) {
                    asm volatile("v_mul_f64 %0, %1, %2\n\ts_nop 1" : "=v"(dep) : "v"(b), "v"(a));
                }
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(dep));
            } else {
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (NX >= 16) {
#pragma unroll
                    for (int r = 0; r < NX / 16; ++r) extra<KIND>(t0, t1, a, b, ldsaddr, w, gp, gl, dep);
                } else if (NX > 0 && (i % (16 / (NX > 0 ? NX : 1))) == 0) extra<KIND>(t0, t1, a, b, ldsaddr, w, gp, gl, dep);
            }
        }
        if (KIND == X_DSREAD || KIND == X_DSWRITE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == X_GLOAD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = t0 + t1 + gl[0] + dep;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

static double *g_out, *g_in; static unsigned long long *g_cyc;
static double g_base[4];
template <int KIND, int NX>
static void run(const char *name, int W)
{
    const int blocks = 256 * W, iters = 2000;
    hipLaunchKernelGGL((k<KIND, NX>), dim3(blocks), dim3(256), 0, 0, g_out, g_cyc, 50, g_in);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NX>), dim3(blocks), dim3(256), 0, 0, g_out, g_cyc, iters, g_in);
    (void)hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); exit(1); }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c((size_t)blocks * 4);
    (void)hipMemcpy(c.data(), g_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[c.size() / 2];
    // every wave spends `med` cycles on iters * 16 MFMAs while W waves share the SIMD
    const double per_mfma = med / ((double)iters * 16.0 * W);
    const double wall_per_mfma = ms * 1e-3 * 2.4e9 / ((double)iters * 16.0 * W);
    if (KIND == X_NONE) g_base[W] = per_mfma;
    const double added = NX > 0 ? (per_mfma - g_base[W]) * 16.0 / NX : 0.0;
    printf("%-34s W=%d  %6.1f cycles per MFMA (wall at 2.4 GHz: %6.1f)   each extra instruction adds %6.1f cycles of SIMD time\n", name, W, per_mfma, wall_per_mfma, added);
    fflush(stdout);
}
#define ALLW(KIND, NX, NAME) do { run<KIND, NX>(NAME, 1); run<KIND, NX>(NAME, 2); run<KIND, NX>(NAME, 3); } while (0)
int main()
{
    (void)hipMalloc(&g_out, sizeof(double) * 256 * 256 * 4);
    (void)hipMalloc(&g_cyc, 8 * 256 * 4 * 4);
    (void)hipMalloc(&g_in, sizeof(double) * 1024);
    std::vector<double> hin(1024);
    for (int i = 0; i < 1024; ++i) hin[i] = 0.5 + (double)((i * 2654435761u) % 1000u) * 1e-3;
    (void)hipMemcpy(g_in, hin.data(), sizeof(double) * 1024, hipMemcpyHostToDevice);
    printf("# 16 MFMAs per iteration + extras; s_memtime cycles of the median wave / (MFMAs of all W waves of its SIMD)\n");
    ALLW(X_NONE, 0, "16 MFMA");
    ALLW(X_MUL64, 8, "+ 8 v_mul_f64");
    ALLW(X_MUL64, 16, "+ 16 v_mul_f64");
    ALLW(X_MUL64, 32, "+ 32 v_mul_f64");
    ALLW(X_ADD64, 16, "+ 16 v_add_f64");
    ALLW(X_FMA64, 16, "+ 16 v_fma_f64 (dependent chain)");
    ALLW(X_MOV32, 16, "+ 16 v_mov_b32");
    ALLW(X_MUL32, 16, "+ 16 v_mul_f32");
    ALLW(X_SALU, 16, "+ 16 s_add_u32");
    ALLW(X_DSREAD, 8, "+ 8 ds_read_b64");
    ALLW(X_DSREAD, 16, "+ 16 ds_read_b64");
    ALLW(X_DSWRITE, 4, "+ 4 ds_write_b128");
    ALLW(X_GLOAD, 4, "+ 4 global_load_dwordx4 (L1 hit)");
    ALLW(X_BARRIER, 1, "+ 1 s_barrier");
    ALLW(X_MUL64_DEP, 8, "+ 8 v_mul_f64 feeding the MFMAs");
    ALLW(X_MUL64_DEP, 16, "+ 16 v_mul_f64 feeding the MFMAs");
    return 0;
}

// How fast can workgroups that sit on the SAME XCD hand data to each other through that XCD's L2 while a kernel runs?
//   - ping-pong of an 8 KB payload + flag between two workgroups of one XCD: flag by an atomic that stays in the L2 (workgroup scope: no sc
//     bits, the L2 executes it), payload stored plainly (write-through L1 -> L2, s_waitcnt vmcnt(0) before the flag) and read back past the
//     reader's L1 (sc1 loads) -- against the agent-scope release/acquire version of tools/flag_handoff.hip;
//   - a barrier among all resident workgroups of an XCD (128 of 256 threads) on one L2 word.
// Decides whether a persistent, XCD-local factorization (one chain per XCD, E resident in its L2) can beat one launch per panel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define SPIN_MAX 4000000
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }
__device__ __forceinline__ unsigned l2_add(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ double ld_l2(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// MODE 0: L2-local flag (l2_add), payload read with sc1 loads; 1: the same with PLAIN payload loads (may see a stale L1 line);
// 2: agent-scope release / acquire on the flag, plain payload loads
template <int MODE>
__global__ __launch_bounds__(256) void k_pingpong(double *bufA, double *bufB, unsigned *flags, int wgB, int rounds, unsigned long long *out)
{
    __shared__ int ok;
    const int tid = threadIdx.x;
    const bool isA = blockIdx.x == 0, isB = (int)blockIdx.x == wgB;
    if (!isA && !isB) return;
    double *mine = isA ? bufA : bufB;
    const double *theirs = isA ? bufB : bufA;
    unsigned *fmine = flags + (isA ? 0 : 64), *ftheirs = flags + (isA ? 64 : 0);
    unsigned long long t0 = 0, bad = 0;
    if (tid == 0) { ok = 1; t0 = __builtin_amdgcn_s_memrealtime(); out[3 + (isA ? 0 : 1)] = xcc_id(); }
    __syncthreads();
    for (int r = 1; r <= rounds; ++r) {
        if (isB || r > 1) {
            const unsigned want = isA ? (unsigned)(r - 1) : (unsigned)r;
            if (tid == 0) {
                bool g = false;
                for (int i = 0; i < SPIN_MAX && !g; ++i) {
                    // (a workgroup-scope RMW of 0 is folded into a plain load by the compiler and then spins on a stale L1 line: the flag is read past the L1)
                    const unsigned v = MODE == 2 ? __hip_atomic_load(ftheirs, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(ftheirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g = v >= want;
                }
                if (!g) ok = 0;
            }
            __syncthreads();
            if (!ok) break;
            for (int i = 0; i < 4; ++i) {
                const double v = MODE == 0 ? ld_l2(theirs + tid + 256 * i) : theirs[tid + 256 * i];
                if (v != (double)want) ++bad;
            }
        }
        for (int i = 0; i < 4; ++i) mine[tid + 256 * i] = (double)r;
        if (MODE != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (MODE == 2) __hip_atomic_store(fmine, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else l2_add(fmine, 1u);
        }
    }
    if (bad) atomicAdd(&out[1], bad);
    if (tid == 0 && isA) { out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[2] = ok ? 0 : 1; }
}
template <int MODE> void pingpong(int wgB, int rounds, const char *what)
{
    double *a, *b; unsigned *f; unsigned long long *out, h[5];
    (void)hipMalloc(&a, 8192); (void)hipMalloc(&b, 8192); (void)hipMalloc(&f, 1024); (void)hipMalloc(&out, 64);
    (void)hipMemset(a, 0, 8192); (void)hipMemset(b, 0, 8192); (void)hipMemset(f, 0, 1024); (void)hipMemset(out, 0, 64);
    hipLaunchKernelGGL((k_pingpong<MODE>), dim3(64), dim3(256), 0, 0, a, b, f, wgB, rounds, out);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, 40, hipMemcpyDeviceToHost);
    printf("%-58s wgB=%2d (XCDs %llu, %llu): %.3f us per one-way hand-off (payload mismatches %llu, timeout %llu)\n", what, wgB, h[3], h[4],
           (double)h[0] * 0.01 / (2.0 * rounds), h[1], h[2]);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(f); (void)hipFree(out);
}
// barrier among the workgroups of one XCD: every workgroup adds 1 to the XCD's word and waits until it reaches round * members.
// members = workgroups of the grid on that XCD (counted in a first pass).  AGENT = 1: the same with agent-scope atomics (for comparison).
template <int AGENT>
__global__ __launch_bounds__(256) void k_barrier(unsigned *words, int rounds, unsigned long long *out)
{
    const unsigned x = xcc_id();
    unsigned *cnt = words + 64 * x, *arr = words + 64 * x + 32;
    __shared__ unsigned members;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = 1;
        if (AGENT) atomicAdd(cnt, 1u); else l2_add(cnt, 1u);
        // everybody of the grid is resident (host sizes the grid so): wait until the census is complete chip-wide
        atomicAdd(words + 1000, 1u);
        int i = 0;
        while (__hip_atomic_load(words + 1000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++i < SPIN_MAX) __builtin_amdgcn_s_sleep(8);
        if (i >= SPIN_MAX) ok = 0;
        members = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned m = members;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds && ok; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned want = (unsigned)r * m;
            if (AGENT) atomicAdd(arr, 1u); else l2_add(arr, 1u);
            int i = 0;
            while (__hip_atomic_load(arr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++i < SPIN_MAX) ;
            if (i >= SPIN_MAX) ok = 0;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
        atomicMax(&out[x], dt);
        out[8 + x] = m;
        if (!ok) out[16] = 1;
    }
}
template <int AGENT> void barrier_test(int grid, int rounds, const char *what)
{
    unsigned *w; unsigned long long *out, h[17];
    (void)hipMalloc(&w, 8192); (void)hipMalloc(&out, 17 * 8);
    (void)hipMemset(w, 0, 8192); (void)hipMemset(out, 0, 17 * 8);
    hipLaunchKernelGGL((k_barrier<AGENT>), dim3(grid), dim3(256), 0, 0, w, rounds, out);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-40s grid %4d: per barrier", what, grid);
    for (int x = 0; x < 8; ++x) printf(" %.2f us (%llu wg)", (double)h[x] * 0.01 / rounds, h[8 + x]);
    printf("%s\n", h[16] ? "  TIMED OUT" : "");
    (void)hipFree(w); (void)hipFree(out);
}
int main()
{
    for (int wgB : {8, 16, 1, 3}) {
        pingpong<0>(wgB, 2000, "L2-local flag, payload by sc1 loads");
        pingpong<1>(wgB, 2000, "L2-local flag, payload by plain loads");
        pingpong<2>(wgB, 2000, "agent-scope release/acquire flag, plain payload loads");
    }
    for (int grid : {256, 1024}) {
        barrier_test<0>(grid, 500, "XCD barrier, atomics in the L2");
        barrier_test<1>(grid, 500, "XCD barrier, agent-scope atomics");
    }
    return 0;
}

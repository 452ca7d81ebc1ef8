// Latency of a producer -> consumer hand-off between two workgroups of ONE running kernel through global memory:
// an 8 KB payload + a flag, agent-scope release/acquire.  Workgroup ids are chosen so that the two workgroups sit on the
// same XCD (id difference 8) or on different XCDs (difference 1).  Decides whether a persistent data-flow Cholesky can beat
// one launch per panel (tools/trace_gaps.py: ~2 us ramp per launch).
#include <hip/hip_runtime.h>
#include <cstdio>

#define SPIN_MAX 20000000

__device__ __forceinline__ bool wait_flag(const unsigned int *f, unsigned int v)
{
    for (int i = 0; i < SPIN_MAX; ++i) {
        if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= v) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// mode 0: agent-scope release/acquire;  mode 1: relaxed flag + explicit __threadfence on both sides
template <int MODE>
__global__ __launch_bounds__(256) void k_pingpong(double *bufA, double *bufB, unsigned int *flags, int wgB, int rounds, unsigned long long *out)
{
    __shared__ int ok;
    const int tid = threadIdx.x;
    const bool isA = blockIdx.x == 0, isB = (int)blockIdx.x == wgB;
    if (!isA && !isB) return;
    double *mine = isA ? bufA : bufB;
    const double *theirs = isA ? bufB : bufA;
    unsigned int *fmine = flags + (isA ? 0 : 64), *ftheirs = flags + (isA ? 64 : 0);
    unsigned long long t0 = 0, bad = 0;
    if (tid == 0) { ok = 1; t0 = __builtin_amdgcn_s_memrealtime(); }
    __syncthreads();
    for (int r = 1; r <= rounds; ++r) {
        if (isB || r > 1) {
            // wait for the other side's round (A waits for B's r-1, B for A's r)
            unsigned int want = isA ? (unsigned int)(r - 1) : (unsigned int)r;
            if (tid == 0) {
                bool g;
                if (MODE == 0) g = wait_flag(ftheirs, want);
                else {
                    g = false;
                    for (int i = 0; i < SPIN_MAX; ++i) {
                        if (__hip_atomic_load(ftheirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { g = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    __threadfence();
                }
                if (!g) ok = 0;
            }
            __syncthreads();
            if (!ok) break;
            for (int i = 0; i < 4; ++i) if (theirs[tid + 256 * i] != (double)want) ++bad;
        }
        for (int i = 0; i < 4; ++i) mine[tid + 256 * i] = (double)r;
        __syncthreads();
        if (tid == 0) {
            if (MODE == 0) __hip_atomic_store(fmine, (unsigned int)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else { __threadfence(); __hip_atomic_store(fmine, (unsigned int)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
    }
    if (bad) atomicAdd(&out[1], bad);
    if (tid == 0 && isA) {
        unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        out[0] = t1 - t0; out[2] = ok ? 0 : 1;
    }
}

template <int MODE> void run(int wgB, int rounds, const char *what)
{
    double *a, *b; unsigned int *f; unsigned long long *out, h[3];
    (void)hipMalloc(&a, 8192); (void)hipMalloc(&b, 8192); (void)hipMalloc(&f, 1024); (void)hipMalloc(&out, 64);
    (void)hipMemset(a, 0, 8192); (void)hipMemset(b, 0, 8192); (void)hipMemset(f, 0, 1024); (void)hipMemset(out, 0, 64);
    hipLaunchKernelGGL((k_pingpong<MODE>), dim3(64), dim3(256), 0, 0, a, b, f, wgB, rounds, out);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    // s_memrealtime counts at 100 MHz
    printf("%-34s wgB=%2d: %.3f us per one-way hand-off (payload mismatches %llu, timeout %llu)\n", what, wgB,
           (double)h[0] * 0.01 / (2.0 * rounds), h[1], h[2]);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(f); (void)hipFree(out);
}
int main()
{
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(8, 2000, "release/acquire, same XCD");
        run<0>(1, 2000, "release/acquire, other XCD");
        run<0>(4, 2000, "release/acquire, other XCD");
        run<1>(8, 2000, "relaxed + __threadfence, same XCD");
        run<1>(1, 2000, "relaxed + __threadfence, other XCD");
    }
    return 0;
}

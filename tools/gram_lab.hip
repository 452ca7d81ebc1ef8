// Lab for the round-5 Gram of a lockstep group: G_c = X diag(S_c) X' for C chains that share X, stream-K over a static item list.
//   * one 256-thread workgroup = 4 waves, each wave one 32 x 32 block of G for C chains (C x 4 MFMA accumulators);
//   * X panels staged UNSCALED through LDS (three buffers, global loads three batches ahead), S applied to the B fragment after ds_read;
//   * work = (quad, batch of 16 columns) items, cut into equal contiguous ranges per workgroup (stream-K): no tail round.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/gram_lab tools/gram_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <type_traits>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int KB = 16;                    // columns per batch (4 MFMA k-steps)
constexpr int COLSTRIDE = 4 * 32;         // doubles per LDS column: 4 units of 32 rows
constexpr int XBUF = KB * COLSTRIDE;      // doubles per X buffer (16 KiB)
constexpr int MAXSEG = 4;

struct gquad_t { int urow[4]; int ua[4], ub[4], kind[4]; };      // kind: 0 dead, 1 full block, 2 diagonal block
struct seg_t { int quad, b0, b1, slot[4]; };
struct wg_t { int nseg, nitem; seg_t seg[MAXSEG]; };
struct kargs_t { const double *X; const double *S[8]; double *G[8]; const gquad_t *quads; const wg_t *wgs; int ld, q, nwg_per_group, ngroups; unsigned long long *stamps; };
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> __device__ __forceinline__ T *sgpr_ptr(T *p)
{
    unsigned long long v = (unsigned long long)p;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}

template <int C, int WPS, int EXP>
__global__ __launch_bounds__(256, WPS) void k_gramc(const kargs_t A)
{
    // workgroup id -> (XCD label, group, slot): the groups' workgroups of one slot sit next to each other on one XCD (they read the same X panels)
    const int gid = blockIdx.x, gx = gid & 7, gr = gid >> 3;
    const int group = gr % A.ngroups, wslot = (gr / A.ngroups) * 8 + gx;
    if (wslot >= A.nwg_per_group) return;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    __shared__ double sX[3 * XBUF];
    __shared__ double sS[3 * KB * C];
    __shared__ int sW[sizeof(wg_t) / 4];
    __shared__ int sQ[MAXSEG][16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ln = lane & 15, lq = lane >> 4;
    {
        const int *wsrc = (const int *)(A.wgs + wslot);
        if (tid < (int)(sizeof(wg_t) / 4)) sW[tid] = wsrc[tid];
        __syncthreads();
        const int nseg0 = sW[0];
        if (tid < 16 * nseg0) { const int sgi = tid >> 4; sQ[sgi][tid & 15] = ((const int *)(A.quads + sW[2 + 7 * sgi]))[tid & 15]; }
        __syncthreads();
    }
    const int nseg = sgpr(sW[0]);
    // seg s: sW[2 + 7 s + {0 quad, 1 b0, 2 b1, 3.. slot[4]}]; quad of seg s: sQ[s][{0..3 urow, 4..7 ua, 8..11 ub, 12..15 kind}]
    const int ld = A.ld;
    const double *Sp[C]; double *Gp[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { Sp[c] = A.S[group * C + c]; Gp[c] = A.G[group * C + c]; }
    // staging map: thread -> column tid >> 4, row pair tid & 15 of every unit
    const int scol = tid >> 4, srp = tid & 15;
    const int woff = scol * COLSTRIDE + ((2 * srp) ^ ((scol & 1) << 4));
    const unsigned goff = (unsigned)scol * (unsigned)ld + 2u * srp;          // per-thread offset inside a batch (doubles)
    const int sc = (tid >> 4) & (C - 1), scl = tid & 15;           // S staging: chain sc, column scl (threads beyond 16 C repeat)
    const double *Smine = Sp[0];
#pragma unroll
    for (int c = 1; c < C; ++c) Smine = sc == c ? Sp[c] : Smine;
    d2 rx[4];
    double rs;
    // load cursor (three items ahead of the compute cursor): all scalar
    int lseg = 0, lb = sgpr(sW[3]), lb1 = sgpr(sW[4]);
    int lu0 = sgpr(sQ[0][0]), lu1 = sgpr(sQ[0][1]), lu2 = sgpr(sQ[0][2]), lu3 = sgpr(sQ[0][3]);
    auto issue_loads = [&](bool inloop = false) {
        if ((EXP & 1) && inloop) return;
        const double *cb = A.X + (size_t)lb * (KB * (size_t)ld);
        rx[0] = *(const d2 *)(cb + lu0 + goff); rx[1] = *(const d2 *)(cb + lu1 + goff);
        rx[2] = *(const d2 *)(cb + lu2 + goff); rx[3] = *(const d2 *)(cb + lu3 + goff);
        int si = lb * KB + scl; si = si < A.q ? si : A.q - 1;
        rs = Smine[si];
        if (lb + 1 < lb1) ++lb;
        else if (lseg + 1 < nseg) {
            ++lseg;
            lb = sgpr(sW[2 + 7 * lseg + 1]); lb1 = sgpr(sW[2 + 7 * lseg + 2]);
            lu0 = sgpr(sQ[lseg][0]); lu1 = sgpr(sQ[lseg][1]); lu2 = sgpr(sQ[lseg][2]); lu3 = sgpr(sQ[lseg][3]);
        }
    };
    auto stage = [&](int buf) {
        double *xb = sX + buf * XBUF + woff;
#pragma unroll
        for (int u = 0; u < 4; ++u) *(d2 *)(xb + u * 32) = rx[u];
        sS[buf * KB * C + scl * C + sc] = rs;
    };
    issue_loads(); stage(0);
    issue_loads(); stage(1);
    issue_loads();
    __syncthreads();
    const unsigned long long t_loop = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_store = 0;
    d4 acc[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c][t] = d4{0, 0, 0, 0};
    int cbuf = 0;
    const int sw = (lq & 1) << 4;
    for (int cseg = 0; cseg < nseg; ++cseg) {
        const int b0s = sgpr(sW[2 + 7 * cseg + 1]), b1s = sgpr(sW[2 + 7 * cseg + 2]);
        const int ua = sgpr(sQ[cseg][4 + wave]), ub = sgpr(sQ[cseg][8 + wave]), kind = sgpr(sQ[cseg][12 + wave]);
        // fragment offsets inside a buffer (doubles): column 4 kk + lq, unit, row
        const int oa0 = lq * COLSTRIDE + ua * 32 + (ln ^ sw), oa1 = lq * COLSTRIDE + ua * 32 + ((16 + ln) ^ sw);
        const int ob0 = lq * COLSTRIDE + ub * 32 + (ln ^ sw), ob1 = lq * COLSTRIDE + ub * 32 + ((16 + ln) ^ sw);
        const int os = lq * C;
        double a0, a1, b0, b1, sv[C];
        {
            const double *xb = sX + cbuf * XBUF;
            const double *sb = sS + cbuf * KB * C;
            a0 = xb[oa0]; a1 = xb[oa1]; b0 = xb[ob0]; b1 = xb[ob1];
#pragma unroll
            for (int c = 0; c < C; ++c) sv[c] = sb[os + c];
        }
        auto batches = [&](auto kind_c) {
            constexpr int KIND = decltype(kind_c)::value;
            for (int b = b0s; b < b1s; ++b) {
                const int nbuf = cbuf == 2 ? 0 : cbuf + 1, wbuf = nbuf == 2 ? 0 : nbuf + 1;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    double na0 = a0, na1 = a1, nb0 = b0, nb1 = b1, nsv[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) nsv[c] = sv[c];
                    if (!(EXP & 4)) {
                        const double *xb = kk < 3 ? sX + cbuf * XBUF + (kk + 1) * 4 * COLSTRIDE : sX + nbuf * XBUF;
                        const double *sb = kk < 3 ? sS + cbuf * KB * C + (kk + 1) * 4 * C : sS + nbuf * KB * C;
                        na0 = xb[oa0]; na1 = xb[oa1]; nb0 = xb[ob0]; nb1 = xb[ob1];
#pragma unroll
                        for (int c = 0; c < C; ++c) nsv[c] = sb[os + c];
                    }
                    if (KIND != 0) {
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const double s0 = (EXP & 2) ? b0 : b0 * sv[c], s1 = (EXP & 2) ? b1 : b1 * sv[c];
                            acc[c][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, s0, acc[c][0], 0, 0, 0);
                            acc[c][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, s1, acc[c][1], 0, 0, 0);
                            if (KIND == 1) acc[c][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, s0, acc[c][2], 0, 0, 0);
                            acc[c][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, s1, acc[c][3], 0, 0, 0);
                        }
                    }
                    if (kk == 0) { stage(wbuf); issue_loads(true); }
                    a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
#pragma unroll
                    for (int c = 0; c < C; ++c) sv[c] = nsv[c];
                }
                __syncthreads();
                cbuf = nbuf;
            }
        };
        if (kind == 1) batches(std::integral_constant<int, 1>{});
        else if (kind == 2) batches(std::integral_constant<int, 2>{});
        else batches(std::integral_constant<int, 0>{});
        // the segment's partial block: element (i, j) of the 32 x 32 block at [j * 32 + i]; this lane: j = jt*16 + lq + 4 r, i = it*16 + ln
        const int slot = sgpr(sW[2 + 7 * cseg + 3 + wave]);
        if (cseg == nseg - 1) t_store = __builtin_amdgcn_s_memrealtime();
        if (slot >= 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                double *out = Gp[c] + (size_t)slot * 1024;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    out[(lq + 4 * r) * 32 + ln] = acc[c][0][r];
                    out[(lq + 4 * r) * 32 + 16 + ln] = acc[c][1][r];
                    out[(16 + lq + 4 * r) * 32 + ln] = acc[c][2][r];
                    out[(16 + lq + 4 * r) * 32 + 16 + ln] = acc[c][3][r];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[c][t] = d4{0, 0, 0, 0};
    }
    if (A.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) { unsigned long long *st = A.stamps + 4 * (size_t)gid; st[0] = t_start; st[1] = t_loop; st[2] = t_store; st[3] = t_end; }
    }
}

// ---------------------------------------------------------------- host
struct layout_t {
    std::vector<gquad_t> quads;
    std::vector<wg_t> wgs;
    std::vector<std::vector<int>> slots_of_block;   // per 32x32 block (bi * nb32 + bj): the partial slots in summation order
    int nslot = 0;
};
static layout_t make_layout(int n_pad, int nbatch, int nwg)
{
    layout_t L;
    const int ntile = n_pad / 64, nb32 = n_pad / 32;
    // off-diagonal 64 x 64 tiles: units [I0, I1, J0, J1]; wave (wi, wj): a = J_wj, b = I_wi
    for (int ti = 0; ti < ntile; ++ti)
        for (int tj = 0; tj < ti; ++tj) {
            gquad_t q{};
            q.urow[0] = ti * 64; q.urow[1] = ti * 64 + 32; q.urow[2] = tj * 64; q.urow[3] = tj * 64 + 32;
            for (int w = 0; w < 4; ++w) { int wi = (w >> 1) & 1, wj = w & 1; q.ua[w] = 2 + wj; q.ub[w] = wi; q.kind[w] = 1; }
            L.quads.push_back(q);
        }
    // diagonal tiles: blocks (d0,d0) (d1,d0) (d1,d1) in a row, cut into fours
    struct blk { int bi, bj; };
    std::vector<blk> db;
    for (int d = 0; d < ntile; ++d) { db.push_back({2 * d, 2 * d}); db.push_back({2 * d + 1, 2 * d}); db.push_back({2 * d + 1, 2 * d + 1}); }
    for (size_t k = 0; k < db.size(); k += 4) {
        gquad_t q{};
        int nu = 0, urows[8];
        auto unit_of = [&](int b32) { for (int u = 0; u < nu; ++u) if (urows[u] == b32 * 32) return u; urows[nu] = b32 * 32; return nu++; };
        for (int w = 0; w < 4; ++w) {
            if (k + w < db.size()) { q.ub[w] = unit_of(db[k + w].bi); q.ua[w] = unit_of(db[k + w].bj); q.kind[w] = db[k + w].bi == db[k + w].bj ? 2 : 1; }
            else { q.ua[w] = q.ub[w] = 0; q.kind[w] = 0; }
        }
        if (nu > 4) { fprintf(stderr, "quad needs %d units\n", nu); exit(1); }
        for (int u = 0; u < 4; ++u) q.urow[u] = u < nu ? urows[u] : 0;
        L.quads.push_back(q);
    }
    const int Q = (int)L.quads.size();
    L.slots_of_block.assign((size_t)nb32 * nb32, {});
    const long total = (long)Q * nbatch;
    L.wgs.resize(nwg);
    for (int w = 0; w < nwg; ++w) {
        long i0 = total * w / nwg, i1 = total * (w + 1) / nwg;
        wg_t &W = L.wgs[w];
        W.nseg = 0; W.nitem = (int)(i1 - i0);
        while (i0 < i1) {
            int qd = (int)(i0 / nbatch), b0 = (int)(i0 % nbatch);
            long e = std::min<long>(i1, (long)(qd + 1) * nbatch);
            if (W.nseg >= MAXSEG) { fprintf(stderr, "too many segments\n"); exit(1); }
            seg_t &s = W.seg[W.nseg++];
            s.quad = qd; s.b0 = b0; s.b1 = b0 + (int)(e - i0);
            const gquad_t &q = L.quads[qd];
            for (int wv = 0; wv < 4; ++wv) {
                if (q.kind[wv] == 0) { s.slot[wv] = -1; continue; }
                s.slot[wv] = L.nslot++;
                int bi = q.urow[q.ub[wv]] / 32, bj = q.urow[q.ua[wv]] / 32;
                L.slots_of_block[(size_t)bi * nb32 + bj].push_back(s.slot[wv]);
            }
            i0 = e;
        }
    }
    return L;
}

template <int C, int WPS, int EXP = 0>
static void run(int n, int q, int nchains, int wg_per_group, int reps, const std::vector<double> &X, int n_pad, const std::vector<std::vector<double>> &S)
{
    const int ld = n_pad, nbatch = (q + KB - 1) / KB, ngroups = nchains / C, nb32 = n_pad / 32;
    layout_t L = make_layout(n_pad, nbatch, wg_per_group);
    double *dX; CK(hipMalloc(&dX, X.size() * 8)); CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    gquad_t *dq; CK(hipMalloc(&dq, L.quads.size() * sizeof(gquad_t))); CK(hipMemcpy(dq, L.quads.data(), L.quads.size() * sizeof(gquad_t), hipMemcpyHostToDevice));
    wg_t *dw; CK(hipMalloc(&dw, L.wgs.size() * sizeof(wg_t))); CK(hipMemcpy(dw, L.wgs.data(), L.wgs.size() * sizeof(wg_t), hipMemcpyHostToDevice));
    kargs_t ka{};
    std::vector<double *> dS(nchains), dG(nchains);
    for (int c = 0; c < nchains; ++c) {
        CK(hipMalloc(&dS[c], S[c].size() * 8)); CK(hipMemcpy(dS[c], S[c].data(), S[c].size() * 8, hipMemcpyHostToDevice));
        CK(hipMalloc(&dG[c], (size_t)L.nslot * 1024 * 8)); CK(hipMemset(dG[c], 0xff, (size_t)L.nslot * 1024 * 8));
    }
    ka.X = dX; ka.quads = dq; ka.wgs = dw; ka.ld = ld; ka.q = q; ka.nwg_per_group = wg_per_group; ka.ngroups = ngroups;
    for (int c = 0; c < nchains; ++c) { ka.S[c] = dS[c]; ka.G[c] = dG[c]; }
    const int grid = ((wg_per_group + 7) / 8) * 8 * ngroups;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) k_gramc<C, WPS, EXP><<<grid, 256>>>(ka);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_gramc<C, WPS, EXP><<<grid, 256>>>(ka);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    {
        unsigned long long *dst; CK(hipMalloc(&dst, (size_t)grid * 32)); CK(hipMemset(dst, 0, (size_t)grid * 32));
        ka.stamps = dst;
        k_gramc<C, WPS, EXP><<<grid, 256>>>(ka);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st((size_t)grid * 4); CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0; double pro = 0, loop = 0, sto = 0, lmin = 1e9, lmax = 0, first_loop_max = 0, last_store_min = 1e18, start_max = 0; int cnt = 0;
        for (int g = 0; g < grid; ++g) { if (!st[4 * g]) continue; t0 = std::min(t0, st[4 * g]); t1 = std::max(t1, st[4 * g + 3]); }
        for (int g = 0; g < grid; ++g) {
            if (!st[4 * g]) continue;
            ++cnt;
            double a = (st[4 * g + 1] - st[4 * g]) / 100.0, b = (st[4 * g + 2] - st[4 * g + 1]) / 100.0, c2 = (st[4 * g + 3] - st[4 * g + 2]) / 100.0;
            pro += a; loop += b; sto += c2; lmin = std::min(lmin, b); lmax = std::max(lmax, b);
            start_max = std::max(start_max, (st[4 * g] - t0) / 100.0);
            first_loop_max = std::max(first_loop_max, (st[4 * g + 1] - t0) / 100.0); last_store_min = std::min(last_store_min, (st[4 * g + 2] - t0) / 100.0);
        }
        printf("   stamps: span %.1f us | per workgroup: prologue %.1f, loop %.1f (min %.1f max %.1f), final store %.1f | latest start +%.1f, latest loop start +%.1f, earliest final store +%.1f\n",
               (t1 - t0) / 100.0, pro / cnt, loop / cnt, lmin, lmax, sto / cnt, start_max, first_loop_max, last_store_min);
        ka.stamps = nullptr; CK(hipFree(dst));
    }
    const double us = ms * 1e3 / reps, flops = (double)n * n * q * nchains;
    // check: sampled entries of chains 0 and last against the plain sum on the host
    double worst = 0;
    for (int c : {0, nchains - 1}) {
        std::vector<double> P((size_t)L.nslot * 1024);
        CK(hipMemcpy(P.data(), dG[c], P.size() * 8, hipMemcpyDeviceToHost));
        srand(7 + c);
        for (int t = 0; t < 600; ++t) {
            int i = rand() % n, j = rand() % (i + 1);
            if (t < 40) j = i;                                          // diagonal entries too
            double ref = 0, mag = 0;
            for (int k = 0; k < q; ++k) { double v = X[(size_t)k * ld + i] * S[c][k] * X[(size_t)k * ld + j]; ref += v; mag += fabs(v); }
            double got = 0;
            for (int sl : L.slots_of_block[(size_t)(i / 32) * nb32 + j / 32]) got += P[(size_t)sl * 1024 + (j % 32) * 32 + i % 32];
            worst = std::max(worst, fabs(got - ref) / (mag + 1e-300));
        }
    }
    size_t maxparts = 0; for (auto &v : L.slots_of_block) maxparts = std::max(maxparts, v.size());
    printf("EXP=%d C=%d waves/SIMD=%d  groups %d x %d workgroups (grid %d)  quads %zu  slots %d (%.1f MB partials for %d chains, max %zu per block)  %8.2f us per launch  %6.2f TFLOP/s algorithmic  frac %.3f  worst rel err %.2e\n",
           EXP, C, WPS, ngroups, wg_per_group, grid, L.quads.size(), L.nslot, (double)L.nslot * 8192 * nchains / 1e6, nchains, maxparts, us, flops / us / 1e6, flops / us / 1e6 / 78.6, worst);
    fflush(stdout);
    CK(hipFree(dX)); CK(hipFree(dq)); CK(hipFree(dw));
    for (int c = 0; c < nchains; ++c) { CK(hipFree(dS[c])); CK(hipFree(dG[c])); }
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 500, V = argc > 2 ? atoi(argv[2]) : 100, nchains = 8, reps = 30;
    const int q = V * (V + 1) / 2, n_pad = (n + 63) / 64 * 64, q_alloc = (q + 63) / 64 * 64 + 64;
    std::vector<double> X((size_t)q_alloc * n_pad, 0.0);
    srand(1);
    for (int k = 0; k < q; ++k) for (int i = 0; i < n; ++i) X[(size_t)k * n_pad + i] = (rand() % 2001 - 1000) / 1000.0;
    std::vector<std::vector<double>> S(nchains, std::vector<double>(q));
    for (int c = 0; c < nchains; ++c) for (int k = 0; k < q; ++k) S[c][k] = 0.01 + (rand() % 1000) / 100.0;
    printf("# n=%d V=%d q=%d n_pad=%d, %d chains sharing X\n", n, V, q, n_pad, nchains);
    run<4, 1>(n, q, nchains, 128, reps, X, n_pad, S);
    run<4, 1, 1>(n, q, nchains, 128, reps, X, n_pad, S);
    run<4, 1, 2>(n, q, nchains, 128, reps, X, n_pad, S);
    run<4, 1, 4>(n, q, nchains, 128, reps, X, n_pad, S);
    run<4, 1, 7>(n, q, nchains, 128, reps, X, n_pad, S);
    run<4, 2>(n, q, nchains, 256, reps, X, n_pad, S);
    run<4, 2, 1>(n, q, nchains, 256, reps, X, n_pad, S);
    run<4, 2, 2>(n, q, nchains, 256, reps, X, n_pad, S);
    run<4, 2, 4>(n, q, nchains, 256, reps, X, n_pad, S);
    run<4, 2, 7>(n, q, nchains, 256, reps, X, n_pad, S);
    run<8, 1>(n, q, nchains, 256, reps, X, n_pad, S);
    run<8, 1, 7>(n, q, nchains, 256, reps, X, n_pad, S);
    return 0;
}

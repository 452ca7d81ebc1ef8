// Lab, second design for the group Gram (round 5): "wide strips".
// Finding behind it (tools/mfma_f64_mix.hip): on gfx950 every VALU instruction issued between v_mfma_f64_16x16x4_f64 costs the
// matrix pipe 6-14 cycles (LDS, VMEM, SALU and barriers cost nothing), so the loop must be built from MFMA + ds_read + a minimum
// of VALU.  Here a wave owns 16 rows of G (one B fragment, scaled by S with ONE v_mul_f64 per k-step) times up to 256 columns
// (W <= 16 A fragments = 16 accumulators): 1/W VALU instructions per MFMA instead of k_gram8's ~2.4.
//   * 256-thread workgroup = 4 waves = the four 16-row slices of a 64-row strip; X panels unscaled in LDS (padded column stride:
//     conflict-free ds_read_b64 without a swizzle), S in LDS; two LDS buffers, loads two batches ahead;
//   * diagonal 64 x 64 tiles: two per workgroup, two waves per tile (rows {0,3} and {1,2}: five MFMAs per k-step each);
//   * work = (task, batch of 8 columns) items with cost W (5 for a diagonal pair), cut into equal-cost contiguous ranges per
//     workgroup (stream-K): every workgroup of a chain gets the same number of MFMAs, no tail round.
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o tools/bin/gram_lab2 tools/gram_lab2.hip
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

constexpr int KB = 8;                     // columns per batch (2 MFMA k-steps)
constexpr int CS = 336;                   // doubles per LDS column: 64 I rows + 256 J rows + 16 pad (2688 B = 128 mod 256: lanes of columns k, k+1 on disjoint bank halves)
constexpr int XBUF = KB * CS;             // doubles per X buffer (21 KiB)
constexpr int MAXSEG = 4;

struct gtask_t { int kind, irow, jrow, w, tile0, ntile; };     // kind 0: strip part (64 rows at irow) x (16 w rows at jrow); kind 1: diagonal tiles at irow and jrow (jrow < 0: one tile)
struct seg_t { int task, b0, b1, slot[4]; };
struct wg_t { int nseg, pad; seg_t seg[MAXSEG]; };
struct kargs_t { const double *X; const double *S[16]; double *G[16]; const gtask_t *tasks; const wg_t *wgs; int ld, q, nwg_per_chain, nchains; unsigned long long *stamps; };
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// one segment of a strip part of width W (16-row column blocks): batches [b0, b1)
template <int W, bool DIAG, int EXP = 0>
__device__ __forceinline__ void gram_segment(const double *__restrict__ X, const double *__restrict__ Sp, int ld, int irow, int jrow, int b0, int b1,
                                             double *sX, double *sS, d4 (&acc)[16], int role)
{
    constexpr int R = 64 + 16 * W, RP = R / 2, NL = (KB * RP) / 256;       // rows per column, row pairs, 16-byte loads per thread and batch
    static_assert((KB * RP) % 256 == 0, "staging map");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ln = lane & 15, lq = lane >> 4;
    unsigned goff[NL]; int loff[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int e = l * 256 + tid, col = e / RP, rp = e % RP;
        const int grow = rp < 32 ? irow + 2 * rp : jrow + 2 * (rp - 32);
        goff[l] = (unsigned)col * (unsigned)ld + (unsigned)grow;
        loff[l] = col * CS + 2 * rp;
    }
    const int soff = tid & 7;
    d2 rx[NL];
    double rs;
    auto load = [&](int b, bool inloop = false) {
        if ((EXP & 1) && inloop) return;
        const double *cb = X + (size_t)b * (KB * (size_t)ld);
#pragma unroll
        for (int l = 0; l < NL; ++l) rx[l] = *(const d2 *)(cb + goff[l]);
        rs = Sp[b * KB + soff];
    };
    auto store = [&](int buf, bool inloop = false) {
        if ((EXP & 2) && inloop) return;
#pragma unroll
        for (int l = 0; l < NL; ++l) *(d2 *)(sX + buf * XBUF + loff[l]) = rx[l];
        if (wave == 0) sS[buf * KB + soff] = rs;
    };
    // fragment offsets (doubles) inside a buffer for k-step 0: column lq
    const int fb = lq * CS + ln;
    double keep_a = Sp[lane], keep_b = Sp[lane + 64];
    auto kstep = [&](int buf, int kk) {
        const double *xb = sX + buf * XBUF + kk * 4 * CS + fb;
        const double s = sS[buf * KB + kk * 4 + lq];
        if (!DIAG && (EXP & 8)) {
            // timing experiment: operands from registers
            const double bs = (EXP & 4) ? keep_b : keep_b * keep_a;
#pragma unroll
            for (int jc = 0; jc < W; ++jc) acc[jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(keep_a, bs, acc[jc], 0, 0, 0);
        } else if (!DIAG) {
            const double bs = (EXP & 4) ? xb[16 * wave] : xb[16 * wave] * s;
#pragma unroll
            for (int jc = 0; jc < W; ++jc) acc[jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[64 + 16 * jc], bs, acc[jc], 0, 0, 0);
        } else {
            // two waves per diagonal tile: role bit 0 = pairing (0: rows {0, 3}, 1: rows {1, 2}), bit 1 = tile (0: rows 0..63 of the panel, 1: rows 64..127)
            const double *tb = xb + ((role & 2) ? 64 : 0);
            const double a0 = tb[0], a1 = tb[16], a2 = tb[32], a3 = tb[48];
            if (role & 1) {
                const double bx = a1 * s, by = a2 * s;
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bx, acc[0], 0, 0, 0);       // (1, 0)
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bx, acc[1], 0, 0, 0);       // (1, 1)
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, by, acc[2], 0, 0, 0);       // (2, 0)
                acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, by, acc[3], 0, 0, 0);       // (2, 1)
                acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, by, acc[4], 0, 0, 0);       // (2, 2)
            } else {
                const double bx = a0 * s, by = a3 * s;
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bx, acc[0], 0, 0, 0);       // (0, 0)
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, by, acc[1], 0, 0, 0);       // (3, 0)
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, by, acc[2], 0, 0, 0);       // (3, 1)
                acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, by, acc[3], 0, 0, 0);       // (3, 2)
                acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, by, acc[4], 0, 0, 0);       // (3, 3)
            }
        }
    };
    load(b0); store(0);
    load(b0 + 1);                                  // (past the segment's end: the next columns of X, allocated and harmless)
    __syncthreads();
    auto batch = [&](auto cur_c, int b) {
        constexpr int cur = decltype(cur_c)::value;
        if (!DIAG || role >= 0) kstep(cur, 0);
        store(cur ^ 1, true);
        load(b + 2, true);
        if (!DIAG || role >= 0) kstep(cur, 1);
        if (!(EXP & 16)) __syncthreads();
    };
    int b = b0;
    for (; b + 1 < b1; b += 2) { batch(std::integral_constant<int, 0>{}, b); batch(std::integral_constant<int, 1>{}, b + 1); }
    if (b < b1) batch(std::integral_constant<int, 0>{}, b);
}

template <int WPS, int EXP = 0>
__global__ __launch_bounds__(256, WPS) void k_gramw(const kargs_t A)
{
    // workgroup id -> (XCD label, chain, slot): the chains' workgroups of one slot sit next to each other on one XCD (same X panels)
    const int gid = blockIdx.x, gx = gid & 7, gr = gid >> 3;
    const int chain = gr % A.nchains, wslot = (gr / A.nchains) * 8 + gx;
    if (wslot >= A.nwg_per_chain) return;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime(), c_start = __builtin_amdgcn_s_memtime();
    __shared__ double sX[2 * XBUF];
    __shared__ double sS[2 * KB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ln = lane & 15, lq = lane >> 4;
    const wg_t *Wd = A.wgs + wslot;
    const int nseg = sgpr(Wd->nseg);
    const double *Sp = A.S[chain];
    double *Gp = A.G[chain];
    d4 acc[16];
    for (int sgi = 0; sgi < nseg; ++sgi) {
        const seg_t *sg = Wd->seg + sgi;
        const int task = sgpr(sg->task), b0 = sgpr(sg->b0), b1 = sgpr(sg->b1);
        const gtask_t *T = A.tasks + task;
        const int kind = sgpr(T->kind), irow = sgpr(T->irow), jrow = sgpr(T->jrow), w = sgpr(T->w), ntile = sgpr(T->ntile);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
        if (sgi > 0) __syncthreads();                      // the previous segment's last reads of the LDS buffers
        if (kind == 0) {
            if (w == 16) gram_segment<16, false, EXP>(A.X, Sp, A.ld, irow, jrow, b0, b1, sX, sS, acc, 0);
            else if (w == 12) gram_segment<12, false>(A.X, Sp, A.ld, irow, jrow, b0, b1, sX, sS, acc, 0);
            else if (w == 8) gram_segment<8, false>(A.X, Sp, A.ld, irow, jrow, b0, b1, sX, sS, acc, 0);
            else gram_segment<4, false>(A.X, Sp, A.ld, irow, jrow, b0, b1, sX, sS, acc, 0);
            // tile k of the part: columns 64 k .. 64 k + 63 = accumulators 4 k .. 4 k + 3; element (i, j) of a tile at [j * 64 + i]
            for (int k = 0; k < ntile; ++k) {
                double *out = Gp + (size_t)sgpr(sg->slot[k]) * 4096 + 16 * wave + ln;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const d4 v = k == 0 ? acc[c] : k == 1 ? acc[4 + c] : k == 2 ? acc[8 + c] : acc[12 + c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[(16 * c + lq + 4 * r) * 64] = v[r];
                }
            }
        } else {
            const int role = (wave >> 1) < ntile ? wave : -1;            // waves 0, 1: first tile; 2, 3: second tile
            gram_segment<4, true>(A.X, Sp, A.ld, irow, jrow < 0 ? irow : jrow, b0, b1, sX, sS, acc, role);
            if (role >= 0) {
                double *out = Gp + (size_t)sgpr(sg->slot[wave >> 1]) * 4096 + ln;
                const d4 z = {0, 0, 0, 0};
                // rows of this wave: role & 1 ? {1, 2} : {0, 3}; block (u, c) -> out[(16 c + lq + 4 r) * 64 + 16 u]
#define PUT(U, Cc, V) do { _Pragma("unroll") for (int r = 0; r < 4; ++r) out[(16 * (Cc) + lq + 4 * r) * 64 + 16 * (U)] = (V)[r]; } while (0)
                if (role & 1) { PUT(1, 0, acc[0]); PUT(1, 1, acc[1]); PUT(1, 2, z); PUT(1, 3, z); PUT(2, 0, acc[2]); PUT(2, 1, acc[3]); PUT(2, 2, acc[4]); PUT(2, 3, z); }
                else { PUT(0, 0, acc[0]); PUT(0, 1, z); PUT(0, 2, z); PUT(0, 3, z); PUT(3, 0, acc[1]); PUT(3, 1, acc[2]); PUT(3, 2, acc[3]); PUT(3, 3, acc[4]); }
#undef PUT
            }
        }
    }
    if (A.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        const unsigned long long c_end = __builtin_amdgcn_s_memtime();
        if (tid == 0) { unsigned long long *st = A.stamps + 2 * (size_t)gid; st[0] = t_start; st[1] = t_end; A.stamps[2 * (size_t)gridDim.x + gid] = c_end - c_start; }
    }
}

// ---------------------------------------------------------------- host
struct layout_t {
    std::vector<gtask_t> tasks;
    std::vector<wg_t> wgs;
    std::vector<std::vector<int>> slots_of_tile;    // per lower 64x64 tile t = ti (ti + 1) / 2 + tj: partial slots in summation order
    int nslot = 0;
    long cost_total = 0, cost_max = 0;
};
static layout_t make_layout(int n_pad, int nbatch, int nwg, double diag_cost)
{
    layout_t L;
    const int ntile = n_pad / 64;
    std::vector<double> cost;
    for (int ti = 1; ti < ntile; ++ti) {
        int left = 4 * ti, jc = 0;                                  // off-diagonal extent of strip ti in 16-row column blocks
        while (left > 0) {
            int w = left >= 16 ? 16 : left;                         // 16, 12, 8 or 4 (left is a multiple of 4)
            if (left > 16 && left < 32) w = left >= 24 ? (left == 28 ? 16 : 12) : (left == 20 ? 12 : 16);   // 20 = 12 + 8, 24 = 12 + 12, 28 = 16 + 12
            gtask_t t{0, ti * 64, jc * 16, w, ti * (ti + 1) / 2 + jc / 4, w / 4};
            L.tasks.push_back(t); cost.push_back(w);
            left -= w; jc += w;
        }
    }
    for (int d = 0; d < ntile; d += 2) {
        const bool two = d + 1 < ntile;
        gtask_t t{1, d * 64, two ? (d + 1) * 64 : -1, 4, d * (d + 1) / 2 + d, two ? 2 : 1};
        L.tasks.push_back(t); cost.push_back(diag_cost);
    }
    const int NT = (int)L.tasks.size();
    L.slots_of_tile.assign((size_t)ntile * (ntile + 1) / 2, {});
    // items = (task, batch), cost per item = cost[task]; workgroup w gets the items whose cumulative cost midpoint falls into its share
    double total = 0; for (int t = 0; t < NT; ++t) total += cost[t] * nbatch;
    L.wgs.assign(nwg, wg_t{});
    double cum = 0; int w = 0;
    for (int t = 0; t < NT; ++t) {
        int b = 0;
        while (b < nbatch) {
            // how many batches of this task still fit into workgroup w's share
            const double limit = total * (w + 1) / nwg;
            int nb = (int)floor((limit - cum) / cost[t] + 0.5);
            if (w == nwg - 1) nb = nbatch - b;
            nb = std::max(0, std::min(nb, nbatch - b));
            if (nb == 0) { if (w < nwg - 1) { ++w; continue; } nb = nbatch - b; }
            wg_t &W = L.wgs[w];
            if (W.nseg >= MAXSEG) { fprintf(stderr, "too many segments\n"); exit(1); }
            seg_t &s = W.seg[W.nseg++];
            s.task = t; s.b0 = b; s.b1 = b + nb;
            const gtask_t &T = L.tasks[t];
            for (int k = 0; k < 4; ++k) s.slot[k] = -1;
            for (int k = 0; k < T.ntile; ++k) {
                s.slot[k] = L.nslot++;
                const int tile = T.kind == 0 ? T.tile0 + k : (k == 0 ? T.tile0 : (T.jrow / 64) * (T.jrow / 64 + 1) / 2 + T.jrow / 64);
                L.slots_of_tile[tile].push_back(s.slot[k]);
            }
            cum += cost[t] * nb; b += nb;
            if (cum >= limit - 1e-9 && w < nwg - 1) ++w;
        }
    }
    return L;
}

template <int WPS>
static void run(int n, int q, int nchains, int wg_per_chain, double diag_cost, int reps, const std::vector<double> &X, int n_pad, const std::vector<std::vector<double>> &S)
{
    const int ld = n_pad, nbatch = (q + KB - 1) / KB, ntile = n_pad / 64;
    layout_t L = make_layout(n_pad, nbatch, wg_per_chain, diag_cost);
    double *dX; CK(hipMalloc(&dX, X.size() * 8)); CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    gtask_t *dt; CK(hipMalloc(&dt, L.tasks.size() * sizeof(gtask_t))); CK(hipMemcpy(dt, L.tasks.data(), L.tasks.size() * sizeof(gtask_t), hipMemcpyHostToDevice));
    wg_t *dw; CK(hipMalloc(&dw, L.wgs.size() * sizeof(wg_t))); CK(hipMemcpy(dw, L.wgs.data(), L.wgs.size() * sizeof(wg_t), hipMemcpyHostToDevice));
    kargs_t ka{};
    std::vector<double *> dS(nchains), dG(nchains);
    for (int c = 0; c < nchains; ++c) {
        CK(hipMalloc(&dS[c], S[c].size() * 8)); CK(hipMemcpy(dS[c], S[c].data(), S[c].size() * 8, hipMemcpyHostToDevice));
        CK(hipMalloc(&dG[c], (size_t)L.nslot * 4096 * 8)); CK(hipMemset(dG[c], 0xff, (size_t)L.nslot * 4096 * 8));
        ka.S[c] = dS[c]; ka.G[c] = dG[c];
    }
    ka.X = dX; ka.tasks = dt; ka.wgs = dw; ka.ld = ld; ka.q = q; ka.nwg_per_chain = wg_per_chain; ka.nchains = nchains;
    const int grid = ((wg_per_chain + 7) / 8) * 8 * nchains;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) k_gramw<WPS><<<grid, 256>>>(ka);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_gramw<WPS><<<grid, 256>>>(ka);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double span = 0, dmin = 1e9, dmax = 0, davg = 0;
    {
        unsigned long long *dst; CK(hipMalloc(&dst, (size_t)grid * 16)); CK(hipMemset(dst, 0, (size_t)grid * 16));
        ka.stamps = dst;
        k_gramw<WPS><<<grid, 256>>>(ka);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st((size_t)grid * 2); CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0; int cnt = 0;
        for (int g = 0; g < grid; ++g) { if (!st[2 * g]) continue; t0 = std::min(t0, st[2 * g]); t1 = std::max(t1, st[2 * g + 1]); }
        for (int g = 0; g < grid; ++g) { if (!st[2 * g]) continue; ++cnt; double d = (st[2 * g + 1] - st[2 * g]) / 100.0; davg += d; dmin = std::min(dmin, d); dmax = std::max(dmax, d); }
        span = (t1 - t0) / 100.0; davg /= cnt;
        if (getenv("LAB_VERBOSE")) {
            for (int w = 0; w < wg_per_chain; ++w) {
                // gid of (chain 0, wslot w): gx = w & 7, gr = (w >> 3) * nchains
                const int g = ((w >> 3) * nchains) * 8 + (w & 7);
                printf("   wg %3d: %6.1f us (start +%.1f) |", w, (st[2 * g + 1] - st[2 * g]) / 100.0, (st[2 * g] - t0) / 100.0);
                for (int k = 0; k < L.wgs[w].nseg; ++k) { const seg_t &sg = L.wgs[w].seg[k]; const gtask_t &T = L.tasks[sg.task]; printf(" %s%d x %d batches;", T.kind ? "diag" : "w", T.kind ? T.ntile : T.w, sg.b1 - sg.b0); }
                printf("\n");
            }
        }
        ka.stamps = nullptr; CK(hipFree(dst));
    }
    const double us = ms * 1e3 / reps, flops = (double)n * n * q * nchains;
    // check: sampled entries of chains 0 and last against the plain sum on the host
    double worst = 0;
    for (int c : {0, nchains - 1}) {
        std::vector<double> P((size_t)L.nslot * 4096);
        CK(hipMemcpy(P.data(), dG[c], P.size() * 8, hipMemcpyDeviceToHost));
        srand(7 + c);
        for (int t = 0; t < 800; ++t) {
            int i = rand() % n, j = rand() % (i + 1);
            if (t < 60) j = i;
            if (t >= 60 && t < 200) j = (i / 64) * 64 + rand() % (i % 64 + 1);      // inside the diagonal tiles
            double ref = 0, mag = 0;
            for (int k = 0; k < q; ++k) { double v = X[(size_t)k * ld + i] * S[c][k] * X[(size_t)k * ld + j]; ref += v; mag += fabs(v); }
            double got = 0;
            const int ti = i / 64, tj = j / 64;
            for (int sl : L.slots_of_tile[(size_t)ti * (ti + 1) / 2 + tj]) got += P[(size_t)sl * 4096 + (j % 64) * 64 + i % 64];
            worst = std::max(worst, fabs(got - ref) / (mag + 1e-300));
        }
    }
    size_t maxparts = 0; for (auto &v : L.slots_of_tile) maxparts = std::max(maxparts, v.size());
    (void)ntile;
    printf("waves/SIMD=%d  %d chains x %d workgroups (grid %d)  tasks %zu  diag cost %.1f  slots %d (%.1f MB partials, max %zu per tile)  %8.2f us per launch  %6.2f TFLOP/s algorithmic  frac %.3f  | in-kernel span %.1f us, workgroup min %.1f avg %.1f max %.1f | worst rel err %.2e\n",
           WPS, nchains, wg_per_chain, grid, L.tasks.size(), diag_cost, L.nslot, (double)L.nslot * 32768 * nchains / 1e6, maxparts, us, flops / us / 1e6, flops / us / 1e6 / 78.6, span, dmin, davg, dmax, worst);
    fflush(stdout);
    CK(hipFree(dX)); CK(hipFree(dt)); CK(hipFree(dw));
    for (int c = 0; c < nchains; ++c) { CK(hipFree(dS[c])); CK(hipFree(dG[c])); }
}


// uniform synthetic workload for ablations: every workgroup runs ONE w = 16 segment of `nb` batches (results meaningless)
template <int WPS, int EXP>
static void bench_uniform(const char *what, int nb, const std::vector<double> &X, int n_pad, const std::vector<std::vector<double>> &S, int q)
{
    const int nchains = 8, wg_per_chain = 32 * WPS, grid = 256 * WPS;
    std::vector<gtask_t> tasks(1, gtask_t{0, 448, 0, 16, 28, 4});
    std::vector<wg_t> wgs(wg_per_chain);
    for (int w = 0; w < wg_per_chain; ++w) { wgs[w].nseg = 1; wgs[w].seg[0] = seg_t{0, (w * 7) % 300, (w * 7) % 300 + nb, {0, 1, 2, 3}}; }
    double *dX; CK(hipMalloc(&dX, X.size() * 8)); CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    gtask_t *dt; CK(hipMalloc(&dt, sizeof(gtask_t))); CK(hipMemcpy(dt, tasks.data(), sizeof(gtask_t), hipMemcpyHostToDevice));
    wg_t *dw; CK(hipMalloc(&dw, wgs.size() * sizeof(wg_t))); CK(hipMemcpy(dw, wgs.data(), wgs.size() * sizeof(wg_t), hipMemcpyHostToDevice));
    kargs_t ka{};
    double *dS, *dG; CK(hipMalloc(&dS, S[0].size() * 8)); CK(hipMemcpy(dS, S[0].data(), S[0].size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dG, (size_t)4 * 4096 * 8 * nchains));
    for (int c = 0; c < nchains; ++c) { ka.S[c] = dS; ka.G[c] = dG + (size_t)c * 4 * 4096; }
    ka.X = dX; ka.tasks = dt; ka.wgs = dw; ka.ld = n_pad; ka.q = q; ka.nwg_per_chain = wg_per_chain; ka.nchains = nchains;
    unsigned long long *dst; CK(hipMalloc(&dst, (size_t)grid * 24)); CK(hipMemset(dst, 0, (size_t)grid * 24));
    ka.stamps = dst;
    for (int i = 0; i < 3; ++i) k_gramw<WPS, EXP><<<grid, 256>>>(ka);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)grid * 3); CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    double us = 0, cyc = 0, cmax = 0;
    for (int g = 0; g < grid; ++g) { us += (st[2 * g + 1] - st[2 * g]) / 100.0; cyc += (double)st[2 * grid + g]; cmax = std::max(cmax, (double)st[2 * grid + g]); }
    us /= grid; cyc /= grid;
    const double mf = (double)nb * 32.0 * WPS;          // MFMAs per SIMD
    printf("%-46s waves/SIMD=%d  workgroup %.1f us = %.0f cycles (max %.0f): %.1f cycles per MFMA per SIMD (max %.1f), clock %.2f GHz\n", what, WPS, us, cyc, cmax, cyc / mf, cmax / mf, cyc / us / 1e3);
    fflush(stdout);
    CK(hipFree(dX)); CK(hipFree(dt)); CK(hipFree(dw)); CK(hipFree(dS)); CK(hipFree(dG)); CK(hipFree(dst));
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 500, V = argc > 2 ? atoi(argv[2]) : 100, nchains = argc > 3 ? atoi(argv[3]) : 8, reps = 30;
    const int q = V * (V + 1) / 2, n_pad = (n + 63) / 64 * 64, q_alloc = (q + 63) / 64 * 64 + 64;
    std::vector<double> X((size_t)q_alloc * n_pad, 0.0);
    srand(1);
    for (int k = 0; k < q; ++k) for (int i = 0; i < n; ++i) X[(size_t)k * n_pad + i] = (rand() % 2001 - 1000) / 1000.0;
    std::vector<std::vector<double>> S(nchains, std::vector<double>(q_alloc, 1.0));
    for (int c = 0; c < nchains; ++c) for (int k = 0; k < q; ++k) S[c][k] = 0.01 + (rand() % 1000) / 100.0;
    printf("# n=%d V=%d q=%d n_pad=%d, %d chains sharing X\n", n, V, q, n_pad, nchains);
    if (getenv("LAB_ABLATE")) {
        const int nb = 160;
#define AB(E, NAME) do { bench_uniform<1, E>(NAME, nb, X, n_pad, S, q); bench_uniform<2, E>(NAME, nb, X, n_pad, S, q); } while (0)
        AB(0, "full loop");
        AB(1, "no global loads");
        AB(3, "no global loads, no ds_write");
        AB(4, "no v_mul");
        AB(7, "no loads / ds_write / v_mul");
        AB(8, "operands from registers (no ds_read)");
        AB(16, "no barrier");
        AB(15, "MFMA + barrier only");
        AB(31, "MFMA only");
        return 0;
    }
    run<2>(n, q, nchains, 512 / nchains, 5.5, reps, X, n_pad, S);
    run<2>(n, q, nchains, 512 / nchains, 5.0, reps, X, n_pad, S);
    run<2>(n, q, nchains, 512 / nchains, 6.0, reps, X, n_pad, S);
    run<1>(n, q, nchains, 256 / nchains, 5.5, reps, X, n_pad, S);
    run<2>(n, q, nchains, 256 / nchains, 5.5, reps, X, n_pad, S);
    return 0;
}

// bnr_kernels.h -- CDNA4 (gfx950) kernels of the Gibbs sweep (reference: src/gibbs.jl:267-677).
//
// Data layout in HBM (one chain):
//   X      n_pad x q_pad doubles, column-major, zero padded (n_pad % 64 == 0): the reference's n x q model matrix
//   trace  tot x rowlen doubles, ROW-major: row = one Gibbs state
//          [tau2, theta, Delta, mu | xi(V) | lam(R) | pi(R x 3) | M(R x R) | u(R x V) | gamma(q) | S(q)]  (col-major pieces)
//          -> every kernel writes its part of the new state straight into the trace row (coalesced); the
//          reference's iteration-fastest Table layout is produced by a transpose on fetch.
//   plan   one entry per iteration of a run call: {rng iteration id, row to write, row to read, ring wrap flag}
//
// Sweep order and what each kernel covers (gibbs.jl:663-677):
//   k_node      update_tau2! (scalar draw from carried sums) + update_u_xi! (one wave per node, log-space weights)
//   k_xpass     W = lowtri(u' L u), sz = sqrt(S) z1, partial GEMVs X W and X sz          (reads X once)
//   k_gram      X diag(S) X' by v_mfma_f64_16x16x4_f64, split-K partial tiles             (reads X once)
//   k_gram_reduce, k_chol_panel   G + I = L L'
//   k_solve     a4 = (G+I)^-1 (a1 - a3); X gamma_new from n-vectors (no third pass over X)
//   k_backproj  gamma (back-projection X' a4), update_D! (GIG draws), partial sums for theta and Lambda (reads X once)
//   k_tail      update_theta!, update_Delta!, update_M!, update_mu!, update_Lambda!, update_pi!, carried sums
#pragma once
#include "bnr_rng.h"

#define BNR_RMAX 32          // latent dimension limit of the small-matrix routines
#define BNR_NB 32            // Cholesky block size
#define BNR_GT 64            // Gram workgroup tile

struct bnr_plan_entry { uint32_t it; int32_t row; int32_t prev; int32_t wrap; };   // rows 0-based

struct bnr_dev {
    // sizes
    int n, n_pad, V, R, q, q_pad, tot;
    // row layout (offsets in doubles)
    int o_xi, o_lam, o_pi, o_M, o_u, o_gamma, o_S, rowlen;
    // hyper
    double eta, zeta, iota, aDelta, bDelta, nu;
    uint64_t seed;
    // inputs
    const double *X, *y;
    const int *ek, *el;          // edge e -> column node k, row node l (l >= k)
    // state
    double *trace;
    const bnr_plan_entry *plan;
    const int *pbase;            // plan[pbase[0] + s] is the entry of slot s (lets a captured graph be replayed)
    // work
    double *Wbuf, *sz;           // q_pad each
    double *PW, *PA;             // nblk_x x n_pad GEMV partials (X W, X sz)
    double *PG;                  // nblk_x x n_pad GEMV partials (X gamma, refresh path)
    int nblk_x, chunk_x;
    double *Gpart, *G, *Winv;    // Gram partial tiles, G (n_pad x n_pad col-major; holds L after the factorization), Winv = L^-1
    int ksplit, ntile;           // ntile = n_pad/64
    double *a3, *xw, *a4, *res, *xg, *bw;   // n_pad each (bw: rhs b, overwritten by w = L^-1 b)
    double *scal;                // [0]=rr (sum res^2), [1]=sig_q (sum (g^2/2)/S), [2]=tau (sqrt tau2 of current row)
    double *Psum;                // nblk_bp x (1+3R) partial sums from k_backproj
    int nblk_bp, chunk_bp;
    long long *counters;         // [0] jitter, [1] nan_w, [2] sampler cap, [3] chol fail
};

enum { ROW_TAU2 = 0, ROW_THETA = 1, ROW_DELTA = 2, ROW_MU = 3 };
enum { SC_RR = 0, SC_SIGQ = 1, SC_TAU = 2 };

// ----------------------------------------------------------------------------------------- helpers
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double block_sum(double v, double *sh /* >= blockDim/64 doubles */)
{
    v = wave_sum(v);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += sh[i];
    __syncthreads();
    return s;
}

// ---- single-wave small dense routines on LDS matrices (column-major R x R), blockDim.x == 64 or wave 0 of a block.
// `sync` must be a barrier valid for the participating threads: we always call these from code paths where the
// WHOLE block executes them (other waves just take part in the barriers).
// in-place lower Cholesky, right-looking; returns 0 ok / 1 not positive definite (uniform across the block)
__device__ inline int lds_chol(double *A, int R, int lane_in_block)
{
    for (int j = 0; j < R; ++j) {
        double d = A[j + R * j];
        if (!(d > 0.0) || !isfinite(d)) return 1;
        d = sqrt(d);
        __syncthreads();
        if (lane_in_block == j) A[j + R * j] = d;
        else if (lane_in_block > j && lane_in_block < R) A[lane_in_block + R * j] = A[lane_in_block + R * j] / d;
        __syncthreads();
        int m = R - j - 1;
        for (int idx = lane_in_block; idx < m * m; idx += blockDim.x) {
            int c = j + 1 + idx / m, i = j + 1 + idx % m;
            if (i >= c) A[i + R * c] = A[i + R * c] - A[i + R * j] * A[c + R * j];
        }
        __syncthreads();
    }
    return 0;
}
// x (one value per lane, lane i < R holds b_i) <- L^-1 b ; column oriented, same operation order as a row sweep
__device__ __forceinline__ double wave_fwd_solve(const double *L, int R, int lane, double b)
{
    for (int k = 0; k < R; ++k) {
        double xk = __shfl(b, k, 64) / L[k + R * k];
        if (lane == k) b = xk;
        else if (lane > k && lane < R) b = b - L[lane + R * k] * xk;
    }
    return b;
}
// x <- L^-T b
__device__ __forceinline__ double wave_bwd_solve_T(const double *L, int R, int lane, double b)
{
    for (int k = R - 1; k >= 0; --k) {
        double xk = __shfl(b, k, 64) / L[k + R * k];
        if (lane == k) b = xk;
        else if (lane < k) b = b - L[k + R * lane] * xk;
    }
    return b;
}

// W_e = sum_r lam_r u[r,l] u[r,k]   (utils.jl:50-55 order; gibbs.jl:271,421,455)
__device__ __forceinline__ double edge_W(const double *u, const double *lam, int R, int l, int k)
{
    double s = 0.0;
    for (int r = 0; r < R; ++r) s += u[r + R * l] * lam[r] * u[r + R * k];
    return s;
}

// ===================================================================================== k_node
// grid = V blocks, 64 threads (one wavefront per node).  mode bit0: draw tau2; bit1: node update.
// update_tau2! (gibbs.jl:267-277): tau2 ~ InverseGamma(n/2 + V(V+1)/4, rr/2 + sig_q) from the carried sums.
// update_u_xi! (gibbs.jl:293-371) for node k = blockIdx.x, all inputs from row prev (no Gauss-Seidel).  Weights in
// log space (determinant lemma + Woodbury): log w_bot - log w_top = log(D/(1-D)) - 1/2[logdet M + logdet Sigma^-1]
// + 1/2 b' Sigma b,  b = U'H^-1 gamma_k / tau2 -- equal to the reference's ratio of dense (V-1)-dim pdfs whenever
// those do not under/overflow (gibbs.jl:349-351).
__global__ __launch_bounds__(64) void k_node(bnr_dev cd, int s, int mode)
{
    __shared__ double sM[BNR_RMAX * BNR_RMAX], sMinv[BNR_RMAX * BNR_RMAX], sS[BNR_RMAX * BNR_RMAX], sL[BNR_RMAX * BNR_RMAX];
    __shared__ double slam[BNR_RMAX], sc[BNR_RMAX];
    const int lane = threadIdx.x, k = blockIdx.x, V = cd.V, R = cd.R;
    const bnr_plan_entry P = cd.plan[cd.pbase[0] + s];
    double *row = cd.trace + (size_t)P.row * cd.rowlen;
    const double *prev = cd.trace + (size_t)P.prev * cd.rowlen;
    int cap = 0;
    double tau2;
    if (mode & 1) {
        double sigma = cd.scal[SC_RR] / 2.0 + cd.scal[SC_SIGQ];
        double shape = (cd.n / 2.0) + (V * (V + 1) / 4.0);
        tau2 = sigma / bnr_gamma(cd.seed, shape, P.it, SITE_TAU2, 0, &cap);
        if (k == 0 && lane == 0) { row[ROW_TAU2] = tau2; cd.scal[SC_TAU] = sqrt(tau2); }
    } else {
        tau2 = row[ROW_TAU2];
        if (k == 0 && lane == 0) cd.scal[SC_TAU] = sqrt(tau2);
    }
    if (!(mode & 2)) { if (cap && lane == 0 && k == 0) atomicAdd((unsigned long long *)&cd.counters[2], 1ull); return; }

    const double Delta = prev[ROW_DELTA];
    const double *pu = prev + cd.o_u, *pg = prev + cd.o_gamma, *pS = prev + cd.o_S;
    if (lane < R) slam[lane] = prev[cd.o_lam + lane];
    for (int i = lane; i < R * R; i += 64) sM[i] = prev[cd.o_M + i];
    __syncthreads();

    // A[x,y] = sum_a U[a,x] U[a,y] / h_a ,  c[x] = sum_a U[a,x] g_a / h_a   over the V-1 other nodes
    for (int x = 0; x < R; ++x) {
        for (int y = x; y < R; ++y) {
            double acc = 0.0;
            for (int a = lane; a < V - 1; a += 64) {
                int l = a < k ? a : a + 1;
                int e = l > k ? bnr_edge_index(V, l, k) : bnr_edge_index(V, k, l);
                acc += (pu[x + R * l] * slam[x]) * ((pu[y + R * l] * slam[y]) / pS[e]);
            }
            acc = wave_sum(acc);
            if (lane == 0) { sS[x + R * y] = acc; sS[y + R * x] = acc; }
        }
        double acc = 0.0;
        for (int a = lane; a < V - 1; a += 64) {
            int l = a < k ? a : a + 1;
            int e = l > k ? bnr_edge_index(V, l, k) : bnr_edge_index(V, k, l);
            acc += (pu[x + R * l] * slam[x]) * (pg[e] / pS[e]);
        }
        acc = wave_sum(acc);
        if (lane == 0) sc[x] = acc;
    }
    __syncthreads();

    // inv(M) and logdet M via Cholesky of M_prev (gibbs.jl:315)
    for (int i = lane; i < R * R; i += 64) sL[i] = sM[i];
    __syncthreads();
    int fail = lds_chol(sL, R, lane);
    double logdetM = 0.0;
    for (int i = 0; i < R; ++i) logdetM += 2.0 * log(sL[i + R * i]);
    // column j of inv(M): forward + backward solve of e_j
    for (int j = 0; j < R; ++j) {
        double b = (lane == j) ? 1.0 : 0.0;
        b = wave_fwd_solve(sL, R, lane, b);
        b = wave_bwd_solve_T(sL, R, lane, b);
        if (lane < R) sMinv[lane + R * j] = b;
    }
    __syncthreads();
    // Sigma^-1 = A / tau2 + inv(M), Cholesky with the reference's jitter ladder (gibbs.jl:322-347)
    for (int i = lane; i < R * R; i += 64) sS[i] = sS[i] / tau2 + sMinv[i];
    __syncthreads();
    for (int i = lane; i < R * R; i += 64) sL[i] = sS[i];
    __syncthreads();
    int f2 = lds_chol(sL, R, lane);
    if (f2) {
        if (lane == 0) atomicAdd((unsigned long long *)&cd.counters[0], 1ull);
        __syncthreads();
        for (int i = lane; i < R; i += 64) sS[i + R * i] += 1e-5;
        __syncthreads();
        for (int i = lane; i < R * R; i += 64) sL[i] = sS[i];
        __syncthreads();
        f2 = lds_chol(sL, R, lane);
        if (f2) {
            __syncthreads();
            for (int i = lane; i < R; i += 64) sS[i + R * i] += 4e-5;
            __syncthreads();
            for (int i = lane; i < R * R; i += 64) sL[i] = sS[i];
            __syncthreads();
            f2 = lds_chol(sL, R, lane);
        }
    }
    if (fail || f2) {
        if (lane == 0) atomicAdd((unsigned long long *)&cd.counters[3], 1ull);
        if (lane < R) row[cd.o_u + lane + R * k] = NAN;
        if (lane == 0) row[cd.o_xi + k] = NAN;
        return;
    }
    double ldS = 0.0;
    for (int i = 0; i < R; ++i) ldS += 2.0 * log(sL[i + R * i]);
    // b = c / tau2 ; mu_t = Sigma b (gibbs.jl:364)
    double b = (lane < R) ? sc[lane] / tau2 : 0.0;
    double mt = wave_fwd_solve(sL, R, lane, b);
    mt = wave_bwd_solve_T(sL, R, lane, mt);
    double qf = wave_sum((lane < R) ? b * mt : 0.0);
    double logit = log(Delta) - log1p(-Delta) - 0.5 * (logdetM + ldS) + 0.5 * qf;
    double w = 1.0 / (1.0 + exp(logit));
    // update_xi (gibbs.jl:385-402)
    double xi;
    if (w <= 0.0) xi = 1.0;
    else if (w >= 1.0) xi = 0.0;
    else {
        double ua, ub;
        bnr_draw2(cd.seed, P.it, SITE_XI, (uint32_t)k, 0, ua, ub);
        if (isnan(w)) { xi = (ua <= 0.5) ? 1.0 : 0.0; if (lane == 0) atomicAdd((unsigned long long *)&cd.counters[1], 1ull); }
        else xi = (ua <= 1.0 - w) ? 1.0 : 0.0;
    }
    // u_k = xi (mu_t + inv(C.U) z), z ~ N(0, I_R) (gibbs.jl:365-367)
    double z = (lane < R) ? bnr_normal(cd.seed, P.it, SITE_U_Z, (uint32_t)(k * R + lane), 0) : 0.0;
    z = wave_bwd_solve_T(sL, R, lane, z);
    if (lane < R) row[cd.o_u + lane + R * k] = xi * (mt + z);
    if (lane == 0) row[cd.o_xi + k] = xi;
    if (cap && lane == 0 && k == 0) atomicAdd((unsigned long long *)&cd.counters[2], 1ull);
}

// ===================================================================================== k_xpass
// One pass over X.  grid = nblk_x blocks of 256 threads; block b owns columns [b*chunk, (b+1)*chunk).
//   W_e  = lowtri(u_new' diag(lam_prev) u_new)_e               (gibbs.jl:421)           -> Wbuf
//   sz_e = sqrt(S_prev,e) * z1_e   (Delta_gamma1 = tau * sz)   (gibbs.jl:429)           -> sz
//   PW[b][i] = sum_{e in chunk} X[i,e] W_e ;  PA[b][i] = sum X[i,e] sz_e                 (gibbs.jl:432-433)
// which: bit0 -> W/PW, bit1 -> sz/PA, bit2 -> PG = partial X*gamma(row `P.prev` if bit3 else row P.row)
__global__ __launch_bounds__(256) void k_xpass(bnr_dev cd, int s, int which)
{
    extern __shared__ double sh[];
    double *sW = sh, *sZ = sh + cd.chunk_x, *sG = sh + 2 * cd.chunk_x;
    const bnr_plan_entry P = cd.plan[cd.pbase[0] + s];
    const double *row = cd.trace + (size_t)P.row * cd.rowlen;
    const double *prev = cd.trace + (size_t)P.prev * cd.rowlen;
    const int R = cd.R;
    const int e0 = blockIdx.x * cd.chunk_x;
    const int ne = min(cd.chunk_x, cd.q - e0);
    const double *grow = (which & 8) ? prev : row;
    for (int t = threadIdx.x; t < cd.chunk_x; t += blockDim.x) {
        double w = 0.0, zz = 0.0, g = 0.0;
        int e = e0 + t;
        if (t < ne) {
            if (which & 1) { w = edge_W(row + cd.o_u, prev + cd.o_lam, R, cd.el[e], cd.ek[e]); cd.Wbuf[e] = w; }
            if (which & 2) { zz = sqrt(prev[cd.o_S + e]) * bnr_normal(cd.seed, P.it, SITE_G_Z1, (uint32_t)e, 0); cd.sz[e] = zz; }
            if (which & 4) g = grow[cd.o_gamma + e];
        }
        sW[t] = w; sZ[t] = zz; sG[t] = g;
    }
    __syncthreads();
    const size_t ld = cd.n_pad;
    for (int i = threadIdx.x; i < cd.n_pad; i += blockDim.x) {
        const double *xp = cd.X + (size_t)e0 * ld + i;
        double aw = 0.0, aa = 0.0, ag = 0.0;
        if ((which & 7) == 3) {
#pragma unroll 4
            for (int t = 0; t < ne; ++t) { double x = xp[(size_t)t * ld]; aw = fma(x, sW[t], aw); aa = fma(x, sZ[t], aa); }
        } else {
            for (int t = 0; t < ne; ++t) {
                double x = xp[(size_t)t * ld];
                aw = fma(x, sW[t], aw); aa = fma(x, sZ[t], aa); ag = fma(x, sG[t], ag);
            }
        }
        if (which & 1) cd.PW[(size_t)blockIdx.x * ld + i] = aw;
        if (which & 2) cd.PA[(size_t)blockIdx.x * ld + i] = aa;
        if (which & 4) cd.PG[(size_t)blockIdx.x * ld + i] = ag;
    }
}

// ===================================================================================== k_gram
// G = X diag(S_prev) X'  (the n x n matrix of gibbs.jl:434 without the identity; tau cancels: Xt tau2 D Xt' = X D X').
// v_mfma_f64_16x16x4_f64, D = A*B + C with A[m][k] (lane l: m = l&15, k = l>>4), B[k][n] (k = l>>4, n = l&15),
// C[m][n]: lane l holds rows m = (l>>4) + 4*reg, column n = l&15.
// Here m indexes the tile's COLUMN (j) and n its ROW (i): A = X[j-rows], B = S_e X[i-rows], so that a lane's results are
// consecutive in i and the tile is written column-major (i fastest) with coalesced 128-byte segments.
// Workgroup = 1024 threads = 16 waves on one 64x64 tile of the LOWER triangle: 4 K-groups x (2x2 waves of 32x32).
// Measured issue rate of the f64 MFMA rises with waves per SIMD (1 wave: 139, 2: 103, 4: 92 cycles/MFMA;
// profiles/round1_mfma_f64_peak.txt), hence 4 waves per SIMD and the in-workgroup K split, reduced through LDS.
// blockIdx.y = K slice across workgroups (split-K partials, summed by k_gram_reduce).
typedef double bnr_d4 __attribute__((ext_vector_type(4)));
#define BNR_GRAM_KG 4

__global__ __launch_bounds__(1024) void k_gram(bnr_dev cd, int s)
{
    __shared__ double sred[BNR_GRAM_KG * BNR_GT * BNR_GT];     // 128 KiB: one 64x64 tile per K-group
    const bnr_plan_entry P = cd.plan[cd.pbase[0] + s];
    const double *Sp = cd.trace + (size_t)P.prev * cd.rowlen + cd.o_S;
    int t = blockIdx.x, ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    int tj = t - ti * (ti + 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kg = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
    const int i0 = ti * BNR_GT + wi * 32, j0 = tj * BNR_GT + wj * 32;
    const int ks = blockIdx.y;
    const int kchunk = cd.q_pad / cd.ksplit;          // multiple of 16 (host guarantees)
    const int ksub = kchunk / BNR_GRAM_KG;            // multiple of 4
    const int eb = ks * kchunk + kg * ksub, ee = eb + ksub;
    const size_t ld = cd.n_pad;
    const int li = lane & 15, lk = lane >> 4;
    const double *xi = cd.X + (size_t)(i0 + li) + (size_t)(eb + lk) * ld;
    const double *xj = cd.X + (size_t)(j0 + li) + (size_t)(eb + lk) * ld;
    bnr_d4 c00 = {0, 0, 0, 0}, c01 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};   // c[jt][it]
#pragma unroll 4
    for (int e = eb; e < ee; e += 4) {
        int ecol = e + lk;
        double sv = (ecol < cd.q) ? Sp[ecol] : 0.0;
        double a0 = xj[0], a1 = xj[16];
        double b0 = xi[0] * sv, b1 = xi[16] * sv;
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
        xi += 4 * ld; xj += 4 * ld;
    }
    // tile element (i,j) lives at [j*64 + i]; this lane: j = wj*32 + jt*16 + (lane>>4) + 4 r, i = wi*32 + it*16 + (lane&15)
    double *mine = sred + (size_t)kg * (BNR_GT * BNR_GT);
    const int jb = wj * 32 + (lane >> 4), ib = wi * 32 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        mine[(jb + 4 * r) * BNR_GT + ib] = c00[r];
        mine[(jb + 4 * r) * BNR_GT + ib + 16] = c01[r];
        mine[(jb + 16 + 4 * r) * BNR_GT + ib] = c10[r];
        mine[(jb + 16 + 4 * r) * BNR_GT + ib + 16] = c11[r];
    }
    __syncthreads();
    double *out = cd.Gpart + ((size_t)ks * (cd.ntile * (cd.ntile + 1) / 2) + t) * (BNR_GT * BNR_GT);
#pragma unroll
    for (int idx = threadIdx.x; idx < BNR_GT * BNR_GT; idx += 1024)
        out[idx] = (sred[idx] + sred[BNR_GT * BNR_GT + idx]) + (sred[2 * BNR_GT * BNR_GT + idx] + sred[3 * BNR_GT * BNR_GT + idx]);
}

// G = sum_ks partial + I, lower tiles of the n_pad x n_pad column-major matrix.  grid = (lower tiles, 4), 256 threads.
__global__ __launch_bounds__(256) void k_gram_reduce(bnr_dev cd)
{
    int t = blockIdx.x, ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    int tj = t - ti * (ti + 1) / 2;
    const int ntl = cd.ntile * (cd.ntile + 1) / 2;
    const size_t ld = cd.n_pad, tsz = BNR_GT * BNR_GT;
    const int idx0 = blockIdx.y * (BNR_GT * BNR_GT / 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int idx = idx0 + u * 256 + threadIdx.x;
        double sacc = 0.0;
        for (int ks = 0; ks < cd.ksplit; ++ks) sacc += cd.Gpart[((size_t)ks * ntl + t) * tsz + idx];
        int i = ti * BNR_GT + idx % BNR_GT, j = tj * BNR_GT + idx / BNR_GT;
        if (i == j) sacc += 1.0;
        cd.G[(size_t)i + ld * j] = sacc;
    }
}

// ===================================================================================== blocked Cholesky + solve
// (G + I) a4 = b,  b = a1 - a3  (gibbs.jl:434; the reference's generic `\` is an LU solve of this SPD system).
// Right-looking blocked Cholesky, NB = 32, ONE launch per panel with lookahead 1 (k_chol_step(p), p = 0..nbk-1).
// The matrix is extended by two kinds of extra rows that ride through the same sweeps:
//     [ G + I ]  block rows 0..nbk-1        -> L
//     [  b'   ]  one row                    -> w' = (L^-1 b)'          (the forward substitution, for free)
//     [   I   ]  block rows 0..nbk-1        -> L^-T                    (so the back substitution becomes one GEMV)
//   role A (panel workgroups): a block row first applies panel p-1's update to its blocks (p,p) and (i,p), then one
//       wavefront sweeps the 64 x 32 register-resident panel [A_pp ; A_ip] column by column: pivot broadcast by
//       v_readlane, rsqrt by v_rsq_f64 + 2 Newton steps, rank-1 updates by v_fma_f64 with the broadcast in SGPRs.
//       Lanes 0-31 redo the diagonal block in every workgroup (no cross-workgroup hand-off inside a launch).
//   role B (update workgroups): apply panel p-1's update to every block (.,j), j >= p+1.
// Identity block row r is zero left of column block r, so it takes part from panel r on.  L^-T is stored transposed,
// Winv = L^-1 column-major, so that row r of L^-T is the contiguous column r of Winv.
// k_solve_gemv: a4 = L^-T w, then the n-vector bookkeeping for X gamma_new.
__device__ __forceinline__ double bnr_readlane(double v, int srclane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bnr_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    return y;
}

#define BNR_LP (BNR_NB + 1)
// number of workgroups of launch p
__host__ __device__ inline int bnr_chol_npanel(int nbk, int p) { return (nbk - p) + 1 + (p + 1); }
__host__ __device__ inline int bnr_chol_ntile(int nbk, int p)
{
    if (p == 0) return 0;
    int m = nbk - (p + 1);
    return m * (m + 1) / 2 + m + p * m;          // A tiles, b-row tiles, identity-row tiles (r <= p-1)
}

__global__ __launch_bounds__(256) void k_chol_step(bnr_dev cd, int p)
{
    __shared__ double sD[BNR_NB * BNR_LP], sB[BNR_NB * BNR_LP], sLp[BNR_NB * BNR_LP], sLi[BNR_NB * BNR_LP];
    const int tid = threadIdx.x, nbk = cd.n_pad / BNR_NB;
    const size_t ld = cd.n_pad;
    const int npanel = bnr_chol_npanel(nbk, p);
    const int r = tid & 31, c0 = tid >> 5;                 // thread owns elements (r, c0 + 8 m), m = 0..3
    const int pc = p * BNR_NB;
    if ((int)blockIdx.x >= npanel) {
        // ------------------------------------------------ role B: blk[.,j] -= blk[.,p-1] L[j,p-1]'
        const int m = nbk - (p + 1);
        int t = blockIdx.x - npanel;
        const int ntri = m * (m + 1) / 2;
        const int kc = pc - BNR_NB;
        if (t >= ntri && t < ntri + m) {
            const int j = p + 1 + (t - ntri);              // b row
            if (tid < BNR_NB) {
                double acc = cd.bw[j * BNR_NB + tid];
                for (int k = 0; k < BNR_NB; ++k) acc = fma(-cd.bw[kc + k], cd.G[(size_t)(j * BNR_NB + tid) + ld * (kc + k)], acc);
                cd.bw[j * BNR_NB + tid] = acc;
            }
            return;
        }
        int i, j, ident;
        if (t < ntri) {
            int ti = 0;
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            int tj = t - ti * (ti + 1) / 2;
            i = p + 1 + ti; j = p + 1 + tj; ident = 0;
        } else {
            t -= ntri + m;
            i = t / m; j = p + 1 + t % m; ident = 1;       // identity block row i (<= p-1)
        }
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            int c = c0 + 8 * mm;
            // Linv^T block (i, kc-block) is stored transposed in Winv: element (r,c) at Winv[(kc + c) + ld*(i*NB + r)]
            sLi[r + BNR_LP * c] = ident ? cd.Winv[(size_t)(kc + c) + ld * (i * BNR_NB + r)] : cd.G[(size_t)(i * BNR_NB + r) + ld * (kc + c)];
            sLp[r + BNR_LP * c] = cd.G[(size_t)(j * BNR_NB + r) + ld * (kc + c)];
        }
        __syncthreads();
        if (!ident) {
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                int c = c0 + 8 * mm;
                double *dst = cd.G + (size_t)(i * BNR_NB + r) + ld * (j * BNR_NB + c);
                double acc = *dst;
#pragma unroll 8
                for (int k = 0; k < BNR_NB; ++k) acc = fma(-sLi[r + BNR_LP * k], sLp[c + BNR_LP * k], acc);
                *dst = acc;
            }
        } else {
            // transposed store: thread owns (row c0 + 8 mm of the block, column r) so that stores are contiguous in r
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                int rr2 = c0 + 8 * mm;                      // row of the (i,j) block
                double *dst = cd.Winv + (size_t)(j * BNR_NB + r) + ld * (i * BNR_NB + rr2);   // element (rr2, r)
                double acc = (p - 1 == i) ? 0.0 : *dst;     // first update of this identity row: the block starts at zero
#pragma unroll 8
                for (int k = 0; k < BNR_NB; ++k) acc = fma(-sLi[rr2 + BNR_LP * k], sLp[r + BNR_LP * k], acc);
                *dst = acc;
            }
        }
        return;
    }
    // ---------------------------------------------------- role A: panel workgroup
    const int b = blockIdx.x;
    int kind, i;
    if (b < nbk - p) { kind = 0; i = p + b; }              // block row i of G + I
    else if (b == nbk - p) { kind = 1; i = 0; }            // b row
    else { kind = 2; i = b - (nbk - p) - 1; }              // identity block row i (0..p)
    const int kc = pc - BNR_NB;
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
        int c = c0 + 8 * mm;
        sD[r + BNR_LP * c] = cd.G[(size_t)(pc + r) + ld * (pc + c)];
        double v;
        if (kind == 0) v = cd.G[(size_t)(i * BNR_NB + r) + ld * (pc + c)];
        else if (kind == 1) v = (r == 0) ? cd.bw[pc + c] : 0.0;
        else v = (i == p) ? ((r == c) ? 1.0 : 0.0) : ((i == p - 1) ? 0.0 : cd.Winv[(size_t)(pc + c) + ld * (i * BNR_NB + r)]);
        sB[r + BNR_LP * c] = v;
        if (p > 0) {
            sLp[r + BNR_LP * c] = cd.G[(size_t)(pc + r) + ld * (kc + c)];
            double w;
            if (kind == 0) w = cd.G[(size_t)(i * BNR_NB + r) + ld * (kc + c)];
            else if (kind == 1) w = (r == 0) ? cd.bw[kc + c] : 0.0;
            else w = (i <= p - 1) ? cd.Winv[(size_t)(kc + c) + ld * (i * BNR_NB + r)] : 0.0;
            sLi[r + BNR_LP * c] = w;
        }
    }
    __syncthreads();
    if (p > 0) {
        double accD[4], accB[4];
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            int c = c0 + 8 * mm;
            double d = sD[r + BNR_LP * c], bb = sB[r + BNR_LP * c];
#pragma unroll 8
            for (int k = 0; k < BNR_NB; ++k) {
                double lpc = sLp[c + BNR_LP * k];
                d = fma(-sLp[r + BNR_LP * k], lpc, d);
                bb = fma(-sLi[r + BNR_LP * k], lpc, bb);
            }
            accD[mm] = d; accB[mm] = bb;
        }
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) { int c = c0 + 8 * mm; sD[r + BNR_LP * c] = accD[mm]; sB[r + BNR_LP * c] = accB[mm]; }
        __syncthreads();
    }
    if (tid < 64) {
        // panel sweep, lane = row: lanes 0..31 rows of the diagonal block, lanes 32..63 rows of the own block
        const int lane = tid, rr = lane & 31;
        const double *src = (lane < 32) ? sD : sB;
        double a[BNR_NB];
#pragma unroll
        for (int c = 0; c < BNR_NB; ++c) a[c] = src[rr + BNR_LP * c];
        int bad = 0;
#pragma unroll
        for (int j = 0; j < BNR_NB; ++j) {
            double piv = bnr_readlane(a[j], j);
            if (!(piv > 0.0)) { bad = 1; piv = 1.0; }
            double rinv = bnr_rsqrt(piv);
            double lj = a[j] * rinv;
            a[j] = lj;
#pragma unroll
            for (int k = j + 1; k < BNR_NB; ++k) {
                double lk = bnr_readlane(lj, k);
                a[k] = fma(-lj, lk, a[k]);
            }
        }
        if (bad && lane == 0 && b == 0) atomicAdd((unsigned long long *)&cd.counters[3], 1ull);
        double *dstl = (lane < 32) ? sD : sB;
#pragma unroll
        for (int c = 0; c < BNR_NB; ++c) dstl[rr + BNR_LP * c] = (lane < 32 && c > rr) ? 0.0 : a[c];
    }
    __syncthreads();
    if (kind == 0) {
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            int c = c0 + 8 * mm;
            if (b == 0) cd.G[(size_t)(pc + r) + ld * (pc + c)] = sD[r + BNR_LP * c];
            else cd.G[(size_t)(i * BNR_NB + r) + ld * (pc + c)] = sB[r + BNR_LP * c];
        }
    } else if (kind == 1) {
        if (tid < BNR_NB) cd.bw[pc + tid] = sB[BNR_LP * tid];
    } else {
        // (L^-T)[i-block row rr2, p-block col r] -> Winv[(pc + r) + ld*(i*NB + rr2)]  (contiguous in r)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            int rr2 = c0 + 8 * mm;
            cd.Winv[(size_t)(pc + r) + ld * (i * BNR_NB + rr2)] = sB[rr2 + BNR_LP * r];
        }
    }
}

// Right-hand side: finishes the GEMVs of k_xpass and forms b = a1 - a3 (gibbs.jl:432-434):
//   a1 = (y - X W - mu_prev)/tau, a3 = X sz + z2  (note (X/tau) Delta_gamma1 = X sz).
// grid = n_pad/64 blocks of 256 threads: 64 rows x 4 partial groups; fixed summation order (deterministic).
__global__ __launch_bounds__(256) void k_rhs(bnr_dev cd, int s)
{
    __shared__ double sw[4][64], sa[4][64];
    const bnr_plan_entry P = cd.plan[cd.pbase[0] + s];
    const double *prev = cd.trace + (size_t)P.prev * cd.rowlen;
    const double tau = cd.scal[SC_TAU], mu = prev[ROW_MU];
    const int n = cd.n;
    const size_t ld = cd.n_pad;
    const int il = threadIdx.x & 63, g = threadIdx.x >> 6, i = blockIdx.x * 64 + il;
    double xw = 0.0, xs = 0.0;
    const int nb = cd.nblk_x;
    int b = g;
    for (; b + 28 < nb; b += 32) {
        double w[8], a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { w[u] = cd.PW[(size_t)(b + 4 * u) * ld + i]; a[u] = cd.PA[(size_t)(b + 4 * u) * ld + i]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { xw += w[u]; xs += a[u]; }
    }
    for (; b < nb; b += 4) { xw += cd.PW[(size_t)b * ld + i]; xs += cd.PA[(size_t)b * ld + i]; }
    sw[g][il] = xw; sa[g][il] = xs;
    __syncthreads();
    if (g == 0) {
        xw = (sw[0][il] + sw[1][il]) + (sw[2][il] + sw[3][il]);
        xs = (sa[0][il] + sa[1][il]) + (sa[2][il] + sa[3][il]);
        double z2 = (i < n) ? bnr_normal(cd.seed, P.it, SITE_G_Z2, (uint32_t)i, 0) : 0.0;
        double bb = (i < n) ? ((cd.y[i] - xw - mu) / tau - (xs + z2)) : 0.0;
        cd.xw[i] = xw; cd.a3[i] = xs;          // a3 buffer keeps X sz (without z2)
        cd.bw[i] = bb; cd.res[i] = bb;         // bw is overwritten by w = L^-1 b; res keeps b
    }
}

// a4 = L^-T w : a4_r = sum_{c >= r} Linv[c, r] w_c = <column r of Winv, w>   (one wavefront per row r), then
//   X gamma_new = X W + tau X sz + tau G a4,  G a4 = b - a4   (no third pass over X).  grid = n_pad/4 blocks of 256.
__global__ __launch_bounds__(256) void k_solve_gemv(bnr_dev cd)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rrow = blockIdx.x * 4 + wave;
    const size_t ld = cd.n_pad;
    const double *col = cd.Winv + ld * (size_t)rrow;
    const int cstart = (rrow / BNR_NB) * BNR_NB;       // Winv is lower triangular: entries c >= r (block granularity)
    double acc = 0.0;
    for (int cc = cstart + lane; cc < cd.n_pad; cc += 64) acc = fma(col[cc], cd.bw[cc], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
        const double tau = cd.scal[SC_TAU];
        double b = cd.res[rrow];
        cd.a4[rrow] = acc;
        cd.xg[rrow] = (rrow < cd.n) ? (cd.xw[rrow] + tau * cd.a3[rrow] + tau * (b - acc)) : 0.0;
    }
}

// ===================================================================================== k_backproj
// grid = nblk_bp blocks of 256 threads; block owns chunk_bp (<= 64) consecutive edges.
// flags bit0: compute gamma (else read from row); bit1: draw S (else read from row); bit2: partial sums.
//   gamma_e = W_e + tau (sz_e + S_prev,e x_e' a4)                           (gibbs.jl:435-436)
//   S_e ~ GIG(1/2, chi = (gamma_e - W_e)^2 / tau2, psi = theta_prev)         (gibbs.jl:454-458, gig.jl)
//   Psum[b][0] = sum_e S_e ; Psum[b][1+3r+c] = sum_e logpdf(Normal(W_c,e, sqrt(tau2 S_e)), gamma_e)  (gibbs.jl:603-605)
__global__ __launch_bounds__(256) void k_backproj(bnr_dev cd, int s, int flags)
{
    extern __shared__ double sh[];          // n_pad (a4) + 64 (dots)
    double *sa = sh, *sdot = sh + cd.n_pad;
    const bnr_plan_entry P = cd.plan[cd.pbase[0] + s];
    double *row = cd.trace + (size_t)P.row * cd.rowlen;
    const double *prev = cd.trace + (size_t)P.prev * cd.rowlen;
    const int R = cd.R, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int e0 = blockIdx.x * cd.chunk_bp, ne = min(cd.chunk_bp, cd.q - e0);
    const size_t ld = cd.n_pad;
    const double tau2 = row[ROW_TAU2], tau = sqrt(tau2);
    if (flags & 1) {
        for (int i = tid; i < cd.n_pad; i += blockDim.x) sa[i] = cd.a4[i];
        __syncthreads();
        for (int t = wave; t < ne; t += 4) {
            const double *xc = cd.X + (size_t)(e0 + t) * ld;
            double acc = 0.0;
            for (int i = lane; i < cd.n_pad; i += 64) acc = fma(xc[i], sa[i], acc);
            acc = wave_sum(acc);
            if (lane == 0) sdot[t] = acc;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    int cap = 0;
    const int e = e0 + lane;
    const bool act = lane < ne;
    double gam = 0.0, Snew = 1.0, W = 0.0;
    int l = 0, k = 0;
    if (act) {
        W = cd.Wbuf[e]; l = cd.el[e]; k = cd.ek[e];
        if (flags & 1) {
            double Sp = prev[cd.o_S + e];
            gam = tau * (cd.sz[e] + Sp * sdot[lane]) + W;
            row[cd.o_gamma + e] = gam;
        } else gam = row[cd.o_gamma + e];
        if (flags & 2) {
            double g = gam - W;
            double chi = (g * g) / tau2;
            Snew = bnr_gig(cd.seed, 0.5, chi, prev[ROW_THETA], P.it, (uint32_t)e, &cap);
            row[cd.o_S + e] = Snew;
        } else Snew = row[cd.o_S + e];
    }
    if (!(flags & 4)) { if (cap) atomicAdd((unsigned long long *)&cd.counters[2], 1ull); return; }
    double *ps = cd.Psum + (size_t)blockIdx.x * (1 + 3 * R);
    double ssum = wave_sum(act ? Snew : 0.0);
    if (lane == 0) ps[0] = ssum;
    const double *un = row + cd.o_u, *lamp = prev + cd.o_lam;
    const double sd = sqrt(tau2 * Snew), lsd = log(sd) + 0.5 * log(2.0 * BNR_PI);
    for (int r = 0; r < R; ++r) {
        double dr = act ? un[r + R * l] * un[r + R * k] : 0.0;
        double lr = lamp[r];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double Wc = W + (bnr_lambda_value(c) - lr) * dr;
            double zz = (gam - Wc) / sd;
            double term = act ? (-0.5 * zz * zz - lsd) : 0.0;
            term = wave_sum(term);
            if (lane == 0) ps[1 + 3 * r + c] = term;
        }
    }
    if (cap) atomicAdd((unsigned long long *)&cd.counters[2], 1ull);
}

// ===================================================================================== k_tail
// One block of 1024 threads.  mask bits: 1 theta, 2 Delta, 4 M, 8 mu, 16 Lambda, 32 pi, 64 carried sums (res, rr,
// sig_q for the next tau2), 128 ring wrap copy.  When a bit is clear the value already in `row` is kept.
// xg_src: 0 = cd.xg (from k_solve); 1 = sum of the PG partials (X*gamma by k_xpass bit2).
__global__ __launch_bounds__(1024) void k_tail(bnr_dev cd, int s, int mask, int xg_src)
{
    __shared__ double sred[32];
    __shared__ double sPsi[BNR_RMAX * BNR_RMAX], sA[BNR_RMAX * BNR_RMAX], sT[BNR_RMAX * BNR_RMAX], sBm[BNR_RMAX * BNR_RMAX];
    __shared__ double sll[3 * BNR_RMAX], slam[BNR_RMAX];
    __shared__ double sval[8];
    const bnr_plan_entry P = cd.plan[cd.pbase[0] + s];
    double *row = cd.trace + (size_t)P.row * cd.rowlen;
    const double *prev = cd.trace + (size_t)P.prev * cd.rowlen;
    const int R = cd.R, V = cd.V, q = cd.q, n = cd.n, tid = threadIdx.x;
    const double tau2 = row[ROW_TAU2];
    int cap = 0;
    if (tid < R) slam[tid] = row[cd.o_lam + tid];
    if (tid == 0) sval[1] = row[ROW_MU];

    // ---- reduce partial sums of k_backproj
    if (mask & (1 | 16)) {
        for (int j = tid; j < 1 + 3 * R; j += blockDim.x) {
            double sacc = 0.0;
            for (int b = 0; b < cd.nblk_bp; ++b) sacc += cd.Psum[(size_t)b * (1 + 3 * R) + j];
            if (j == 0) sval[0] = sacc; else sll[j - 1] = sacc;
        }
    }
    __syncthreads();
    // ---- theta (gibbs.jl:476-479)
    if ((mask & 1) && tid == 0) {
        double g = bnr_gamma(cd.seed, cd.zeta + (V * (V + 1)) / 2.0, P.it, SITE_THETA, 0, &cap);
        row[ROW_THETA] = g * (2.0 / (2.0 * cd.iota + sval[0]));
    }
    // ---- Delta (gibbs.jl:496-499, 130-140)
    if (mask & 2) {
        double sx = 0.0;
        for (int v = tid; v < V; v += blockDim.x) sx += row[cd.o_xi + v];
        sx = block_sum(sx, sred);
        if (tid == 0) {
            double a = cd.aDelta + sx, b = cd.bDelta + ((double)V - sx), out;
            if (a > 0.0 && b > 0.0) {
                double g1 = bnr_gamma(cd.seed, a, P.it, SITE_DELTA, 0, &cap), g2 = bnr_gamma(cd.seed, b, P.it, SITE_DELTA, 1, &cap);
                out = g1 / (g1 + g2);
            } else if (a > 0.0) out = 1.0;
            else if (b > 0.0) out = 0.0;
            else { double ua, ub; bnr_draw2(cd.seed, P.it, SITE_DELTA_COIN, 0, 0, ua, ub); out = (ua < 0.5) ? 0.0 : 1.0; }
            row[ROW_DELTA] = out;
        }
    }
    // ---- M (gibbs.jl:516-547): Psi = I + sum_v u_v u_v', df = nu + #{xi != 0}, M ~ InverseWishart(df, Psi)
    if (mask & 4) {
        const double *un = row + cd.o_u;
        for (int idx = tid; idx < R * R; idx += blockDim.x) {
            int a = idx % R, b = idx / R;
            double sacc = (a == b) ? 1.0 : 0.0;
            for (int v = 0; v < V; ++v) sacc += un[a + R * v] * un[b + R * v];
            sPsi[idx] = sacc;
        }
        double nz = 0.0;
        for (int v = tid; v < V; v += blockDim.x) nz += (!(fabs(row[cd.o_xi + v]) <= 0.1)) ? 1.0 : 0.0;
        nz = block_sum(nz, sred);
        const double df = cd.nu + nz;
        for (int idx = tid; idx < R * R; idx += blockDim.x) sA[idx] = sPsi[idx];     // sA <- C (chol of Psi)
        __syncthreads();
        int f = lds_chol(sA, R, tid);
        if (f) {                                                                      // retry ladder :529-543
            if (tid == 0) atomicAdd((unsigned long long *)&cd.counters[0], 1ull);
            __syncthreads();
            for (int i = tid; i < R; i += blockDim.x) sPsi[i + R * i] += 1e-5;
            __syncthreads();
            for (int idx = tid; idx < R * R; idx += blockDim.x) sA[idx] = sPsi[idx];
            __syncthreads();
            f = lds_chol(sA, R, tid);
            if (f && tid == 0) atomicAdd((unsigned long long *)&cd.counters[3], 1ull);
        }
        for (int idx = tid; idx < R * R; idx += blockDim.x) { int a = idx % R, b = idx / R; if (a < b) sA[idx] = 0.0; }
        // Bartlett factor A -> sBm (lower): A_jj = sqrt(2 Gamma((df-j)/2)), A_ij ~ N(0,1) i > j
        for (int idx = tid; idx < R * R; idx += blockDim.x) {
            int i = idx % R, j = idx / R;
            double v = 0.0;
            if (i == j) v = sqrt(2.0 * bnr_gamma(cd.seed, 0.5 * (df - j), P.it, SITE_M_CHI, (uint32_t)j, &cap));
            else if (i > j) v = bnr_normal(cd.seed, P.it, SITE_M_N, (uint32_t)(i * R + j), 0);
            sBm[idx] = v;
        }
        __syncthreads();
        // T = A^-1 (lower): column j by thread j
        if (tid < R) {
            int j = tid;
            for (int i = 0; i < R; ++i) {
                double v = 0.0;
                if (i == j) v = 1.0 / sBm[j + R * j];
                else if (i > j) {
                    double sacc = 0.0;
                    for (int k = j; k < i; ++k) sacc += sBm[i + R * k] * sT[k + R * j];
                    v = -sacc / sBm[i + R * i];
                }
                sT[i + R * j] = v;
            }
        }
        __syncthreads();
        // B = C T' -> sPsi ; M = B B'
        for (int idx = tid; idx < R * R; idx += blockDim.x) {
            int a = idx % R, b = idx / R;
            double sacc = 0.0;
            for (int k = 0; k < R; ++k) sacc += sA[a + R * k] * sT[b + R * k];
            sPsi[idx] = sacc;
        }
        __syncthreads();
        for (int idx = tid; idx < R * R; idx += blockDim.x) {
            int a = idx % R, b = idx / R;
            double sacc = 0.0;
            for (int k = 0; k < R; ++k) sacc += sPsi[a + R * k] * sPsi[b + R * k];
            row[cd.o_M + idx] = sacc;
        }
    }
    // ---- X gamma_new
    if ((mask & (8 | 64)) && xg_src == 1) {
        for (int i = tid; i < cd.n_pad; i += blockDim.x) {
            double sacc = 0.0;
            for (int b = 0; b < cd.nblk_x; ++b) sacc += cd.PG[(size_t)b * cd.n_pad + i];
            cd.xg[i] = sacc;
        }
        __syncthreads();
    }
    // ---- mu (gibbs.jl:565-570)
    if (mask & 8) {
        double sacc = 0.0;
        for (int i = tid; i < n; i += blockDim.x) sacc += cd.y[i] - cd.xg[i];
        sacc = block_sum(sacc, sred);
        if (tid == 0) { double m = sacc / n + sqrt(tau2 / n) * bnr_normal(cd.seed, P.it, SITE_MU, 0, 0); row[ROW_MU] = m; sval[1] = m; }
    }
    // ---- Lambda (gibbs.jl:586-613)
    if ((mask & 16) && tid < R) {
        int r = tid;
        double l0 = sll[3 * r], l1 = sll[3 * r + 1], l2 = sll[3 * r + 2];
        double pmax = fmax(l0, fmax(l1, l2));
        double w0 = prev[cd.o_pi + r] * exp(l0 - pmax), w1 = prev[cd.o_pi + r + R] * exp(l1 - pmax), w2 = prev[cd.o_pi + r + 2 * R] * exp(l2 - pmax);
        double ua, ub;
        bnr_draw2(cd.seed, P.it, SITE_LAMBDA, (uint32_t)r, 0, ua, ub);
        double lv = bnr_lambda_value(bnr_categorical3(w0, w1, w2, ua));
        row[cd.o_lam + r] = lv; slam[r] = lv;
    }
    __syncthreads();
    // ---- pi (gibbs.jl:630-636, 159-169)
    if ((mask & 32) && tid < 3 * R) {
        int r = tid / 3, c = tid % 3;
        double lam = slam[r];
        double base = pow((double)(r + 1), cd.eta);
        double alpha;
        if (lam == 1.0) alpha = (c == 0) ? base : (c == 1 ? 2.0 : 1.0);
        else if (lam == 0.0) alpha = (c == 0) ? base + 1.0 : 1.0;
        else alpha = (c == 0) ? base : (c == 1 ? 1.0 : 2.0);
        double g = bnr_gamma(cd.seed, alpha, P.it, SITE_PI, (uint32_t)(3 * r + c), &cap);
        sT[tid] = g;
    }
    __syncthreads();
    if ((mask & 32) && tid < 3 * R) {
        int r = tid / 3, c = tid % 3;
        double ssum = sT[3 * r] + sT[3 * r + 1] + sT[3 * r + 2];
        row[cd.o_pi + r + R * c] = sT[tid] / ssum;
    }
    __syncthreads();
    // ---- carried sums for the next update_tau2! (gibbs.jl:270-273): res = y - mu - X gamma, rr = res'res,
    //      sig_q = sum_e ((gamma_e - W(u,lam)_e)^2 / 2) / S_e  with the NEW lambda
    if (mask & 64) {
        const double mu = sval[1];
        double racc = 0.0;
        for (int i = tid; i < cd.n_pad; i += blockDim.x) {
            double rv = (i < n) ? (cd.y[i] - mu - cd.xg[i]) : 0.0;
            cd.res[i] = rv;
            racc += rv * rv;
        }
        racc = block_sum(racc, sred);
        double qacc = 0.0;
        const double *un = row + cd.o_u;
        for (int e = tid; e < q; e += blockDim.x) {
            double g = row[cd.o_gamma + e] - edge_W(un, slam, R, cd.el[e], cd.ek[e]);
            qacc += ((g * g) / 2.0) / row[cd.o_S + e];
        }
        qacc = block_sum(qacc, sred);
        if (tid == 0) { cd.scal[SC_RR] = racc; cd.scal[SC_SIGQ] = qacc; }
    }
    // ---- purge ring (gibbs.jl:857-860): copy_table!(state, 1, j)
    if ((mask & 128) && P.wrap) {
        __syncthreads();
        double *dst = cd.trace;
        for (int i = tid; i < cd.rowlen; i += blockDim.x) dst[i] = row[i];
    }
    if (cap && tid == 0) atomicAdd((unsigned long long *)&cd.counters[2], 1ull);
}

// advances the plan base after a batch of sweeps (last node of the captured graph)
__global__ void k_advance(int *pbase, int by) { if (threadIdx.x == 0 && blockIdx.x == 0) pbase[0] += by; }

// ===================================================================================== k_init_prior
// initialize_variables! (gibbs.jl:191-224) into row 0.  One block of 256 threads.
__global__ __launch_bounds__(256) void k_init_prior(bnr_dev cd)
{
    __shared__ double sT[3 * BNR_RMAX], sA[BNR_RMAX * BNR_RMAX], sTi[BNR_RMAX * BNR_RMAX], slam[BNR_RMAX];
    double *row = cd.trace;
    const int R = cd.R, V = cd.V, q = cd.q, tid = threadIdx.x;
    const uint32_t it = 1;
    int cap = 0;
    double eta = cd.eta;
    if (eta <= 1.0) eta = 1.01;                       // local-only reset (gibbs.jl:193-196)
    if (tid == 0) { row[ROW_THETA] = 0.5; row[ROW_DELTA] = 0.5; row[ROW_MU] = 1.0; row[ROW_TAU2] = 1.0; }
    for (int e = tid; e < q; e += blockDim.x) {
        double ua, ub;
        bnr_draw2(cd.seed, it, SITE_INIT_S, (uint32_t)e, 0, ua, ub);
        row[cd.o_S + e] = -(0.5 / 2.0) * log(ua);
    }
    if (tid < 3 * R) {
        int r = tid / 3, c = tid % 3;
        double alpha = (c == 0) ? pow((double)(r + 1), eta) : 1.0;
        sT[tid] = bnr_gamma(cd.seed, alpha, it, SITE_INIT_PI, (uint32_t)(3 * r + c), &cap);
    }
    __syncthreads();
    if (tid < 3 * R) {
        int r = tid / 3, c = tid % 3;
        row[cd.o_pi + r + R * c] = sT[tid] / (sT[3 * r] + sT[3 * r + 1] + sT[3 * r + 2]);
    }
    __syncthreads();
    if (tid < R) {
        double ua, ub;
        bnr_draw2(cd.seed, it, SITE_INIT_LAM, (uint32_t)tid, 0, ua, ub);
        double lv = bnr_lambda_value(bnr_categorical3(row[cd.o_pi + tid], row[cd.o_pi + tid + R], row[cd.o_pi + tid + 2 * R], ua));
        row[cd.o_lam + tid] = lv; slam[tid] = lv;
    }
    for (int v = tid; v < V; v += blockDim.x) {
        double ua, ub;
        bnr_draw2(cd.seed, it, SITE_INIT_XI, (uint32_t)v, 0, ua, ub);
        row[cd.o_xi + v] = (ua <= 0.5) ? 1.0 : 0.0;
    }
    // M ~ InverseWishart(nu, I): M = T' T... (C = I): B = T', M = B B' with T = A^-1
    for (int idx = tid; idx < R * R; idx += blockDim.x) {
        int i = idx % R, j = idx / R;
        double v = 0.0;
        if (i == j) v = sqrt(2.0 * bnr_gamma(cd.seed, 0.5 * (cd.nu - j), it, SITE_INIT_M_CHI, (uint32_t)j, &cap));
        else if (i > j) v = bnr_normal(cd.seed, it, SITE_INIT_M_N, (uint32_t)(i * R + j), 0);
        sA[idx] = v;
    }
    __syncthreads();
    if (tid < R) {
        int j = tid;
        for (int i = 0; i < R; ++i) {
            double v = 0.0;
            if (i == j) v = 1.0 / sA[j + R * j];
            else if (i > j) {
                double sacc = 0.0;
                for (int k = j; k < i; ++k) sacc += sA[i + R * k] * sTi[k + R * j];
                v = -sacc / sA[i + R * i];
            }
            sTi[i + R * j] = v;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < R * R; idx += blockDim.x) {
        int a = idx % R, b = idx / R;       // B[a,k] = T[k,a] ; M[a,b] = sum_k T[k,a] T[k,b]
        double sacc = 0.0;
        for (int k = 0; k < R; ++k) sacc += sTi[k + R * a] * sTi[k + R * b];
        row[cd.o_M + idx] = sacc;
    }
    for (int idx = tid; idx < R * V; idx += blockDim.x) {
        int r = idx % R, v = idx / R;
        row[cd.o_u + idx] = bnr_normal(cd.seed, it, SITE_INIT_U, (uint32_t)(v * R + r), 0);
    }
    __syncthreads();
    for (int e = tid; e < q; e += blockDim.x) {
        double W = edge_W(row + cd.o_u, slam, R, cd.el[e], cd.ek[e]);
        row[cd.o_gamma + e] = W + sqrt(1.0 * row[cd.o_S + e]) * bnr_normal(cd.seed, it, SITE_INIT_GAMMA, (uint32_t)e, 0);
    }
    if (cap) atomicAdd((unsigned long long *)&cd.counters[2], 1ull);
}

// ===================================================================================== table transposes
// out[r + nrows*d] = trace[(first + r)*rowlen + off + d]   (device row-major -> reference iteration-fastest)
__global__ void k_fetch_cols(const double *trace, int rowlen, int off, int ncols, int first, int nrows, double *out)
{
    __shared__ double tile[32][33];
    int d0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int r = r0 + j, d = d0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < nrows && d < ncols) ? trace[(size_t)(first + r) * rowlen + off + d] : 0.0;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int d = d0 + j, r = r0 + threadIdx.x;
        if (r < nrows && d < ncols) out[(size_t)r + (size_t)nrows * d] = tile[threadIdx.x][j];
    }
}
__global__ void k_load_cols(double *trace, int rowlen, int off, int ncols, int first, int nrows, const double *in)
{
    __shared__ double tile[32][33];
    int d0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int d = d0 + j, r = r0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < nrows && d < ncols) ? in[(size_t)r + (size_t)nrows * d] : 0.0;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int r = r0 + j, d = d0 + threadIdx.x;
        if (r < nrows && d < ncols) trace[(size_t)(first + r) * rowlen + off + d] = tile[threadIdx.x][j];
    }
}

// ===================================================================================== k_rhat_stats
// First half of rhat() (convergence.jl:4-65) for one chain: per parameter p in [gamma(q) | xi(V)], mean and corrected
// variance of the first and the last floor(nsamp/2) of rows first..first+nsamp-1.  out: [mean0 | var0 | mean1 | var1] x np
__global__ void k_rhat_stats(bnr_dev cd, int first, int nsamp, double *out)
{
    const int np = cd.q + cd.V;
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    const int off = (p < cd.q) ? cd.o_gamma + p : cd.o_xi + (p - cd.q);
    const int h = nsamp / 2;
    for (int half = 0; half < 2; ++half) {
        int r0 = first + (half == 0 ? 0 : nsamp - h);
        double sacc = 0.0;
        for (int r = 0; r < h; ++r) sacc += cd.trace[(size_t)(r0 + r) * cd.rowlen + off];
        double mean = sacc / h, v = 0.0;
        for (int r = 0; r < h; ++r) { double d = cd.trace[(size_t)(r0 + r) * cd.rowlen + off] - mean; v += d * d; }
        out[(size_t)(2 * half) * np + p] = mean;
        out[(size_t)(2 * half + 1) * np + p] = v / (h - 1);
    }
}

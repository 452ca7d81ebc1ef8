/*
 * oracle/bnr_oracle.c -- CPU restatement of BayesianNetworkRegression.jl's Gibbs hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (bayesiannetworkregression.jl_amd/, the
 * C-ABI library) includes, links or calls this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker.
 *
 * What it restates (reference @ /root/reference, v0.1.1), function by function:
 *   src/gibbs.jl:191-224  initialize_variables!      -> orc_init_prior
 *   src/gibbs.jl:267-277  update_tau2!               -> orc_update_tau2
 *   src/gibbs.jl:293-371  update_u_xi! (+385-402)    -> orc_update_u_xi
 *   src/gibbs.jl:420-438  update_gamma!              -> orc_update_gamma
 *   src/gibbs.jl:454-458  update_D! (+116-118)       -> orc_update_D
 *   src/gibbs.jl:476-479  update_theta!              -> orc_update_theta
 *   src/gibbs.jl:496-499  update_Delta! (+130-140)   -> orc_update_Delta
 *   src/gibbs.jl:516-547  update_M!                  -> orc_update_M
 *   src/gibbs.jl:565-570  update_mu!                 -> orc_update_mu
 *   src/gibbs.jl:586-613  update_Lambda!             -> orc_update_Lambda
 *   src/gibbs.jl:630-636  update_pi! (+159-169)      -> orc_update_pi
 *   src/gibbs.jl:663-677  gibbs_sample!              -> orc_gibbs_sample
 *   src/gibbs.jl:849-864  run! (purge ring)          -> orc_run
 *   src/gig.jl:8-176      sample_gig & helpers       -> orc_sample_gig
 *   src/utils.jl:17-57    lower_triangle index map   -> edge_index / orc_compute_W
 *   src/utils.jl:72-84    copy_table!                -> copy_row
 *   src/convergence.jl:4-65  rhat                    -> orc_rhat
 *
 * Parity pinning: the reference is Julia and cannot run here (no julia binary), and its
 * sample PATH depends on Random.Xoshiro + Distributions.jl internals that are not in
 * /root/reference.  This restatement therefore keeps every DISTRIBUTION and every
 * deterministic formula of the reference, but draws its variates from a counter-based
 * Philox-4x32-10 stream (the "draw-site contract") so that the HIP
 * kernels can reproduce the very same variates.  It is pinned to the reference by
 * tests/test_oracle_golden.py: (i) rhat reproduces the golden rhat vectors of
 * test/data/gen_test_results.jld2 to 1e-13; (ii) every conditional's parameters computed
 * by this file on consecutive golden rows make the golden draws pass PIT/KS calibration;
 * (iii) posterior summaries of an oracle chain on test/data/test1.csv agree with golden
 * res2 within MCSE tolerances.
 *
 * Table layout = the reference's: every column is Array{Float64,3}(tot_save,d1,d2),
 * column-major, ITERATION INDEX FASTEST (gibbs.jl:835-841).  Rows here are 0-based.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ draw sites */
enum {
    SITE_INIT_S = 1, SITE_INIT_PI = 2, SITE_INIT_LAM = 3, SITE_INIT_XI = 4,
    SITE_INIT_M_CHI = 5, SITE_INIT_M_N = 6, SITE_INIT_U = 7, SITE_INIT_GAMMA = 8,
    SITE_TAU2 = 16, SITE_XI = 17, SITE_U_Z = 18, SITE_G_Z1 = 19, SITE_G_Z2 = 20,
    SITE_D_GIG = 21, SITE_D_GAMMA = 22, SITE_THETA = 23, SITE_DELTA = 24, SITE_DELTA_COIN = 25,
    SITE_M_CHI = 26, SITE_M_N = 27, SITE_MU = 28, SITE_LAMBDA = 29, SITE_PI = 30
};
#define ORC_PI 3.14159265358979323846
#define ORC_MAX_ATTEMPTS 100000u
#define ATT_BOOST 0xFFFFFFFFu

typedef struct {
    int32_t n, V, R, q;
    double eta, zeta, iota, aDelta, bDelta, nu;
    uint64_t seed;        /* stream seed = user seed + chain index c (gibbs.jl:928) */
    const double *X;      /* n x q, column-major (Julia Matrix memory) */
    const double *y;
    int32_t pdf_mode;     /* 0 = reference: dense (V-1)-dim pdf, exp(), NaN->coin (gibbs.jl:349-360)
                             1 = log-space weights via Woodbury/determinant lemma (what the HIP path does) */
    int32_t cost_mode;    /* 1 = reference cost: full 2n^2q GEMM for X D X' (gibbs.jl:434)
                             0 = symmetric half only */
    int32_t tot;          /* tot_save rows of every column */
    int32_t status;       /* 0 ok; 3 = Cholesky failed after jitter ladder; 4 = sampler attempt cap */
    double *tau2, *u, *xi, *gamma, *S, *theta, *Delta, *M, *mu, *lam, *pi;
    int64_t iter;         /* global iteration counter of the LAST completed draw row (init = 1) */
    int64_t jitter_events, nan_w_events, gig_branch[5];
} orc_t;

#define IDX(i, d1, a, b) ((size_t)(i) + (size_t)o->tot * ((size_t)(a) + (size_t)(d1) * (size_t)(b)))
#define U_(i, r, v)   o->u[IDX(i, o->R, r, v)]
#define XI_(i, v)     o->xi[IDX(i, o->V, v, 0)]
#define G_(i, e)      o->gamma[IDX(i, o->q, e, 0)]
#define S_(i, e)      o->S[IDX(i, o->q, e, 0)]
#define M_(i, a, b)   o->M[IDX(i, o->R, a, b)]
#define LAM_(i, r)    o->lam[IDX(i, o->R, r, 0)]
#define PI_(i, r, c)  o->pi[IDX(i, o->R, r, c)]

/* ------------------------------------------------------------------ Philox-4x32-10 */
static inline void philox4x32_10(const uint32_t c_in[4], const uint32_t k_in[2], uint32_t out[4])
{
    uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3];
    uint32_t k0 = k_in[0], k1 = k_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void orc_philox(const uint32_t c[4], const uint32_t k[2], uint32_t out[4]) { philox4x32_10(c, k, out); }

/* one draw = two uniforms in the open interval (0,1), 53 bits each */
static inline void draw2(uint64_t seed, uint32_t it, uint32_t site, uint32_t elem, uint32_t att,
                         double *ua, double *ub)
{
    uint32_t c[4] = { it, site, elem, att }, k[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) }, r[4];
    philox4x32_10(c, k, r);
    uint64_t a = (uint64_t)r[0] | ((uint64_t)r[1] << 32);
    uint64_t b = (uint64_t)r[2] | ((uint64_t)r[3] << 32);
    *ua = ((double)(a >> 11) + 0.5) * 0x1.0p-53;
    *ub = ((double)(b >> 11) + 0.5) * 0x1.0p-53;
}
void orc_uniform2(uint64_t seed, uint32_t it, uint32_t site, uint32_t elem, uint32_t att, double *out)
{ draw2(seed, it, site, elem, att, &out[0], &out[1]); }

/* cos(2*pi*u) with an exact quadrant reduction so host and device evaluate sin/cos on the
 * same small argument */
static inline double cos2pi(double u)
{
    double x = 4.0 * u;                 /* in (0,4): quadrant units */
    double qd = floor(x + 0.5);         /* nearest quadrant boundary 0..4 */
    double t = (x - qd) * 1.57079632679489661923;         /* (x-q)*pi/2 in [-pi/4, pi/4] */
    int qi = ((int)qd) & 3;
    switch (qi) {
        case 0: return cos(t);
        case 1: return -sin(t);
        case 2: return -cos(t);
        default: return sin(t);
    }
}
static inline double normal_from(double ua, double ub) { return sqrt(-2.0 * log(ua)) * cos2pi(ub); }
static inline double draw_normal(uint64_t seed, uint32_t it, uint32_t site, uint32_t elem, uint32_t att)
{ double a, b; draw2(seed, it, site, elem, att, &a, &b); return normal_from(a, b); }
double orc_normal(uint64_t seed, uint32_t it, uint32_t site, uint32_t elem, uint32_t att)
{ return draw_normal(seed, it, site, elem, att); }

/* Gamma(shape a, scale 1): Marsaglia-Tsang (a>=1) with the U^(1/a) boost for a<1.
 * Distribution-identical to Distributions.jl's GammaMTSampler / GammaIPSampler. */
static double draw_gamma(orc_t *o, double a, uint32_t it, uint32_t site, uint32_t elem)
{
    double boost = 1.0;
    if (a < 1.0) {
        double ua, ub; draw2(o->seed, it, site, elem, ATT_BOOST, &ua, &ub);
        boost = pow(ua, 1.0 / a);
        a += 1.0;
    }
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t t = 0; t < ORC_MAX_ATTEMPTS; ++t) {
        double x = draw_normal(o->seed, it, site, elem, 2 * t);
        double ua, ub; draw2(o->seed, it, site, elem, 2 * t + 1, &ua, &ub);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (log(ua) < 0.5 * x * x + d - d * v + d * log(v)) return d * v * boost;
    }
    o->status = 4;
    return d * boost;
}
double orc_gamma(orc_t *o, double a, uint32_t it, uint32_t site, uint32_t elem) { return draw_gamma(o, a, it, site, elem); }

/* KahanSummation.sum_kbn (Kahan-Babuska / Neumaier), as used at gibbs.jl:273,477,497,603-605 */
typedef struct { double s, c; } kbn_t;
static inline void kbn_add(kbn_t *k, double x)
{
    double t = k->s + x;
    if (fabs(k->s) >= fabs(x)) k->c += (k->s - t) + x; else k->c += (x - t) + k->s;
    k->s = t;
}
static inline double kbn_val(const kbn_t *k) { return k->s + k->c; }

/* ------------------------------------------------------------------ index map (utils.jl:17-57) */
/* edge e <-> (l,k), l>=k, column-wise lower triangle incl. diagonal; all 0-based here */
static inline int edge_index(int V, int l, int k) { return k * V - (k * (k - 1)) / 2 + (l - k); }
int orc_edge_index(int V, int l, int k) { return l >= k ? edge_index(V, l, k) : edge_index(V, k, l); }

/* W = lower_triangle(u' * Diagonal(lam) * u)  (gibbs.jl:219-221, 271, 421, 455) */
void orc_compute_W(orc_t *o, int row_u, int row_lam, double *W)
{
    int V = o->V, R = o->R, e = 0;
    for (int k = 0; k < V; ++k)
        for (int l = k; l < V; ++l, ++e) {
            double s = 0.0;
            for (int r = 0; r < R; ++r) s += U_(row_u, r, l) * LAM_(row_lam, r) * U_(row_u, r, k);
            W[e] = s;
        }
}

/* ------------------------------------------------------------------ small dense helpers */
/* in-place lower Cholesky of an m x m column-major matrix; returns 0 ok, 1 not PD */
static int chol_lower(double *A, int m)
{
    for (int j = 0; j < m; ++j) {
        double d = A[j + (size_t)m * j];
        for (int k = 0; k < j; ++k) d -= A[j + (size_t)m * k] * A[j + (size_t)m * k];
        if (!(d > 0.0) || !isfinite(d)) return 1;
        d = sqrt(d);
        A[j + (size_t)m * j] = d;
        for (int i = j + 1; i < m; ++i) {
            double s = A[i + (size_t)m * j];
            for (int k = 0; k < j; ++k) s -= A[i + (size_t)m * k] * A[j + (size_t)m * k];
            A[i + (size_t)m * j] = s / d;
        }
    }
    for (int j = 1; j < m; ++j) for (int i = 0; i < j; ++i) A[i + (size_t)m * j] = 0.0;
    return 0;
}
static void fwd_solve(const double *L, int m, double *b)      /* L x = b */
{
    for (int i = 0; i < m; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i + (size_t)m * k] * b[k];
        b[i] = s / L[i + (size_t)m * i];
    }
}
static void bwd_solve_T(const double *L, int m, double *b)    /* L' x = b */
{
    for (int i = m - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < m; ++k) s -= L[k + (size_t)m * i] * b[k];
        b[i] = s / L[i + (size_t)m * i];
    }
}
/* inverse of an SPD R x R matrix via Cholesky (inv(M), gibbs.jl:315) */
static int spd_inverse(const double *A, int m, double *Ainv, double *logdet)
{
    double *L = (double *)malloc(sizeof(double) * m * m);
    memcpy(L, A, sizeof(double) * m * m);
    if (chol_lower(L, m)) { free(L); return 1; }
    double ld = 0.0;
    for (int i = 0; i < m; ++i) ld += 2.0 * log(L[i + (size_t)m * i]);
    if (logdet) *logdet = ld;
    for (int j = 0; j < m; ++j) {
        double *col = Ainv + (size_t)m * j;
        for (int i = 0; i < m; ++i) col[i] = (i == j);
        fwd_solve(L, m, col);
        bwd_solve_T(L, m, col);
    }
    free(L);
    return 0;
}
/* LU with partial pivoting, solve A x = b in place (Julia's generic `\`, gibbs.jl:434) */
static int lu_solve(double *A, int m, double *b)
{
    for (int k = 0; k < m; ++k) {
        int p = k; double best = fabs(A[k + (size_t)m * k]);
        for (int i = k + 1; i < m; ++i) { double v = fabs(A[i + (size_t)m * k]); if (v > best) { best = v; p = i; } }
        if (best == 0.0) return 1;
        if (p != k) {
            for (int j = 0; j < m; ++j) { double t = A[k + (size_t)m * j]; A[k + (size_t)m * j] = A[p + (size_t)m * j]; A[p + (size_t)m * j] = t; }
            double t = b[k]; b[k] = b[p]; b[p] = t;
        }
        double piv = 1.0 / A[k + (size_t)m * k];
        for (int i = k + 1; i < m; ++i) A[i + (size_t)m * k] *= piv;
        for (int j = k + 1; j < m; ++j) {
            double akj = A[k + (size_t)m * j];
            if (akj == 0.0) continue;
            double *cj = A + (size_t)m * j; const double *ck = A + (size_t)m * k;
            for (int i = k + 1; i < m; ++i) cj[i] -= ck[i] * akj;
        }
    }
    for (int i = 0; i < m; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[i + (size_t)m * k] * b[k]; b[i] = s; }
    for (int i = m - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < m; ++k) s -= A[i + (size_t)m * k] * b[k]; b[i] = s / A[i + (size_t)m * i]; }
    return 0;
}

/* ------------------------------------------------------------------ GIG sampler (gig.jl) */
static double gig_mode(double lambda, double omega)            /* gig.jl:170-176 */
{
    if (lambda >= 1.0) return (sqrt((lambda - 1.0) * (lambda - 1.0) + omega * omega) + lambda - 1.0) / omega;
    return omega / (sqrt((1.0 - lambda) * (1.0 - lambda) + omega * omega) + (1.0 - lambda));
}
static double gig_rou_shift(orc_t *o, double lambda, double lambda_old, double omega, double alpha,
                            uint32_t it, uint32_t elem)        /* gig.jl:44-78 */
{
    double t = 0.5 * (lambda - 1.0), s = 0.25 * omega;
    double xm = gig_mode(lambda, omega);
    double nc = t * log(xm) - s * (xm + 1.0 / xm);
    double a = -(2.0 * (lambda + 1.0) / omega + xm);
    double b = (2.0 * (lambda - 1.0) * xm / omega - 1.0);
    double c = xm;
    double p = b - a * a / 3.0;
    double q = 2.0 * a * a * a / 27.0 - a * b / 3.0 + c;
    double fi = acos(-q / (2.0 * sqrt(-p * p * p / 27.0)));
    double fak = 2.0 * sqrt(-p / 3.0);
    double y1 = fak * cos(fi / 3.0) - a / 3.0;
    double y2 = fak * cos(fi / 3.0 + 4.0 / 3.0 * ORC_PI) - a / 3.0;
    double uplus = (y1 - xm) * exp(t * log(y1) - s * (y1 + 1.0 / y1) - nc);
    double uminus = (y2 - xm) * exp(t * log(y2) - s * (y2 + 1.0 / y2) - nc);
    for (uint32_t k = 0; k < ORC_MAX_ATTEMPTS; ++k) {
        double ru, rv; draw2(o->seed, it, SITE_D_GIG, elem, k, &ru, &rv);
        double U = uminus + ru * (uplus - uminus);
        double Vv = rv;
        double X = U / Vv + xm;
        if (X > 0.0 && log(Vv) <= t * log(X) - s * (X + 1.0 / X) - nc)
            return lambda_old < 0.0 ? alpha / X : alpha * X;
    }
    o->status = 4;
    return alpha * xm;
}
static double gig_rou_noshift(orc_t *o, double lambda, double lambda_old, double omega, double alpha,
                              uint32_t it, uint32_t elem)      /* gig.jl:80-100 */
{
    double t = 0.5 * (lambda - 1.0), s = 0.25 * omega;
    double xm = gig_mode(lambda, omega);
    double nc = t * log(xm) - s * (xm + 1.0 / xm);
    double ym = ((lambda + 1.0) + sqrt((lambda + 1.0) * (lambda + 1.0) + omega * omega)) / omega;
    double um = exp(0.5 * (lambda + 1.0) * log(ym) - s * (ym + 1.0 / ym) - nc);
    for (uint32_t k = 0; k < ORC_MAX_ATTEMPTS; ++k) {
        double ru, rv; draw2(o->seed, it, SITE_D_GIG, elem, k, &ru, &rv);
        double U = um * ru;
        double Vv = rv;
        double X = U / Vv;
        if (log(Vv) <= (t * log(X) - s * (X + 1.0 / X) - nc))
            return lambda_old < 0.0 ? alpha / X : alpha * X;
    }
    o->status = 4;
    return alpha * xm;
}
static double gig_concave(orc_t *o, double lambda, double lambda_old, double omega, double alpha,
                          uint32_t it, uint32_t elem)          /* gig.jl:102-168 */
{
    double xm = gig_mode(lambda, omega);
    double x0 = omega / (1.0 - lambda);
    double k0 = exp((lambda - 1.0) * log(xm) - 0.5 * omega * (xm + 1.0 / xm));
    double A[3], k1, k2;
    A[0] = k0 * x0;
    if (x0 >= 2.0 / omega) {
        k1 = 0.0; A[1] = 0.0;
        k2 = pow(x0, lambda - 1.0);
        A[2] = k2 * 2.0 * exp(-omega * x0 / 2.0) / omega;
    } else {
        k1 = exp(-omega);
        if (lambda == 0.0) A[1] = k1 * log(2.0 / (omega * omega));
        else A[1] = k1 / lambda * (pow(2.0 / omega, lambda) - pow(x0, lambda));
        k2 = pow(2.0 / omega, lambda - 1.0);
        A[2] = k2 * 2.0 * exp(-1.0) / omega;
    }
    double Atot = A[0] + A[1] + A[2];
    for (uint32_t k = 0; k < ORC_MAX_ATTEMPTS; ++k) {
        double ru, rv; draw2(o->seed, it, SITE_D_GIG, elem, k, &ru, &rv);
        double Vv = Atot * ru, hx, X;
        if (Vv <= A[0]) { X = x0 * Vv / A[0]; hx = k0; }
        else {
            Vv -= A[0];
            if (Vv <= A[1]) {
                if (lambda == 0.0) { X = omega * exp(exp(omega) * Vv); hx = k1 / X; }
                else { X = pow(pow(x0, lambda) + (lambda / k1 * Vv), 1.0 / lambda); hx = k1 * pow(X, lambda - 1.0); }
            } else {
                Vv -= A[1];
                double a = (x0 > 2.0 / omega) ? x0 : 2.0 / omega;
                X = -2.0 / omega * log(exp(-omega / 2.0 * a) - omega / (2.0 * k2) * Vv);
                hx = k2 * exp(-omega / 2.0 * X);
            }
        }
        double U = rv * hx;
        if (log(U) <= (lambda - 1.0) * log(X) - omega / 2.0 * (X + 1.0 / X))
            return lambda_old < 0.0 ? alpha / X : alpha * X;
    }
    o->status = 4;
    return alpha * xm;
}
/* sample_gig(rng, lambda, chi, psi)  gig.jl:8-42.  Quirks kept: the invalid-parameter
 * ArgumentError is constructed but not thrown (9-13); Gamma SCALE psi/2 in the chi~0
 * branch (17), chi/2 in the psi~0 branch (23). */
double orc_sample_gig(orc_t *o, double lambda, double chi, double psi, uint32_t it, uint32_t elem)
{
    const double eps10 = 2.220446049250313e-16 * 10.0;
    if (chi < eps10) {
        o->gig_branch[3]++;
        if (lambda > 0.0) return draw_gamma(o, lambda, it, SITE_D_GAMMA, elem) * (psi / 2.0);
        return 1.0 / (draw_gamma(o, -lambda, it, SITE_D_GAMMA, elem) * (psi / 2.0));
    } else if (psi < eps10) {
        o->gig_branch[4]++;
        if (lambda > 0.0) return 1.0 / (draw_gamma(o, lambda, it, SITE_D_GAMMA, elem) * (chi / 2.0));
        return draw_gamma(o, -lambda, it, SITE_D_GAMMA, elem) * (chi / 2.0);
    }
    double lambda_old = lambda;
    if (lambda < 0.0) lambda = -lambda;
    double alpha = sqrt(chi / psi), omega = sqrt(psi * chi);
    if (lambda > 2.0 || omega > 3.0) { o->gig_branch[0]++; return gig_rou_shift(o, lambda, lambda_old, omega, alpha, it, elem); }
    if (lambda >= 1.0 - 2.25 * (omega * omega) || omega > 0.2) { o->gig_branch[1]++; return gig_rou_noshift(o, lambda, lambda_old, omega, alpha, it, elem); }
    if (lambda >= 0.0 && omega > 0.0) { o->gig_branch[2]++; return gig_concave(o, lambda, lambda_old, omega, alpha, it, elem); }
    return NAN; /* the reference returns `nothing` here (gig.jl:41) */
}

/* ------------------------------------------------------------------ samplers built on Gamma */
static void draw_dirichlet3(orc_t *o, const double alpha[3], uint32_t it, uint32_t site, uint32_t r, double out[3])
{
    double g[3], s = 0.0;
    for (int c = 0; c < 3; ++c) { g[c] = draw_gamma(o, alpha[c], it, site, (uint32_t)(3 * r + c)); s += g[c]; }
    for (int c = 0; c < 3; ++c) out[c] = g[c] / s;
}
/* StatsBase.sample(rng, vals, weights): linear scan of the cumulative weights */
static int draw_categorical3(const double w[3], double u01)
{
    double t = u01 * (w[0] + w[1] + w[2]);
    int i = 0; double cw = w[0];
    while (cw < t && i < 2) { ++i; cw += w[i]; }
    return i;
}
static const double LAMBDA_VALUES[3] = { 0.0, 1.0, -1.0 };    /* order [0,1,-1]: gibbs.jl:207,610 */

/* InverseWishart(df, Psi = C C') draw: M = (C A^-T)(C A^-T)', A lower Bartlett factor
 * (A_jj = sqrt(chi2(df-j)), A_ij ~ N(0,1), i>j).  Distribution = Distributions.jl's
 * rand(InverseWishart(df, Psi)) = inv(rand(Wishart(df, inv(Psi)))) (gibbs.jl:212,545). */
static void draw_inverse_wishart(orc_t *o, double df, const double *C, uint32_t it,
                                 uint32_t site_chi, uint32_t site_n, double *Mout)
{
    int R = o->R;
    double *A = (double *)calloc((size_t)R * R, sizeof(double));
    double *T = (double *)calloc((size_t)R * R, sizeof(double));
    double *B = (double *)calloc((size_t)R * R, sizeof(double));
    for (int j = 0; j < R; ++j) {
        A[j + R * j] = sqrt(2.0 * draw_gamma(o, 0.5 * (df - j), it, site_chi, (uint32_t)j));
        for (int i = j + 1; i < R; ++i) A[i + R * j] = draw_normal(o->seed, it, site_n, (uint32_t)(i * R + j), 0);
    }
    /* T = A^-1 (lower) */
    for (int j = 0; j < R; ++j) {
        T[j + R * j] = 1.0 / A[j + R * j];
        for (int i = j + 1; i < R; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s += A[i + R * k] * T[k + R * j];
            T[i + R * j] = -s / A[i + R * i];
        }
    }
    /* B = C * T'  ;  B[a,b] = sum_k C[a,k] T[b,k] */
    for (int a = 0; a < R; ++a) for (int b = 0; b < R; ++b) {
        double s = 0.0;
        for (int k = 0; k < R; ++k) s += C[a + R * k] * T[b + R * k];
        B[a + R * b] = s;
    }
    for (int a = 0; a < R; ++a) for (int b = 0; b < R; ++b) {
        double s = 0.0;
        for (int k = 0; k < R; ++k) s += B[a + R * k] * B[b + R * k];
        Mout[a + R * b] = s;
    }
    free(A); free(T); free(B);
}

/* ------------------------------------------------------------------ initialize_variables! (gibbs.jl:191-224) */
void orc_init_prior(orc_t *o)
{
    const int V = o->V, R = o->R, q = o->q;
    const uint32_t it = 1;
    double eta = o->eta;
    if (eta <= 1.0) eta = 1.01;                        /* local-only reset, gibbs.jl:193-196 */
    o->theta[IDX(0, 1, 0, 0)] = 0.5;
    for (int e = 0; e < q; ++e) {                      /* S ~ Exponential(scale theta/2) :201 */
        double ua, ub; draw2(o->seed, it, SITE_INIT_S, (uint32_t)e, 0, &ua, &ub);
        S_(0, e) = -(0.5 / 2.0) * log(ua);
    }
    for (int r = 0; r < R; ++r) {                      /* pi_r ~ Dirichlet([r^eta,1,1]) :205 */
        double alpha[3] = { pow((double)(r + 1), eta), 1.0, 1.0 }, p[3];
        draw_dirichlet3(o, alpha, it, SITE_INIT_PI, (uint32_t)r, p);
        for (int c = 0; c < 3; ++c) PI_(0, r, c) = p[c];
    }
    for (int r = 0; r < R; ++r) {                      /* lambda_r ~ Categorical([0,1,-1]; pi_r) :207 */
        double w[3] = { PI_(0, r, 0), PI_(0, r, 1), PI_(0, r, 2) }, ua, ub;
        draw2(o->seed, it, SITE_INIT_LAM, (uint32_t)r, 0, &ua, &ub);
        LAM_(0, r) = LAMBDA_VALUES[draw_categorical3(w, ua)];
    }
    o->Delta[IDX(0, 1, 0, 0)] = 0.5;
    for (int v = 0; v < V; ++v) {                      /* xi ~ Binomial(1, Delta) :211 */
        double ua, ub; draw2(o->seed, it, SITE_INIT_XI, (uint32_t)v, 0, &ua, &ub);
        XI_(0, v) = (ua <= 0.5) ? 1.0 : 0.0;
    }
    {                                                  /* M ~ InverseWishart(nu, I) :212 */
        double *C = (double *)calloc((size_t)R * R, sizeof(double));
        double *Mo = (double *)calloc((size_t)R * R, sizeof(double));
        for (int a = 0; a < R; ++a) C[a + R * a] = 1.0;
        draw_inverse_wishart(o, o->nu, C, it, SITE_INIT_M_CHI, SITE_INIT_M_N, Mo);
        for (int a = 0; a < R; ++a) for (int b = 0; b < R; ++b) M_(0, a, b) = Mo[a + R * b];
        free(C); free(Mo);
    }
    for (int v = 0; v < V; ++v)                        /* u_v ~ N(0, I_R) :213-215 */
        for (int r = 0; r < R; ++r) U_(0, r, v) = draw_normal(o->seed, it, SITE_INIT_U, (uint32_t)(v * R + r), 0);
    o->mu[IDX(0, 1, 0, 0)] = 1.0;
    o->tau2[IDX(0, 1, 0, 0)] = 1.0;
    double *W = (double *)malloc(sizeof(double) * q);
    orc_compute_W(o, 0, 0, W);
    for (int e = 0; e < q; ++e)                        /* gamma ~ N(W, tau2*diag(S)) :223 */
        G_(0, e) = W[e] + sqrt(1.0 * S_(0, e)) * draw_normal(o->seed, it, SITE_INIT_GAMMA, (uint32_t)e, 0);
    free(W);
    o->iter = 1;
}

/* ------------------------------------------------------------------ update_tau2! (gibbs.jl:267-277) */
/* params[0]=shape, params[1]=scale (sigma_t^2) */
void orc_tau2_params(orc_t *o, int j, double *params)
{
    const int n = o->n, q = o->q, V = o->V;
    double *W = (double *)malloc(sizeof(double) * q);
    double *res = (double *)malloc(sizeof(double) * n);
    double mu = o->mu[IDX(j - 1, 1, 0, 0)];
    for (int i = 0; i < n; ++i) res[i] = 0.0;
    for (int e = 0; e < q; ++e) {
        double g = G_(j - 1, e); const double *xc = o->X + (size_t)n * e;
        if (g != 0.0) for (int i = 0; i < n; ++i) res[i] += xc[i] * g;
    }
    double rr = 0.0;
    for (int i = 0; i < n; ++i) { double r = o->y[i] - mu - res[i]; rr += r * r; }
    orc_compute_W(o, j - 1, j - 1, W);
    kbn_t k = { 0.0, 0.0 };
    for (int e = 0; e < q; ++e) { double g = G_(j - 1, e) - W[e]; kbn_add(&k, ((g * g) / 2.0) / S_(j - 1, e)); }
    params[0] = (n / 2.0) + (V * (V + 1) / 4.0);
    params[1] = rr / 2.0 + kbn_val(&k);
    free(W); free(res);
}
void orc_update_tau2(orc_t *o, int j, uint32_t it)
{
    double p[2]; orc_tau2_params(o, j, p);
    /* InverseGamma(shape, scale) = scale / Gamma(shape, 1) */
    o->tau2[IDX(j, 1, 0, 0)] = p[1] / draw_gamma(o, p[0], it, SITE_TAU2, 0);
}

/* ------------------------------------------------------------------ update_u_xi! (gibbs.jl:293-371) */
/* Deterministic part for node k using row j-1 and tau2[j]:
 *   w      = w_top/(w_top+w_bot)  (probability that xi_k = 0)
 *   mu_t   = Sigma * U' H^-1 gamma_k / tau2            (R)
 *   Lc     = lower Cholesky factor of Sigma^-1 (C.U = Lc')   (R x R col-major)
 * returns 0 ok / 3 Cholesky failure.  logit_out (optional) = log w_bot - log w_top. */
int orc_node_params(orc_t *o, int j, int k, double *w_out, double *mu_t, double *Lc, double *logit_out)
{
    const int V = o->V, R = o->R, m = V - 1;
    const double tau2 = o->tau2[IDX(j, 1, 0, 0)], Delta = o->Delta[IDX(j - 1, 1, 0, 0)];
    double *Um = (double *)malloc(sizeof(double) * m * R);     /* U: m x R, U[a,r] = u[r,node_a]*lam[r]  :296 */
    double *gk = (double *)malloc(sizeof(double) * m);
    double *h = (double *)malloc(sizeof(double) * m);
    double *Mprev = (double *)malloc(sizeof(double) * R * R);
    double *Minv = (double *)malloc(sizeof(double) * R * R);
    double *Sinv = (double *)malloc(sizeof(double) * R * R);
    double *b = (double *)malloc(sizeof(double) * R);
    int rc = 0, a = 0;
    for (int l = 0; l < V; ++l) {
        if (l == k) continue;                                   /* self-loop (k,k) excluded :300-309 */
        int e = (l > k) ? edge_index(V, l, k) : edge_index(V, k, l);
        gk[a] = G_(j - 1, e); h[a] = S_(j - 1, e);
        for (int r = 0; r < R; ++r) Um[a + (size_t)m * r] = U_(j - 1, r, l) * LAM_(j - 1, r);
        ++a;
    }
    for (int x = 0; x < R; ++x) for (int y2 = 0; y2 < R; ++y2) Mprev[x + R * y2] = M_(j - 1, x, y2);
    double logdetM = 0.0;
    if (spd_inverse(Mprev, R, Minv, &logdetM)) { rc = 3; goto done; }
    /* Sigma^-1 = U' H^-1 U / tau2 + inv(M)   :315 */
    for (int x = 0; x < R; ++x) for (int y2 = 0; y2 < R; ++y2) {
        double s = 0.0;
        for (int t = 0; t < m; ++t) s += Um[t + (size_t)m * x] * (Um[t + (size_t)m * y2] / h[t]);
        Sinv[x + R * y2] = s / tau2 + Minv[x + R * y2];
    }
    /* Cholesky with the reference's jitter ladder :322-347 */
    memcpy(Lc, Sinv, sizeof(double) * R * R);
    if (chol_lower(Lc, R)) {
        o->jitter_events++;
        for (int x = 0; x < R; ++x) Sinv[x + R * x] += 1e-5;
        memcpy(Lc, Sinv, sizeof(double) * R * R);
        if (chol_lower(Lc, R)) {
            for (int x = 0; x < R; ++x) Sinv[x + R * x] += 4e-5;
            memcpy(Lc, Sinv, sizeof(double) * R * R);
            if (chol_lower(Lc, R)) { rc = 3; goto done; }
        }
    }
    /* b = U' H^-1 gamma_k / tau2 ; mu_t = Sigma b   :364 */
    for (int x = 0; x < R; ++x) {
        double s = 0.0;
        for (int t = 0; t < m; ++t) s += Um[t + (size_t)m * x] * (gk[t] / h[t]);
        b[x] = s / tau2;
    }
    memcpy(mu_t, b, sizeof(double) * R);
    fwd_solve(Lc, R, mu_t); bwd_solve_T(Lc, R, mu_t);

    double w, logit;
    if (o->pdf_mode == 0) {
        /* reference: two dense (V-1)-dim MvNormal pdfs :349-351 */
        double lt = 0.0;                 /* logpdf N(gk; 0, tau2 H) */
        for (int t = 0; t < m; ++t) lt += -0.5 * (log(2.0 * ORC_PI) + log(tau2 * h[t]) + gk[t] * gk[t] / (tau2 * h[t]));
        double *Cov = (double *)malloc(sizeof(double) * m * m);
        double *UM = (double *)malloc(sizeof(double) * m * R);
        for (int t = 0; t < m; ++t) for (int x = 0; x < R; ++x) {
            double s = 0.0;
            for (int y2 = 0; y2 < R; ++y2) s += Um[t + (size_t)m * y2] * Mprev[y2 + R * x];
            UM[t + (size_t)m * x] = s;
        }
        for (int c = 0; c < m; ++c) for (int t = 0; t < m; ++t) {
            double s = 0.0;
            for (int x = 0; x < R; ++x) s += UM[t + (size_t)m * x] * Um[c + (size_t)m * x];
            Cov[t + (size_t)m * c] = s + (t == c ? tau2 * h[t] : 0.0);
        }
        double lb;
        if (chol_lower(Cov, m)) { lb = NAN; }
        else {
            double ld = 0.0; for (int t = 0; t < m; ++t) ld += 2.0 * log(Cov[t + (size_t)m * t]);
            double *z = (double *)malloc(sizeof(double) * m);
            memcpy(z, gk, sizeof(double) * m); fwd_solve(Cov, m, z);
            double qf = 0.0; for (int t = 0; t < m; ++t) qf += z[t] * z[t];
            lb = -0.5 * (m * log(2.0 * ORC_PI) + ld + qf);
            free(z);
        }
        free(Cov); free(UM);
        double w_top = (1.0 - Delta) * exp(lt), w_bot = Delta * exp(lb);
        w = w_top / (w_bot + w_top);
        logit = (log(Delta) + lb) - (log1p(-Delta) + lt);
    } else {
        /* log space: log w_bot - log w_top = log(D/(1-D)) - 1/2[logdet M + logdet Sigma^-1] + 1/2 b' Sigma b */
        double ldS = 0.0; for (int x = 0; x < R; ++x) ldS += 2.0 * log(Lc[x + R * x]);
        double qf = 0.0; for (int x = 0; x < R; ++x) qf += b[x] * mu_t[x];
        logit = log(Delta) - log1p(-Delta) - 0.5 * (logdetM + ldS) + 0.5 * qf;
        w = 1.0 / (1.0 + exp(logit));
    }
    *w_out = w;
    if (logit_out) *logit_out = logit;
done:
    free(Um); free(gk); free(h); free(Mprev); free(Minv); free(Sinv); free(b);
    return rc;
}
void orc_update_u_xi(orc_t *o, int j, uint32_t it)
{
    const int V = o->V, R = o->R;
    double *mu_t = (double *)malloc(sizeof(double) * R);
    double *z = (double *)malloc(sizeof(double) * R);
    double *Lc = (double *)malloc(sizeof(double) * R * R);
    for (int k = 0; k < V; ++k) {
        double w;
        if (orc_node_params(o, j, k, &w, mu_t, Lc, NULL)) { o->status = 3; break; }
        /* update_xi :385-402 */
        double xi;
        if (w <= 0.0) xi = 1.0;
        else if (w >= 1.0) xi = 0.0;
        else {
            double ua, ub; draw2(o->seed, it, SITE_XI, (uint32_t)k, 0, &ua, &ub);
            if (isnan(w)) { o->nan_w_events++; xi = (ua <= 0.5) ? 1.0 : 0.0; }
            else xi = (ua <= 1.0 - w) ? 1.0 : 0.0;
        }
        XI_(j, k) = xi;
        /* u_tmp = mu_t + inv(C.U) z :365 ; inv(C.U) z = Lc'^-1 z */
        for (int r = 0; r < R; ++r) z[r] = draw_normal(o->seed, it, SITE_U_Z, (uint32_t)(k * R + r), 0);
        bwd_solve_T(Lc, R, z);
        for (int r = 0; r < R; ++r) U_(j, r, k) = xi * (mu_t[r] + z[r]);
    }
    free(mu_t); free(z); free(Lc);
}

/* Optional BLAS/LAPACK back end for the reference-cost timing mode (cost_mode = 2): Fortran-interface dgemm / dgesv with 64-bit
 * integers (numpy's bundled OpenBLAS: scipy_dgemm_64_, scipy_dgesv_64_), handed in as plain function addresses by the Python
 * side -- the reference's `Xt*tau2D*transpose(Xt)` is an OpenBLAS gemm and its `\` an LAPACK getrf/getrs (gibbs.jl:434). */
typedef void (*orc_dgemm_t)(const char *, const char *, const int64_t *, const int64_t *, const int64_t *, const double *, const double *,
                            const int64_t *, const double *, const int64_t *, const double *, double *, const int64_t *);
typedef void (*orc_dgesv_t)(const int64_t *, const int64_t *, double *, const int64_t *, int64_t *, double *, const int64_t *, int64_t *);
static orc_dgemm_t g_dgemm = 0;
static orc_dgesv_t g_dgesv = 0;
void orc_set_blas(void *dgemm, void *dgesv) { g_dgemm = (orc_dgemm_t)dgemm; g_dgesv = (orc_dgesv_t)dgesv; }
int orc_have_blas(void) { return g_dgemm && g_dgesv; }

/* ------------------------------------------------------------------ update_gamma! (gibbs.jl:420-438) */
void orc_update_gamma(orc_t *o, int j, uint32_t it)
{
    const int n = o->n, q = o->q;
    const double tau2 = o->tau2[IDX(j, 1, 0, 0)], tau = sqrt(tau2), mu = o->mu[IDX(j - 1, 1, 0, 0)];
    double *W = (double *)malloc(sizeof(double) * q);
    double *dg1 = (double *)malloc(sizeof(double) * q);
    double *d = (double *)malloc(sizeof(double) * q);        /* tau2*D diagonal */
    double *a1 = (double *)calloc(n, sizeof(double));
    double *a3 = (double *)calloc(n, sizeof(double));
    double *A = (double *)calloc((size_t)n * n, sizeof(double));
    orc_compute_W(o, j, j - 1, W);                            /* u[i], lambda[i-1] :421 */
    for (int e = 0; e < q; ++e) {
        d[e] = tau2 * S_(j - 1, e);
        dg1[e] = sqrt(d[e]) * draw_normal(o->seed, it, SITE_G_Z1, (uint32_t)e, 0);   /* :429 */
    }
    /* a1 = (y - X W - mu)/tau :432 ; a3 = (X/tau) dg1 + dg2 :433 */
    for (int e = 0; e < q; ++e) {
        const double *xc = o->X + (size_t)n * e; double we = W[e], ge = dg1[e];
        for (int i = 0; i < n; ++i) { a1[i] += xc[i] * we; a3[i] += (xc[i] / tau) * ge; }
    }
    for (int i = 0; i < n; ++i) {
        a1[i] = (o->y[i] - a1[i] - mu) / tau;
        a3[i] += draw_normal(o->seed, it, SITE_G_Z2, (uint32_t)i, 0);               /* :430 */
    }
    /* A = Xt * tau2D * Xt' + I :434  (Xt = X/tau) */
    if (o->cost_mode == 2 && orc_have_blas()) {
        /* as the reference executes it: Xt = X ./ tau (a copy, :425), Xt * tau2D (Diagonal product: another n x q copy), then a
         * dense gemm with transpose(Xt), the identity added, and a general LU solve */
        double *Xt = (double *)malloc(sizeof(double) * (size_t)n * q), *XD = (double *)malloc(sizeof(double) * (size_t)n * q);
        for (size_t e = 0; e < (size_t)q; ++e) {
            const double *xc = o->X + (size_t)n * e; double *tc = Xt + (size_t)n * e, *dc = XD + (size_t)n * e, de = d[e];
            for (int i = 0; i < n; ++i) { tc[i] = xc[i] / tau; dc[i] = tc[i] * de; }
        }
        const int64_t nn = n, qq = q, one = 1; const double alpha = 1.0, beta = 0.0;
        g_dgemm("N", "T", &nn, &nn, &qq, &alpha, XD, &nn, Xt, &nn, &beta, A, &nn);
        for (int i = 0; i < n; ++i) A[i + (size_t)n * i] += 1.0;
        for (int i = 0; i < n; ++i) a1[i] -= a3[i];
        int64_t *ipiv = (int64_t *)malloc(sizeof(int64_t) * n), info = 0;
        g_dgesv(&nn, &one, A, &nn, ipiv, a1, &nn, &info);
        if (info != 0) o->status = 3;
        free(ipiv); free(Xt); free(XD);
        goto solved;
    }
    {
        const int EB = 64;
#pragma omp parallel for schedule(dynamic, 1) if ((double)n * n * q > 2e7)   /* small problems: a parallel region costs more than it saves (hundreds of host threads on a GPU box) */
        for (int jb = 0; jb < n; jb += 32) {
            int jend = jb + 32 < n ? jb + 32 : n;
            for (int e0 = 0; e0 < q; e0 += EB) {
                int e1 = e0 + EB < q ? e0 + EB : q;
                for (int c = jb; c < jend; ++c) {
                    double *Ac = A + (size_t)n * c;
                    int i0 = o->cost_mode ? 0 : c;
                    for (int e = e0; e < e1; ++e) {
                        const double *xc = o->X + (size_t)n * e;
                        double f = (xc[c] / tau) * d[e] / tau;
                        if (f == 0.0 && !o->cost_mode) continue;   /* reference-cost mode stays dense */
                        for (int i = i0; i < n; ++i) Ac[i] += xc[i] * f;
                    }
                }
            }
        }
        if (!o->cost_mode)
            for (int c = 0; c < n; ++c) for (int i = c + 1; i < n; ++i) A[c + (size_t)n * i] = A[i + (size_t)n * c];
        for (int i = 0; i < n; ++i) A[i + (size_t)n * i] += 1.0;
    }
    for (int i = 0; i < n; ++i) a1[i] -= a3[i];
    if (lu_solve(A, n, a1)) o->status = 3;                    /* a4 = A \ (a1 - a3) */
solved:
    /* a5 = dg1 + tau2D * Xt' a4 :435 ; gamma = a5 + W :436 */
    for (int e = 0; e < q; ++e) {
        const double *xc = o->X + (size_t)n * e; double s = 0.0;
        for (int i = 0; i < n; ++i) s += (xc[i] / tau) * a1[i];
        G_(j, e) = (dg1[e] + d[e] * s) + W[e];
    }
    free(W); free(dg1); free(d); free(a1); free(a3); free(A);
}

/* ------------------------------------------------------------------ update_D! (gibbs.jl:454-458) */
void orc_update_D(orc_t *o, int j, uint32_t it)
{
    const int q = o->q;
    const double tau2 = o->tau2[IDX(j, 1, 0, 0)], theta = o->theta[IDX(j - 1, 1, 0, 0)];
    double *W = (double *)malloc(sizeof(double) * q);
    orc_compute_W(o, j, j - 1, W);
    for (int e = 0; e < q; ++e) {
        double g = G_(j, e) - W[e];
        double a_ = (g * g) / tau2;
        S_(j, e) = orc_sample_gig(o, 0.5, a_, theta, it, (uint32_t)e);   /* sample_rgig(theta, a_) :116-118 */
    }
    free(W);
}

/* ------------------------------------------------------------------ update_theta! (gibbs.jl:476-479) */
void orc_theta_params(orc_t *o, int j, double *params)
{
    kbn_t k = { 0, 0 };
    for (int e = 0; e < o->q; ++e) kbn_add(&k, S_(j, e));
    params[0] = o->zeta + (o->V * (o->V + 1)) / 2.0;
    params[1] = 2.0 / (2.0 * o->iota + kbn_val(&k));
}
void orc_update_theta(orc_t *o, int j, uint32_t it)
{
    double p[2]; orc_theta_params(o, j, p);
    o->theta[IDX(j, 1, 0, 0)] = draw_gamma(o, p[0], it, SITE_THETA, 0) * p[1];
}

/* ------------------------------------------------------------------ update_Delta! (gibbs.jl:496-499, 130-140) */
void orc_update_Delta(orc_t *o, int j, uint32_t it)
{
    kbn_t k1 = { 0, 0 }, k0 = { 0, 0 };
    for (int v = 0; v < o->V; ++v) { kbn_add(&k1, XI_(j, v)); kbn_add(&k0, 1.0 - XI_(j, v)); }
    double a = o->aDelta + kbn_val(&k1), b = o->bDelta + kbn_val(&k0), out;
    if (a > 0.0 && b > 0.0) {
        double g1 = draw_gamma(o, a, it, SITE_DELTA, 0), g2 = draw_gamma(o, b, it, SITE_DELTA, 1);
        out = g1 / (g1 + g2);
    } else if (a > 0.0) out = 1.0;
    else if (b > 0.0) out = 0.0;
    else { double ua, ub; draw2(o->seed, it, SITE_DELTA_COIN, 0, 0, &ua, &ub); out = (ua < 0.5) ? 0.0 : 1.0; }
    o->Delta[IDX(j, 1, 0, 0)] = out;
}

/* ------------------------------------------------------------------ update_M! (gibbs.jl:516-547) */
/* Psi (R x R col-major) and df */
void orc_M_params(orc_t *o, int j, double *Psi, double *df)
{
    const int V = o->V, R = o->R;
    int nz = 0;
    for (int a = 0; a < R; ++a) for (int b = 0; b < R; ++b) Psi[a + R * b] = (a == b);
    for (int v = 0; v < V; ++v) {
        for (int a = 0; a < R; ++a) for (int b = 0; b < R; ++b) Psi[a + R * b] += U_(j, a, v) * U_(j, b, v);
        if (!(fabs(XI_(j, v)) <= 0.1)) ++nz;                    /* !isapprox(xi,0,atol=0.1) :522 */
    }
    *df = o->nu + nz;
}
void orc_update_M(orc_t *o, int j, uint32_t it)
{
    const int R = o->R;
    double *Psi = (double *)malloc(sizeof(double) * R * R);
    double *C = (double *)malloc(sizeof(double) * R * R);
    double *Mo = (double *)malloc(sizeof(double) * R * R);
    double df; orc_M_params(o, j, Psi, &df);
    memcpy(C, Psi, sizeof(double) * R * R);
    if (chol_lower(C, R)) {                                     /* retry ladder :529-543 */
        o->jitter_events++;
        for (int a = 0; a < R; ++a) Psi[a + R * a] += 1e-5;
        memcpy(C, Psi, sizeof(double) * R * R);
        if (chol_lower(C, R)) { o->status = 3; }
    }
    draw_inverse_wishart(o, df, C, it, SITE_M_CHI, SITE_M_N, Mo);
    for (int a = 0; a < R; ++a) for (int b = 0; b < R; ++b) M_(j, a, b) = Mo[a + R * b];
    free(Psi); free(C); free(Mo);
}

/* ------------------------------------------------------------------ update_mu! (gibbs.jl:565-570) */
void orc_mu_params(orc_t *o, int j, double *params)
{
    const int n = o->n, q = o->q;
    double *xg = (double *)calloc(n, sizeof(double));
    for (int e = 0; e < q; ++e) {
        double g = G_(j, e); const double *xc = o->X + (size_t)n * e;
        for (int i = 0; i < n; ++i) xg[i] += xc[i] * g;
    }
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += o->y[i] - xg[i];
    params[0] = s / n;
    params[1] = sqrt(o->tau2[IDX(j, 1, 0, 0)] / n);
    free(xg);
}
void orc_update_mu(orc_t *o, int j, uint32_t it)
{
    double p[2]; orc_mu_params(o, j, p);
    o->mu[IDX(j, 1, 0, 0)] = p[0] + p[1] * draw_normal(o->seed, it, SITE_MU, 0, 0);
}

/* ------------------------------------------------------------------ update_Lambda! (gibbs.jl:586-613) */
/* probs: R x 3 (row-major here: probs[3*r+c]), unnormalised weights pi_prev[r,c]*exp(l_c - max), order (0,1,-1) */
void orc_Lambda_params(orc_t *o, int j, double *probs)
{
    const int V = o->V, R = o->R, q = o->q;
    const double tau2 = o->tau2[IDX(j, 1, 0, 0)];
    double *lamv = (double *)malloc(sizeof(double) * R);
    double *Wc = (double *)malloc(sizeof(double) * q);
    for (int r = 0; r < R; ++r) {
        double ll[3];
        for (int c = 0; c < 3; ++c) {
            for (int x = 0; x < R; ++x) lamv[x] = LAM_(j - 1, x);     /* Lambda built once from row i-1 :587 */
            lamv[r] = LAMBDA_VALUES[c];
            int e = 0;
            for (int k = 0; k < V; ++k) for (int l = k; l < V; ++l, ++e) {
                double s = 0.0;
                for (int x = 0; x < R; ++x) s += U_(j, x, l) * lamv[x] * U_(j, x, k);
                Wc[e] = s;
            }
            kbn_t kk = { 0, 0 };
            for (e = 0; e < q; ++e) {
                double sd = sqrt(tau2 * S_(j, e)), zz = (G_(j, e) - Wc[e]) / sd;
                kbn_add(&kk, -0.5 * zz * zz - log(sd) - 0.5 * log(2.0 * ORC_PI));   /* logpdf(Normal) :603-605 */
            }
            ll[c] = kbn_val(&kk);
        }
        double pmax = fmax(ll[0], fmax(ll[1], ll[2]));
        for (int c = 0; c < 3; ++c) probs[3 * r + c] = PI_(j - 1, r, c) * exp(ll[c] - pmax);
    }
    free(lamv); free(Wc);
}
void orc_update_Lambda(orc_t *o, int j, uint32_t it)
{
    const int R = o->R;
    double *probs = (double *)malloc(sizeof(double) * 3 * R);
    orc_Lambda_params(o, j, probs);
    for (int r = 0; r < R; ++r) {
        double ua, ub; draw2(o->seed, it, SITE_LAMBDA, (uint32_t)r, 0, &ua, &ub);
        LAM_(j, r) = LAMBDA_VALUES[draw_categorical3(probs + 3 * r, ua)];
    }
    free(probs);
}

/* ------------------------------------------------------------------ update_pi! (gibbs.jl:630-636, 159-169) */
void orc_pi_alpha(orc_t *o, int j, int r, double alpha[3])
{
    double base = pow((double)(r + 1), o->eta);                  /* caller's eta (no local reset here) */
    double lam = LAM_(j, r);
    if (lam == 1.0) { alpha[0] = base; alpha[1] = 2.0; alpha[2] = 1.0; }
    else if (lam == 0.0) { alpha[0] = base + 1.0; alpha[1] = 1.0; alpha[2] = 1.0; }
    else { alpha[0] = base; alpha[1] = 1.0; alpha[2] = 2.0; }
}
void orc_update_pi(orc_t *o, int j, uint32_t it)
{
    for (int r = 0; r < o->R; ++r) {
        double alpha[3], p[3];
        orc_pi_alpha(o, j, r, alpha);
        draw_dirichlet3(o, alpha, it, SITE_PI, (uint32_t)r, p);
        for (int c = 0; c < 3; ++c) PI_(j, r, c) = p[c];
    }
}

/* ------------------------------------------------------------------ gibbs_sample! (gibbs.jl:663-677) */
void orc_gibbs_sample(orc_t *o, int j, uint32_t it)
{
    orc_update_tau2(o, j, it);
    orc_update_u_xi(o, j, it);
    orc_update_gamma(o, j, it);
    orc_update_D(o, j, it);
    orc_update_theta(o, j, it);
    orc_update_Delta(o, j, it);
    orc_update_M(o, j, it);
    orc_update_mu(o, j, it);
    orc_update_Lambda(o, j, it);
    orc_update_pi(o, j, it);
}

/* copy_table!(table,to,from)  utils.jl:72-84 (11 live columns) */
static void copy_row(orc_t *o, int to, int from)
{
    const int V = o->V, R = o->R, q = o->q;
    o->tau2[IDX(to, 1, 0, 0)] = o->tau2[IDX(from, 1, 0, 0)];
    o->theta[IDX(to, 1, 0, 0)] = o->theta[IDX(from, 1, 0, 0)];
    o->Delta[IDX(to, 1, 0, 0)] = o->Delta[IDX(from, 1, 0, 0)];
    o->mu[IDX(to, 1, 0, 0)] = o->mu[IDX(from, 1, 0, 0)];
    for (int v = 0; v < V; ++v) { XI_(to, v) = XI_(from, v); for (int r = 0; r < R; ++r) U_(to, r, v) = U_(from, r, v); }
    for (int e = 0; e < q; ++e) { G_(to, e) = G_(from, e); S_(to, e) = S_(from, e); }
    for (int r = 0; r < R; ++r) {
        LAM_(to, r) = LAM_(from, r);
        for (int c = 0; c < 3; ++c) PI_(to, r, c) = PI_(from, r, c);
        for (int b = 0; b < R; ++b) M_(to, r, b) = M_(from, r, b);
    }
}

/* run!(X,y,state,c,first_index,nburn,total,...,purge_burn,...)  gibbs.jl:849-864.
 * first_index/total/purge_burn are the reference's 1-based values; purge_burn<=0 means `nothing`.
 * Returns the 1-based row index j the NEXT call would write (for continuation). */
int orc_run(orc_t *o, int first_index, int nburn, int total, int purge_burn)
{
    int j = first_index;
    for (int i = first_index; i <= total; ++i) {
        if (j > o->tot) return -9;                 /* Julia: BoundsError on state.X[j,...] (e.g. purge_burn = 1 with the
                                                      nsamp + purge_burn rows initialize_and_run! allocates, gibbs.jl:827-830) */
        o->iter += 1;
        orc_gibbs_sample(o, j - 1, (uint32_t)o->iter);
        if (o->status) return -o->status;
        if (purge_burn > 0 && i < nburn && j == purge_burn + 1) { copy_row(o, 0, j - 1); j = 1; }
        j = j + 1;
    }
    return j;
}

/* ------------------------------------------------------------------ rhat (convergence.jl:4-65) */
/* chains: (niter_in, nparams, nchains) column-major.  out: nparams */
void orc_rhat(const double *chains, int niter_in, int nparams, int nchains, double *out)
{
    int niter = niter_in / 2, m = 2 * nchains;
    if (niter - 1 <= 0) { for (int p = 0; p < nparams; ++p) out[p] = NAN; return; }
    double cf = (double)(niter - 1) / niter;
    double *mean = (double *)malloc(sizeof(double) * m), *var = (double *)malloc(sizeof(double) * m);
    for (int p = 0; p < nparams; ++p) {
        for (int c = 0; c < nchains; ++c) {
            const double *x = chains + (size_t)niter_in * ((size_t)p + (size_t)nparams * c);
            /* copyto_split!: first half = rows [0,niter), second half = LAST niter rows */
            for (int h = 0; h < 2; ++h) {
                const double *xs = h == 0 ? x : x + (niter_in - niter);
                double s = 0.0; for (int i = 0; i < niter; ++i) s += xs[i];
                double mu = s / niter, v = 0.0;
                for (int i = 0; i < niter; ++i) v += (xs[i] - mu) * (xs[i] - mu);
                mean[2 * c + h] = mu; var[2 * c + h] = v / (niter - 1);
            }
        }
        double W = 0.0, mm = 0.0;
        for (int k = 0; k < m; ++k) { W += var[k]; mm += mean[k]; }
        W /= m; mm /= m;
        double B = 0.0; for (int k = 0; k < m; ++k) B += (mean[k] - mm) * (mean[k] - mm);
        B /= (m - 1);
        double varp = cf * W + B;
        if (varp == 0.0 && W == 0.0) out[p] = 1.0;
        else if (W == 0.0) out[p] = INFINITY;
        else out[p] = sqrt(varp / W);
    }
    free(mean); free(var);
}

int orc_sizeof(void) { return (int)sizeof(orc_t); }
int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

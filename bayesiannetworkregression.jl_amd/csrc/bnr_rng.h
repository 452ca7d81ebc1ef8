// bnr_rng.h -- draw-site contract of the MI355X Gibbs sampler: counter-based Philox-4x32-10 and the
// scalar samplers built on it (uniform, normal, Gamma, GIG).  __host__ __device__ so that the same source
// runs inside the kernels and behind the bnr_host_* exports that the CPU test-suite checks.
//
// One draw = philox(counter = {iteration, site, element, attempt}, key = {lo32(s), hi32(s)}), s = seed + chain
// (the reference keys chain c with Xoshiro(seed+c), gibbs.jl:928).  Results never depend on launch geometry.
//
// The distributions are the reference's: GIG by the three GIGrvg branches the reference translated
// (gig.jl:8-176), Gamma by Marsaglia-Tsang (what Distributions.jl uses for rand(Gamma)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define BNR_HD __host__ __device__ __forceinline__

enum {
    SITE_INIT_S = 1, SITE_INIT_PI = 2, SITE_INIT_LAM = 3, SITE_INIT_XI = 4,
    SITE_INIT_M_CHI = 5, SITE_INIT_M_N = 6, SITE_INIT_U = 7, SITE_INIT_GAMMA = 8,
    SITE_TAU2 = 16, SITE_XI = 17, SITE_U_Z = 18, SITE_G_Z1 = 19, SITE_G_Z2 = 20,
    SITE_D_GIG = 21, SITE_D_GAMMA = 22, SITE_THETA = 23, SITE_DELTA = 24, SITE_DELTA_COIN = 25,
    SITE_M_CHI = 26, SITE_M_N = 27, SITE_MU = 28, SITE_LAMBDA = 29, SITE_PI = 30
};
#define BNR_MAX_ATTEMPTS 100000u
#define BNR_ATT_BOOST 0xFFFFFFFFu
#define BNR_PI 3.14159265358979323846

struct bnr_u4 { uint32_t x, y, z, w; };

BNR_HD bnr_u4 bnr_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
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
    bnr_u4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}

// two uniforms in the open interval (0,1), 53 bits each
BNR_HD void bnr_draw2(uint64_t seed, uint32_t it, uint32_t site, uint32_t elem, uint32_t att, double &ua, double &ub)
{
    bnr_u4 r = bnr_philox4x32_10(it, site, elem, att, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t a = (uint64_t)r.x | ((uint64_t)r.y << 32);
    uint64_t b = (uint64_t)r.z | ((uint64_t)r.w << 32);
    ua = ((double)(a >> 11) + 0.5) * 0x1.0p-53;
    ub = ((double)(b >> 11) + 0.5) * 0x1.0p-53;
}

// cos(2 pi u) with an exact quadrant reduction: sin/cos are only evaluated on [-pi/4, pi/4]
BNR_HD double bnr_cos2pi(double u)
{
    double x = 4.0 * u;
    double qd = floor(x + 0.5);
    double t = (x - qd) * 1.57079632679489661923;
    int qi = ((int)qd) & 3;
    double r;
    if (qi == 0) r = cos(t);
    else if (qi == 1) r = -sin(t);
    else if (qi == 2) r = -cos(t);
    else r = sin(t);
    return r;
}
BNR_HD double bnr_normal_from(double ua, double ub) { return sqrt(-2.0 * log(ua)) * bnr_cos2pi(ub); }
BNR_HD double bnr_normal(uint64_t seed, uint32_t it, uint32_t site, uint32_t elem, uint32_t att)
{
    double a, b;
    bnr_draw2(seed, it, site, elem, att, a, b);
    return bnr_normal_from(a, b);
}

// Gamma(shape a, scale 1).  *cap is set to 1 when the attempt cap is hit.
BNR_HD double bnr_gamma(uint64_t seed, double a, uint32_t it, uint32_t site, uint32_t elem, int *cap)
{
    double boost = 1.0;
    if (a < 1.0) {
        double ua, ub;
        bnr_draw2(seed, it, site, elem, BNR_ATT_BOOST, ua, ub);
        boost = pow(ua, 1.0 / a);
        a += 1.0;
    }
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t t = 0; t < BNR_MAX_ATTEMPTS; ++t) {
        double x = bnr_normal(seed, it, site, elem, 2 * t);
        double ua, ub;
        bnr_draw2(seed, it, site, elem, 2 * t + 1, ua, ub);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (log(ua) < 0.5 * x * x + d - d * v + d * log(v)) return d * v * boost;
    }
    if (cap) *cap = 1;
    return d * boost;
}

// ---------------------------------------------------------------------------------- GIG (gig.jl)
BNR_HD double bnr_gig_mode(double lambda, double omega)   // gig.jl:170-176
{
    if (lambda >= 1.0) return (sqrt((lambda - 1.0) * (lambda - 1.0) + omega * omega) + lambda - 1.0) / omega;
    return omega / (sqrt((1.0 - lambda) * (1.0 - lambda) + omega * omega) + (1.0 - lambda));
}

// sample_gig(rng, lambda, chi, psi), gig.jl:8-42, with the three rejection branches (44-168), split into the setup of a
// draw and ONE rejection attempt, so that a kernel can evaluate several attempts of the same draw side by side (attempt k
// uses the variates of counter {it, SITE_D_GIG, elem, k}: the first accepted attempt is the draw, whoever evaluates it).
// Quirks kept: no exception for invalid parameters (9-13); Gamma SCALE psi/2 in the chi~0 branch (17).
struct bnr_gig_ctx {
    int kind;                    // 0: chi ~ 0 (Gamma), 1: psi ~ 0 (inverse Gamma), 2: ratio of uniforms, 3: concave, 4: none (NaN)
    double lambda_old, lambda, alpha, omega, xm;
    double t, s, nc, ulo, uhi, xoff;                       // ratio of uniforms (shift and no-shift share one loop)
    double x0, k0, A0, A1, A2, k1, k2, Atot, x0l;          // concave
    int half;
};
BNR_HD void bnr_gig_setup(bnr_gig_ctx &c, double lambda, double chi, double psi)
{
    const double eps10 = 2.220446049250313e-16 * 10.0;
    c.lambda_old = lambda; c.lambda = lambda; c.alpha = 0.0; c.omega = 0.0; c.xm = 0.0;
    if (chi < eps10) { c.kind = 0; return; }
    if (psi < eps10) { c.kind = 1; return; }
    if (lambda < 0.0) lambda = -lambda;
    c.lambda = lambda;
    const double alpha = sqrt(chi / psi), omega = sqrt(psi * chi);
    c.alpha = alpha; c.omega = omega;
    const bool shift = (lambda > 2.0 || omega > 3.0);
    if (shift || lambda >= 1.0 - 2.25 * (omega * omega) || omega > 0.2) {
        // gig_ROU_shift (gig.jl:44-78) and gig_ROU_noshift (gig.jl:80-100) share one rejection loop:
        //   U = ulo + ru (uhi - ulo), X = U / V + xoff;  no-shift is (ulo, uhi, xoff) = (0, um, 0), bit-identical to um*ru, U/V.
        // One loop instead of two keeps a wavefront whose lanes are split over the branches from running them back to back.
        c.kind = 2;
        double t = 0.5 * (lambda - 1.0), s = 0.25 * omega;
        double xm = bnr_gig_mode(lambda, omega);
        double nc = t * log(xm) - s * (xm + 1.0 / xm);
        double ulo, uhi, xoff;
        if (shift) {
            double a = -(2.0 * (lambda + 1.0) / omega + xm);
            double b = (2.0 * (lambda - 1.0) * xm / omega - 1.0);
            double cc = xm;
            double p = b - a * a / 3.0;
            double q = 2.0 * a * a * a / 27.0 - a * b / 3.0 + cc;
            double fi = acos(-q / (2.0 * sqrt(-p * p * p / 27.0)));
            double fak = 2.0 * sqrt(-p / 3.0);
            double y1 = fak * cos(fi / 3.0) - a / 3.0;
            double y2 = fak * cos(fi / 3.0 + 4.0 / 3.0 * BNR_PI) - a / 3.0;
            uhi = (y1 - xm) * exp(t * log(y1) - s * (y1 + 1.0 / y1) - nc);
            ulo = (y2 - xm) * exp(t * log(y2) - s * (y2 + 1.0 / y2) - nc);
            xoff = xm;
        } else {
            double ym = ((lambda + 1.0) + sqrt((lambda + 1.0) * (lambda + 1.0) + omega * omega)) / omega;
            uhi = exp(0.5 * (lambda + 1.0) * log(ym) - s * (ym + 1.0 / ym) - nc);
            ulo = 0.0;
            xoff = 0.0;
        }
        c.t = t; c.s = s; c.xm = xm; c.nc = nc; c.ulo = ulo; c.uhi = uhi; c.xoff = xoff;
        return;
    }
    if (lambda >= 0.0 && omega > 0.0) {
        // gig_concave, gig.jl:102-168.  For lambda = 1/2 (the only value update_D! uses) the powers are square roots.
        c.kind = 3;
        const bool half = (lambda == 0.5);
        double xm = bnr_gig_mode(lambda, omega);
        double x0 = omega / (1.0 - lambda);
        double k0 = exp((lambda - 1.0) * log(xm) - 0.5 * omega * (xm + 1.0 / xm));
        double A0, A1, A2, k1, k2;
        A0 = k0 * x0;
        const double x0l = half ? sqrt(x0) : pow(x0, lambda);                 // x0^lambda
        if (x0 >= 2.0 / omega) {
            k1 = 0.0; A1 = 0.0;
            k2 = half ? 1.0 / x0l : pow(x0, lambda - 1.0);
            A2 = k2 * 2.0 * exp(-omega * x0 / 2.0) / omega;
        } else {
            k1 = exp(-omega);
            const double tw = 2.0 / omega, twl = half ? sqrt(tw) : pow(tw, lambda);
            if (lambda == 0.0) A1 = k1 * log(2.0 / (omega * omega));
            else A1 = k1 / lambda * (twl - x0l);
            k2 = half ? 1.0 / twl : pow(tw, lambda - 1.0);
            A2 = k2 * 2.0 * exp(-1.0) / omega;
        }
        c.half = half ? 1 : 0; c.xm = xm; c.x0 = x0; c.k0 = k0; c.A0 = A0; c.A1 = A1; c.A2 = A2; c.k1 = k1; c.k2 = k2;
        c.Atot = A0 + A1 + A2; c.x0l = x0l;
        return;
    }
    c.kind = 4;
}
// which of bnr_gig_setup's branches a draw takes (the same tests on the same expressions), without the setup's arithmetic
BNR_HD int bnr_gig_kind(double lambda, double chi, double psi)
{
    const double eps10 = 2.220446049250313e-16 * 10.0;
    if (chi < eps10) return 0;
    if (psi < eps10) return 1;
    if (lambda < 0.0) lambda = -lambda;
    const double omega = sqrt(psi * chi);
    const bool shift = (lambda > 2.0 || omega > 3.0);
    if (shift || lambda >= 1.0 - 2.25 * (omega * omega) || omega > 0.2) return 2;
    if (lambda >= 0.0 && omega > 0.0) return 3;
    return 4;
}
// attempt k of a kind 2 / 3 draw: true and the value (already scaled by alpha) when accepted
BNR_HD bool bnr_gig_try(const bnr_gig_ctx &c, uint64_t seed, uint32_t it, uint32_t elem, uint32_t k, double &out)
{
    double ru, rv;
    bnr_draw2(seed, it, SITE_D_GIG, elem, k, ru, rv);
    if (c.kind == 2) {
        double U = c.ulo + ru * (c.uhi - c.ulo);
        double X = U / rv + c.xoff;
        if (X > 0.0 && log(rv) <= c.t * log(X) - c.s * (X + 1.0 / X) - c.nc) {
            out = c.lambda_old < 0.0 ? c.alpha / X : c.alpha * X;
            return true;
        }
        return false;
    }
    const double lambda = c.lambda, omega = c.omega;
    double Vv = c.Atot * ru, hx, X;
    if (Vv <= c.A0) { X = c.x0 * Vv / c.A0; hx = c.k0; }
    else {
        Vv -= c.A0;
        if (Vv <= c.A1) {
            if (lambda == 0.0) { X = omega * exp(exp(omega) * Vv); hx = c.k1 / X; }
            else if (c.half) { double r = c.x0l + (lambda / c.k1 * Vv); X = r * r; hx = c.k1 / r; }
            else { X = pow(c.x0l + (lambda / c.k1 * Vv), 1.0 / lambda); hx = c.k1 * pow(X, lambda - 1.0); }
        } else {
            Vv -= c.A1;
            double a = (c.x0 > 2.0 / omega) ? c.x0 : 2.0 / omega;
            X = -2.0 / omega * log(exp(-omega / 2.0 * a) - omega / (2.0 * c.k2) * Vv);
            hx = c.k2 * exp(-omega / 2.0 * X);
        }
    }
    double U = rv * hx;
    if (log(U) <= (lambda - 1.0) * log(X) - omega / 2.0 * (X + 1.0 / X)) {
        out = c.lambda_old < 0.0 ? c.alpha / X : c.alpha * X;
        return true;
    }
    return false;
}
// the draws that need no rejection loop of their own (kinds 0, 1, 4)
BNR_HD double bnr_gig_degenerate(const bnr_gig_ctx &c, uint64_t seed, double chi, double psi, uint32_t it, uint32_t elem, int *cap)
{
    const double lambda = c.lambda_old;
    if (c.kind == 0) {
        if (lambda > 0.0) return bnr_gamma(seed, lambda, it, SITE_D_GAMMA, elem, cap) * (psi / 2.0);
        return 1.0 / (bnr_gamma(seed, -lambda, it, SITE_D_GAMMA, elem, cap) * (psi / 2.0));
    }
    if (c.kind == 1) {
        if (lambda > 0.0) return 1.0 / (bnr_gamma(seed, lambda, it, SITE_D_GAMMA, elem, cap) * (chi / 2.0));
        return bnr_gamma(seed, -lambda, it, SITE_D_GAMMA, elem, cap) * (chi / 2.0);
    }
    return NAN;   // the reference returns `nothing` here (gig.jl:41)
}
BNR_HD double bnr_gig(uint64_t seed, double lambda, double chi, double psi, uint32_t it, uint32_t elem, int *cap)
{
    bnr_gig_ctx c;
    bnr_gig_setup(c, lambda, chi, psi);
    if (c.kind != 2 && c.kind != 3) return bnr_gig_degenerate(c, seed, chi, psi, it, elem, cap);
    for (uint32_t k = 0; k < BNR_MAX_ATTEMPTS; ++k) {
        double out;
        if (bnr_gig_try(c, seed, it, elem, k, out)) return out;
    }
    if (cap) *cap = 1;
    return c.alpha * c.xm;
}

// StatsBase.sample(rng, vals, weights): linear scan of the cumulative weights; order [0,1,-1] (gibbs.jl:207,610)
BNR_HD int bnr_categorical3(double w0, double w1, double w2, double u01)
{
    double t = u01 * (w0 + w1 + w2);
    int i = 0;
    double cw = w0;
    if (cw < t) { i = 1; cw += w1; if (cw < t) { i = 2; } }
    return i;
}
BNR_HD double bnr_lambda_value(int c) { return c == 0 ? 0.0 : (c == 1 ? 1.0 : -1.0); }

// edge e <-> (l,k), l >= k, column-wise lower triangle incl. diagonal (utils.jl:50-55), all 0-based
BNR_HD int bnr_edge_index(int V, int l, int k) { return k * V - (k * (k - 1)) / 2 + (l - k); }

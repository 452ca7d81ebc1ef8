// Lab for the binary-X Gram: G_c = X diag(S_c) X' for a 0/1 model matrix on the i8 matrix pipe.
//   S_c[k] = sum_l d_l[k] 2^(e + 1 - 7 (l + 1)) + r,  d_l in 0..127, e = exponent of max S, |r| < 2^(e + 1 - 7 L)      (k_sdigits)
//   T_l = X diag(d_l) X' exactly in i32 (v_mfma_i32_16x16x64_i8: A = byte mask of X (0xFF = -1), B = mask AND digits = x d)
//   G = - sum_l T_l 2^(e + 1 - 7 (l + 1)) in f64 (Horner, L roundings).          |G - G_f64| <= q 2^(e + 1 - 7 L) <= 1e-12 max|G| for L = 8 .. 9
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/gram_i8_lab tools/gram_i8_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
typedef int i4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct args_t {
    const unsigned char *XM;            // [n_pad][kslab]: 0xFF where X = 1; column k = ks * kchunk + kk at byte ks * kcp + kk
    const double *S[8];
    unsigned char *D[8];                // [L][kslab] digits of S
    double *scale[8];                   // [1]: 2^(e + 1 - 7 L)
    double *G[8];                       // [ksplit][ntl][64 x 64] partial tiles, element (i, j) at [j * 64 + i]
    int n_pad, q, kslab, kcp, kchunk, ksplit, ntile, nchains;
};

template <int L, bool BAL = false>
__global__ __launch_bounds__(1024) void k_sdigits(const args_t A)
{
    const int c = blockIdx.x, tid = threadIdx.x;
    const double *S = A.S[c];
    __shared__ double red[16];
    double m = 0.0;
    for (int k = tid; k < A.q; k += 1024) m = fmax(m, S[k]);
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
    for (int w = 1; w < 16; ++w) m = fmax(m, red[w]);
    int e;
    (void)frexp(m, &e);                                   // m = f 2^e, f in [0.5, 1): every S < 2^e
    // plain: 7-bit digits 0..127, S up < 2^(7 L).  balanced: base-256 digits -128..127 (one plane fewer for the same bits): S up < 2^(8 L - 2), the top digit stays below 64
    const int bits = BAL ? 8 * L - 2 : 7 * L;
    const double up = ldexp(1.0, bits - e);
    if (tid == 0) A.scale[c][0] = ldexp(1.0, e - bits);
    unsigned char *D = A.D[c];
    for (int idx = tid; idx < A.kslab; idx += 1024) {
        const int ks = idx / A.kcp, kk = idx % A.kcp, k = ks * A.kchunk + kk;
        unsigned long long N = 0;
        if (kk < A.kchunk && k < A.q) N = (unsigned long long)(S[k] * up);
        if (!BAL) {
#pragma unroll
            for (int l = 0; l < L; ++l) D[(size_t)l * A.kslab + idx] = (unsigned char)((N >> (7 * (L - 1 - l))) & 127ull);
        } else {
            // least significant digit first: a byte >= 128 becomes byte - 256 and carries one into the next
            unsigned carry = 0;
#pragma unroll
            for (int l = L - 1; l >= 0; --l) {
                unsigned b = (unsigned)((N >> (8 * (L - 1 - l))) & 255ull) + carry;
                carry = b >= 128u ? 1u : 0u;
                D[(size_t)l * A.kslab + idx] = (unsigned char)(b & 255u);          // two's complement byte of b - 256 carry
            }
        }
    }
}

// version 2: X tiles staged through LDS (full 128-byte lines, each byte fetched once per workgroup instead of once per wave pair), two k-steps per
// batch, double buffered; balanced base-256 digits (BAL) one plane fewer.  Row stride 144 bytes: conflict-free ds_read_b128 fragments.
template <int L, bool BAL, int EXP = 0>
__global__ __launch_bounds__(256, 2) void k_gram_i8v2(const args_t A)
{
    const int gid = blockIdx.x, gx = gid & 7, gr = gid >> 3;
    const int chain = gr % A.nchains, slot = (gr / A.nchains) * 8 + gx;
    const int ntl = A.ntile * (A.ntile + 1) / 2;
    if (slot >= ntl * A.ksplit) return;
    const int ks = slot % A.ksplit, t = slot / A.ksplit;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    constexpr int RS = 144;                                 // bytes per staged row (128 + 16 pad)
    constexpr int TB = 64 * RS;                             // one 64-row tile of a batch
    extern __shared__ i4 smem[];
    unsigned char *sX = (unsigned char *)smem;              // [buf][I | J][64 rows][144]
    i4 *sD = (i4 *)(sX + 4 * TB);                           // digits: [l][kcp / 16]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ln = lane & 15, lq = lane >> 4;
    const int wi = (wave >> 1) & 1, wj = wave & 1;
    const int ng = A.kcp / 16;
    for (int idx = tid; idx < L * ng; idx += 256) {
        const int l = idx / ng, g = idx % ng;
        sD[idx] = *(const i4 *)(A.D[chain] + (size_t)l * A.kslab + (size_t)ks * A.kcp + 16 * g);
    }
    // staging map: thread -> row tid / 8 (+ 32), 16-byte column tid % 8 of the 128-byte batch
    const int srow = tid >> 3, scol = tid & 7;
    const unsigned char *gI = A.XM + (size_t)(ti * 64 + srow) * A.kslab + (size_t)ks * A.kcp + 16 * scol;
    const unsigned char *gJ = A.XM + (size_t)(tj * 64 + srow) * A.kslab + (size_t)ks * A.kcp + 16 * scol;
    const size_t r32 = (size_t)32 * A.kslab;
    const int soff = srow * RS + 16 * scol;
    i4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
    const int nbatch = A.kcp / 128, tailstep = (A.kcp % 128) / 64;        // kcp is a multiple of 64: a last half batch is possible
    const int nb = nbatch + (tailstep ? 1 : 0);
    auto load = [&](int b) {
        const int o = 128 * b;
        if (b < nbatch || scol < 4) { r0 = *(const i4 *)(gI + o); r1 = *(const i4 *)(gI + r32 + o); r2 = *(const i4 *)(gJ + o); r3 = *(const i4 *)(gJ + r32 + o); }
    };
    auto store = [&](int buf) {
        unsigned char *d = sX + buf * 2 * TB + soff;
        *(i4 *)d = r0; *(i4 *)(d + 32 * RS) = r1; *(i4 *)(d + TB) = r2; *(i4 *)(d + TB + 32 * RS) = r3;
    };
    // wave w: the 16 i rows 16 w .. 16 w + 15 (ONE masked B fragment per digit plane) x all 64 j rows (four A fragments): one mask per four MFMAs
    i4 acc[L][4];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[l][a] = i4{0, 0, 0, 0};
    load(0); store(0);
    if (nb > 1) load(1);
    __syncthreads();
    // fragment addresses inside a buffer: row * 144 + 64 kk + 16 lq; J tile behind the I tile
    const int fa = ln * RS + 16 * lq + TB, fb = (wave * 16 + ln) * RS + 16 * lq;
    (void)wi; (void)wj;
    for (int b = 0; b < nb; ++b) {
        const unsigned char *xb = sX + (b & 1) * 2 * TB;
        const int nk = (b < nbatch) ? 2 : 1;
        for (int kk = 0; kk < nk; ++kk) {
            const i4 a0 = *(const i4 *)(xb + fa + 64 * kk), a1 = *(const i4 *)(xb + fa + 16 * RS + 64 * kk);
            const i4 a2 = *(const i4 *)(xb + fa + 32 * RS + 64 * kk), a3 = *(const i4 *)(xb + fa + 48 * RS + 64 * kk);
            const i4 b0 = *(const i4 *)(xb + fb + 64 * kk);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const i4 d = (EXP & 4) ? a3 : sD[l * ng + (2 * b + kk) * 4 + lq];      // EXP 4: no digit reads (timing only)
                const i4 m0 = (EXP & 2) ? b0 : (b0 & d);
                acc[l][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, m0, acc[l][0], 0, 0, 0);
                acc[l][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, m0, acc[l][1], 0, 0, 0);
                acc[l][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, m0, acc[l][2], 0, 0, 0);
                acc[l][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, m0, acc[l][3], 0, 0, 0);
            }
            if (kk == 0 && b + 1 < nb) { store((b + 1) & 1); if (b + 2 < nb && !(EXP & 1)) load(b + 2); }
        }
        __syncthreads();
    }
    // acc[l][jt][r]: j = jt 16 + 4 lq + r, i = 16 wave + ln
    const double sc = -A.scale[chain][0];
    const double base = BAL ? 256.0 : 128.0;
    double *out = A.G[chain] + ((size_t)ks * ntl + t) * 4096;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = (double)acc[0][jt][r];
#pragma unroll
            for (int l = 1; l < L; ++l) v = v * base + (double)acc[l][jt][r];
            out[(jt * 16 + 4 * lq + r) * 64 + wave * 16 + ln] = v * sc;
        }
}

template <int L, int EXP = 0>
__global__ __launch_bounds__(256, 2) void k_gram_i8(const args_t A)
{
    const int gid = blockIdx.x, gx = gid & 7, gr = gid >> 3;
    const int chain = gr % A.nchains, slot = (gr / A.nchains) * 8 + gx;
    const int ntl = A.ntile * (A.ntile + 1) / 2;
    if (slot >= ntl * A.ksplit) return;
    const int ks = slot % A.ksplit, t = slot / A.ksplit;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    extern __shared__ i4 sD[];                             // digits of this K slice: [l][kcp / 16] 16-byte groups
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ln = lane & 15, lq = lane >> 4;
    const int wi = (wave >> 1) & 1, wj = wave & 1;
    const int ng = A.kcp / 16;
    for (int idx = tid; idx < L * ng; idx += 256) {
        const int l = idx / ng, g = idx % ng;
        sD[l * ng + g] = *(const i4 *)(A.D[chain] + (size_t)l * A.kslab + (size_t)ks * A.kcp + 16 * g);
    }
    __syncthreads();
    const unsigned char *pa0 = A.XM + (size_t)(tj * 64 + wj * 32 + ln) * A.kslab + (size_t)ks * A.kcp + 16 * lq;     // J rows: A operand (raw mask = -x)
    const unsigned char *pb0 = A.XM + (size_t)(ti * 64 + wi * 32 + ln) * A.kslab + (size_t)ks * A.kcp + 16 * lq;     // I rows: B operand (mask & digit = x d)
    const size_t r16 = (size_t)16 * A.kslab;
    i4 acc[L][2][2];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[l][a][b] = i4{0, 0, 0, 0};
    const int nstep = A.kcp / 64;
    // operands two k-steps ahead (an L2 round trip is longer than one k-step's MFMAs)
    i4 a0 = *(const i4 *)pa0, a1 = *(const i4 *)(pa0 + r16), b0 = *(const i4 *)pb0, b1 = *(const i4 *)(pb0 + r16);
    i4 pa_0 = a0, pa_1 = a1, pb_0 = b0, pb_1 = b1;
    if (nstep > 1) { pa_0 = *(const i4 *)(pa0 + 64); pa_1 = *(const i4 *)(pa0 + r16 + 64); pb_0 = *(const i4 *)(pb0 + 64); pb_1 = *(const i4 *)(pb0 + r16 + 64); }
    for (int st = 0; st < nstep; ++st) {
        i4 na0 = pa_0, na1 = pa_1, nb0 = pb_0, nb1 = pb_1;
        if (st + 2 < nstep && !(EXP & 1)) {
            const int o = 64 * (st + 2);
            na0 = *(const i4 *)(pa0 + o); na1 = *(const i4 *)(pa0 + r16 + o); nb0 = *(const i4 *)(pb0 + o); nb1 = *(const i4 *)(pb0 + r16 + o);
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const i4 d = (EXP & 4) ? b0 : sD[l * ng + st * 4 + lq];
            const i4 m0 = (EXP & 2) ? ((l & 1) ? b0 : b1) : (b0 & d), m1 = (EXP & 2) ? ((l & 1) ? b1 : b0) : (b1 & d);
            acc[l][0][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, m0, acc[l][0][0], 0, 0, 0);
            acc[l][0][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, m1, acc[l][0][1], 0, 0, 0);
            acc[l][1][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, m0, acc[l][1][0], 0, 0, 0);
            acc[l][1][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, m1, acc[l][1][1], 0, 0, 0);
        }
        a0 = pa_0; a1 = pa_1; b0 = pb_0; b1 = pb_1;
        pa_0 = na0; pa_1 = na1; pb_0 = nb0; pb_1 = nb1;
    }
    // acc[l][jt][it][r]: j = wj*32 + jt*16 + 4 lq + r, i = wi*32 + it*16 + ln; T_l = - acc (A held -x)
    const double sc = -A.scale[chain][0];
    double *out = A.G[chain] + ((size_t)ks * ntl + t) * 4096;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = (double)acc[0][jt][it][r];
#pragma unroll
                for (int l = 1; l < L; ++l) v = v * 128.0 + (double)acc[l][jt][it][r];
                out[(wj * 32 + jt * 16 + 4 * lq + r) * 64 + wi * 32 + it * 16 + ln] = v * sc;
            }
}

template <int L, int EXP = 0, int VER = 1, bool BAL = false>
static void run(int n, int V, int nchains, int ksplit, double srange)
{
    const int q = V * (V + 1) / 2, n_pad = (n + 63) / 64 * 64, ntile = n_pad / 64, ntl = ntile * (ntile + 1) / 2;
    const int kchunk = ((q + ksplit - 1) / ksplit + 15) / 16 * 16, kcp = (kchunk + 63) / 64 * 64, kslab = ksplit * kcp;
    std::vector<unsigned char> X((size_t)n * q), XM((size_t)n_pad * kslab, 0);
    srand(3);
    for (size_t i = 0; i < X.size(); ++i) X[i] = (rand() & 1);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < q; ++k) if (X[(size_t)i * q + k]) XM[(size_t)i * kslab + (size_t)(k / kchunk) * kcp + k % kchunk] = 0xFF;
    std::vector<std::vector<double>> S(nchains, std::vector<double>(q));
    for (int c = 0; c < nchains; ++c) for (int k = 0; k < q; ++k) S[c][k] = exp(srange * ((rand() % 20001) / 10000.0 - 1.0)) * (0.5 + (rand() % 1000) / 1000.0);
    args_t A{};
    unsigned char *dXM; CK(hipMalloc(&dXM, XM.size())); CK(hipMemcpy(dXM, XM.data(), XM.size(), hipMemcpyHostToDevice));
    A.XM = dXM; A.n_pad = n_pad; A.q = q; A.kslab = kslab; A.kcp = kcp; A.kchunk = kchunk; A.ksplit = ksplit; A.ntile = ntile; A.nchains = nchains;
    std::vector<double *> dS(nchains), dG(nchains), dsc(nchains); std::vector<unsigned char *> dD(nchains);
    const size_t gbytes = (size_t)ksplit * ntl * 4096 * 8;
    for (int c = 0; c < nchains; ++c) {
        CK(hipMalloc(&dS[c], q * 8)); CK(hipMemcpy(dS[c], S[c].data(), q * 8, hipMemcpyHostToDevice));
        CK(hipMalloc(&dG[c], gbytes)); CK(hipMalloc(&dsc[c], 8)); CK(hipMalloc(&dD[c], (size_t)L * kslab));
        A.S[c] = dS[c]; A.G[c] = dG[c]; A.scale[c] = dsc[c]; A.D[c] = dD[c];
    }
    const int grid = (ntl * ksplit + 7) / 8 * 8 * nchains;
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const size_t lds2 = (size_t)4 * 64 * 144 + (size_t)L * kcp;
#define GRAM_LAUNCH() do { if (VER == 1) k_gram_i8<L, EXP><<<grid, 256, (size_t)L * kcp>>>(A); else k_gram_i8v2<L, BAL, EXP><<<grid, 256, lds2>>>(A); } while (0)
    for (int i = 0; i < 3; ++i) { k_sdigits<L, BAL><<<nchains, 1024>>>(A); GRAM_LAUNCH(); }
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_sdigits<L, BAL><<<nchains, 1024>>>(A);
    CK(hipEventRecord(e1));
    for (int i = 0; i < reps; ++i) GRAM_LAUNCH();
    CK(hipEventRecord(e2)); CK(hipDeviceSynchronize());
    float ms0, ms1; CK(hipEventElapsedTime(&ms0, e0, e1)); CK(hipEventElapsedTime(&ms1, e1, e2));
    // check against the plain f64 sum on sampled entries, relative to max |G|
    double worst = 0, gmax = 0;
    for (int c : {0, nchains - 1}) {
        std::vector<double> P(gbytes / 8);
        CK(hipMemcpy(P.data(), dG[c], gbytes, hipMemcpyDeviceToHost));
        double smax = 0; for (int k = 0; k < q; ++k) smax = std::max(smax, S[c][k]);
        srand(11 + c);
        std::vector<double> errs;
        for (int t = 0; t < 400; ++t) {
            int i = rand() % n, j = rand() % (i + 1);
            if (t < 40) j = i;
            long double ref = 0;
            for (int k = 0; k < q; ++k) if (X[(size_t)i * q + k] && X[(size_t)j * q + k]) ref += S[c][k];
            double got = 0;
            const int ti = i / 64, tj = j / 64, tt = ti * (ti + 1) / 2 + tj;
            for (int ks = 0; ks < ksplit; ++ks) got += P[((size_t)ks * ntl + tt) * 4096 + (j % 64) * 64 + i % 64];
            gmax = std::max(gmax, (double)fabsl(ref));
            errs.push_back(fabs(got - (double)ref));
        }
        for (double e : errs) worst = std::max(worst, e / gmax);
        (void)smax;
    }
    printf("v%d%s EXP=%d n=%d V=%d q=%d  %d chains  L=%d slices  ksplit %d (kchunk %d, padded %d)  S range e^+-%.0f:  digits %.2f us  Gram %.2f us per launch   worst |err| / max|G| = %.2e (bound q 2^(1-7L) = %.2e)\n",
           VER, BAL ? " balanced" : "", EXP, n, V, q, nchains, L, ksplit, kchunk, kcp, srange, ms0 * 1e3 / reps, ms1 * 1e3 / reps, worst, BAL ? 8.0 * q * ldexp(1.0, -8 * L) : q * ldexp(1.0, 1 - 7 * L));
    fflush(stdout);
    CK(hipFree(dXM));
    for (int c = 0; c < nchains; ++c) { CK(hipFree(dS[c])); CK(hipFree(dG[c])); CK(hipFree(dsc[c])); CK(hipFree(dD[c])); }
}

int main()
{
    if (getenv("LAB_V2")) {
        run<8, 0, 1>(500, 100, 8, 7, 3.0); run<8, 0, 2>(500, 100, 8, 7, 3.0); run<7, 0, 2, true>(500, 100, 8, 7, 3.0); run<7, 0, 2, true>(500, 100, 8, 7, 30.0);
        run<7, 1, 2, true>(500, 100, 8, 7, 3.0); run<7, 2, 2, true>(500, 100, 8, 7, 3.0); run<7, 4, 2, true>(500, 100, 8, 7, 3.0);
        run<8, 0, 1>(500, 100, 1, 7, 3.0); run<7, 0, 2, true>(500, 100, 1, 7, 3.0);
        run<9, 0, 1>(500, 300, 8, 48, 3.0); run<8, 0, 2, true>(500, 300, 8, 48, 3.0); run<8, 0, 2, true>(500, 300, 1, 48, 3.0);
        run<7, 0, 2, true>(70, 19, 8, 1, 3.0); run<7, 0, 2, true>(130, 40, 3, 2, 3.0);
        return 0;
    }
    if (getenv("LAB_ABLATE")) {
        run<8, 0>(500, 100, 8, 7, 3.0); run<8, 1>(500, 100, 8, 7, 3.0); run<8, 2>(500, 100, 8, 7, 3.0); run<8, 4>(500, 100, 8, 7, 3.0); run<8, 7>(500, 100, 8, 7, 3.0);
        run<9, 0>(500, 300, 8, 48, 3.0); run<9, 1>(500, 300, 8, 48, 3.0); run<9, 2>(500, 300, 8, 48, 3.0); run<9, 7>(500, 300, 8, 48, 3.0);
        return 0;
    }
    for (int ks : {1, 2, 3, 4, 7}) run<8>(500, 100, 8, ks, 3.0);
    for (int ks : {1, 2, 3, 7}) run<8>(500, 100, 1, ks, 3.0);
    for (int ks : {4, 8, 16, 48}) run<9>(500, 300, 8, ks, 3.0);
    for (int ks : {4, 8, 16, 48}) run<9>(500, 300, 1, ks, 3.0);
    run<8>(500, 100, 8, 2, 30.0);
    run<8>(70, 19, 8, 1, 3.0);
    return 0;
}

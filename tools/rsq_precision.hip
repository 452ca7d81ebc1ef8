// Accuracy of v_rsq_f64 on gfx950 and of one / two Newton steps on top of it (decides how many steps the pivot chain needs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *o, double *o3, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    o[3 * i] = y;
    double e = fma(-v * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    o[3 * i + 1] = y;
    e = fma(-v * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    o[3 * i + 2] = y;
    // one third-order step: y0 (1 + e/2 + 3 e^2/8), e = 1 - x y0^2
    double y0 = __builtin_amdgcn_rsq(v);
    double e3 = fma(-(v * y0), y0, 1.0);
    double p3 = fma(0.375, e3, 0.5);
    o3[i] = fma(y0 * e3, p3, y0);
}
int main()
{
    const int n = 1 << 20;
    double *hx = new double[n], *ho = new double[3 * n], *h3 = new double[n], *dx, *dout, *d3;
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); hx[i] = ldexp(1.0 + u, (int)(s % 41) - 20); }
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dout, 3 * n * 8); (void)hipMalloc(&d3, n * 8);
    (void)hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, d3, n);
    (void)hipMemcpy(ho, dout, 3 * n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h3, d3, n * 8, hipMemcpyDeviceToHost);
    double m[3] = {0, 0, 0}, m3 = 0;
    for (int i = 0; i < n; ++i) {
        long double ref = 1.0L / sqrtl((long double)hx[i]);
        { double r = fabs((double)(((long double)h3[i] - ref) / ref)); if (r > m3) m3 = r; }
        for (int j = 0; j < 3; ++j) { double r = fabs((double)(((long double)ho[3 * i + j] - ref) / ref)); if (r > m[j]) m[j] = r; }
    }
    printf("max relative error over %d inputs: v_rsq_f64 %.3e (2^%.1f), + 1 Newton step %.3e (%.2f ulp), + 2 steps %.3e (%.2f ulp)\n", n,
           m[0], log2(m[0]), m[1], m[1] / 1.11e-16, m[2], m[2] / 1.11e-16);
    printf("one third-order step: %.3e (%.2f ulp)\n", m3, m3 / 1.11e-16);
    return 0;
}

// Accuracy of v_rcp_f64 on gfx950 and of the refinements built on it (kernels.h: rcp_nr), over 4M doubles
// with exponents -200..200, against long double; and of the reciprocals that two or four denominators in
// [1, 2^220] -- the range BOUNDS_FAST admits -- take from ONE rcp_nr of their product (kernels.h: rcp_joint, the
// pair of frequencies 2k, 2k+1 of ColeCole / Shin / Dias).
//   hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -o rcp_accuracy rcp_accuracy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, double *r3, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i];
    double r = __builtin_amdgcn_rcp(a);
    r0[i] = r;
    double e = fma(-a, r, 1.0);
    r = fma(r, e, r);
    r1[i] = r;
    e = fma(-a, r, 1.0);
    r = fma(r, e, r);
    r2[i] = r;
    // one cubic step: 1/a = r0 (1 + e + e^2 + ...), e = 1 - a r0
    const double r00 = r0[i];
    const double e0 = fma(-a, r00, 1.0);
    r3[i] = fma(r00, fma(e0, e0, e0), r00);
}
__device__ double rcp_nr(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}
// x: n groups of four; out2: the first two of each group by the pair form, out4: all four by the tree
__global__ void k_joint(const double *x, double *out2, double *out4, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x0 = x[4 * i], x1 = x[4 * i + 1], x2 = x[4 * i + 2], x3 = x[4 * i + 3];
    const double t2 = rcp_nr(x0 * x1);
    out2[2 * i] = t2 * x1;
    out2[2 * i + 1] = t2 * x0;
    const double p01 = x0 * x1, p23 = x2 * x3;
    const double t = rcp_nr(p01 * p23);
    const double i01 = t * p23, i23 = t * p01;
    out4[4 * i] = i01 * x1;
    out4[4 * i + 1] = i01 * x0;
    out4[4 * i + 2] = i23 * x3;
    out4[4 * i + 3] = i23 * x2;
}
static void joint()
{
    const int n = 1 << 20;
    std::vector<double> x(4 * n), o2(2 * n), o4(4 * n);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 4 * n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (double)(s >> 11) / 9007199254740992.0;
        x[i] = std::ldexp(1.0 + u, (int)(s % 221));      // [1, 2^221): products of four up to 2^884
    }
    double *dx, *d2, *d4;
    hipMalloc(&dx, 4 * n * 8); hipMalloc(&d2, 2 * n * 8); hipMalloc(&d4, 4 * n * 8);
    hipMemcpy(dx, x.data(), 4 * n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_joint, dim3(n / 256), dim3(256), 0, 0, dx, d2, d4, n);
    hipMemcpy(o2.data(), d2, 2 * n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o4.data(), d4, 4 * n * 8, hipMemcpyDeviceToHost);
    double m2 = 0, m4 = 0;
    for (int i = 0; i < n; ++i)
        for (int q = 0; q < 4; ++q) {
            long double t = 1.0L / (long double)x[4 * i + q];
            if (q < 2) m2 = fmax(m2, (double)fabsl(((long double)o2[2 * i + q] - t) / t));
            m4 = fmax(m4, (double)fabsl(((long double)o4[4 * i + q] - t) / t));
        }
    printf("one rcp_nr for TWO denominators in [1, 2^221): max rel err %.3e (%.2f ulp); for FOUR, as the tree "
           "(x0 x1)(x2 x3): %.3e (%.2f ulp)  [4M values, against long double]\n", m2, m2 / 1.11e-16, m4, m4 / 1.11e-16);
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n), d(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (double)(s >> 11) / 9007199254740992.0;
        x[i] = std::ldexp(1.0 + u, (int)(s % 400) - 200);
    }
    double *dx, *d0, *d1, *d2, *d3;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), d3, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        m0 = fmax(m0, (double)fabsl(((long double)a[i] - t) / t));
        m1 = fmax(m1, (double)fabsl(((long double)b[i] - t) / t));
        m2 = fmax(m2, (double)fabsl(((long double)c[i] - t) / t));
        m3 = fmax(m3, (double)fabsl(((long double)d[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err %.3e (2^%.1f); after one Newton step %.3e (%.2f ulp); after two %.3e (%.2f ulp); "
           "after ONE cubic step r (1 + e + e^2) %.3e (%.2f ulp)\n",
           m0, log2(m0), m1, m1 / 1.11e-16, m2, m2 / 1.11e-16, m3, m3 / 1.11e-16);
    joint();
    return 0;
}

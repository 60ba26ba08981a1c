// Accuracy of v_rcp_f64 on gfx950 and of the refinements built on it (kernels.h: rcp_nr), over 4M doubles
// with exponents -200..200, against long double.
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
    return 0;
}

// host_precompute.cpp -- see host_precompute.h.  Plain C++ (no HIP); runs once per
// context, never per walker.
#include "host_precompute.h"

#include <cmath>
#include <cstddef>

namespace bisip {

typedef long double ld;

static const ld PI_L = 3.141592653589793238462643383279502884L;

double loglike_const(int n2, const double *zn_err)
{
    ld s = 0;
    for (int i = 0; i < n2; ++i) {
        ld v = (ld)zn_err[i] * (ld)zn_err[i];
        s += logl(v);
    }
    return (double)(-s);
}

void common_operands(int N, const double *w, const double *zn_err, std::vector<double> &lnw,
                     std::vector<double> &inv_var)
{
    lnw.resize(N);
    inv_var.resize(2 * (size_t)N);
    for (int j = 0; j < N; ++j) lnw[j] = (double)logl((ld)w[j]);
    for (int i = 0; i < 2 * N; ++i) {
        ld v = (ld)zn_err[i] * (ld)zn_err[i];
        inv_var[i] = (double)(1.0L / v);
    }
}

void polydecomp_operands(int N, const double *w, int S, const double *taus, int D,
                         const double *log_taus, double c_exp, const double *zn,
                         const double *zn_err, PolyDecompOperands &o)
{
    o.N = N; o.S = S; o.D = D;
    std::vector<ld> Kr((size_t)N * S), Ki((size_t)N * S);
    // (i*w*tau)^c = (w*tau)^c * (cos(c*pi/2) + i sin(c*pi/2)); the base is purely
    // imaginary and positive so its argument is exactly pi/2.
    const ld ang = (ld)c_exp * PI_L / 2;
    const ld ca = cosl(ang), sa = sinl(ang);
    for (int j = 0; j < N; ++j)
        for (int k = 0; k < S; ++k) {
            ld x = powl((ld)w[j] * (ld)taus[k], (ld)c_exp);
            ld xr = x * ca, xi = x * sa;
            ld dr = 1 + xr;
            ld den = dr * dr + xi * xi;
            // 1 - 1/(1+x) = x/(1+x) = x*conj(1+x)/|1+x|^2
            Kr[(size_t)j * S + k] = (xr * dr + xi * xi) / den;
            Ki[(size_t)j * S + k] = xi / den;
        }
    o.K_re.resize((size_t)N * S); o.K_im.resize((size_t)N * S);
    for (size_t i = 0; i < (size_t)N * S; ++i) { o.K_re[i] = (double)Kr[i]; o.K_im[i] = (double)Ki[i]; }

    std::vector<ld> Gr((size_t)N * D), Gi((size_t)N * D);
    for (int j = 0; j < N; ++j)
        for (int p = 0; p < D; ++p) {
            ld sr = 0, si = 0;
            for (int k = 0; k < S; ++k) {
                ld L = (ld)log_taus[(size_t)p * S + k];
                sr += L * Kr[(size_t)j * S + k];
                si += L * Ki[(size_t)j * S + k];
            }
            Gr[(size_t)j * D + p] = sr;
            Gi[(size_t)j * D + p] = si;
        }
    o.G_re.resize((size_t)N * D); o.G_im.resize((size_t)N * D);
    for (size_t i = 0; i < (size_t)N * D; ++i) { o.G_re[i] = (double)Gr[i]; o.G_im[i] = (double)Gi[i]; }

    // Weighted design matrix of the linear model Z = R0 - sum_p (R0 a_p) G_p, built from
    // the ROUNDED G (what the collapsed kernel uses), rows = (real j..., imag j...).
    const int n = D + 1, m = 2 * N;
    std::vector<ld> A((size_t)m * n), y(m);
    for (int i = 0; i < m; ++i) {
        const int j = i % N;
        const bool im = i >= N;
        ld s = 1.0L / (ld)zn_err[i];
        y[i] = (ld)zn[i] * s;
        A[(size_t)i * n + 0] = im ? 0.0L : s;
        for (int p = 0; p < D; ++p)
            A[(size_t)i * n + 1 + p] = -s * (ld)(im ? o.G_im[(size_t)j * D + p] : o.G_re[(size_t)j * D + p]);
    }
    // Householder QR, applied to y as well (column-wise backward stable).
    const int steps = n < m ? n : m;
    for (int c = 0; c < steps; ++c) {
        ld nrm = 0;
        for (int i = c; i < m; ++i) nrm += A[(size_t)i * n + c] * A[(size_t)i * n + c];
        nrm = sqrtl(nrm);
        if (nrm == 0) continue;
        ld alpha = A[(size_t)c * n + c] > 0 ? -nrm : nrm;
        std::vector<ld> v(m - c);
        for (int i = c; i < m; ++i) v[i - c] = A[(size_t)i * n + c];
        v[0] -= alpha;
        ld vv = 0;
        for (int i = 0; i < m - c; ++i) vv += v[i] * v[i];
        if (vv == 0) continue;
        for (int cc = c; cc < n; ++cc) {
            ld dot = 0;
            for (int i = c; i < m; ++i) dot += v[i - c] * A[(size_t)i * n + cc];
            ld f = 2 * dot / vv;
            for (int i = c; i < m; ++i) A[(size_t)i * n + cc] -= f * v[i - c];
        }
        ld dot = 0;
        for (int i = c; i < m; ++i) dot += v[i - c] * y[i];
        ld f = 2 * dot / vv;
        for (int i = c; i < m; ++i) y[i] -= f * v[i - c];
    }
    o.R.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n && i < m; ++i)
        for (int j = i; j < n; ++j) o.R[(size_t)i * n + j] = (double)A[(size_t)i * n + j];
    ld rest = 0;
    for (int i = n; i < m; ++i) rest += y[i] * y[i];
    o.rest = (double)rest;
    // Any bhat gives an exact identity once e = c - R_d*bhat_d is carried; take the
    // least-squares solution where the triangle is well conditioned, 0 elsewhere.
    std::vector<ld> bh(n, 0.0L);
    ld rmax = 0;
    for (int i = 0; i < n && i < m; ++i) { ld a = fabsl((ld)o.R[(size_t)i * n + i]); if (a > rmax) rmax = a; }
    for (int i = (n < m ? n : m) - 1; i >= 0; --i) {
        ld rii = (ld)o.R[(size_t)i * n + i];
        if (fabsl(rii) <= 1e-13L * rmax) { bh[i] = 0; continue; }
        ld s = y[i];
        for (int j = i + 1; j < n; ++j) s -= (ld)o.R[(size_t)i * n + j] * bh[j];
        bh[i] = s / rii;
    }
    o.bhat.resize(n);
    for (int i = 0; i < n; ++i) o.bhat[i] = (double)bh[i];
    o.e.assign(n, 0.0);
    for (int i = 0; i < n && i < m; ++i) {
        ld s = y[i];
        for (int j = i; j < n; ++j) s -= (ld)o.R[(size_t)i * n + j] * (ld)o.bhat[j];
        o.e[i] = (double)s;
    }
    o.qty.assign(n, 0.0L);
    for (int i = 0; i < n && i < m; ++i) o.qty[i] = y[i];
    o.bhat_ls = bh;
}

void reduced_center(int n, const std::vector<double> &R, const std::vector<long double> &qty,
                    const std::vector<long double> &bhat_ls, const double *b_lo, const double *b_hi,
                    double *out_bhat, double *out_e)
{
    // Is the least-squares solution where the walkers can be (within the box inflated 2x about
    // its centre; |b| <= 1e3 for a component without finite limits)?
    bool use_ls = true;
    for (int j = 0; j < n; ++j) {
        const ld lo = (ld)b_lo[j], hi = (ld)b_hi[j], b = bhat_ls[j];
        if (std::isfinite(b_lo[j]) && std::isfinite(b_hi[j])) {
            const ld c = 0.5L * (lo + hi), h = 0.5L * (hi - lo);
            if (!(fabsl(b - c) <= 2.0L * h)) use_ls = false;
        } else if (!(fabsl(b) <= 1e3L)) {
            use_ls = false;
        }
    }
    for (int j = 0; j < n; ++j) {
        ld b = bhat_ls[j];
        if (!use_ls) b = (std::isfinite(b_lo[j]) && std::isfinite(b_hi[j])) ? 0.5L * ((ld)b_lo[j] + (ld)b_hi[j]) : 0.0L;
        out_bhat[j] = (double)b;
    }
    for (int i = 0; i < n; ++i) {
        ld s = qty[i];
        for (int j = i; j < n; ++j) s -= (ld)R[(size_t)i * n + j] * (ld)out_bhat[j];
        out_e[i] = (double)s;
    }
}

}  // namespace bisip

// host_precompute.cpp -- see host_precompute.h.  Plain C++ (no HIP); runs once per
// context, never per walker.
#include "host_precompute.h"

#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <thread>

namespace bisip {

typedef long double ld;
constexpr int BISIP_HOST_MAXN = 32;    // unknowns of the reduced form: poly_deg + 2 <= 12

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
    polydecomp_kernel_sums(N, w, S, taus, D, log_taus, c_exp, o);
    polydecomp_reduce(zn, zn_err, o);
}

void polydecomp_kernel_sums(int N, const double *w, int S, const double *taus, int D,
                            const double *log_taus, double c_exp, PolyDecompOperands &o)
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
    o.Gl_re = Gr;
    o.Gl_im = Gi;
}

void polydecomp_reduce(const double *zn, const double *zn_err, PolyDecompOperands &o)
{
    const int N = o.N, D = o.D;
    // Weighted design matrix of the linear model Z = R0 - sum_p (R0 a_p) G_p, rows = (real j..., imag
    // j...), built from the UNROUNDED kernel sums: the reduced form stands for the reference's formula,
    // not for the per-frequency kernel's rounded operands (rounding G to double moves a log-probability
    // on the shell logp = 0 of a degree-9 design by 1e-9, like rounding R does).
    const int n = D + 1, m = 2 * N;
    std::vector<ld> A((size_t)m * n), y(m);
    for (int i = 0; i < m; ++i) {
        const int j = i % N;
        const bool im = i >= N;
        ld s = 1.0L / (ld)zn_err[i];
        y[i] = (ld)zn[i] * s;
        A[(size_t)i * n + 0] = im ? 0.0L : s;
        for (int p = 0; p < D; ++p)
            A[(size_t)i * n + 1 + p] = -s * (im ? o.Gl_im[(size_t)j * D + p] : o.Gl_re[(size_t)j * D + p]);
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
    o.Rl.assign((size_t)n * n, 0.0L);
    for (int i = 0; i < n && i < m; ++i)
        for (int j = i; j < n; ++j) {
            o.Rl[(size_t)i * n + j] = A[(size_t)i * n + j];
            o.R[(size_t)i * n + j] = (double)A[(size_t)i * n + j];
        }
    ld rest = 0;
    for (int i = n; i < m; ++i) rest += y[i] * y[i];
    o.rest = (double)rest;
    // Any bhat gives an exact identity once e = c - R_d*bhat_d is carried; take the
    // least-squares solution where the triangle is well conditioned, 0 elsewhere.
    std::vector<ld> bh(n, 0.0L);
    ld rmax = 0;
    for (int i = 0; i < n && i < m; ++i) { ld a = fabsl(o.Rl[(size_t)i * n + i]); if (a > rmax) rmax = a; }
    for (int i = (n < m ? n : m) - 1; i >= 0; --i) {
        ld rii = o.Rl[(size_t)i * n + i];
        if (fabsl(rii) <= 1e-13L * rmax) { bh[i] = 0; continue; }
        ld s = y[i];
        for (int j = i + 1; j < n; ++j) s -= o.Rl[(size_t)i * n + j] * bh[j];
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

namespace {

// logprob_row_reduced<P, COMP> (kernels.h) in the same double arithmetic, operation for operation
double reduced_chi2_double(int n, const std::vector<double> &R, const std::vector<ld> &Rl, const double *bhat,
                           const double *e, const double *elo, double rest, const double *th, bool comp)
{
    double chi2 = rest;
    std::vector<double> d(n), dl(n);
    auto two_diff = [](double a, double b, double &s, double &err) {
        s = a - b;
        const double bb = s - a;
        err = (a - (s - bb)) + ((-b) - bb);
    };
    if (!comp) {
        d[0] = bhat[0] - th[0];
        for (int q = 1; q < n; ++q) {
            const double prod = th[0] * th[q];
            d[q] = bhat[q] - prod;
        }
        for (int i = 0; i < n; ++i) {
            double u = e[i];
            for (int j = i; j < n; ++j) u = std::fma(R[(size_t)i * n + j], d[j], u);
            chi2 = std::fma(u, u, chi2);
        }
        return chi2;
    }
    two_diff(bhat[0], th[0], d[0], dl[0]);
    for (int q = 1; q < n; ++q) {
        const double p = th[0] * th[q];
        const double pe = std::fma(th[0], th[q], -p);
        double err;
        two_diff(bhat[q], p, d[q], err);
        dl[q] = err - pe;
    }
    for (int i = 0; i < n; ++i) {
        double s = e[i], c = elo[i];
        for (int j = i; j < n; ++j) {
            const double Rk = R[(size_t)i * n + j];
            const double h = Rk * d[j];
            const double l = std::fma(Rk, d[j], -h);
            const double t = s + h;
            const double bb = t - s;
            const double er = (s - (t - bb)) + (h - bb);
            s = t;
            c += er + l;
            c = std::fma(Rk, dl[j], c);
            c = std::fma((double)(float)(Rl[(size_t)i * n + j] - (ld)Rk), d[j], c);     // Rlo, as the kernel holds it (a float: <= 11 bits)
        }
        const double u = s + c;
        chi2 = std::fma(u, u, chi2);
    }
    return chi2;
}

// the same quantity from the unrounded operands, to about twice the precision of long double: every row
// qty_i - sum_j R_ij b_j is accumulated as an unevaluated sum of two long doubles (TwoProduct by fmal,
// TwoSum; Ogita-Rump-Oishi Dot2, b_j = R0 a_j split the same way).  Plain long double is NOT enough to
// judge the compensated kernel: on degree 9-10 designs the terms reach 1e8 against a row sum of ~10 and
// a 64-bit mantissa leaves 1e-10 of error in a log-probability on the shell logp = 0 -- more than the
// kernel being judged (whose double-double rows are good to ~1e-30).
static inline void two_sum(ld a, ld b, ld &s, ld &err)
{
    s = a + b;
    const ld bb = s - a;
    err = (a - (s - bb)) + (b - bb);
}

// a*b = p + err exactly, by Dekker's splitting of the 64-bit mantissas into 32 + 32 bits (glibc's fmal is a
// software routine on x86: ~350 ns a call, which made a context's estimate take 60 ms)
static inline void split32(ld a, ld &hi, ld &lo)
{
    const ld c = 4294967297.0L * a;      // 2^32 + 1
    hi = c - (c - a);
    lo = a - hi;
}

static inline void two_prod(ld a, ld b, ld &p, ld &err)
{
    p = a * b;
    ld ah, al, bh, bl;
    split32(a, ah, al);
    split32(b, bh, bl);
    err = ((ah * bh - p) + ah * bl + al * bh) + al * bl;
}

ld reduced_chi2_exact(int n, const std::vector<ld> &R, const std::vector<ld> &qty, double rest,
                      const double *th)
{
    ld chi2 = rest;
    // b = b_h + b_l exactly: R0 and a_j are doubles, their product two doubles (one hardware fma)
    ld bh[BISIP_HOST_MAXN], bl[BISIP_HOST_MAXN];
    bh[0] = (ld)th[0];
    bl[0] = 0.0L;
    for (int j = 1; j < n; ++j) {
        const double p = th[0] * th[j];
        bh[j] = (ld)p;
        bl[j] = (ld)std::fma(th[0], th[j], -p);
    }
    for (int i = 0; i < n; ++i) {
        ld hi = qty[i], lo = 0.0L;
        for (int j = i; j < n; ++j) {
            const ld r = -R[(size_t)i * n + j];
            ld p, pe, s, se;
            two_prod(r, bh[j], p, pe);
            two_sum(hi, p, s, se);
            hi = s;
            lo += se + pe + r * bl[j];
        }
        const ld u = hi + lo;
        chi2 += u * u;
    }
    return chi2;
}

struct Lcg {   // deterministic probe points, no <random>
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    double uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }
    double sym() { return 2.0 * uni() - 1.0; }
};

}  // namespace

// Weight of the shell probes in the estimate (see reduced_center).  BISIP_SHELL_WEIGHT=0 reproduces the
// round-2 estimate that never looked at the shell: tests use it to build a context whose estimate passes
// and whose batch then does not (the guard of bisip_logprob).  Read when a context is created.
static double shell_weight()
{
    const char *s = std::getenv("BISIP_SHELL_WEIGHT");
    const double w = s ? std::atof(s) : 0.05;
    return w >= 0.0 && w <= 1.0 ? w : 0.05;
}

double reduced_center(int n, const std::vector<double> &R, const std::vector<long double> &Rl,
                      const std::vector<long double> &qty,
                      const std::vector<long double> &bhat_ls, double rest, double lconst,
                      const double *lo, const double *hi, bool comp, double *out_bhat, double *out_e,
                      double *out_elo)
{
    bool finite_box = true;
    for (int j = 0; j < n; ++j) finite_box = finite_box && std::isfinite(lo[j]) && std::isfinite(hi[j]);
    // image of the theta box under b = R0 * (1, a)
    std::vector<double> blo(n), bhi(n);
    blo[0] = lo[0]; bhi[0] = hi[0];
    for (int j = 1; j < n; ++j) {
        const double c[4] = {lo[0] * lo[j], lo[0] * hi[j], hi[0] * lo[j], hi[0] * hi[j]};
        double a = INFINITY, b = -INFINITY;
        bool nan = false;
        for (double v : c) { if (v != v) nan = true; a = v < a ? v : a; b = v > b ? v : b; }
        if (nan || !(a <= b)) { a = -INFINITY; b = INFINITY; }
        blo[j] = a; bhi[j] = b;
    }
    // probe rows: where walkers are going to be evaluated
    std::vector<std::vector<double>> probes;
    Lcg rng;
    auto inside = [&](const std::vector<double> &t) {
        for (int j = 0; j < n; ++j) if (!(lo[j] < t[j] && t[j] < hi[j])) return false;
        return true;
    };
    const bool ls_ok = std::isfinite((double)bhat_ls[0]) && bhat_ls[0] != 0.0L;
    if (finite_box) {
        for (int k = 0; k < 32; ++k) {   // uniform in the prior box
            std::vector<double> t(n);
            for (int j = 0; j < n; ++j) t[j] = lo[j] + (hi[j] - lo[j]) * rng.uni();
            probes.push_back(t);
        }
        for (int k = 0; k < 16; ++k) {   // small coefficients: where a decent fit usually lies
            std::vector<double> t(n);
            t[0] = lo[0] + (hi[0] - lo[0]) * rng.uni();
            for (int j = 1; j < n; ++j) {
                double v = 1e-3 * rng.sym();
                v = v < lo[j] ? lo[j] : (v > hi[j] ? hi[j] : v);
                t[j] = v;
            }
            if (inside(t)) probes.push_back(t);
        }
    }
    if (ls_ok)
        for (int k = 0; k < 16; ++k) {   // around the least-squares solution (the posterior mode)
            const double s = k < 8 ? 1e-3 : 1e-2;
            std::vector<double> t(n);
            t[0] = (double)bhat_ls[0] * (1.0 + s * rng.sym());
            bool fin = std::isfinite(t[0]);
            for (int j = 1; j < n; ++j) {
                t[j] = (double)(bhat_ls[j] / bhat_ls[0]) * (1.0 + s * rng.sym());
                fin = fin && std::isfinite(t[j]);
            }
            if (fin && (!finite_box || inside(t))) probes.push_back(t);
        }
    // theta of b = b_ls + s R^-1 z for a random direction z (unit variance per component: four uniforms; the
    // shape of the distribution does not matter), s = `scale`, or, for scale < 0, such that the row lies ON
    // the shell rest + s^2 |z|^2 = 2 lconst.  Plain double: where the probe lies need not be exact.  False
    // when the row is not finite or not inside the box.
    bool solvable = ls_ok;
    for (int i = 0; i < n; ++i) solvable = solvable && R[(size_t)i * n + i] != 0.0;
    double bls_d[BISIP_HOST_MAXN];
    for (int j = 0; j < n; ++j) bls_d[j] = (double)bhat_ls[j];
    std::vector<double> tpt(n);
    auto valley_point = [&](double scale, std::vector<double> &t) {
        double z[BISIP_HOST_MAXN], db[BISIP_HOST_MAXN], zz = 0.0;
        for (int j = 0; j < n; ++j) {
            z[j] = 1.7320508075688772 * (rng.uni() + rng.uni() + rng.uni() + rng.uni() - 2.0);
            zz += z[j] * z[j];
        }
        if (!(zz > 0.0)) return false;
        const double sc = scale >= 0.0 ? scale : std::sqrt((2.0 * lconst - rest) / zz);
        for (int i = n - 1; i >= 0; --i) {
            double acc = sc * z[i];
            for (int j = i + 1; j < n; ++j) acc -= R[(size_t)i * n + j] * db[j];
            db[i] = acc / R[(size_t)i * n + i];
        }
        const double b0 = bls_d[0] + db[0];
        t[0] = b0;
        if (!std::isfinite(b0) || b0 == 0.0) return false;
        for (int j = 1; j < n; ++j) {
            t[j] = (bls_d[j] + db[j]) / b0;
            if (!std::isfinite(t[j])) return false;
        }
        return !finite_box || inside(t);
    };
    if (ls_ok) {
        // where an ensemble sampler's walkers actually are: draws from the Gaussian posterior of the
        // linear model, b = b_ls + R^-1 z with z ~ N(0, I), and 3x, 10x and 30x wider: a converged
        // ensemble, and the same ensemble on its way in during burn-in.  On nearly collinear
        // designs these spread far along the flat directions of chi^2 -- the rows of R (bhat - b)
        // then cancel by many orders of magnitude although chi^2 stays within a few units (a few
        // hundred, at 10-30 sigma) of its minimum.  Round 2 probed 1x and 3x only: a NumPy emulation of
        // the plain kernel on 10-sigma rows of degree 8-10 designs it had passed read 1e-10 ... 2e-8.
        // 16 probes INSIDE the box per scale (a narrowed box keeps few of the draws: up to 300 tries each)
        for (int scale_i = 0; solvable && scale_i < 4; ++scale_i) {
            const double sc = scale_i == 0 ? 1.0 : (scale_i == 1 ? 3.0 : (scale_i == 2 ? 10.0 : 30.0));
            int kept = 0;
            for (int k = 0; k < 300 && kept < 16; ++k)
                if (valley_point(sc, tpt)) { probes.push_back(tpt); ++kept; }
        }
    }
    // The shell log-probability = 0.  The parity tolerance is |d logp| <= 1e-10 max(1, |logp|): where the
    // log-probability crosses zero -- chi^2 = 2 lconst, ten or so posterior sigmas out on a typical
    // spectrum, where burn-in passes -- the denominator is 1 and the ABSOLUTE error of a chi^2 of several
    // hundred counts.  Random valley rows land within |logp| < 1 one time in a few hundred, so the probes
    // above practically never see it (measured on the GPU, benchmarks/valley_rows.py: 3000 rows per scale
    // found 1e-11 ... 2.5e-10 on degree 7-9 designs whose 32 probes per scale had read < 1e-12).  These
    // probes are ON the shell: b = b_ls + s R^-1 z with s such that rest + s^2 |z|^2 = 2 lconst.  No double
    // formulation gets below ~1e-12 there (one rounding of chi^2 ~ 1e3 is 1e-13; the plain triangle of the
    // headline's degree-5 design reads 9e-12, of which 5e-12 is the rounding of R itself), so they count
    // at a twentieth: the gate 1e-12 then reads "2e-11 on the shell" -- a fifth of the tolerance, where
    // the region is sampled directly -- the same bar bisip_logprob's guard applies to real batches.
    size_t n_regular = probes.size();
    if (ls_ok && 2.0 * lconst - rest > 0.0) {
        int kept = 0;     // 64 probes inside the box, up to 1500 tries (a narrowed box keeps few of the draws)
        for (int k = 0; solvable && k < 1500 && kept < 64; ++k)
            if (valley_point(-1.0, tpt)) { probes.push_back(tpt); ++kept; }
        // A box that cuts the shell in a small patch keeps none of those draws (a degree-6 design of the
        // fuzz campaign: 9 of 200,000; its estimate then never saw the shell and passed a kernel that read
        // 1.1e-10 on the one row of a 20,000-row batch that lay there).  The box is convex in theta: between
        // a probe with log-probability > 0 and one with < 0, both inside, the segment stays inside and
        // crosses the shell -- found by bisection (plain double: where the probe lies need not be exact).
        if (kept < 64 && finite_box) {
            auto logp_of = [&](const std::vector<double> &t) {
                double chi2 = rest;
                for (int i = 0; i < n; ++i) {
                    double u = (double)qty[i];
                    for (int j = i; j < n; ++j) u -= R[(size_t)i * n + j] * (j ? t[0] * t[j] : t[0]);
                    chi2 += u * u;
                }
                return lconst - 0.5 * chi2;
            };
            std::vector<size_t> pos, neg;
            std::vector<std::vector<double>> ends(probes.begin(), probes.begin() + (long)n_regular);
            for (int k = 0; k < 32; ++k) {       // more of the box, as far ends of the segments only
                std::vector<double> t(n);
                for (int j = 0; j < n; ++j) t[j] = lo[j] + (hi[j] - lo[j]) * rng.uni();
                ends.push_back(t);
            }
            for (size_t ip = 0; ip < ends.size(); ++ip) {
                if (!inside(ends[ip])) continue;
                const double lp = logp_of(ends[ip]);
                if (lp > 0.0) pos.push_back(ip);
                else if (lp < 0.0) neg.push_back(ip);
            }
            for (int k = 0; !pos.empty() && !neg.empty() && kept < 64 && k < 128; ++k) {
                const std::vector<double> &a = ends[pos[(size_t)(rng.uni() * (double)pos.size()) % pos.size()]];
                const std::vector<double> &b = ends[neg[(size_t)(rng.uni() * (double)neg.size()) % neg.size()]];
                double t0 = 0.0, t1 = 1.0;
                for (int it = 0; it < 60; ++it) {
                    const double tm = 0.5 * (t0 + t1);
                    for (int j = 0; j < n; ++j) tpt[j] = a[j] + tm * (b[j] - a[j]);
                    if (logp_of(tpt) > 0.0) t0 = tm; else t1 = tm;
                }
                for (int j = 0; j < n; ++j) tpt[j] = a[j] + t0 * (b[j] - a[j]);
                if (inside(tpt) && std::fabs(logp_of(tpt)) < 1.0) { probes.push_back(tpt); ++kept; }
            }
        }
    }
    // candidates for the expansion point
    std::vector<std::vector<double>> cand;
    {
        bool ok = true;
        std::vector<double> c(n);
        for (int j = 0; j < n; ++j) { c[j] = (double)bhat_ls[j]; ok = ok && std::isfinite(c[j]) && std::fabs(c[j]) <= 1e6; }
        if (ok) cand.push_back(c);
    }
    if (finite_box) {
        std::vector<double> c(n);
        bool ok = true;
        for (int j = 0; j < n; ++j) { c[j] = 0.5 * (blo[j] + bhi[j]); ok = ok && std::isfinite(c[j]); }
        if (ok) cand.push_back(c);
    }
    cand.push_back(std::vector<double>(n, 0.0));
    double best = INFINITY;
    std::vector<double> e(n), elo(n);
    // The compensated kernel's rows are double-doubles: its estimate reads 1e-14 on every design ever probed
    // (TABLE:auto_by_degree); a third of the probes is plenty to notice if that ever stopped being true.
    if (comp) {
        std::vector<std::vector<double>> some;
        size_t shell_from = 0;
        for (size_t ip = 0; ip < probes.size(); ip += 3) {
            if (ip < n_regular) shell_from = some.size() + 1;
            some.push_back(probes[ip]);
        }
        probes.swap(some);
        n_regular = shell_from;
    }
    std::vector<ld> exact_of(probes.size());
    for (size_t ip = 0; ip < probes.size(); ++ip) exact_of[ip] = reduced_chi2_exact(n, Rl, qty, rest, probes[ip].data());
    if (probes.empty()) {
        // nothing to measure the kernel against (a non-finite box with no usable least-squares
        // solution): expand about zero and report "unknown", so AUTO takes the per-frequency form
        for (int j = 0; j < n; ++j) {
            out_bhat[j] = 0.0;
            out_e[j] = (double)qty[j];
            out_elo[j] = (double)(qty[j] - (ld)out_e[j]);
        }
        return INFINITY;
    }
    for (const auto &c : cand) {
        for (int i = 0; i < n; ++i) {
            // plain: with the triangle the kernel holds, so that its identity is exact for THAT triangle;
            // compensated: with the unrounded one, which R + Rlo stands for
            ld s = qty[i];
            for (int j = i; j < n; ++j) s -= (comp ? Rl[(size_t)i * n + j] : (ld)R[(size_t)i * n + j]) * (ld)c[j];
            e[i] = (double)s;
            elo[i] = (double)(s - (ld)e[i]);
        }
        double worst = 0.0;
        const double w_shell = shell_weight();
        for (size_t ip = 0; ip < probes.size(); ++ip) {
            const auto &t = probes[ip];
            const ld exact = exact_of[ip];
            const double got = reduced_chi2_double(n, R, Rl, c.data(), e.data(), elo.data(), rest, t.data(), comp);
            const ld lp = -0.5L * exact + (ld)lconst;
            const ld scale = fabsl(lp) > 1.0L ? fabsl(lp) : 1.0L;
            double rel = (double)(fabsl(-0.5L * ((ld)got - exact)) / scale);
            if (ip >= n_regular) rel *= w_shell;    // shell probes (above)
            if (!(rel <= worst)) worst = rel;   // NaN counts as worst
        }
        if (worst < best || best == INFINITY) {
            best = worst;
            for (int j = 0; j < n; ++j) { out_bhat[j] = c[j]; out_e[j] = e[j]; out_elo[j] = elo[j]; }
        }
    }
    return best;
}

double reduced_logp_reference(int n, const std::vector<long double> &Rl, const std::vector<long double> &qty,
                              double rest, double lconst, const double *theta)
{
    return (double)(-0.5L * reduced_chi2_exact(n, Rl, qty, rest, theta) + (ld)lconst);
}

bool grid_step(int N, const double *w, const double *lnw, double *dlnw)
{
    *dlnw = 0.0;
    if (N < 8) return false;
    for (int j = 0; j < N; ++j)
        if (!(w[j] > 0.0) || !std::isfinite(w[j])) return false;
    const ld first = logl((ld)w[0]), step = (logl((ld)w[N - 1]) - first) / (ld)(N - 1);
    const double d = (double)step;
    if (!std::isfinite(d) || d == 0.0) return false;
    for (int j = 0; j < N; ++j) {
        const ld want = logl((ld)w[j]), got = (ld)lnw[j & ~7] + (ld)(j & 7) * (ld)d;
        if (!(fabsl(want - got) <= 4e-15L)) return false;
    }
    *dlnw = d;
    return true;
}

int host_threads()
{
    if (const char *env = std::getenv("BISIP_HOST_THREADS")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1) return (int)(v > 256 ? 256 : v);
    }
    long n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n < 1) n = 1;
    // cgroup v2 "quota period" (or "max"), then v1: a container is throttled, not helped, by more
    // runnable threads than its quota
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0 && quota / period < n)
            n = (long)(quota / period < 1 ? 1 : quota / period);
        std::fclose(f);
    } else {
        long long quota = -1, period = 0;
        if (FILE *q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(q, "%lld", &quota) != 1) quota = -1; std::fclose(q); }
        if (FILE *q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(q, "%lld", &period) != 1) period = 0; std::fclose(q); }
        if (quota > 0 && period > 0 && quota / period < n) n = (long)(quota / period < 1 ? 1 : quota / period);
    }
    return (int)(n > 16 ? 16 : n);
}

void parallel_blocks(int64_t n, int64_t min_per_thread, const std::function<void(int64_t, int64_t)> &fn)
{
    if (n <= 0) return;
    if (min_per_thread < 1) min_per_thread = 1;
    int64_t threads = host_threads();
    if (threads > n / min_per_thread) threads = n / min_per_thread;
    if (threads <= 1) { fn(0, n); return; }
    std::vector<std::exception_ptr> errors((size_t)threads);
    auto block = [&](int64_t t) {
        try {
            fn(n * t / threads, n * (t + 1) / threads);
        } catch (...) {
            errors[(size_t)t] = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    for (int64_t t = 1; t < threads; ++t) pool.emplace_back(block, t);
    block(0);
    for (auto &th : pool) th.join();
    for (auto &e : errors)
        if (e) std::rethrow_exception(e);
}

}  // namespace bisip

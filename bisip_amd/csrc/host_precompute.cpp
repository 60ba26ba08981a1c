// host_precompute.cpp -- see host_precompute.h.  Plain C++ (no HIP); runs once per
// context, never per walker.
#include "host_precompute.h"

#include <immintrin.h>
#include <quadmath.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <thread>

namespace bisip {

typedef long double ld;
typedef __float128 qd;
constexpr int BISIP_HOST_MAXN = 32;    // unknowns of the reduced form: poly_deg + 2 <= 12

// ---------------------------------------------------------------------------------------------
// the two working precisions of the precompute: the same algorithms, written once
// ---------------------------------------------------------------------------------------------
namespace {

inline ld x_sqrt(ld x) { return sqrtl(x); }
inline qd x_sqrt(qd x) { return sqrtq(x); }
inline ld x_abs(ld x) { return fabsl(x); }
inline qd x_abs(qd x) { return fabsq(x); }
inline ld x_pow(ld x, ld y) { return powl(x, y); }
inline qd x_pow(qd x, qd y) { return powq(x, y); }
inline void x_sincos_half_pi(ld c, ld &sn, ld &cs)
{
    const ld ang = c * 3.141592653589793238462643383279502884L / 2;
    cs = cosl(ang);
    sn = sinl(ang);
}
inline void x_sincos_half_pi(qd c, qd &sn, qd &cs)
{
    const qd ang = c * M_PIq / 2;
    cs = cosq(ang);
    sn = sinq(ang);
}

// K[j,k] = 1 - 1/(1 + (i w_j tau_k)^c) and G[j,p] = sum_k log_taus[p,k] K[j,k] in the working precision T.
// (i*w*tau)^c = (w*tau)^c * (cos(c*pi/2) + i sin(c*pi/2)); the base is purely imaginary and positive so its
// argument is exactly pi/2.
template <class T>
void kernel_sums_T(int N, const double *w, int S, const double *taus, int D, const double *log_taus, double c_exp,
                   std::vector<T> &Kr, std::vector<T> &Ki, std::vector<T> &Gr, std::vector<T> &Gi, bool threads = false)
{
    Kr.assign((size_t)N * S, T(0));
    Ki.assign((size_t)N * S, T(0));
    T ca, sa;
    x_sincos_half_pi((T)c_exp, sa, ca);
    // long double: (w tau)^c of the product rounded to 64 bits, one power per (j, k) -- as in every earlier round,
    // bit for bit.  binary128: the product of two doubles is exact there, and w^c tau^c needs N + S powers
    // instead of N S (a power in binary128 is a microsecond); c = 1 needs none.
    std::vector<T> wc, tc;
    if constexpr (!std::is_same<T, ld>::value) {
        wc.resize((size_t)N);
        tc.resize((size_t)S);
        for (int j = 0; j < N; ++j) wc[(size_t)j] = c_exp == 1.0 ? (T)w[j] : x_pow((T)w[j], (T)c_exp);
        for (int k = 0; k < S; ++k) tc[(size_t)k] = c_exp == 1.0 ? (T)taus[k] : x_pow((T)taus[k], (T)c_exp);
    }
    Gr.assign((size_t)N * D, T(0));
    Gi.assign((size_t)N * D, T(0));
    // every frequency's row of K and G on its own (binary128 arithmetic is software: the rows of one frequency
    // list are spread over the host threads when the caller has nothing else running on them)
    auto rows = [&](int64_t j_lo, int64_t j_hi) {
        for (int64_t j = j_lo; j < j_hi; ++j) {
            for (int k = 0; k < S; ++k) {
                T x;
                if constexpr (std::is_same<T, ld>::value) x = x_pow((T)w[j] * (T)taus[k], (T)c_exp);
                else x = wc[(size_t)j] * tc[(size_t)k];
                const T xr = x * ca, xi = x * sa;
                const T dr = 1 + xr;
                const T den = dr * dr + xi * xi;
                // 1 - 1/(1+x) = x/(1+x) = x*conj(1+x)/|1+x|^2
                Kr[(size_t)j * S + k] = (xr * dr + xi * xi) / den;
                Ki[(size_t)j * S + k] = xi / den;
            }
            for (int p = 0; p < D; ++p) {
                T sr = 0, si = 0;
                for (int k = 0; k < S; ++k) {
                    const T L = (T)log_taus[(size_t)p * S + k];
                    sr += L * Kr[(size_t)j * S + k];
                    si += L * Ki[(size_t)j * S + k];
                }
                Gr[(size_t)j * D + p] = sr;
                Gi[(size_t)j * D + p] = si;
            }
        }
    };
    if (threads) parallel_blocks(N, 2, rows);
    else rows(0, N);
}

// Weighted design matrix of the linear model Z = R0 - sum_p (R0 a_p) G_p, rows = (real j..., imag j...), built
// from the UNROUNDED kernel sums (the reduced form stands for the reference's formula, not for the per-frequency
// kernel's rounded operands: rounding G to double moves a log-probability on the shell logp = 0 of a degree-9
// design by 1e-9, like rounding R does); Householder QR, applied to y as well (column-wise backward stable);
// the least-squares solution where the triangle is well conditioned, 0 elsewhere.
// Rl: (n,n) upper triangle; qty: first n entries of Q^T y; rest = sum of the squares of the others.
template <class T>
void reduce_T(int N, int D, const std::vector<T> &Gr, const std::vector<T> &Gi, const double *zn, const double *zn_err,
              std::vector<T> &Rl, std::vector<T> &qty, std::vector<T> &bhat_ls, T &rest_out)
{
    const int n = D + 1, m = 2 * N;
    std::vector<T> A((size_t)m * n), y(m), v((size_t)m);
    for (int i = 0; i < m; ++i) {
        const int j = i % N;
        const bool im = i >= N;
        const T s = T(1) / (T)zn_err[i];
        y[i] = (T)zn[i] * s;
        A[(size_t)i * n + 0] = im ? T(0) : s;
        for (int p = 0; p < D; ++p)
            A[(size_t)i * n + 1 + p] = -s * (im ? Gi[(size_t)j * D + p] : Gr[(size_t)j * D + p]);
    }
    const int steps = n < m ? n : m;
    for (int c = 0; c < steps; ++c) {
        T nrm = 0;
        for (int i = c; i < m; ++i) nrm += A[(size_t)i * n + c] * A[(size_t)i * n + c];
        nrm = x_sqrt(nrm);
        if (nrm == 0) continue;
        const T alpha = A[(size_t)c * n + c] > 0 ? -nrm : nrm;
        for (int i = c; i < m; ++i) v[(size_t)(i - c)] = A[(size_t)i * n + c];
        v[0] -= alpha;
        T vv = 0;
        for (int i = 0; i < m - c; ++i) vv += v[(size_t)i] * v[(size_t)i];
        if (vv == 0) continue;
        for (int cc = c; cc < n; ++cc) {
            T dot = 0;
            for (int i = c; i < m; ++i) dot += v[(size_t)(i - c)] * A[(size_t)i * n + cc];
            const T f = 2 * dot / vv;
            for (int i = c; i < m; ++i) A[(size_t)i * n + cc] -= f * v[(size_t)(i - c)];
        }
        T dot = 0;
        for (int i = c; i < m; ++i) dot += v[(size_t)(i - c)] * y[i];
        const T f = 2 * dot / vv;
        for (int i = c; i < m; ++i) y[i] -= f * v[(size_t)(i - c)];
    }
    Rl.assign((size_t)n * n, T(0));
    for (int i = 0; i < n && i < m; ++i)
        for (int j = i; j < n; ++j) Rl[(size_t)i * n + j] = A[(size_t)i * n + j];
    T rest = 0;
    for (int i = n; i < m; ++i) rest += y[i] * y[i];
    rest_out = rest;
    bhat_ls.assign((size_t)n, T(0));
    T rmax = 0;
    for (int i = 0; i < n && i < m; ++i) { const T a = x_abs(Rl[(size_t)i * n + i]); if (a > rmax) rmax = a; }
    for (int i = (n < m ? n : m) - 1; i >= 0; --i) {
        const T rii = Rl[(size_t)i * n + i];
        if (x_abs(rii) <= T(1e-13L) * rmax) { bhat_ls[(size_t)i] = 0; continue; }
        T s = y[i];
        for (int j = i + 1; j < n; ++j) s -= Rl[(size_t)i * n + j] * bhat_ls[(size_t)j];
        bhat_ls[(size_t)i] = s / rii;
    }
    qty.assign((size_t)n, T(0));
    for (int i = 0; i < n && i < m; ++i) qty[(size_t)i] = y[i];
}

}  // namespace

struct QuadKernelSums {
    int N = 0, D = 0;
    std::vector<qd> Gr, Gi;
};

struct QuadReduced {
    std::vector<qd> Rq, qty;     // (n,n), (n,)
    qd rest = 0;
};

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
    std::vector<ld> Kr, Ki;
    kernel_sums_T<ld>(N, w, S, taus, D, log_taus, c_exp, Kr, Ki, o.Gl_re, o.Gl_im);
    o.K_re.resize((size_t)N * S); o.K_im.resize((size_t)N * S);
    for (size_t i = 0; i < (size_t)N * S; ++i) { o.K_re[i] = (double)Kr[i]; o.K_im[i] = (double)Ki[i]; }
    o.G_re.resize((size_t)N * D); o.G_im.resize((size_t)N * D);
    for (size_t i = 0; i < (size_t)N * D; ++i) { o.G_re[i] = (double)o.Gl_re[i]; o.G_im[i] = (double)o.Gl_im[i]; }
}

void polydecomp_reduce(const double *zn, const double *zn_err, PolyDecompOperands &o)
{
    const int N = o.N, D = o.D, n = D + 1, m = 2 * N;
    ld rest = 0;
    reduce_T<ld>(N, D, o.Gl_re, o.Gl_im, zn, zn_err, o.Rl, o.qty, o.bhat_ls, rest);
    o.rest = (double)rest;
    o.R.assign((size_t)n * n, 0.0);
    for (size_t i = 0; i < (size_t)n * n; ++i) o.R[i] = (double)o.Rl[i];
    // Any bhat gives an exact identity once e = c - R_d*bhat_d is carried; the least-squares solution here
    // (reduced_center_* choose the one a context runs with)
    o.bhat.resize(n);
    for (int i = 0; i < n; ++i) o.bhat[i] = (double)o.bhat_ls[i];
    o.e.assign(n, 0.0);
    for (int i = 0; i < n && i < m; ++i) {
        ld s = o.qty[i];
        for (int j = i; j < n; ++j) s -= (ld)o.R[(size_t)i * n + j] * (ld)o.bhat[j];
        o.e[i] = (double)s;
    }
}

void reduced_from_operands(const PolyDecompOperands &o, double lconst, ReducedProblem &p)
{
    p.n = o.D + 1;
    p.R = o.R; p.Rl = o.Rl; p.qty = o.qty; p.bhat_ls = o.bhat_ls;
    p.rest = o.rest;
    p.lconst = lconst;
    p.Rc.clear(); p.Rc_lo.clear(); p.rest_c = 0.0; p.quad.reset();
}

std::shared_ptr<const QuadKernelSums> polydecomp_kernel_sums_quad(int N, const double *w, int S, const double *taus, int D,
                                                                  const double *log_taus, double c_exp, bool threads)
{
    auto ks = std::make_shared<QuadKernelSums>();
    ks->N = N; ks->D = D;
    std::vector<qd> Kr, Ki;
    kernel_sums_T<qd>(N, w, S, taus, D, log_taus, c_exp, Kr, Ki, ks->Gr, ks->Gi, threads);
    return ks;
}

void reduced_make_quad(const QuadKernelSums &ks, const double *zn, const double *zn_err, ReducedProblem &p)
{
    auto q = std::make_shared<QuadReduced>();
    std::vector<qd> bls;
    reduce_T<qd>(ks.N, ks.D, ks.Gr, ks.Gi, zn, zn_err, q->Rq, q->qty, bls, q->rest);
    const int n = ks.D + 1;
    p.n = n;
    p.Rc.assign((size_t)n * n, 0.0);
    p.Rc_lo.assign((size_t)n * n, 0.0f);
    for (size_t i = 0; i < (size_t)n * n; ++i) {
        p.Rc[i] = (double)q->Rq[i];
        p.Rc_lo[i] = (float)(q->Rq[i] - (qd)p.Rc[i]);
    }
    p.rest_c = (double)q->rest;
    p.quad = q;
}

namespace {

// ---------------------------------------------------------------------------------------------
// yardsticks
// ---------------------------------------------------------------------------------------------
// the reduced form from long-double operands, to about twice the precision of long double: every row
// qty_i - sum_j R_ij b_j is accumulated as an unevaluated sum of two long doubles (TwoProduct, TwoSum;
// Ogita-Rump-Oishi Dot2, b_j = R0 a_j split the same way).  Plain long double is NOT enough to
// judge the compensated kernel: on degree 9-10 designs the terms reach 1e8 against a row sum of ~10 and
// a 64-bit mantissa leaves 1e-10 of error in a log-probability on the shell logp = 0.
static inline void two_sum(ld a, ld b, ld &s, ld &err)
{
    s = a + b;
    const ld bb = s - a;
    err = (a - (s - bb)) + (b - bb);
}

// a*b = p + err exactly, by Dekker's splitting of the 64-bit mantissas into 32 + 32 bits (glibc's fmal is a
// software routine on x86: ~350 ns a call)
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
    // b = b_h + b_l exactly: R0 and a_j are doubles, their product two doubles (one fma)
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

// the same from binary128 operands, in binary128: 113 bits against terms 1e8 times the row sums
qd reduced_chi2_quad(int n, const QuadReduced &q, const double *th)
{
    qd b[BISIP_HOST_MAXN];
    b[0] = (qd)th[0];
    for (int j = 1; j < n; ++j) b[j] = (qd)th[0] * (qd)th[j];      // exact: 106 bits
    qd chi2 = q.rest;
    for (int i = 0; i < n; ++i) {
        qd u = q.qty[(size_t)i];
        for (int j = i; j < n; ++j) u -= q.Rq[(size_t)i * n + j] * b[j];
        chi2 += u * u;
    }
    return chi2;
}

// the kernels' arithmetic and the plain tier's yardstick, as plain x86-64 code and with hardware fma
#define EMU(name) name##_base
#define EMU_ATTR
#include "host_emulate.inc"
#undef EMU
#undef EMU_ATTR
#define EMU(name) name##_fma
#define EMU_ATTR __attribute__((target("fma")))
#include "host_emulate.inc"
#undef EMU
#undef EMU_ATTR

bool have_fma()
{
    static const bool yes = __builtin_cpu_supports("fma");
    return yes;
}

// FOUR probe rows at a time (AVX2 + FMA): lane k carries probe k through exactly the operations, in exactly the
// order, of the scalar functions above -- IEEE lane-wise arithmetic, so the same bits -- at a quarter of the
// instructions (a survey's estimates: 4096 spectra x ~800 kernel emulations + 200 yardstick rows each).
// BISIP_HOST_SCALAR_ESTIMATE=1 (tests) keeps the scalar functions.
bool have_avx2()
{
    static const bool yes = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma") &&
                            std::getenv("BISIP_HOST_SCALAR_ESTIMATE") == nullptr;
    return yes;
}

#define BISIP_X4 __attribute__((target("avx2,fma")))

BISIP_X4 static inline __m256d column4(const double *rows, int n, int q)
{
    return _mm256_set_pd(rows[3 * (size_t)n + q], rows[2 * (size_t)n + q], rows[(size_t)n + q], rows[q]);
}

// chi2_kernel, plain form, probes rows[0..4n)
BISIP_X4 static void chi2_kernel_plain_x4(int n, const double *R, const double *bhat, const double *e, double rest,
                                          const double *rows, double *out)
{
    __m256d d[BISIP_HOST_MAXN];
    const __m256d th0 = column4(rows, n, 0);
    d[0] = _mm256_sub_pd(_mm256_set1_pd(bhat[0]), th0);
    for (int q = 1; q < n; ++q) {
        const __m256d prod = _mm256_mul_pd(th0, column4(rows, n, q));
        d[q] = _mm256_sub_pd(_mm256_set1_pd(bhat[q]), prod);
    }
    __m256d chi2 = _mm256_set1_pd(rest);
    for (int i = 0; i < n; ++i) {
        __m256d u = _mm256_set1_pd(e[i]);
        for (int j = i; j < n; ++j) u = _mm256_fmadd_pd(_mm256_set1_pd(R[(size_t)i * n + j]), d[j], u);
        chi2 = _mm256_fmadd_pd(u, u, chi2);
    }
    _mm256_storeu_pd(out, chi2);
}

// chi2_dd, probes rows[0..4n)
BISIP_X4 static void chi2_dd_x4(int n, const double *Rhi, const double *Rlo, const double *qhi, const double *qlo, double rest,
                                const double *rows, long double *out)
{
    __m256d bh[BISIP_HOST_MAXN], bl[BISIP_HOST_MAXN];
    const __m256d th0 = column4(rows, n, 0);
    bh[0] = th0;
    bl[0] = _mm256_setzero_pd();
    for (int j = 1; j < n; ++j) {
        const __m256d thj = column4(rows, n, j);
        const __m256d p = _mm256_mul_pd(th0, thj);
        bh[j] = p;
        bl[j] = _mm256_fmsub_pd(th0, thj, p);
    }
    long double chi2[4] = {rest, rest, rest, rest};
    for (int i = 0; i < n; ++i) {
        __m256d hi = _mm256_set1_pd(qhi[i]), lo = _mm256_set1_pd(qlo[i]);
        for (int j = i; j < n; ++j) {
            const __m256d r = _mm256_set1_pd(-Rhi[(size_t)i * n + j]);
            const __m256d p = _mm256_mul_pd(r, bh[j]);
            const __m256d pe = _mm256_fmsub_pd(r, bh[j], p);
            const __m256d s = _mm256_add_pd(hi, p);
            const __m256d bb = _mm256_sub_pd(s, hi);
            const __m256d se = _mm256_add_pd(_mm256_sub_pd(hi, _mm256_sub_pd(s, bb)), _mm256_sub_pd(p, bb));
            hi = s;
            const __m256d cross = _mm256_sub_pd(_mm256_mul_pd(r, bl[j]), _mm256_mul_pd(_mm256_set1_pd(Rlo[(size_t)i * n + j]), bh[j]));
            lo = _mm256_add_pd(lo, _mm256_add_pd(_mm256_add_pd(se, pe), cross));
        }
        double h[4], l[4];
        _mm256_storeu_pd(h, hi);
        _mm256_storeu_pd(l, lo);
        for (int k = 0; k < 4; ++k) {
            const long double u = (long double)h[k] + (long double)l[k];
            chi2[k] += u * u;
        }
    }
    for (int k = 0; k < 4; ++k) out[k] = chi2[k];
}

long double chi2_dd(int n, const double *Rhi, const double *Rlo, const double *qhi, const double *qlo, double rest,
                    const double *th)
{
    return have_fma() ? chi2_dd_fma(n, Rhi, Rlo, qhi, qlo, rest, th) : chi2_dd_base(n, Rhi, Rlo, qhi, qlo, rest, th);
}

// worst_error of host_emulate.inc for the plain form, four probes at a time (the remainder by the scalar function)
static double worst_error_plain_x4(int n, const double *R, const double *bhat, const double *e, double rest,
                                   const ReducedProbes &pr, const long double *exact, double lconst, double w_shell)
{
    const size_t count = pr.count(), full = count & ~(size_t)3;
    double worst = 0.0;
    for (size_t ip = 0; ip < full; ip += 4) {
        double got[4];
        chi2_kernel_plain_x4(n, R, bhat, e, rest, &pr.rows[ip * (size_t)n], got);
        for (size_t k = 0; k < 4; ++k) {
            const long double lp = -0.5L * exact[ip + k] + (long double)lconst;
            const long double scale = fabsl(lp) > 1.0L ? fabsl(lp) : 1.0L;
            double rel = (double)(fabsl(-0.5L * ((long double)got[k] - exact[ip + k])) / scale);
            if (ip + k >= pr.n_regular) rel *= w_shell;    // shell probes
            if (!(rel <= worst)) worst = rel;              // NaN counts as worst
        }
    }
    if (full < count) {
        const double tail = worst_error_fma(n, R, nullptr, bhat, e, nullptr, rest, false, &pr.rows[full * (size_t)n], count - full,
                                            pr.n_regular > full ? pr.n_regular - full : 0, exact + full, lconst, w_shell);
        if (!(tail <= worst)) worst = tail;
    }
    return worst;
}

double worst_error(int n, const double *R, const float *Rlo, const double *bhat, const double *e, const double *elo,
                   double rest, bool comp, const ReducedProbes &pr, const long double *exact, double lconst, double w_shell)
{
    if (!comp && have_avx2()) return worst_error_plain_x4(n, R, bhat, e, rest, pr, exact, lconst, w_shell);
    return have_fma() ? worst_error_fma(n, R, Rlo, bhat, e, elo, rest, comp, pr.rows.data(), pr.count(), pr.n_regular, exact, lconst, w_shell)
                      : worst_error_base(n, R, Rlo, bhat, e, elo, rest, comp, pr.rows.data(), pr.count(), pr.n_regular, exact, lconst, w_shell);
}

struct Lcg {   // deterministic probe points, no <random>
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    double uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }
    double sym() { return 2.0 * uni() - 1.0; }
};

// image of the theta box under b = R0 * (1, a)
void box_image(int n, const double *lo, const double *hi, double *blo, double *bhi)
{
    blo[0] = lo[0]; bhi[0] = hi[0];
    for (int j = 1; j < n; ++j) {
        const double c[4] = {lo[0] * lo[j], lo[0] * hi[j], hi[0] * lo[j], hi[0] * hi[j]};
        double a = INFINITY, b = -INFINITY;
        bool nan = false;
        for (double v : c) { if (v != v) nan = true; a = v < a ? v : a; b = v > b ? v : b; }
        if (nan || !(a <= b)) { a = -INFINITY; b = INFINITY; }
        blo[j] = a; bhi[j] = b;
    }
}

// candidates for the expansion point: least squares, centre of the box's image, zero
int candidates(int n, const std::vector<ld> &bhat_ls, const double *lo, const double *hi, double (*cand)[BISIP_HOST_MAXN])
{
    int k = 0;
    bool finite_box = true;
    for (int j = 0; j < n; ++j) finite_box = finite_box && std::isfinite(lo[j]) && std::isfinite(hi[j]);
    {
        bool ok = true;
        for (int j = 0; j < n; ++j) { cand[k][j] = (double)bhat_ls[(size_t)j]; ok = ok && std::isfinite(cand[k][j]) && std::fabs(cand[k][j]) <= 1e6; }
        if (ok) ++k;
    }
    if (finite_box) {
        double blo[BISIP_HOST_MAXN], bhi[BISIP_HOST_MAXN];
        box_image(n, lo, hi, blo, bhi);
        bool ok = true;
        for (int j = 0; j < n; ++j) { cand[k][j] = 0.5 * (blo[j] + bhi[j]); ok = ok && std::isfinite(cand[k][j]); }
        if (ok) ++k;
    }
    for (int j = 0; j < n; ++j) cand[k][j] = 0.0;
    return k + 1;
}

}  // namespace

// BISIP_SHELL_WEIGHT (a test hook: tests force an estimate to pass with 0) changes which reduced tier AUTO
// selects, so an override is never silent: it is validated ([0, 1], else ignored) and announced on stderr once
// per process.  Read on every (re)estimate; BISIP_HOST_SCALAR_ESTIMATE (have_avx2) is latched on first use.
double reduced_shell_weight()
{
    const char *s = std::getenv("BISIP_SHELL_WEIGHT");
    if (!s) return 0.05;
    char *end = nullptr;
    const double w = std::strtod(s, &end);
    const bool ok = end != s && w >= 0.0 && w <= 1.0;
    static std::atomic<bool> said{false};
    if (!said.exchange(true))
        std::fprintf(stderr, "bisip: BISIP_SHELL_WEIGHT=%s %s (default 0.05): the error estimate behind BISIP_VARIANT_AUTO "
                             "weighs shell probes differently -- a test / measurement hook\n", s, ok ? "overrides the shell-probe weight" : "ignored");
    return ok ? w : 0.05;
}

void reduced_probes(const ReducedProblem &p, const double *lo, const double *hi, ReducedProbes &out)
{
    const int n = p.n;
    const std::vector<double> &R = p.R;
    const std::vector<ld> &qty = p.qty, &bhat_ls = p.bhat_ls;
    const double rest = p.rest, lconst = p.lconst;
    out.n = n;
    out.rows.clear();
    out.rows.reserve((size_t)n * 260);
    auto push = [&](const double *t) { out.rows.insert(out.rows.end(), t, t + n); };
    bool finite_box = true;
    for (int j = 0; j < n; ++j) finite_box = finite_box && std::isfinite(lo[j]) && std::isfinite(hi[j]);
    Lcg rng;
    auto inside = [&](const double *t) {
        for (int j = 0; j < n; ++j) if (!(lo[j] < t[j] && t[j] < hi[j])) return false;
        return true;
    };
    double t[BISIP_HOST_MAXN];
    const bool ls_ok = std::isfinite((double)bhat_ls[0]) && bhat_ls[0] != 0.0L;
    if (finite_box) {
        for (int k = 0; k < 32; ++k) {   // uniform in the prior box
            for (int j = 0; j < n; ++j) t[j] = lo[j] + (hi[j] - lo[j]) * rng.uni();
            push(t);
        }
        for (int k = 0; k < 16; ++k) {   // small coefficients: where a decent fit usually lies
            t[0] = lo[0] + (hi[0] - lo[0]) * rng.uni();
            for (int j = 1; j < n; ++j) {
                double v = 1e-3 * rng.sym();
                v = v < lo[j] ? lo[j] : (v > hi[j] ? hi[j] : v);
                t[j] = v;
            }
            if (inside(t)) push(t);
        }
    }
    if (ls_ok)
        for (int k = 0; k < 16; ++k) {   // around the least-squares solution (the posterior mode)
            const double s = k < 8 ? 1e-3 : 1e-2;
            t[0] = (double)bhat_ls[0] * (1.0 + s * rng.sym());
            bool fin = std::isfinite(t[0]);
            for (int j = 1; j < n; ++j) {
                t[j] = (double)(bhat_ls[(size_t)j] / bhat_ls[0]) * (1.0 + s * rng.sym());
                fin = fin && std::isfinite(t[j]);
            }
            if (fin && (!finite_box || inside(t))) push(t);
        }
    // theta of b = b_ls + s R^-1 z for a random direction z (unit variance per component: four uniforms; the
    // shape of the distribution does not matter), s = `scale`, or, for scale < 0, such that the row lies ON
    // the shell rest + s^2 |z|^2 = 2 lconst.  Plain double: where the probe lies need not be exact.  False
    // when the row is not finite or not inside the box.
    bool solvable = ls_ok;
    for (int i = 0; i < n; ++i) solvable = solvable && R[(size_t)i * n + i] != 0.0;
    double bls_d[BISIP_HOST_MAXN];
    for (int j = 0; j < n; ++j) bls_d[j] = (double)bhat_ls[(size_t)j];
    auto valley_point = [&](double scale, double *tp) {
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
        tp[0] = b0;
        if (!std::isfinite(b0) || b0 == 0.0) return false;
        for (int j = 1; j < n; ++j) {
            tp[j] = (bls_d[j] + db[j]) / b0;
            if (!std::isfinite(tp[j])) return false;
        }
        return !finite_box || inside(tp);
    };
    if (ls_ok) {
        // where an ensemble sampler's walkers actually are: draws from the Gaussian posterior of the
        // linear model, b = b_ls + R^-1 z with z ~ N(0, I), and 3x, 10x and 30x wider: a converged
        // ensemble, and the same ensemble on its way in during burn-in.  On nearly collinear
        // designs these spread far along the flat directions of chi^2 -- the rows of R (bhat - b)
        // then cancel by many orders of magnitude although chi^2 stays within a few units (a few
        // hundred, at 10-30 sigma) of its minimum.
        // 16 probes INSIDE the box per scale (a narrowed box keeps few of the draws: up to 300 tries each)
        for (int scale_i = 0; solvable && scale_i < 4; ++scale_i) {
            const double sc = scale_i == 0 ? 1.0 : (scale_i == 1 ? 3.0 : (scale_i == 2 ? 10.0 : 30.0));
            int kept = 0;
            for (int k = 0; k < 300 && kept < 16; ++k)
                if (valley_point(sc, t)) { push(t); ++kept; }
        }
    }
    // The shell log-probability = 0.  The parity tolerance is |d logp| <= 1e-10 max(1, |logp|): where the
    // log-probability crosses zero -- chi^2 = 2 lconst, ten or so posterior sigmas out on a typical
    // spectrum, where burn-in passes -- the denominator is 1 and the ABSOLUTE error of a chi^2 of several
    // hundred counts.  Random valley rows land within |logp| < 1 one time in a few hundred, so the probes
    // above practically never see it.  These probes are ON the shell: b = b_ls + s R^-1 z with s such that
    // rest + s^2 |z|^2 = 2 lconst.  No double formulation gets below ~1e-12 there (one rounding of chi^2 ~ 1e3
    // is 1e-13; the plain triangle of the headline's degree-5 design reads 9e-12, of which 5e-12 is the rounding
    // of R itself), so they count at a twentieth (reduced_shell_weight): the gate 1e-12 then reads "2e-11 on the
    // shell" -- a fifth of the tolerance, where the region is sampled directly -- the same bar bisip_logprob's
    // guard applies to real batches.
    out.n_regular = out.count();
    if (ls_ok && 2.0 * lconst - rest > 0.0) {
        int kept = 0;     // 64 probes inside the box, up to 1500 tries (a narrowed box keeps few of the draws)
        for (int k = 0; solvable && k < 1500 && kept < 64; ++k)
            if (valley_point(-1.0, t)) { push(t); ++kept; }
        // A box that cuts the shell in a small patch keeps none of those draws (a degree-6 design of the
        // fuzz campaign: 9 of 200,000).  The box is convex in theta: between a probe with log-probability > 0 and
        // one with < 0, both inside, the segment stays inside and crosses the shell -- found by bisection (plain
        // double: where the probe lies need not be exact).
        if (kept < 64 && finite_box) {
            auto logp_of = [&](const double *tt) {
                double chi2 = rest;
                for (int i = 0; i < n; ++i) {
                    double u = (double)qty[(size_t)i];
                    for (int j = i; j < n; ++j) u -= R[(size_t)i * n + j] * (j ? tt[0] * tt[j] : tt[0]);
                    chi2 += u * u;
                }
                return lconst - 0.5 * chi2;
            };
            std::vector<size_t> pos, neg;
            std::vector<double> ends(out.rows.begin(), out.rows.begin() + (long)(out.n_regular * (size_t)n));
            for (int k = 0; k < 32; ++k) {       // more of the box, as far ends of the segments only
                for (int j = 0; j < n; ++j) t[j] = lo[j] + (hi[j] - lo[j]) * rng.uni();
                ends.insert(ends.end(), t, t + n);
            }
            const size_t n_ends = ends.size() / (size_t)n;
            for (size_t ip = 0; ip < n_ends; ++ip) {
                const double *ep = &ends[ip * (size_t)n];
                if (!inside(ep)) continue;
                const double lp = logp_of(ep);
                if (lp > 0.0) pos.push_back(ip);
                else if (lp < 0.0) neg.push_back(ip);
            }
            for (int k = 0; !pos.empty() && !neg.empty() && kept < 64 && k < 128; ++k) {
                const double *a = &ends[pos[(size_t)(rng.uni() * (double)pos.size()) % pos.size()] * (size_t)n];
                const double *b = &ends[neg[(size_t)(rng.uni() * (double)neg.size()) % neg.size()] * (size_t)n];
                double t0 = 0.0, t1 = 1.0;
                for (int it = 0; it < 60; ++it) {
                    const double tm = 0.5 * (t0 + t1);
                    for (int j = 0; j < n; ++j) t[j] = a[j] + tm * (b[j] - a[j]);
                    if (logp_of(t) > 0.0) t0 = tm; else t1 = tm;
                }
                for (int j = 0; j < n; ++j) t[j] = a[j] + t0 * (b[j] - a[j]);
                if (inside(t) && std::fabs(logp_of(t)) < 1.0) { push(t); ++kept; }
            }
        }
    }
}

double reduced_center_plain(const ReducedProblem &p, const ReducedProbes &probes, const double *lo, const double *hi,
                            double shell_weight, double *out_bhat, double *out_e, double *out_elo)
{
    const int n = p.n;
    if (probes.count() == 0) {
        // nothing to measure the kernel against (a non-finite box with no usable least-squares
        // solution): expand about zero and report "unknown", so AUTO takes the per-frequency form
        for (int j = 0; j < n; ++j) {
            out_bhat[j] = 0.0;
            out_e[j] = (double)p.qty[(size_t)j];
            out_elo[j] = (double)(p.qty[(size_t)j] - (ld)out_e[j]);
        }
        return INFINITY;
    }
    // yardstick: the reduced form with the UNROUNDED triangle, rows as double-doubles (exact splits of Rl, qty)
    std::vector<double> Rhi((size_t)n * n), Rlo((size_t)n * n), qhi((size_t)n), qlo((size_t)n);
    for (size_t i = 0; i < (size_t)n * n; ++i) { Rhi[i] = (double)p.Rl[i]; Rlo[i] = (double)(p.Rl[i] - (ld)Rhi[i]); }
    for (size_t i = 0; i < (size_t)n; ++i) { qhi[i] = (double)p.qty[i]; qlo[i] = (double)(p.qty[i] - (ld)qhi[i]); }
    std::vector<ld> exact(probes.count());
    size_t done = 0;
    if (have_avx2())
        for (; done + 4 <= probes.count(); done += 4)
            chi2_dd_x4(n, Rhi.data(), Rlo.data(), qhi.data(), qlo.data(), p.rest, &probes.rows[done * (size_t)n], &exact[done]);
    for (size_t ip = done; ip < probes.count(); ++ip)
        exact[ip] = chi2_dd(n, Rhi.data(), Rlo.data(), qhi.data(), qlo.data(), p.rest, &probes.rows[ip * (size_t)n]);
    double cand[3][BISIP_HOST_MAXN];
    const int nc = candidates(n, p.bhat_ls, lo, hi, cand);
    double best = INFINITY, e[BISIP_HOST_MAXN], elo[BISIP_HOST_MAXN];
    for (int ic = 0; ic < nc; ++ic) {
        const double *c = cand[ic];
        for (int i = 0; i < n; ++i) {
            // with the triangle the kernel holds, so that its identity is exact for THAT triangle
            ld s = p.qty[(size_t)i];
            for (int j = i; j < n; ++j) s -= (ld)p.R[(size_t)i * n + j] * (ld)c[j];
            e[i] = (double)s;
            elo[i] = (double)(s - (ld)e[i]);
        }
        const double worst = worst_error(n, p.R.data(), nullptr, c, e, elo, p.rest, false, probes, exact.data(), p.lconst, shell_weight);
        if (worst < best || best == INFINITY) {
            best = worst;
            for (int j = 0; j < n; ++j) { out_bhat[j] = c[j]; out_e[j] = e[j]; out_elo[j] = elo[j]; }
        }
    }
    return best;
}

namespace {

// The compensated kernel's rows are double-doubles: its estimate reads 1e-14 on every design ever probed; a
// third of the probes is plenty to notice if that ever stopped being true.
void every_third(const ReducedProbes &all, ReducedProbes &some)
{
    const size_t n = (size_t)all.n;
    some.n = all.n;
    some.rows.clear();
    size_t shell_from = 0;
    for (size_t ip = 0; ip < all.count(); ip += 3) {
        if (ip < all.n_regular) shell_from = some.count() + 1;
        some.rows.insert(some.rows.end(), all.rows.begin() + (long)(ip * n), all.rows.begin() + (long)((ip + 1) * n));
    }
    some.n_regular = shell_from;
}

// compensated tier, operands R = Rc + Rlo (as the kernel holds them), e + elo formed with the unrounded triangle
// RT / qtyT of precision T, yardstick `exact` (chi^2 per probe) from the same unrounded operands
template <class T>
double center_comp_T(int n, const double *Rc, const float *Rlo, const std::vector<T> &RT, const std::vector<T> &qtyT,
                     const std::vector<ld> &bhat_ls, double rest, double lconst, const ReducedProbes &some,
                     const std::vector<ld> &exact, const double *lo, const double *hi, double shell_weight,
                     double *out_bhat, double *out_e, double *out_elo)
{
    double cand[3][BISIP_HOST_MAXN];
    const int nc = candidates(n, bhat_ls, lo, hi, cand);
    double best = INFINITY, e[BISIP_HOST_MAXN], elo[BISIP_HOST_MAXN];
    for (int ic = 0; ic < nc; ++ic) {
        const double *c = cand[ic];
        for (int i = 0; i < n; ++i) {
            T s = qtyT[(size_t)i];
            for (int j = i; j < n; ++j) s -= RT[(size_t)i * n + j] * (T)c[j];
            e[i] = (double)s;
            elo[i] = (double)(s - (T)e[i]);
        }
        const double worst = worst_error(n, Rc, Rlo, c, e, elo, rest, true, some, exact.data(), lconst, shell_weight);
        if (worst < best || best == INFINITY) {
            best = worst;
            for (int j = 0; j < n; ++j) { out_bhat[j] = c[j]; out_e[j] = e[j]; out_elo[j] = elo[j]; }
        }
    }
    return best;
}

}  // namespace

double reduced_center_comp(const ReducedProblem &p, const ReducedProbes &probes, const double *lo, const double *hi,
                           double shell_weight, double *out_bhat, double *out_e, double *out_elo)
{
    const int n = p.n;
    if (probes.count() == 0) {
        for (int j = 0; j < n; ++j) {
            out_bhat[j] = 0.0;
            out_e[j] = (double)p.qty[(size_t)j];
            out_elo[j] = (double)(p.qty[(size_t)j] - (ld)out_e[j]);
        }
        return INFINITY;
    }
    ReducedProbes some;
    every_third(probes, some);
    std::vector<ld> exact(some.count());
    if (p.has_quad()) {
        const QuadReduced &q = *p.quad;
        for (size_t ip = 0; ip < some.count(); ++ip) exact[ip] = (ld)reduced_chi2_quad(n, q, &some.rows[ip * (size_t)n]);
        // (the yardstick's chi^2 is rounded to long double: 64 bits of a number of a few hundred, 1e-17)
        return center_comp_T<qd>(n, p.Rc.data(), p.Rc_lo.data(), q.Rq, q.qty, p.bhat_ls, p.rest_c, p.lconst, some, exact,
                                 lo, hi, shell_weight, out_bhat, out_e, out_elo);
    }
    // no binary128 operands (reduced_center's old signature): the long-double triangle and its 11-bit low word
    std::vector<float> Rlo((size_t)n * n);
    for (size_t i = 0; i < (size_t)n * n; ++i) Rlo[i] = (float)(p.Rl[i] - (ld)p.R[i]);
    for (size_t ip = 0; ip < some.count(); ++ip) exact[ip] = reduced_chi2_exact(n, p.Rl, p.qty, p.rest, &some.rows[ip * (size_t)n]);
    return center_comp_T<ld>(n, p.R.data(), Rlo.data(), p.Rl, p.qty, p.bhat_ls, p.rest, p.lconst, some, exact, lo, hi,
                             shell_weight, out_bhat, out_e, out_elo);
}

double reduced_center(int n, const std::vector<double> &R, const std::vector<long double> &Rl,
                      const std::vector<long double> &qty,
                      const std::vector<long double> &bhat_ls, double rest, double lconst,
                      const double *lo, const double *hi, bool comp, double *out_bhat, double *out_e,
                      double *out_elo)
{
    ReducedProblem p;
    p.n = n; p.R = R; p.Rl = Rl; p.qty = qty; p.bhat_ls = bhat_ls; p.rest = rest; p.lconst = lconst;
    ReducedProbes probes;
    reduced_probes(p, lo, hi, probes);
    const double w = reduced_shell_weight();
    return comp ? reduced_center_comp(p, probes, lo, hi, w, out_bhat, out_e, out_elo)
                : reduced_center_plain(p, probes, lo, hi, w, out_bhat, out_e, out_elo);
}

double reduced_logp_reference(const ReducedProblem &p, const double *theta)
{
    if (p.has_quad()) return (double)(-0.5Q * reduced_chi2_quad(p.n, *p.quad, theta) + (qd)p.lconst);
    return (double)(-0.5L * reduced_chi2_exact(p.n, p.Rl, p.qty, p.rest, theta) + (ld)p.lconst);
}

// The yardstick for `count` contiguous rows.  A spectrum with binary128 operands (the compensated tier's): row by
// row in binary128.  Otherwise the plain tier's own yardstick -- the one its estimate is made against
// (reduced_center_plain): rows as double-doubles from exact hi / lo splits of the long-double QR, four rows per
// AVX2 instruction where the host has it (0.08 us per row against 0.45 for the long-double pairs).  What the
// guards measure a few hundred rows against while a sampler waits (bisip_ctx_reduced_guard_rows).
void reduced_logp_reference_rows(const ReducedProblem &p, const double *theta, int64_t count, double *out)
{
    const int n = p.n;
    if (p.has_quad() || p.Rl.empty()) {
        for (int64_t i = 0; i < count; ++i) out[i] = reduced_logp_reference(p, theta + i * n);
        return;
    }
    double Rhi[BISIP_HOST_MAXN * BISIP_HOST_MAXN], Rlo[BISIP_HOST_MAXN * BISIP_HOST_MAXN], qhi[BISIP_HOST_MAXN], qlo[BISIP_HOST_MAXN];
    for (int i = 0; i < n * n; ++i) { Rhi[i] = (double)p.Rl[(size_t)i]; Rlo[i] = (double)(p.Rl[(size_t)i] - (ld)Rhi[i]); }
    for (int i = 0; i < n; ++i) { qhi[i] = (double)p.qty[(size_t)i]; qlo[i] = (double)(p.qty[(size_t)i] - (ld)qhi[i]); }
    int64_t done = 0;
    if (have_avx2())
        for (; done + 4 <= count; done += 4) {
            ld c4[4];
            chi2_dd_x4(n, Rhi, Rlo, qhi, qlo, p.rest, theta + done * n, c4);
            for (int k = 0; k < 4; ++k) out[done + k] = (double)(-0.5L * c4[k] + (ld)p.lconst);
        }
    for (; done < count; ++done)
        out[done] = (double)(-0.5L * chi2_dd(n, Rhi, Rlo, qhi, qlo, p.rest, theta + done * n) + (ld)p.lconst);
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
        const ld want = logl((ld)w[j]), got = (ld)lnw[j & ~(HOST_GRID_BLOCK - 1)] + (ld)(j & (HOST_GRID_BLOCK - 1)) * (ld)d;
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

/*
 * bisip_oracle.c -- CPU restatement of the reference's log-probability hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; bisip_amd/ never does.
 *
 * Every function restates one reference symbol, loop for loop, in C99
 * `double complex` arithmetic with glibc cpow/exp -- the same lowering the
 * reference's Cython takes under gcc (CYTHON_CCOMPLEX: `**` -> cpow,
 * src/bisip/cython_funcs.c:808-816,1537-1557 of the reference).  Citations are
 * relative to /root/reference.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against the golden
 * vectors in the tests/golden npz files, which tests/golden/make_golden.py produced by
 * importing the real reference (pyx rebuilt out of tree) in the build container.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { ORACLE_POLYDECOMP = 0, ORACLE_COLECOLE = 1, ORACLE_DIAS = 2, ORACLE_SHIN = 3 };

typedef struct {
    int model_id;
    int N;               /* frequencies */
    int ndim;            /* parameters per walker */
    const double *w;     /* (N,) angular frequencies */
    const double *zn;    /* (2,N) row 0 real, row 1 imag  (src/bisip/utils.py:141) */
    const double *zn_err;/* (2,N)                          (src/bisip/utils.py:142) */
    const double *lo;    /* (ndim,) */
    const double *hi;    /* (ndim,) */
    /* PolynomialDecomposition only (src/bisip/models.py:200-209) */
    int S;               /* number of relaxation times */
    int D;               /* poly_deg + 1 */
    double c_exp;
    const double *taus;      /* (S,) */
    const double *log_taus;  /* (D,S) row-major */
    /* PeltonColeCole only */
    int n_modes;
} oracle_problem;

/* src/bisip/cython_funcs.pyx:33-34 */
static double complex C_ColeCole(double w_, double m_, double lt_, double c_)
{
    return m_ * (1.0 - 1.0 / (1.0 + cpow(I * w_ * exp(lt_), c_)));
}

/* src/bisip/cython_funcs.pyx:36-40 */
static double complex C_Dias(double w_, double R0_, double m_, double log_tau_,
                             double eta_, double delta_)
{
    double tau_p = exp(log_tau_) * (1 / delta_ - 1) / (1 - m_);
    double e = exp(log_tau_);
    double tau_pp = (e * e) * (eta_ * eta_);
    double complex mu = I * w_ * exp(log_tau_) + cpow(I * w_ * tau_pp, 0.5);
    return R0_ * (1 - m_ * (1 - 1.0 / (1 + I * w_ * tau_p * (1 + 1 / mu))));
}

/* src/bisip/cython_funcs.pyx:42-44 */
static double complex C_Shin(double w_, double R_, double log_Q_, double n_)
{
    double complex z_cpe = 1 / (exp(log_Q_) * cpow(I * w_, n_));
    return cpow(1 / z_cpe + 1 / R_, -1); /* `**-1` lowers to cpow as well */
}

/* src/bisip/cython_funcs.pyx:46-47 */
static double complex C_Debye(double w_, double m_, double tau_, double c_)
{
    return m_ * (1 - 1.0 / (1 + cpow(I * w_ * tau_, c_)));
}

/* src/bisip/cython_funcs.pyx:49-62 */
void oracle_colecole_forward(int N, const double *w, double R0, int D, const double *m,
                             const double *lt, const double *c, double *Z)
{
    for (int j = 0; j < N; ++j) {
        double complex z_ = 0;
        for (int i = 0; i < D; ++i)
            z_ += C_ColeCole(w[j], m[i], lt[i], c[i]);
        z_ = R0 * (1 - z_);
        Z[j] = creal(z_);
        Z[N + j] = cimag(z_);
    }
}

/* src/bisip/cython_funcs.pyx:64-73 */
void oracle_dias_forward(int N, const double *w, double R0, double m, double log_tau,
                         double eta, double delta, double *Z)
{
    for (int j = 0; j < N; ++j) {
        double complex z_ = C_Dias(w[j], R0, m, log_tau, eta, delta);
        Z[j] = creal(z_);
        Z[N + j] = cimag(z_);
    }
}

/* src/bisip/cython_funcs.pyx:75-94 */
void oracle_decomp_forward(int N, const double *w, int S, const double *taus, int D,
                           const double *log_taus, double c_exp, double R0,
                           const double *a, double *Z)
{
    double *M = (double *)calloc((size_t)S, sizeof(double));
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < S; ++k)
            M[k] = M[k] + a[i] * log_taus[(size_t)i * S + k];
    for (int j = 0; j < N; ++j) {
        double complex z_ = 0;
        for (int k = 0; k < S; ++k)
            z_ += C_Debye(w[j], M[k], taus[k], c_exp);
        z_ = R0 * (1 - z_);
        Z[j] = creal(z_);
        Z[N + j] = cimag(z_);
    }
    free(M);
}

/* src/bisip/cython_funcs.pyx:96-108 */
void oracle_shin_forward(int N, const double *w, int D, const double *R,
                         const double *log_Q, const double *n, double *Z)
{
    for (int j = 0; j < N; ++j) {
        double complex z_ = 0;
        for (int i = 0; i < D; ++i)
            z_ += C_Shin(w[j], R[i], log_Q[i], n[i]);
        Z[j] = creal(z_);
        Z[N + j] = cimag(z_);
    }
}

/* src/bisip/models.py:64-69 -- strict open box; NaN compares false -> -inf */
double oracle_log_prior(int ndim, const double *theta, const double *lo, const double *hi)
{
    for (int q = 0; q < ndim; ++q)
        if (!(lo[q] < theta[q])) return -INFINITY;
    for (int q = 0; q < ndim; ++q)
        if (!(theta[q] < hi[q])) return -INFINITY;
    return 0.0;
}

/* NumPy's float64 add-reduce over a contiguous run (pairwise_sum in
 * numpy/_core/src/umath/loops_utils.h.src): plain loop below 8 terms, eight
 * strided accumulators up to 128 terms, recursive halving above.  np.sum of the
 * contiguous (2,N) term array in src/bisip/models.py:62 takes this path. */
static double np_pairwise_sum(const double *a, size_t n)
{
    if (n < 8) {
        double res = 0.;
        for (size_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        size_t i;
        for (i = 0; i < 8; ++i) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (size_t q = 0; q < 8; ++q) r[q] += a[i + q];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

/* src/bisip/models.py:59-62 : -0.5*sum((y-Z)^2/sigma2 + 2*log(sigma2)) */
double oracle_log_likelihood(int N, const double *Z, const double *zn, const double *zn_err)
{
    size_t n = 2 * (size_t)N;
    double stack_terms[256] = {0};
    double *t = n <= 256 ? stack_terms : (double *)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; ++i) {
        double sigma2 = zn_err[i] * zn_err[i];
        double r = zn[i] - Z[i];
        t[i] = (r * r) / sigma2 + 2 * log(sigma2);
    }
    double s = np_pairwise_sum(t, n);
    if (t != stack_terms) free(t);
    return -0.5 * s;
}

/* forward dispatch: src/bisip/models.py:217-229, 256-271, 295-305, 335-349 */
static void forward_one(const oracle_problem *p, const double *th, double *Z)
{
    switch (p->model_id) {
    case ORACLE_POLYDECOMP:
        oracle_decomp_forward(p->N, p->w, p->S, p->taus, p->D, p->log_taus, p->c_exp,
                              th[0], th + 1, Z);
        break;
    case ORACLE_COLECOLE: {
        int D = p->n_modes;
        oracle_colecole_forward(p->N, p->w, th[0], D, th + 1, th + 1 + D, th + 1 + 2 * D, Z);
        break;
    }
    case ORACLE_DIAS:
        oracle_dias_forward(p->N, p->w, th[0], th[1], th[2], th[3], th[4], Z);
        break;
    default:
        oracle_shin_forward(p->N, p->w, 2, th, th + 2, th + 4, Z);
        break;
    }
}

/* src/bisip/models.py:71-76, one walker */
double oracle_log_probability(const oracle_problem *p, const double *th, double *Zwork)
{
    double lp = oracle_log_prior(p->ndim, th, p->lo, p->hi);
    if (!isfinite(lp)) return -INFINITY;
    forward_one(p, th, Zwork);
    return lp + oracle_log_likelihood(p->N, Zwork, p->zn, p->zn_err);
}

/* The per-walker loop emcee runs with vectorize=False (src/bisip/models.py:111-118),
 * optionally spread over host threads the way fit(pool=...) spreads it over
 * processes (src/bisip/models.py:91-94,115). */
int oracle_logprob_batch(const oracle_problem *p, const double *theta, int64_t W,
                         double *logp, int n_threads)
{
    if (!p || !theta || !logp || W < 0) return -1;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
#endif
    {
        double *Z = (double *)malloc(2 * (size_t)p->N * sizeof(double));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t i = 0; i < W; ++i)
            logp[i] = oracle_log_probability(p, theta + (size_t)i * p->ndim, Z);
        free(Z);
    }
    return 0;
}

/* batched forward(): theta (W,ndim) -> Z (W,2,N); the loop of src/bisip/utils.py:33-34 */
int oracle_forward_batch(const oracle_problem *p, const double *theta, int64_t W, double *Z)
{
    if (!p || !theta || !Z || W < 0) return -1;
    for (int64_t i = 0; i < W; ++i)
        forward_one(p, theta + (size_t)i * p->ndim, Z + (size_t)i * 2 * p->N);
    return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

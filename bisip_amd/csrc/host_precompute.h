// host_precompute.h -- walker-independent operands, computed once per context in
// x87 80-bit long double and rounded once to binary64.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace bisip {

// What the reference precomputes or recomputes per call but that does not depend on
// the walker (SURVEY.md §0 "the one fact that shapes the kernel design").
struct PolyDecompOperands {
    int N = 0, S = 0, D = 0;          // D = poly_deg + 1
    std::vector<double> K_re, K_im;   // (N,S)  K[j,k] = 1 - 1/(1+(i w_j tau_k)^c)
    std::vector<double> G_re, G_im;   // (N,D)  G[j,p] = sum_k log_taus[p,k]*K[j,k]
    // QR-reduced chi^2 (n = D+1 unknowns b = R0*(1, a_0..a_P)):
    //   chi2(b) = rest + sum_i ( e_i + sum_{j>=i} R[i][j]*(bhat_j - b_j) )^2
    std::vector<double> R;            // (n,n) row-major upper triangle (zeros below)
    std::vector<double> bhat;         // (n,)  expansion point (least-squares solution)
    std::vector<double> e;            // (n,)  Q^T y - R bhat
    double rest = 0.0;
    // unrounded, for reduced_center(): first n entries of Q^T y and the least-squares solution,
    // the triangle itself (R = (double)Rl) and the kernel sums the design matrix is built from
    std::vector<long double> qty, bhat_ls, Rl, Gl_re, Gl_im;
};

// reference: C_Debye at src/bisip/cython_funcs.pyx:46-47, Decomp_cyth :75-94,
// likelihood weights at src/bisip/models.py:59-62.
void polydecomp_operands(int N, const double *w, int S, const double *taus, int D,
                         const double *log_taus, double c_exp, const double *zn,
                         const double *zn_err, PolyDecompOperands &out);

// Chooses the expansion point of the reduced form and says how far the kernel can be trusted.
// The identity
//   chi2(b) = rest + | e + R (bhat - b) |^2,  e = Q^T y - R bhat
// holds for ANY bhat; what bhat decides is the rounding error.  R has entries up to 1e10 for
// high polynomial degrees (nearly collinear columns), each row of R (bhat - b) cancels by many
// orders of magnitude, and forming bhat - b in double discards the low bits of a small b when
// bhat is O(1).  No single rule is best everywhere (measured: least squares 2e-10 .. 8.6e-3
// when it lies far outside the prior box; box centre 2.5e-10 where zero gives 4e-14; zero 3x
// worse than least squares around a well-determined mode), so this function EMULATES the
// kernel's double arithmetic on the host -- the plain form (comp = false) or the compensated
// one (comp = true; kernels.h: logprob_row_reduced<P, COMP>) -- for ~200 probe rows: uniform in
// the prior box [lo, hi] (theta space), clouds of small coefficients, clouds around the
// least-squares solution, and draws from the Gaussian posterior b_ls + R^-1 z (where a
// sampler's walkers sit: the flat valley of an ill-conditioned design) -- against long double,
// for each candidate (least squares, centre of the box's image, zero), keeps the best, and
// returns its worst relative log-probability error.  out_e + out_elo = Q^T y - R bhat to twice
// the working precision.  The caller (AUTO variant) takes the plain kernel while its estimate
// is <= 1e-12, else the compensated one, else the per-frequency form.
// R is the triangle as the kernels hold it (double); Rl the unrounded one: the yardstick is the reduced
// form with Rl, so the plain kernel's estimate includes what rounding R costs; the compensated kernel
// carries Rlo = Rl - R and its e + elo is formed with Rl.
double reduced_center(int n, const std::vector<double> &R, const std::vector<long double> &Rl,
                      const std::vector<long double> &qty,
                      const std::vector<long double> &bhat_ls, double rest, double lconst,
                      const double *lo, const double *hi, bool comp, double *out_bhat, double *out_e,
                      double *out_elo);

// The part of polydecomp_operands that depends on the frequencies and the tau grid only (K, G):
// the spectra of a survey usually share one frequency list, and the N*S long-double powers are
// most of the cost of a spectrum's operands.
void polydecomp_kernel_sums(int N, const double *w, int S, const double *taus, int D,
                            const double *log_taus, double c_exp, PolyDecompOperands &out);
// The rest (weighted design matrix from out's rounded G, Householder QR, least squares).
void polydecomp_reduce(const double *zn, const double *zn_err, PolyDecompOperands &out);

// Host threads for per-spectrum work (batch contexts, file ingest): BISIP_HOST_THREADS, else the
// CPUs this process may use -- affinity mask and cgroup CPU quota -- capped at 16.
int host_threads();
// fn(begin, end) over contiguous blocks of [0, n) on host_threads() threads (the caller's included);
// serial when n is small.  An exception in any block is rethrown on the caller's thread.
void parallel_blocks(int64_t n, int64_t min_per_thread, const std::function<void(int64_t, int64_t)> &fn);

// host_ingest.cpp: the body of bisip_read_tables (include/bisip_hip.h)
void read_tables(const char *const *paths, int64_t n_files, int headers, int64_t n_rows, double *tables,
                 int32_t *status, int threads);

// The reduced form's log-likelihood of one theta row from the unrounded operands, in long double:
// what the reduced kernels compute, without their rounding (reduced_center's yardstick).
double reduced_logp_reference(int n, const std::vector<long double> &Rl, const std::vector<long double> &qty,
                              double rest, double lconst, const double *theta);

// -0.5 * sum_i 2*ln(zn_err_i^2), the walker-independent term of src/bisip/models.py:62
double loglike_const(int n2, const double *zn_err);

// Geometric frequency grid (kernels.h: BOUNDS_GRID): true when, in blocks of eight frequencies,
// ln w_{8k+q} = lnw[8k] + q * dlnw to 4e-15 -- lnw[] the ROUNDED values the kernels hold, ln w_f on the left
// in long double -- for the common step dlnw = (ln w_{N-1} - ln w_0)/(N-1); N >= 8.  *dlnw is set either way
// (0 when there is no such grid).
bool grid_step(int N, const double *w, const double *lnw, double *dlnw);

// ln(w_j) and 1/sigma^2 rounded from long double
void common_operands(int N, const double *w, const double *zn_err, std::vector<double> &lnw,
                     std::vector<double> &inv_var);

}  // namespace bisip

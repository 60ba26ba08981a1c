// host_precompute.h -- walker-independent operands, computed once per context in
// x87 80-bit long double (and, for the compensated tier of the QR-reduced form, in IEEE binary128)
// and rounded once to binary64.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
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

// The part of polydecomp_operands that depends on the frequencies and the tau grid only (K, G):
// the spectra of a survey usually share one frequency list, and the N*S long-double powers are
// most of the cost of a spectrum's operands.
void polydecomp_kernel_sums(int N, const double *w, int S, const double *taus, int D,
                            const double *log_taus, double c_exp, PolyDecompOperands &out);
// The rest (weighted design matrix from out's unrounded G, Householder QR, least squares).
void polydecomp_reduce(const double *zn, const double *zn_err, PolyDecompOperands &out);

// ---------------------------------------------------------------------------------------------
// The reduced form of ONE spectrum as the host keeps it (per spectrum of a context).
//
// Plain tier: R = (double)Rl from the long-double QR above.
// Compensated tier: its operands come from the SAME computation carried out in IEEE binary128
// (__float128: kernel sums G, weighted design matrix, Householder QR, Q^T y), because on nearly
// collinear designs (degree 9, 64 frequencies, c = 0.5: terms 6e7 times the row sums) a 64-bit
// mantissa leaves 2e-10 in a log-probability on the shell logp = 0 -- in the OPERANDS, where no
// arithmetic of the kernel can undo it.  The kernel holds Rc = (double)Rq and Rc_lo = (float)(Rq - Rc)
// (77 bits of the triangle) and e + elo formed with Rq.  Built on demand (reduced_make_quad): a spectrum
// that passes the plain tier never pays for it.
// ---------------------------------------------------------------------------------------------
struct QuadKernelSums;     // kernel sums G of one frequency list in binary128 (opaque: host_precompute.cpp)
struct QuadReduced;        // Rq, Q^T y, least squares of one spectrum in binary128 (opaque)

struct ReducedProblem {
    int n = 0;
    std::vector<double> R;                          // (n,n) plain tier's triangle
    std::vector<long double> Rl, qty, bhat_ls;      // long-double QR
    double rest = 0.0, lconst = 0.0;
    std::vector<double> Rc;                         // (n,n) compensated tier's triangle, empty until reduced_make_quad
    std::vector<float> Rc_lo;                       // (n,n) its low word
    double rest_c = 0.0;
    std::shared_ptr<const QuadReduced> quad;
    bool has_quad() const { return (bool)quad; }
};

// threads: spread the frequencies' rows over the host threads (a caller that is not itself inside parallel_blocks)
std::shared_ptr<const QuadKernelSums> polydecomp_kernel_sums_quad(int N, const double *w, int S, const double *taus, int D,
                                                                  const double *log_taus, double c_exp, bool threads = false);
// fills p.Rc, p.Rc_lo, p.rest_c, p.quad (p.n must be set; zn, zn_err: the spectrum's (2N,) rows)
void reduced_make_quad(const QuadKernelSums &ks, const double *zn, const double *zn_err, ReducedProblem &p);
// a ReducedProblem's long-double part from operands computed by polydecomp_reduce
void reduced_from_operands(const PolyDecompOperands &o, double lconst, ReducedProblem &p);

// Probe rows of one spectrum for one prior box: where walkers are going to be evaluated -- uniform in the
// box, clouds of small coefficients, clouds around the least-squares solution, draws from the Gaussian
// posterior b_ls + s R^-1 z at 1, 3, 10 and 30 sigma (the flat valley of an ill-conditioned design), and rows
// ON the shell log-probability = 0 (rows [n_regular, count)), found by scaling random directions or, where the
// box cuts the shell in a small patch, by bisection.  Deterministic (a fixed LCG): the same rows for both
// tiers and for every call with the same operands and box.
struct ReducedProbes {
    int n = 0;
    std::vector<double> rows;      // count x n, theta space
    size_t n_regular = 0;
    size_t count() const { return n ? rows.size() / (size_t)n : 0; }
};
void reduced_probes(const ReducedProblem &p, const double *lo, const double *hi, ReducedProbes &out);

// Chooses the expansion point of the reduced form and says how far the kernel can be trusted.
// The identity
//   chi2(b) = rest + | e + R (bhat - b) |^2,  e = Q^T y - R bhat
// holds for ANY bhat; what bhat decides is the rounding error.  R has entries up to 1e10 for
// high polynomial degrees (nearly collinear columns), each row of R (bhat - b) cancels by many
// orders of magnitude, and forming bhat - b in double discards the low bits of a small b when
// bhat is O(1).  No single rule is best everywhere (measured: least squares 2e-10 .. 8.6e-3
// when it lies far outside the prior box; box centre 2.5e-10 where zero gives 4e-14; zero 3x
// worse than least squares around a well-determined mode), so these functions EMULATE the
// kernel's double arithmetic on the host -- the plain form or the compensated one (kernels.h:
// logprob_row_reduced<P, COMP>) -- on the probe rows against a yardstick of about twice the precision,
// for each candidate (least squares, centre of the box's image, zero), keep the best, and
// return its worst relative log-probability error (shell probes weighted by shell_weight: see the .cpp).
// out_e + out_elo = Q^T y - R bhat to twice the working precision.  The caller (AUTO variant) takes the
// plain kernel while its estimate is <= 1e-12, else the compensated one, else the per-frequency form.
//   plain: the kernel's triangle is p.R; the yardstick is the reduced form with the unrounded p.Rl (so the
//          estimate includes what rounding R costs), rows accumulated as double-doubles.
//   comp : needs p.has_quad(); the kernel holds p.Rc + p.Rc_lo, the yardstick is the reduced form in
//          binary128; a third of the probes (the compensated rows read 1e-14 on every design ever probed).
double reduced_center_plain(const ReducedProblem &p, const ReducedProbes &probes, const double *lo, const double *hi,
                            double shell_weight, double *out_bhat, double *out_e, double *out_elo);
double reduced_center_comp(const ReducedProblem &p, const ReducedProbes &probes, const double *lo, const double *hi,
                           double shell_weight, double *out_bhat, double *out_e, double *out_elo);
// Weight of the shell probes: BISIP_SHELL_WEIGHT (a test hook: 0 reproduces an estimate that never looks at
// the shell) or 0.05.  Read by the caller ONCE per (re)estimation, on its own thread.
double reduced_shell_weight();

// One call for a lone spectrum, the signature of earlier rounds (tests/native/sanitize_host.cpp): probes +
// centre of one tier from long-double operands; comp = true builds nothing in binary128 (its yardstick is then
// the long-double one).
double reduced_center(int n, const std::vector<double> &R, const std::vector<long double> &Rl,
                      const std::vector<long double> &qty,
                      const std::vector<long double> &bhat_ls, double rest, double lconst,
                      const double *lo, const double *hi, bool comp, double *out_bhat, double *out_e,
                      double *out_elo);

// The reduced form's log-likelihood of one theta row from the unrounded operands: what the reduced kernels
// compute, without their rounding.  From the binary128 operands when the spectrum has them, else from the
// long-double ones with rows accumulated as pairs of long doubles.
double reduced_logp_reference(const ReducedProblem &p, const double *theta);
// ... of `count` contiguous rows, through the yardstick the guards and checks of a context use: binary128 where the
// spectrum has such operands, else rows as double-doubles from exact splits of the long-double QR (the yardstick
// the plain tier's estimate is made against; four rows per AVX2 instruction, the same bits as one by one)
void reduced_logp_reference_rows(const ReducedProblem &p, const double *theta, int64_t count, double *out);
double reduced_logp_reference(int n, const std::vector<long double> &Rl, const std::vector<long double> &qty,
                              double rest, double lconst, const double *theta);

// Host threads for per-spectrum work (batch contexts, file ingest): BISIP_HOST_THREADS, else the
// CPUs this process may use -- affinity mask and cgroup CPU quota -- capped at 16.
int host_threads();
// fn(begin, end) over contiguous blocks of [0, n) on host_threads() threads (the caller's included);
// serial when n is small.  An exception in any block is rethrown on the caller's thread.
void parallel_blocks(int64_t n, int64_t min_per_thread, const std::function<void(int64_t, int64_t)> &fn);

// host_ingest.cpp: the body of bisip_read_tables (include/bisip_hip.h)
void read_tables(const char *const *paths, int64_t n_files, int headers, int64_t n_rows, double *tables,
                 int32_t *status, int threads);

// -0.5 * sum_i 2*ln(zn_err_i^2), the walker-independent term of src/bisip/models.py:62
double loglike_const(int n2, const double *zn_err);

// Geometric frequency grid (kernels.h: BOUNDS_GRID, GRID_BLOCK): true when, in blocks of GRID_BLOCK = 16 frequencies,
// ln w_{16k+q} = lnw[16k] + q * dlnw to 4e-15 -- lnw[] the ROUNDED values the kernels hold, ln w_f on the left
// in long double -- for the common step dlnw = (ln w_{N-1} - ln w_0)/(N-1); N >= 8.  *dlnw is set either way
// (0 when there is no such grid).
constexpr int HOST_GRID_BLOCK = 16;     // == GRID_BLOCK of kernels.h (static_assert in bisip_hip.hip)
bool grid_step(int N, const double *w, const double *lnw, double *dlnw);

// ln(w_j) and 1/sigma^2 rounded from long double
void common_operands(int N, const double *w, const double *zn_err, std::vector<double> &lnw,
                     std::vector<double> &inv_var);

}  // namespace bisip

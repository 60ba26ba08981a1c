// host_precompute.h -- walker-independent operands, computed once per context in
// x87 80-bit long double and rounded once to binary64.
#pragma once
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
    // unrounded, for reduced_center(): first n entries of Q^T y and the least-squares solution
    std::vector<long double> qty, bhat_ls;
};

// Chooses the expansion point of the reduced form.  The identity
//   chi2(b) = rest + | e + R (bhat - b) |^2,  e = Q^T y - R bhat
// holds for ANY bhat; what bhat decides is the rounding error.  R has entries up to 1e10 for
// high polynomial degrees (nearly collinear columns), so each row of R (bhat - b) cancels by
// many orders of magnitude; the part of that cancellation that does not depend on the walker
// is folded into e here, in long double, and what is left for the kernel grows with
// |bhat - b|.  The least-squares solution is the best point when the walkers can reach it
// (posterior rows: 1e-15), but when it lies far outside the prior box (ill-conditioned
// designs: |bhat| ~ 1e2 .. 1e12, measured 2e-10 .. 1e-2 relative error) the centre of the box
// is (5e-14 on the same rows).  [b_lo, b_hi] is the image of the prior box; out_* are (n,).
void reduced_center(int n, const std::vector<double> &R, const std::vector<long double> &qty,
                    const std::vector<long double> &bhat_ls, const double *b_lo, const double *b_hi,
                    double *out_bhat, double *out_e);

// reference: C_Debye at src/bisip/cython_funcs.pyx:46-47, Decomp_cyth :75-94,
// likelihood weights at src/bisip/models.py:59-62.
void polydecomp_operands(int N, const double *w, int S, const double *taus, int D,
                         const double *log_taus, double c_exp, const double *zn,
                         const double *zn_err, PolyDecompOperands &out);

// -0.5 * sum_i 2*ln(zn_err_i^2), the walker-independent term of src/bisip/models.py:62
double loglike_const(int n2, const double *zn_err);

// ln(w_j) and 1/sigma^2 rounded from long double
void common_operands(int N, const double *w, const double *zn_err, std::vector<double> &lnw,
                     std::vector<double> &inv_var);

}  // namespace bisip

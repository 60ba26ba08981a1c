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
    std::vector<double> bhat;         // (n,)
    std::vector<double> e;            // (n,)
    double rest = 0.0;
};

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

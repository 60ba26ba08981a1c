/* Plain-C consumer of libbisip_hip.so: no Python, no framework types.
 *
 *   gcc -std=c99 -I include examples/c_abi_demo.c -L bisip_amd -lbisip_hip -Wl,-rpath,$PWD/bisip_amd -lm -o c_abi_demo
 *   ./c_abi_demo            # prints one log-probability per theta row
 *
 * Pelton Cole-Cole, one mode, 8 frequencies; spectrum generated from theta_true so the
 * first row scores the walker-independent constant exactly.
 */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include "bisip_hip.h"

int main(void)
{
    enum { N = 8, NDIM = 4, W = 5 };
    const double theta_true[NDIM] = {1.0, 0.3, -2.0, 0.5};
    double w[N], zn[2 * N], zn_err[2 * N];
    for (int j = 0; j < N; ++j) {
        w[j] = 2 * 3.14159265358979323846 * 1000.0 / pow(4.0, j);
        double complex z = 1.0 - 1.0 / (1.0 + cpow(I * w[j] * exp(theta_true[2]), theta_true[3]));
        z = theta_true[0] * (1.0 - theta_true[1] * z);
        zn[j] = creal(z);
        zn[N + j] = cimag(z);
        zn_err[j] = 0.01;
        zn_err[N + j] = 0.002;
    }
    const double lo[NDIM] = {0.9, 0.0, -15.0, 0.0}, hi[NDIM] = {1.1, 1.0, 5.0, 1.0};
    bisip_model_desc desc = {0};
    desc.n_modes = 1;
    bisip_ctx *ctx = NULL;
    if (bisip_ctx_create(&ctx, 0, BISIP_MODEL_COLECOLE, N, w, zn, zn_err, NDIM, lo, hi, &desc)) {
        fprintf(stderr, "bisip_ctx_create: %s\n", bisip_last_error());
        return 1;
    }
    const double theta[W * NDIM] = {1.0, 0.3, -2.0, 0.5,    /* the truth */
                                    1.0, 0.3, -2.0, 0.6,
                                    1.05, 0.2, -3.0, 0.4,
                                    1.2, 0.3, -2.0, 0.5,    /* r0 outside the prior -> -inf */
                                    1.0, 1.0, -2.0, 0.5};   /* m on the bound -> -inf */
    double logp[W];
    if (bisip_logprob(ctx, theta, W, logp)) {
        fprintf(stderr, "bisip_logprob: %s\n", bisip_last_error());
        return 1;
    }
    printf("const %.17g\n", bisip_ctx_loglike_const(ctx));
    for (int i = 0; i < W; ++i) printf("logp[%d] %.17g\n", i, logp[i]);
    bisip_ctx_destroy(ctx);
    return 0;
}

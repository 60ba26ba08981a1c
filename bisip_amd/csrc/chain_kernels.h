// Posterior summaries of a device-resident chain (bisip_chain_moments_dev).
//
// The reference summarises a fit with np.mean / np.std over the flattened chain
// (src/bisip/utils.py:55-85, get_param_mean / get_param_std).  For a batch of spectra the
// chain is (n_samples, E*Wp, ndim) and lives in HBM; copying it to the host to take two
// moments per parameter would cost more than sampling it, so the two passes run here.
//
// Mapping: workgroup (e, split) sweeps a contiguous range of samples of ensemble e.  One
// sample of one ensemble is Wp*ndim contiguous doubles; the first `per*ndim` lanes of the
// workgroup (per = 256/ndim) read them as consecutive doubles, so lane t always sees
// parameter t % ndim and keeps ONE running sum -- no dynamic register indexing, coalesced
// loads.  Partial sums are combined in a fixed order (lanes, then splits): the result does
// not depend on scheduling.  Pass 0: mean.  Pass 1: sum (x - mean)^2 -> population std.
#pragma once
#include <hip/hip_runtime.h>

struct MomentArgs {
    const double *chain;     // first used sample
    long long n_samples;     // used samples
    long long sample_stride; // doubles between consecutive used samples
    long long E, Wp;
    int ndim, splits;
    double *mean, *std;      // (E, ndim)
    double *partial;         // (E, splits, ndim) workspace
};

template <int PASS>
__global__ __launch_bounds__(256) void k_moments_partial(const MomentArgs a)
{
    __shared__ double part[256];
    const int e = blockIdx.x, sp = blockIdx.y;
    const int per = 256 / a.ndim, lanes = per * a.ndim;
    const int t = threadIdx.x;
    const int q = t % a.ndim;
    const long long s0 = a.n_samples * sp / a.splits, s1 = a.n_samples * (sp + 1) / a.splits;
    const long long len = a.Wp * a.ndim;  // doubles of this ensemble per sample
    const double *base = a.chain + (long long)e * len;
    const double m = PASS ? a.mean[(long long)e * a.ndim + q] : 0.0;
    // four samples in flight per lane (independent running sums, combined in a fixed order): a
    // single dependent stream of loads per lane left the sweep latency-bound once the chain no
    // longer fits the last-level cache (1.8 GB: 0.11 TB/s)
    double acc4[4] = {0.0, 0.0, 0.0, 0.0};
    auto add = [&](double &acc, double x) { if (PASS) { const double d = x - m; acc = fma(d, d, acc); } else acc += x; };
    if (t < lanes) {
        long long s = s0;
        for (; s + 4 <= s1; s += 4) {
            const double *p = base + s * a.sample_stride;
            for (long long i = t; i < len; i += lanes) {  // i % ndim == q for every i
                const double x0 = __builtin_nontemporal_load(p + i), x1 = __builtin_nontemporal_load(p + a.sample_stride + i);
                const double x2 = __builtin_nontemporal_load(p + 2 * a.sample_stride + i), x3 = __builtin_nontemporal_load(p + 3 * a.sample_stride + i);
                add(acc4[0], x0); add(acc4[1], x1); add(acc4[2], x2); add(acc4[3], x3);
            }
        }
        for (; s < s1; ++s) {
            const double *p = base + s * a.sample_stride;
            for (long long i = t; i < len; i += lanes) add(acc4[0], __builtin_nontemporal_load(p + i));
        }
    }
    const double acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    part[t] = acc;
    __syncthreads();
    if (t < a.ndim) {
        double sum = 0.0;
        for (int k = 0; k < per; ++k) sum += part[k * a.ndim + t];
        a.partial[((long long)e * a.splits + sp) * a.ndim + t] = sum;
    }
}

template <int PASS>
__global__ __launch_bounds__(256) void k_moments_finish(const MomentArgs a)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;  // (e, q)
    if (idx >= a.E * a.ndim) return;
    const long long e = idx / a.ndim;
    const int q = (int)(idx % a.ndim);
    double sum = 0.0;
    for (int sp = 0; sp < a.splits; ++sp) sum += a.partial[(e * a.splits + sp) * a.ndim + q];
    const double cnt = (double)a.n_samples * (double)a.Wp;
    if (PASS) a.std[idx] = sqrt(sum / cnt);
    else a.mean[idx] = sum / cnt;
}

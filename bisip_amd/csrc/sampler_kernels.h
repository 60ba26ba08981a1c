// sampler_kernels.h -- device-resident stretch move (Goodman & Weare), one lane per
// ACTIVE walker.  This replaces the per-walker Python loop emcee runs around the
// log-probability (reference call sites src/bisip/models.py:111-118; algorithm:
// SURVEY.md Appendix B): proposal, log-probability of the proposal, accept test and
// the state/chain update happen in one launch per half-step, with the ensemble and the
// chain resident in HBM.
//
// A half-step touches Ns "slots"; slot t moves walker active[t] along the line through
// walker partner[t] of the complementary half:
//     q = c - (c - s) * z                      (same three roundings as the host sampler)
//     accept  <=>  factor + logp(q) - logp(s) > ln u
// Active rows are written only by their own lane and partners are never active in the
// same half-step, so the in-place update is race-free.
//
// The log-probability of q is evaluated by the same device functions as the batch
// kernels (kernels.h: logprob_row / logprob_row_reduced) -- identical bits.
#pragma once
#include "kernels.h"
#include "philox.h"

namespace bisip {

struct StretchArgs {
    double *coords;        // (W, NDIM) in/out
    double *logp;          // (W,)      in/out
    const int *active;     // (n_slots,)
    const int *partner;    // (n_slots,)
    const double *zz;      // (n_slots,) stretch factor z
    const double *factor;  // (n_slots,) (ndim-1) ln z
    const double *logu;    // (n_slots,) ln u
    long long n_slots;
    long long slot_lo, slot_hi;  // slots this launch evaluates (eval kernel)
    double *block;         // eval: out (slot_hi-slot_lo, NDIM+2) | apply: in, gathered
    double *chain_row;     // (W, NDIM) chain[step] or null
    double *logp_row;      // (W,)      log_prob[step] or null
    int *naccept;          // (W,) or null
    int *status;           // bit 0: a proposal's log-probability was NaN
    // apply kernel: layout of the gathered block = world slabs of `pad` rows; the first
    // `extra` ranks own base+1 slots, the others base
    long long pad, base, extra;
};

template <class M>
struct GenericLP {
    static constexpr int NDIM = M::NDIM;
    ModelOperands o;
    Bounds b;
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int) const
    {
        return logprob_row<M>(th, o, b);
    }
};

template <int P>
struct ReducedLP {
    static constexpr int NDIM = P + 2;
    ReducedArgs<P> r;
    double lconst;
    Bounds b;
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int) const
    {
        return logprob_row_reduced<P>(th, r, lconst, b);
    }
};

// batch of spectra: operands of the spectrum that owns walker i (i / Wp)
template <class M, bool UNIFORM>
struct BatchGenericLP {
    static constexpr int NDIM = M::NDIM;
    const double *cb;
    long long cb_stride, Wp;
    const double *lconst;
    int N;
    Bounds b;
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int walker) const
    {
        const long long e = spectrum_of<UNIFORM>(walker, Wp);
        const ModelOperands o{cb + e * cb_stride, N, lconst[e]};
        return logprob_row<M>(th, o, b);
    }
};

template <int P, bool UNIFORM>
struct BatchReducedLP {
    static constexpr int NDIM = P + 2;
    const ReducedArgs<P> *red;
    long long Wp;
    const double *lconst;
    Bounds b;
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int walker) const
    {
        const long long e = spectrum_of<UNIFORM>(walker, Wp);
        return logprob_row_reduced<P>(th, red[e], lconst[e], b);
    }
};

// proposal + log-prob + accept for slot t; returns the walker's row/log-prob AFTER the move
template <class LP>
__device__ __forceinline__ bool stretch_slot(const StretchArgs &a, const LP &lp, long long t,
                                             int &walker, double (&row)[LP::NDIM], double &lp_row)
{
    constexpr int NDIM = LP::NDIM;
    const int i = a.active[t], p = a.partner[t];
    const double z = a.zz[t];
    double s[NDIM], q[NDIM];
#pragma unroll
    for (int k = 0; k < NDIM; ++k) {
        const double c = a.coords[(long long)p * NDIM + k];
        s[k] = a.coords[(long long)i * NDIM + k];
        const double d = c - s[k];
        q[k] = c - d * z;
    }
    const double old_lp = a.logp[i];
    const double new_lp = lp(q, i);
    if (new_lp != new_lp) atomicOr(a.status, 1);
    const bool acc = (a.factor[t] + new_lp) - old_lp > a.logu[t];
#pragma unroll
    for (int k = 0; k < NDIM; ++k) row[k] = acc ? q[k] : s[k];
    lp_row = acc ? new_lp : old_lp;
    walker = i;
    return acc;
}

template <int NDIM>
__device__ __forceinline__ void commit_row(const StretchArgs &a, int i, const double (&row)[NDIM],
                                           double lp_row, bool acc)
{
    if (acc) {
#pragma unroll
        for (int k = 0; k < NDIM; ++k) a.coords[(long long)i * NDIM + k] = row[k];
        a.logp[i] = lp_row;
    }
    if (a.chain_row) {
#pragma unroll
        for (int k = 0; k < NDIM; ++k) a.chain_row[(long long)i * NDIM + k] = row[k];
    }
    if (a.logp_row) a.logp_row[i] = lp_row;
    if (a.naccept && acc) a.naccept[i] += 1;
}

// single-rank half-step: evaluate every slot and update the state in place
template <class LP>
__global__ __launch_bounds__(64) void k_stretch_half(const StretchArgs a, const LP lp)
{
    const long long t = (long long)blockIdx.x * 64 + threadIdx.x;
    if (t >= a.n_slots) return;
    int i;
    double row[LP::NDIM], lp_row;
    const bool acc = stretch_slot(a, lp, t, i, row, lp_row);
    commit_row<LP::NDIM>(a, i, row, lp_row, acc);
}

// sharded half-step, part 1: this rank's slots -> block rows (row, logp, accepted)
template <class LP>
__global__ __launch_bounds__(64) void k_stretch_eval(const StretchArgs a, const LP lp)
{
    constexpr int NDIM = LP::NDIM;
    const long long t = a.slot_lo + (long long)blockIdx.x * 64 + threadIdx.x;
    if (t >= a.slot_hi) return;
    int i;
    double row[NDIM], lp_row;
    const bool acc = stretch_slot(a, lp, t, i, row, lp_row);
    double *out = a.block + (t - a.slot_lo) * (NDIM + 2);
#pragma unroll
    for (int k = 0; k < NDIM; ++k) out[k] = row[k];
    out[NDIM] = lp_row;
    out[NDIM + 1] = acc ? 1.0 : 0.0;
}

// sharded half-step, part 2 (after the all-gather): scatter every slot's row into the state
template <int NDIM>
__global__ __launch_bounds__(64) void k_stretch_apply(const StretchArgs a)
{
    const long long t = (long long)blockIdx.x * 64 + threadIdx.x;
    if (t >= a.n_slots) return;
    // owner rank and offset of slot t under shard_range()
    long long r, off;
    const long long big = a.extra * (a.base + 1);
    if (t < big) { r = t / (a.base + 1); off = t - r * (a.base + 1); }
    else { r = a.extra + (t - big) / a.base; off = (t - big) - (r - a.extra) * a.base; }
    const double *in = a.block + (r * a.pad + off) * (NDIM + 2);
    double row[NDIM];
#pragma unroll
    for (int k = 0; k < NDIM; ++k) row[k] = in[k];
    commit_row<NDIM>(a, a.active[t], row, in[NDIM], in[NDIM + 1] > 0.0);
}

// ---------------------------------------------------------------------------------
// Device-side generation of the stretch-move random stream (rng='philox').
// Counter-based, so every (step, half, slot) is independent of launch shape and every
// rank of a sharded run derives the same numbers with no communication.
//   key     = (seed_lo, seed_hi)
//   counter = (slot t, step k, half h, purpose)   purpose 0: (z, partner), 1: accept
//   split   : per-step affine bijection pi(i) = (A*i + B) mod W (host-drawn, gcd(A,W)=1);
//             walker i is in half pi(i)&1 at slot pi(i)>>1, so slot t of half h holds
//             walker Ainv*((2t+h-B) mod W) mod W.
//   z       = ((a-1)*u53(x0,x1) + 1)^2 / a;  partner slot r = (x2*Nc)>>32 in the other half
// Outputs are the same (n_steps, 2, nh) arrays the host-stream mode uploads.
// With E ensembles (batch of spectra) the arrays are (n_steps, 2, E, nh), walker indices
// are global (e*W + i), the counter's third word is h | (e << 1) and all ensembles share
// the step's split.
// ---------------------------------------------------------------------------------
struct DrawArgs {
    long long W, nh, n_steps, step0;  // W = walkers per ensemble
    long long E;                      // ensembles (E > 1 requires W even)
    double a, ndim_m1;
    unsigned int seed_lo, seed_hi;
    const int *perm;  // (n_steps, 3): A, Ainv, B
    int *active, *partner;
    double *zz, *factor, *logu;
};

__device__ __forceinline__ int perm_inverse(long long y, long long W, long long Ainv, long long B)
{
    long long v = (y - B) % W;
    if (v < 0) v += W;
    return (int)((Ainv * v) % W);
}

__global__ __launch_bounds__(256) void k_stretch_draw(const DrawArgs d)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = d.n_steps * 2 * d.E * d.nh;
    if (idx >= total) return;
    const long long t = idx % d.nh;
    const long long e = (idx / d.nh) % d.E;
    const int h = (int)((idx / (d.nh * d.E)) & 1);
    const long long k = idx / (2 * d.E * d.nh);
    const long long n0 = (d.W + 1) / 2, n1 = d.W / 2;
    const long long Ns = h ? n1 : n0, Nc = h ? n0 : n1;
    if (t >= Ns) {  // padding slot of the smaller half
        d.active[idx] = 0; d.partner[idx] = 0; d.zz[idx] = 1.0; d.factor[idx] = 0.0; d.logu[idx] = 0.0;
        return;
    }
    const long long A_inv = d.perm[3 * k + 1], B = d.perm[3 * k + 2];
    const unsigned int step = (unsigned int)(d.step0 + k);
    const unsigned int c2 = (unsigned int)h | ((unsigned int)e << 1);
    const Philox4 r0 = philox4x32_10((unsigned int)t, step, c2, 0u, d.seed_lo, d.seed_hi);
    const Philox4 r1 = philox4x32_10((unsigned int)t, step, c2, 1u, d.seed_lo, d.seed_hi);
    const double uz = u53(r0.v[0], r0.v[1]);
    const long long r = (long long)(((unsigned long long)r0.v[2] * (unsigned long long)Nc) >> 32);
    const double v = (d.a - 1.0) * uz + 1.0;
    const double z = (v * v) / d.a;
    d.active[idx] = (int)(e * d.W) + perm_inverse(2 * t + h, d.W, A_inv, B);
    d.partner[idx] = (int)(e * d.W) + perm_inverse(2 * r + (1 - h), d.W, A_inv, B);
    d.zz[idx] = z;
    d.factor[idx] = d.ndim_m1 * log(z);
    d.logu[idx] = log(u53(r1.v[0], r1.v[1]));
}

}  // namespace bisip

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

// L = lanes per walker (see logprob_row); g = this lane's index inside its group
template <class M, int L_ = 1>
struct GenericLP {
    static constexpr int NDIM = M::NDIM;
    static constexpr int L = L_;
    ModelOperands o;
    Bounds b;
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int, int g) const
    {
        return logprob_row<M, L>(th, o, b, g);
    }
};

template <int P>
struct ReducedLP {
    static constexpr int NDIM = P + 2;
    static constexpr int L = 1;
    ReducedArgs<P> r;
    double lconst;
    Bounds b;
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int, int) const
    {
        return logprob_row_reduced<P>(th, r, lconst, b);
    }
};

// batch of spectra: operands of the spectrum that owns walker i (i / Wp)
template <class M, bool UNIFORM, int L_ = 1>
struct BatchGenericLP {
    static constexpr int NDIM = M::NDIM;
    static constexpr int L = L_;
    const double *cb;
    long long cb_stride, Wp;
    const double *lconst;
    int N;
    Bounds b;
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int walker, int g) const
    {
        const long long e = spectrum_of<UNIFORM>(walker, Wp);
        const ModelOperands o{cb + e * cb_stride, N, lconst[e]};
        return logprob_row<M, L>(th, o, b, g);
    }
};

template <int P, bool UNIFORM>
struct BatchReducedLP {
    static constexpr int NDIM = P + 2;
    static constexpr int L = 1;
    const ReducedArgs<P> *red;
    long long Wp;
    const double *lconst;
    Bounds b;
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int walker, int) const
    {
        const long long e = spectrum_of<UNIFORM>(walker, Wp);
        return logprob_row_reduced<P>(th, red[e], lconst[e], b);
    }
};

// One stretch move: walker i (state row s, log-prob old_lp) along the line through row c.
// `walker` is the global walker id handed to the log-prob functor (batch contexts pick
// the spectrum from it).  Returns the row / log-prob AFTER the move.
template <class LP>
__device__ __forceinline__ bool stretch_move(const double *s_row, const double *c_row, double old_lp,
                                             double z, double factor, double logu, const LP &lp,
                                             int walker, int g, int *status, double (&row)[LP::NDIM],
                                             double &lp_row)
{
    constexpr int NDIM = LP::NDIM;
    double s[NDIM], q[NDIM];
#pragma unroll
    for (int k = 0; k < NDIM; ++k) {
        const double c = c_row[k];
        s[k] = s_row[k];
        const double d = c - s[k];
        q[k] = c - d * z;
    }
    const double new_lp = lp(q, walker, g);
    if (new_lp != new_lp) atomicOr(status, 1);
    const bool acc = (factor + new_lp) - old_lp > logu;
#pragma unroll
    for (int k = 0; k < NDIM; ++k) row[k] = acc ? q[k] : s[k];
    lp_row = acc ? new_lp : old_lp;
    return acc;
}

// proposal + log-prob + accept for slot t of a launch-per-half-step kernel; the LP::L lanes of
// a slot all run it (they share the proposal and split the log-probability's frequencies)
template <class LP>
__device__ __forceinline__ bool stretch_slot(const StretchArgs &a, const LP &lp, long long t, int g,
                                             int &walker, double (&row)[LP::NDIM], double &lp_row)
{
    constexpr int NDIM = LP::NDIM;
    const int i = a.active[t], p = a.partner[t];
    walker = i;
    return stretch_move(a.coords + (long long)i * NDIM, a.coords + (long long)p * NDIM, a.logp[i],
                        a.zz[t], a.factor[t], a.logu[t], lp, i, g, a.status, row, lp_row);
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

// single-rank half-step: evaluate every slot and update the state in place.
// LP::L lanes per slot; a wave holds 64/L slots.  When a wave's last slots do not exist the
// lanes still run (clamped to the last slot) so the wavefront exchanges stay uniform; only
// lane 0 of a live slot commits.  All lanes of a wave read their rows before any commits
// (same instruction stream), and no other wave touches them.
template <class LP>
__global__ __launch_bounds__(64) void k_stretch_half(const StretchArgs a, const LP lp)
{
    constexpr int L = LP::L;
    const long long tid = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long slot = tid / L;
    const int g = (int)(tid % L);
    const bool live = slot < a.n_slots;
    const long long t = live ? slot : a.n_slots - 1;
    int i;
    double row[LP::NDIM], lp_row;
    const bool acc = stretch_slot(a, lp, t, g, i, row, lp_row);
    if (live && g == 0) commit_row<LP::NDIM>(a, i, row, lp_row, acc);
}

// sharded half-step, part 1: this rank's slots -> block rows (row, logp, accepted)
template <class LP>
__global__ __launch_bounds__(64) void k_stretch_eval(const StretchArgs a, const LP lp)
{
    constexpr int NDIM = LP::NDIM;
    constexpr int L = LP::L;
    const long long tid = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long slot = a.slot_lo + tid / L;
    const int g = (int)(tid % L);
    const bool live = slot < a.slot_hi;
    const long long t = live ? slot : a.slot_hi - 1;
    int i;
    double row[NDIM], lp_row;
    const bool acc = stretch_slot(a, lp, t, g, i, row, lp_row);
    if (!live || g != 0) return;
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

// same value as perm_inverse for W <= 2048 (all products < 2^22): 32-bit integers and one
// fp32 reciprocal multiply instead of two 64-bit software divisions -- the persistent
// kernel is latency-bound on a single wave, and the divisions were a third of its path
__device__ __forceinline__ int perm_inverse_small(int y, int W, int Ainv, int B, float invW)
{
    int v = y - B;
    v += (v < 0) ? W : 0;
    const int x = Ainv * v;
    int r = x - (int)((float)x * invW) * W;
    r += (r < 0) ? W : 0;
    r -= (r >= W) ? W : 0;
    return r;
}

// the per-slot draw of the philox contract (shared by k_stretch_draw and the persistent kernel)
struct SlotDraw {
    int active, partner;  // walker indices inside the ensemble
    double z, factor, logu;
};

template <bool SMALL = false>
__device__ __forceinline__ SlotDraw draw_slot(long long W, double a, double ndim_m1, unsigned seed_lo,
                                              unsigned seed_hi, unsigned step, int h, unsigned e,
                                              long long t, long long A_inv, long long B)
{
    const long long Nc = h ? (W + 1) / 2 : W / 2;
    const unsigned c2 = (unsigned)h | (e << 1);
    const Philox4 r0 = philox4x32_10((unsigned)t, step, c2, 0u, seed_lo, seed_hi);
    const Philox4 r1 = philox4x32_10((unsigned)t, step, c2, 1u, seed_lo, seed_hi);
    const double uz = u53(r0.v[0], r0.v[1]);
    const long long r = (long long)(((unsigned long long)r0.v[2] * (unsigned long long)Nc) >> 32);
    const double v = (a - 1.0) * uz + 1.0;
    SlotDraw d;
    d.z = (v * v) / a;
    if constexpr (SMALL) {
        const float invW = 1.0f / (float)W;
        d.active = perm_inverse_small((int)(2 * t + h), (int)W, (int)A_inv, (int)B, invW);
        d.partner = perm_inverse_small((int)(2 * r + (1 - h)), (int)W, (int)A_inv, (int)B, invW);
    } else {
        d.active = perm_inverse(2 * t + h, W, A_inv, B);
        d.partner = perm_inverse(2 * r + (1 - h), W, A_inv, B);
    }
    d.factor = ndim_m1 * log(d.z);
    d.logu = log(u53(r1.v[0], r1.v[1]));
    return d;
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
    const long long Ns = h ? d.W / 2 : (d.W + 1) / 2;
    if (t >= Ns) {  // padding slot of the smaller half
        d.active[idx] = 0; d.partner[idx] = 0; d.zz[idx] = 1.0; d.factor[idx] = 0.0; d.logu[idx] = 0.0;
        return;
    }
    const SlotDraw s = draw_slot(d.W, d.a, d.ndim_m1, d.seed_lo, d.seed_hi, (unsigned)(d.step0 + k), h,
                                 (unsigned)e, t, d.perm[3 * k + 1], d.perm[3 * k + 2]);
    d.active[idx] = (int)(e * d.W) + s.active;
    d.partner[idx] = (int)(e * d.W) + s.partner;
    d.zz[idx] = s.z;
    d.factor[idx] = s.factor;
    d.logu[idx] = s.logu;
}

// ---------------------------------------------------------------------------------
// Persistent sampler: ONE WORKGROUP PER ENSEMBLE runs every iteration of a chunk inside
// one launch.  The ensemble (positions + log-probs) lives in LDS, the random stream is
// drawn in-kernel from the same philox counters as k_stretch_draw, the two half-steps of
// an iteration are separated by a workgroup barrier (no launch, no grid sync), and only
// the stored chain rows go to HBM.  Bit-identical to draw + launch-per-half-step.
// Needs W*(NDIM+1)*8 B of LDS and ceil(W/2) <= blockDim.x lanes.
// ---------------------------------------------------------------------------------
struct PersistArgs {
    double *coords;      // (E*W, NDIM) in/out (global state)
    double *logp;        // (E*W,)
    long long W;         // walkers per ensemble
    long long n_steps, step0, thin_by;
    double a, ndim_m1;
    unsigned int seed_lo, seed_hi;
    const int *perm;     // (n_steps, 3)
    double *chain;       // (n_steps/thin_by, E*W, NDIM) or null
    double *logp_chain;  // (n_steps/thin_by, E*W) or null
    int *naccept;        // (E*W,) or null
    int *status;
    long long E;
};

template <class LP>
__global__ __launch_bounds__(1024) void k_stretch_persistent(const PersistArgs a, const LP lp)
{
    constexpr int NDIM = LP::NDIM;
    extern __shared__ __attribute__((aligned(16))) double lds_state[];  // W*NDIM coords | W logp
    double *xs = lds_state;
    double *ls = lds_state + a.W * NDIM;
    const long long e = blockIdx.x;
    const long long base = e * a.W;           // first global walker of this ensemble
    for (long long i = threadIdx.x; i < a.W * NDIM; i += blockDim.x) xs[i] = a.coords[base * NDIM + i];
    for (long long i = threadIdx.x; i < a.W; i += blockDim.x) ls[i] = a.logp[base + i];
    __syncthreads();
    const long long t = threadIdx.x;
    for (long long k = 0; k < a.n_steps; ++k) {
        const long long A_inv = a.perm[3 * k + 1], B = a.perm[3 * k + 2];
        const bool store = ((k + 1) % a.thin_by) == 0;
        const long long srow = k / a.thin_by;
        for (int h = 0; h < 2; ++h) {
            const long long Ns = h ? a.W / 2 : (a.W + 1) / 2;
            if (t < Ns) {
                const SlotDraw d = draw_slot<true>(a.W, a.a, a.ndim_m1, a.seed_lo, a.seed_hi,
                                                   (unsigned)(a.step0 + k), h, (unsigned)e, t, A_inv, B);
                const int i = d.active;
                double row[NDIM], lp_row;
                const bool acc = stretch_move(xs + (long long)i * NDIM, xs + (long long)d.partner * NDIM,
                                              ls[i], d.z, d.factor, d.logu, lp, (int)(base + i), 0, a.status,
                                              row, lp_row);
                if (acc) {   // own row only; partners are never active in this half
#pragma unroll
                    for (int q = 0; q < NDIM; ++q) xs[(long long)i * NDIM + q] = row[q];
                    ls[i] = lp_row;
                    if (a.naccept) atomicAdd(a.naccept + base + i, 1);
                }
                if (store && a.chain) {
                    double *cr = a.chain + (srow * a.E * a.W + base + i) * NDIM;
#pragma unroll
                    for (int q = 0; q < NDIM; ++q) cr[q] = row[q];
                }
                if (store && a.logp_chain) a.logp_chain[srow * a.E * a.W + base + i] = lp_row;
            }
            __syncthreads();   // the other half reads the rows just written
        }
    }
    for (long long i = threadIdx.x; i < a.W * NDIM; i += blockDim.x) a.coords[base * NDIM + i] = xs[i];
    for (long long i = threadIdx.x; i < a.W; i += blockDim.x) a.logp[base + i] = ls[i];
}

}  // namespace bisip

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
#include <type_traits>
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
    // k_stretch_half_packed: the state as ONE 64-byte row per walker (theta, padding, the log-probability in [7]);
    // null: coords / logp as above
    double *packed;
    // k_stretch_half_packed with the slot's stream entries DRAWN IN PLACE (bisip_stretch_run_philox_dev): perm = the
    // iteration's (A, Ainv, B) of the philox contract (see draw_slot); null: the five arrays above are read
    const int *perm;
    long long draw_W;
    double draw_a, draw_ndim_m1;
    unsigned int seed_lo, seed_hi, draw_step, draw_e;
    int draw_h;
};

// L = lanes per walker (see logprob_row); g = this lane's index inside its group.
// Functors over per-frequency records (CAN_STAGE) also say where a walker's records live and
// evaluate against a copy of them -- the persistent kernel keeps that copy in LDS.
// a batch functor whose waves may straddle two spectra never runs a persistent kernel (a workgroup there IS an ensemble)
template <class LP, class = void>
struct MayPersist : std::true_type {};
template <class LP>
struct MayPersist<LP, std::void_t<decltype(LP::WAVE_IN_ONE_SPECTRUM)>> : std::bool_constant<LP::WAVE_IN_ONE_SPECTRUM> {};

// does the functor evaluate a single-spectrum context (the multi-workgroup sampler exists for those only)?
template <class LP, class = void>
struct SingleSpectrum : std::false_type {};
template <class LP>
struct SingleSpectrum<LP, std::void_t<decltype(LP::SINGLE_SPECTRUM)>> : std::bool_constant<LP::SINGLE_SPECTRUM> {};

template <class M, int L_ = 1>
struct GenericLP {
    static constexpr bool SINGLE_SPECTRUM = true;
    static constexpr int NDIM = M::NDIM;
    static constexpr int L = L_;
    static constexpr bool CAN_STAGE = true;
    static constexpr int REC_DOUBLES = M::REC;
    ModelOperands o;
    Bounds b;
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int, int g) const
    {
        return logprob_row<M, L>(th, o, b, g);
    }
    // persistent kernel: the ensemble index e is known to the caller (no division per call);
    // lds = that ensemble's records in LDS, or null to read them where they are
    __device__ __forceinline__ const double *records(long long) const { return o.cb; }
    __host__ __device__ __forceinline__ int n_freq() const { return o.N; }
    // what the persistent kernel fetches ONCE per launch for its ensemble (nothing here: the
    // operands travel in the kernarg segment)
    struct Local {};
    __device__ __forceinline__ Local local(long long) const { return {}; }
    template <bool STAGED>
    __device__ __forceinline__ double eval_ens(const double (&th)[NDIM], const Local &, int g, const double *lds) const
    {
        if constexpr (STAGED) {
            const ModelOperands oo{lds, o.N, o.lconst};
            return logprob_row<M, L, true>(th, oo, b, g);
        } else {
            return logprob_row<M, L>(th, o, b, g);
        }
    }
};

// ReducedArgs<P, true> with the low words (Rlo, elo) read where they lie -- LDS, in the persistent kernels --
// instead of held in registers: what logprob_row_reduced needs of an operand struct, nothing else
template <int P>
struct ReducedLowInLds {
    static constexpr int n = P + 2;
    double R[n * (n + 1) / 2];
    double bhat[n];
    double e[n];
    double rest;
    const float *Rlo;
    const double *elo;
};

// the compensated row sums of the persistent kernels: R, bhat, e, rest from the register copy, the low words from
// the operand image staged in LDS (layout of ReducedArgs<P, true>: Rlo floats | R | bhat | e | elo | rest)
template <int P>
__device__ __forceinline__ double logprob_row_reduced_staged(const double (&th)[P + 2], const ReducedArgs<P, true> &r,
                                                              const double *lds, double lconst, const Bounds &b)
{
    constexpr int n = P + 2, TRI = n * (n + 1) / 2, LO = ((TRI + 1) & ~1) / 2;
    ReducedLowInLds<P> v;
#pragma unroll
    for (int k = 0; k < TRI; ++k) v.R[k] = r.R[k];
#pragma unroll
    for (int k = 0; k < n; ++k) { v.bhat[k] = r.bhat[k]; v.e[k] = r.e[k]; }
    v.rest = r.rest;
    v.Rlo = reinterpret_cast<const float *>(lds);
    v.elo = lds + LO + TRI + 2 * n;
    return logprob_row_reduced<P, true, ReducedLowInLds<P>>(th, v, lconst, b);
}

// does the functor write its LDS image itself (from kernel arguments) instead of naming a source in memory?
template <class LP, class = void>
struct StagesFromArgs : std::false_type {};
template <class LP>
struct StagesFromArgs<LP, std::void_t<decltype(LP::STAGE_FROM_ARGS)>> : std::bool_constant<LP::STAGE_FROM_ARGS> {};

template <int P, bool COMP = false>
struct ReducedLP {
    static constexpr bool SINGLE_SPECTRUM = true;
    static constexpr int NDIM = P + 2;
    static constexpr int L = 1;
    // compensated tier: the low words live in LDS (see BatchReducedLP); the operands are kernel arguments here,
    // so lane 0 writes the two arrays itself
    static constexpr bool CAN_STAGE = COMP;
    static constexpr bool STAGE_FROM_ARGS = true;
    static constexpr int REC_DOUBLES = (int)(sizeof(ReducedArgs<P, COMP>) / sizeof(double));
    __host__ __device__ __forceinline__ int n_freq() const { return 1; }
    __device__ __forceinline__ const double *records(long long) const { return nullptr; }
    __device__ __forceinline__ void stage_from_args(double *mine, int lane) const
    {
        if constexpr (COMP) {
            constexpr int n = P + 2, TRI = n * (n + 1) / 2, LO = ((TRI + 1) & ~1) / 2;
            if (lane == 0) {
                float *lo = reinterpret_cast<float *>(mine);
#pragma unroll
                for (int k = 0; k < TRI; ++k) lo[k] = r.Rlo[k];
#pragma unroll
                for (int k = 0; k < n; ++k) mine[LO + TRI + 2 * n + k] = r.elo[k];
            }
        }
    }
    ReducedArgs<P, COMP> r;
    double lconst;
    Bounds b;
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int, int) const
    {
        return logprob_row_reduced<P, COMP>(th, r, lconst, b);
    }
    // a copy of the kernarg operands in VECTOR registers (see BatchReducedLP::Local: as scalars
    // they do not fit and return lane by lane in every half-step).  The compensated tier holds 2.5x the
    // operands (low words) and ran out of vector registers instead (256 + 192 bytes of scratch per lane at
    // degree 5): its triangle R -- every use is one operand of a product -- stays in SCALAR registers, the
    // rest goes to vector registers: 225 VGPRs at degree 5, and the persistent sampler 1.2-1.8x faster at
    // degrees 5-8 (benchmarks/micro/persistent_comp_by_degree.py).
    struct Local { ReducedArgs<P, COMP> r; };
    __device__ __forceinline__ Local local(long long) const
    {
        Local loc{r};
        double *v = reinterpret_cast<double *>(&loc.r);
        constexpr int n = P + 2, TRI = n * (n + 1) / 2;
        constexpr int R0 = COMP ? ((TRI + 1) & ~1) / 2 : 0, R1 = R0 + TRI;     // where R lies in the struct, in doubles
#pragma unroll
        for (int i = 0; i < (int)(sizeof(ReducedArgs<P, COMP>) / sizeof(double)); ++i)
            if (!COMP || i < R0 || i >= R1) asm volatile("" : "+v"(v[i]));
        return loc;
    }
    template <bool STAGED>
    __device__ __forceinline__ double eval_ens(const double (&th)[NDIM], const Local &loc, int, const double *lds) const
    {
        if constexpr (COMP && STAGED) return logprob_row_reduced_staged<P>(th, loc.r, lds, lconst, b);
        else return logprob_row_reduced<P, COMP>(th, loc.r, lconst, b);
    }
};

// batch of spectra: operands of the spectrum that owns walker i (i / Wp)
template <class M, bool UNIFORM, int L_ = 1>
struct BatchGenericLP {
    static constexpr bool WAVE_IN_ONE_SPECTRUM = UNIFORM;     // (the persistent kernels are launched for these only)
    static constexpr int NDIM = M::NDIM;
    static constexpr int L = L_;
    static constexpr bool CAN_STAGE = true;
    static constexpr int REC_DOUBLES = M::REC;
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
    __device__ __forceinline__ const double *records(long long e) const { return cb + e * cb_stride; }
    __host__ __device__ __forceinline__ int n_freq() const { return N; }
    // the ensemble is the same for every lane of a wave (a wave never straddles two ensembles)
    struct Local { const double *rec; double lconst; };
    __device__ __forceinline__ Local local(long long e) const
    {
        const long long eu = (long long)__builtin_amdgcn_readfirstlane((int)e);
        return {cb + eu * cb_stride, lconst[eu]};
    }
    template <bool STAGED>
    __device__ __forceinline__ double eval_ens(const double (&th)[NDIM], const Local &loc, int g, const double *lds) const
    {
        const ModelOperands o{STAGED ? lds : loc.rec, N, loc.lconst};
        return logprob_row<M, L, STAGED>(th, o, b, g);
    }
};

template <int P, bool UNIFORM, bool COMP = false>
struct BatchReducedLP {
    static constexpr bool WAVE_IN_ONE_SPECTRUM = UNIFORM;
    static constexpr int NDIM = P + 2;
    static constexpr int L = 1;
    // The compensated tier's register copy of a spectrum's operands (Local) is 130 dwords: with the rest of
    // the kernel 256 VGPRs and 108 bytes of scratch per lane at degree 5.  Its low words -- Rlo as floats and
    // elo, 42 dwords, each read once per evaluation -- are therefore STAGED: the persistent kernel copies the
    // spectrum's operand image into LDS once per launch (the mechanism of the per-frequency models' records)
    // and the row sums read them there, every lane the same address.
    static constexpr bool CAN_STAGE = COMP;
    static constexpr int REC_DOUBLES = (int)(sizeof(ReducedArgs<P, COMP>) / sizeof(double));
    __device__ __forceinline__ const double *records(long long e) const { return reinterpret_cast<const double *>(red + e); }
    __host__ __device__ __forceinline__ int n_freq() const { return 1; }
    const ReducedArgs<P, COMP> *red;
    // COMP only: spectra whose tier[e] is 0 take the plain arithmetic on their plain operands (see BatchArgs)
    const ReducedArgs<P, false> *red_plain;
    const unsigned char *tier;
    long long Wp;
    const double *lconst;
    Bounds b;
    __device__ __forceinline__ bool plain_tier(long long e) const
    {
        if constexpr (COMP) return tier && !tier[e];
        else return false;
    }
    __device__ __forceinline__ double operator()(const double (&th)[NDIM], int walker, int) const
    {
        const long long e = spectrum_of<UNIFORM>(walker, Wp);
        if constexpr (COMP) {
            if (plain_tier(e)) return logprob_row_reduced<P, false>(th, red_plain[e], lconst[e], b);
        }
        return logprob_row_reduced<P, COMP>(th, red[e], lconst[e], b);
    }
    // The spectrum's triangle, expansion point and residual vector (50 doubles at P = 5) are read
    // ONCE per launch into registers: read where they lie they were ~25 vector loads per half-step,
    // each a trip to memory that the 130 instructions of a half-step cannot hide (SQ counters of the
    // 512 x 256 batch: 4,300 cycles per half-step, 1,900 of them waiting; benchmarks/micro/batch_pd_pmc.sh).
    // A spectrum on the plain tier of a compensated launch fills the same registers from its plain operands
    // (its expansion point differs) and leaves the low words unused.
    struct Local { ReducedArgs<P, COMP> r; double lconst; bool plain; };
    __device__ __forceinline__ Local local(long long e) const
    {
        // through the vector path on purpose: as wave-uniform values the 100 dwords would overflow the
        // scalar registers and come back lane by lane (v_readlane) in every half-step
        if constexpr (COMP) {
            if (plain_tier(e)) {
                Local loc;
                const ReducedArgs<P, false> &p = red_plain[e];
                constexpr int n = P + 2;
#pragma unroll
                for (int k = 0; k < n * (n + 1) / 2; ++k) { loc.r.R[k] = p.R[k]; loc.r.Rlo[k] = 0.0f; }
#pragma unroll
                for (int k = 0; k < n; ++k) { loc.r.bhat[k] = p.bhat[k]; loc.r.e[k] = p.e[k]; loc.r.elo[k] = 0.0; }
                loc.r.rest = p.rest;
                loc.lconst = lconst[e];
                loc.plain = true;
                return loc;
            }
        }
        // (the single-spectrum kernel keeps the compensated tier's triangle in scalar registers,
        // ReducedLP::local; here, with the batch's own scalar state, the same 56 values were spilled to
        // vector lanes and read back in the loop: measured no faster, left in vector registers)
        return {red[e], lconst[e], false};
    }
    template <bool STAGED>
    __device__ __forceinline__ double eval_ens(const double (&th)[NDIM], const Local &loc, int, const double *lds) const
    {
        if constexpr (COMP) {
            if (loc.plain) return logprob_row_reduced<P, false, ReducedArgs<P, true>>(th, loc.r, loc.lconst, b);
            if constexpr (STAGED) return logprob_row_reduced_staged<P>(th, loc.r, lds, loc.lconst, b);
        }
        return logprob_row_reduced<P, COMP>(th, loc.r, loc.lconst, b);
    }
};

// One stretch move: walker i (state row s, log-prob old_lp) along the line through row c.
// `walker` is the global walker id handed to the log-prob functor (batch contexts pick
// the spectrum from it).  Returns the row / log-prob AFTER the move.
// ENS = 0: the functor finds the walker's spectrum itself; 1 / 2: the caller (persistent kernel)
// hands over what it fetched for its ensemble (LP::Local), records where they are / staged in LDS.
template <int ENS = 0, class LP>
__device__ __forceinline__ bool stretch_move(const double *s_row, const double *c_row, double old_lp,
                                             double z, double factor, double logu, const LP &lp,
                                             int walker, int g, int *status, double (&row)[LP::NDIM],
                                             double &lp_row, const typename LP::Local *loc = nullptr,
                                             const double *lds_records = nullptr)
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
    double new_lp;
    if constexpr (ENS == 2) new_lp = lp.template eval_ens<true>(q, *loc, g, lds_records);
    else if constexpr (ENS == 1) new_lp = lp.template eval_ens<false>(q, *loc, g, nullptr);
    else new_lp = lp(q, walker, g);
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
    // a return-less atomic: `naccept[i] += 1` is a load the add and the store then wait for -- a third trip to
    // memory at the tail of every half-step launch (one lane per walker touches i: no contention either way)
    if (a.naccept && acc) atomicAdd(a.naccept + i, 1);
}

// pi^-1(y) = Ainv * (y - B) mod W for y, B in [0, W), Ainv in [1, W) (the philox contract: the comment at DrawArgs).
// y - B needs no division at all, and for W <= 65536 (every ensemble that is not one huge one) the product fits
// 32 bits.  Same integers either way.
__device__ __forceinline__ int perm_inverse(long long y, long long W, long long Ainv, long long B)
{
    long long v = y - B;
    if (v < 0) v += W;
    if (W <= 65536) return (int)(((unsigned)Ainv * (unsigned)v) % (unsigned)W);
    // Ainv v < 2^62: the quotient estimated in double (relative error 3 x 2^-53 of a number < 2^31: off by one at
    // most, and only next to an integer), the remainder set right -- the same integer as (Ainv v) % W without the
    // ~100 instructions of a 64-bit division (two per slot were half of the draw kernel's time on a big ensemble)
    const unsigned long long prod = (unsigned long long)Ainv * (unsigned long long)v;
    const unsigned q = (unsigned)((double)prod * (1.0 / (double)W));
    long long r = (long long)(prod - (unsigned long long)q * (unsigned long long)W);
    if (r < 0) r += W;
    if (r >= W) r -= W;
    return (int)r;
}

// the per-slot draw of the philox contract
struct SlotDraw {
    int active, partner;  // walker indices inside the ensemble
    double z, factor, logu;
};

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
    d.active = perm_inverse(2 * t + h, W, A_inv, B);
    d.partner = perm_inverse(2 * r + (1 - h), W, A_inv, B);
    d.factor = ndim_m1 * log(d.z);
    d.logu = log(u53(r1.v[0], r1.v[1]));
    return d;
}

// single-rank half-step: evaluate every slot and update the state in place.
// LP::L lanes per slot; a wave holds 64/L slots.  When a wave's last slots do not exist the
// lanes still run (clamped to the last slot) so the wavefront exchanges stay uniform; only
// lane 0 of a live slot commits.  All lanes of a wave read their rows before any commits
// (same instruction stream), and no other wave touches them.
template <class LP, int BLK = 64>
__global__ __launch_bounds__(BLK) void k_stretch_half(const StretchArgs a, const LP lp)
{
    constexpr int L = LP::L;
    const long long tid = (long long)blockIdx.x * BLK + threadIdx.x;
    const long long slot = tid / L;
    const int g = (int)(tid % L);
    const bool live = slot < a.n_slots;
    const long long t = live ? slot : a.n_slots - 1;
    int i;
    double row[LP::NDIM], lp_row;
    const bool acc = stretch_slot(a, lp, t, g, i, row, lp_row);
    if (live && g == 0) commit_row<LP::NDIM>(a, i, row, lp_row, acc);
}

// The half-step of an ensemble that fills the chip with one lane per slot (131,072 walkers and more) on a PACKED
// state.  With rows of 8 NDIM bytes at random addresses and the log-probabilities in an array of their own, a
// proposal pulls a line or two for its walker's row, as many for its partner's, one more for 8 bytes of
// log-probability, and writes the same way: FETCH_SIZE 242 B + WRITE_SIZE 98 B per proposal for 216 algorithmic
// bytes, and the launch is bound by exactly that -- scattered lines (PolynomialDecomposition and double Cole-Cole
// take the same 65 us per 524,288 proposals).  bisip_stretch_run_dev therefore keeps the state of such a chunk as one
// aligned 64-byte row per walker -- theta[0..NDIM), padding, the log-probability in [7] (NDIM <= 7) -- packed once
// before the chunk's first half-step and unpacked after its last: a walker and its log-probability are ONE line
// to read and ONE full line to write, its partner one line.  Same operations on the same operands as
// k_stretch_half: the chain does not change by a bit.
constexpr int PACKED_ROW = 8;

template <class LP, int BLK = 256>
__global__ __launch_bounds__(BLK) void k_stretch_half_packed(const StretchArgs a, const LP lp)
{
    static_assert(LP::L == 1 && LP::NDIM < PACKED_ROW, "one lane per slot, a walker in one 64-byte row");
    constexpr int NDIM = LP::NDIM;
    const long long slot = (long long)blockIdx.x * BLK + threadIdx.x;
    const bool live = slot < a.n_slots;
    const long long t = live ? slot : a.n_slots - 1;
    int i, p;
    double z, factor, logu;
    if (a.perm) {      // the slot's entries from their Philox counters: nothing of the stream touches memory
        const SlotDraw d = draw_slot(a.draw_W, a.draw_a, a.draw_ndim_m1, a.seed_lo, a.seed_hi, a.draw_step, a.draw_h,
                                     a.draw_e, t, a.perm[1], a.perm[2]);
        i = d.active; p = d.partner; z = d.z; factor = d.factor; logu = d.logu;
    } else {
        i = __builtin_nontemporal_load(a.active + t); p = __builtin_nontemporal_load(a.partner + t);
    }
    const dbl2 *srow = reinterpret_cast<const dbl2 *>(a.packed + (long long)i * PACKED_ROW);
    const dbl2 *crow = reinterpret_cast<const dbl2 *>(a.packed + (long long)p * PACKED_ROW);
    dbl2 sv[4], cv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { sv[q] = srow[q]; cv[q] = crow[q]; }
    if (!a.perm) {
        z = __builtin_nontemporal_load(a.zz + t); factor = __builtin_nontemporal_load(a.factor + t);
        logu = __builtin_nontemporal_load(a.logu + t);
    }
    double s_row[PACKED_ROW], c_row[PACKED_ROW];
#pragma unroll
    for (int q = 0; q < 4; ++q) { s_row[2 * q] = sv[q].x; s_row[2 * q + 1] = sv[q].y; c_row[2 * q] = cv[q].x; c_row[2 * q + 1] = cv[q].y; }
    double row[NDIM], lp_row;
    const bool acc = stretch_move(s_row, c_row, s_row[PACKED_ROW - 1], z, factor, logu, lp, i, 0, a.status, row, lp_row);
    if (!live) return;
    if (acc) {
        double r[PACKED_ROW];
#pragma unroll
        for (int q = 0; q < PACKED_ROW; ++q) r[q] = q < NDIM ? row[q] : 0.0;
        r[PACKED_ROW - 1] = lp_row;
        dbl2 *dst = reinterpret_cast<dbl2 *>(a.packed + (long long)i * PACKED_ROW);
#pragma unroll
        for (int q = 0; q < 4; ++q) { dbl2 v; v.x = r[2 * q]; v.y = r[2 * q + 1]; dst[q] = v; }
        if (a.naccept) atomicAdd(a.naccept + i, 1);
    }
    if (a.chain_row) {
#pragma unroll
        for (int k = 0; k < NDIM; ++k) __builtin_nontemporal_store(row[k], a.chain_row + (long long)i * NDIM + k);
    }
    if (a.logp_row) __builtin_nontemporal_store(lp_row, a.logp_row + i);
}

// (W, NDIM) + (W,) -> (W, 8) and back (a chunk's first and last launch): a tile of 256 walkers goes through LDS so that
// both sides are read and written as whole lines by consecutive lanes -- a lane per walker walking its own row
// (stride 8 NDIM bytes between lanes) moved 128 MB in 103 / 91 us at a million walkers.
template <bool PACK>
__global__ __launch_bounds__(256) void k_state_repack(double *coords, double *logp, double *packed, long long W, int ndim)
{
    __shared__ __attribute__((aligned(16))) double tile[256 * PACKED_ROW];
    const long long row0 = (long long)blockIdx.x * 256;
    const int rows = (int)(W - row0 < 256 ? W - row0 : 256);
    const int tid = threadIdx.x;
    dbl2 *tile2 = reinterpret_cast<dbl2 *>(tile);
    dbl2 *packed2 = reinterpret_cast<dbl2 *>(packed + row0 * PACKED_ROW);
    if (PACK) {
        for (int idx = tid; idx < rows * ndim; idx += 256) {
            const int w = idx / ndim, q = idx - w * ndim;
            tile[w * PACKED_ROW + q] = coords[row0 * ndim + idx];
        }
        if (tid < rows) {
            for (int q = ndim; q < PACKED_ROW - 1; ++q) tile[tid * PACKED_ROW + q] = 0.0;
            tile[tid * PACKED_ROW + PACKED_ROW - 1] = logp[row0 + tid];
        }
        __syncthreads();
        for (int j = tid; j < rows * (PACKED_ROW / 2); j += 256) packed2[j] = tile2[j];
    } else {
        for (int j = tid; j < rows * (PACKED_ROW / 2); j += 256) tile2[j] = packed2[j];
        __syncthreads();
        for (int idx = tid; idx < rows * ndim; idx += 256) {
            const int w = idx / ndim, q = idx - w * ndim;
            coords[row0 * ndim + idx] = tile[w * PACKED_ROW + q];
        }
        if (tid < rows) logp[row0 + tid] = tile[tid * PACKED_ROW + PACKED_ROW - 1];
    }
}

// (What bounds it, round 5: the memory system's rate of RANDOM 64-byte lines.  Parts of a half-step of 524,288 proposals
// taken out one at a time (1,048,576 walkers, PolynomialDecomposition, stream drawn in place: 31 us): rows at
// sequential instead of random places -13 us, no row written -4, no counter bumped -3, no draw -1; the times ADD UP
// and do not depend on the occupancy (2 ... 6 waves per SIMD, limited through LDS), i.e. every part is a share of one
// resource's time.  benchmarks/micro/random_lines.hip asks the chip for the bare pattern: two random rows read per
// proposal 20.6 us, and one written 25.8 -- the half-step is at 0.85-0.9 of that.  Two slots per lane, software-
// pipelined (the second slot's rows travelling while the first is evaluated): 31.7 us, double Cole-Cole 40.7 -> 47.7;
// eight waves per SIMD forced (spills): 33.9 / 126.  Not kept.)
// (Measured and not kept, round 5: the rows of a chip-filling launch fetched and stored by FOUR lanes each through
// LDS, 16 bytes per lane, instead of one lane per row -- same chain; 524,288 proposals of a 1,048,576-walker
// ensemble: PolynomialDecomposition 65.2 -> 63.4 us per half-step, double Cole-Cole 66.3 -> 73.1.  Both models take the
// same 65 us: the half-step is bound by the memory system's rate of SCATTERED lines, not by requests per
// instruction and not by arithmetic -- FETCH_SIZE 127 MB + WRITE_SIZE 51 MB per launch for 113 MB of algorithmic
// bytes: a 56-byte row at a random address costs one or two 64-byte lines each way, an 8-byte log-probability a
// line of its own.  profiles/r05_micro_ab_big_ensemble_rows.jsonl.)

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
    long long e0;                     // index of the context's first spectrum in the whole survey
    int flat;                         // launch shape, see k_stretch_draw
    double a, ndim_m1;
    unsigned int seed_lo, seed_hi;
    const int *perm;  // (n_steps, 3): A, Ainv, B
    int *active, *partner;
    double *zz, *factor, *logu;
};

// Grid: x covers the E*nh slots of one half-step, (y, z) the 2*n_steps half-steps -- the index
// arithmetic of a flat 64-bit grid was four 64-bit divisions per slot.  `flat`: one 32-bit index over
// everything (two 32-bit divisions), for ensembles too small to fill a workgroup per half-step.
static __global__ __launch_bounds__(256) void k_stretch_draw(const DrawArgs d)
{
    const long long per_half = d.E * d.nh;
    long long kh, slot;                     // iteration * 2 + half; (ensemble, t)
    if (d.flat) {                           // small ensembles: a half-step would not fill a workgroup
        const unsigned idx32 = blockIdx.x * 256u + threadIdx.x;
        const unsigned q = idx32 / (unsigned)per_half;
        kh = q; slot = idx32 - q * (unsigned)per_half;
    } else {
        kh = (long long)blockIdx.y + (long long)gridDim.y * blockIdx.z;
        slot = (long long)blockIdx.x * 256 + threadIdx.x;
    }
    if (kh >= 2 * d.n_steps || slot >= per_half) return;
    const long long k = kh >> 1;
    const int h = (int)(kh & 1);
    long long e, t;
    if (per_half <= 0xffffffffLL) {
        const unsigned q = (unsigned)slot / (unsigned)d.nh;
        e = q; t = (long long)((unsigned)slot - q * (unsigned)d.nh);
    } else {
        e = slot / d.nh; t = slot % d.nh;
    }
    const long long idx = kh * per_half + slot;
    const long long Ns = h ? d.W / 2 : (d.W + 1) / 2;
    if (t >= Ns) {  // padding slot of the smaller half
        d.active[idx] = 0; d.partner[idx] = 0; d.zz[idx] = 1.0; d.factor[idx] = 0.0; d.logu[idx] = 0.0;
        return;
    }
    const SlotDraw s = draw_slot(d.W, d.a, d.ndim_m1, d.seed_lo, d.seed_hi, (unsigned)(d.step0 + k), h,
                                 (unsigned)(d.e0 + e), t, d.perm[3 * k + 1], d.perm[3 * k + 2]);
    d.active[idx] = (int)(e * d.W) + s.active;
    d.partner[idx] = (int)(e * d.W) + s.partner;
    d.zz[idx] = s.z;
    d.factor[idx] = s.factor;
    d.logu[idx] = s.logu;
}

// ---------------------------------------------------------------------------------
// Persistent sampler: ONE WORKGROUP PER ENSEMBLE runs every iteration of a chunk inside
// one launch.  The ensemble (positions + log-probs) lives in LDS; the random stream is the
// same pre-drawn (n_steps, 2, E, nh) arrays the launch-per-half-step kernels read (NumPy
// order or k_stretch_draw), so both stream modes share this path; the two half-steps of an
// iteration are separated by a workgroup barrier (no launch, no grid sync) and only the
// stored chain rows go to HBM.  LP::L lanes share a slot exactly as in k_stretch_half, and
// the next half-step's stream entries are requested before the current one is evaluated,
// so the critical path of a half-step is LDS -> proposal -> log-prob -> LDS -> barrier.
// Bit-identical to the launch-per-half-step path.
// Needs W*(NDIM+1)*8 B of LDS and ceil(W/2)*L <= 1024 lanes.
// ---------------------------------------------------------------------------------
struct PersistArgs {
    double *coords;      // (E*W, NDIM) in/out (global state)
    double *logp;        // (E*W,)
    long long W, E;      // walkers per ensemble, ensembles
    long long n_steps, thin_by;
    const int *active, *partner;            // (n_steps, 2, E, nh): global walker ids
    const double *zz, *factor, *logu;
    double *chain;       // (n_steps/thin_by, E*W, NDIM) or null
    double *logp_chain;  // (n_steps/thin_by, E*W) or null
    int *naccept;        // (E*W,) or null
    int *status;
    // launch shape (dispatch_stretch.hip): a workgroup holds `epw` ensembles of `lanes_per_ens`
    // lanes each (a multiple of 64, so no wave straddles two ensembles)
    int lanes_per_ens, epw;
    long long rec_stride;   // STAGED: doubles between two ensembles' record copies in LDS (even)
    // k_stretch_group (one ensemble over several workgroups): its state in memory, 8 doubles per walker, its
    // synchronisation words (see there), workgroups in the group, polls before a barrier gives up
    double *gstate;
    unsigned *gsync;
    int G;
    unsigned spin_limit;
    int spread;          // the group is ALL workgroups of the grid (any XCD) instead of its every eighth
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global access (s_waitcnt vmcnt(0)): here that would put the acknowledgement of
// the chain-row stores and the prefetched stream loads on the critical path of every
// half-step, and nothing after the barrier depends on them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct SlotStream {
    int active, partner;
    double z, factor, logu;
};

__device__ __forceinline__ SlotStream load_slot(const PersistArgs &a, long long idx)
{
    SlotStream d;
    d.active = a.active[idx]; d.partner = a.partner[idx];
    d.z = a.zz[idx]; d.factor = a.factor[idx]; d.logu = a.logu[idx];
    return d;
}

// Workgroup shape.  Many ensembles (a batch of spectra): workgroups of >= 256 lanes, i.e. several
// ensembles each -- the dispatcher spreads four-wave workgroups one per CU with one wave per SIMD,
// whereas 512 two-wave workgroups on 256 CUs land unevenly (measured 1.6x slower,
// benchmarks/micro/issue_latency.hip).  The ensembles of a workgroup share nothing but the
// barrier; lanes of a missing last ensemble shadow ensemble E-1 and never commit.
// STAGED: the frequency records of each ensemble's spectrum are copied into LDS once per launch
// and logprob_row reads them from there (software-pipelined, kernels.h): a scalar load costs
// 250-500 cycles even on a scalar-cache hit, and with one wave per SIMD nothing hides it.
// At most 512 lanes per workgroup: two waves per SIMD leave each lane 256 VGPRs (the pipelined
// log-probability wants ~150; under a 1024-lane bound the compiler must stay below 128 and spills).
template <class LP, bool STAGED>
__global__ __launch_bounds__(512) void k_stretch_persistent(const PersistArgs a, const LP lp)
{
    constexpr int NDIM = LP::NDIM;
    constexpr int L = LP::L;
    extern __shared__ __attribute__((aligned(16))) double lds_state[];  // per ensemble: W*NDIM coords | W logp
    const int ens_local = threadIdx.x / a.lanes_per_ens;
    const int lane = threadIdx.x - ens_local * a.lanes_per_ens;
    const long long e_raw = (long long)blockIdx.x * a.epw + ens_local;
    const bool ens_live = e_raw < a.E;
    const long long e = ens_live ? e_raw : a.E - 1;
    const long long state_doubles = a.W * (NDIM + 1);
    double *xs = lds_state + ens_local * state_doubles;
    double *ls = xs + a.W * NDIM;
    const long long base = e * a.W;           // first global walker of this ensemble
    for (long long i = lane; i < a.W * NDIM; i += a.lanes_per_ens) xs[i] = a.coords[base * NDIM + i];
    for (long long i = lane; i < a.W; i += a.lanes_per_ens) ls[i] = a.logp[base + i];
    const double *recs = nullptr;
    if constexpr (STAGED) {
        double *mine = lds_state + ((a.epw * state_doubles + 1) & ~1LL) + ens_local * a.rec_stride;
        if constexpr (StagesFromArgs<LP>::value) lp.stage_from_args(mine, lane);
        else {
            const double *__restrict__ src = lp.records(e);
            const int n = lp.n_freq() * LP::REC_DOUBLES;
            for (int i = lane; i < n; i += a.lanes_per_ens) mine[i] = src[i];
        }
        recs = mine;
    }
    __syncthreads();
    const long long nh = (a.W + 1) / 2;       // slots of half 0; half 1 has W/2
    const long long slot = lane / L;
    const int g = lane % L;
    // lanes past the last slot of a half run clamped to it (the wavefront exchanges of
    // L > 1 need whole groups) and never commit
    const long long t0 = slot < nh ? slot : nh - 1;
    const long long t1 = slot < a.W / 2 ? slot : a.W / 2 - 1;
    const typename LP::Local loc = lp.local(e);
    SlotStream cur = load_slot(a, (0 * a.E + e) * nh + t0);
    // which iterations are stored, by counting: (k + 1) % thin_by and k / thin_by are two 64-bit
    // divisions per iteration, a few hundred instructions on a path where one wave per SIMD waits
    // for every one of them
    long long srow = 0, until_store = a.thin_by;
    for (long long k = 0; k < a.n_steps; ++k) {
        const bool store = --until_store == 0;
        for (int h = 0; h < 2; ++h) {
            // request the next half-step's entries now; they are consumed after the barrier
            const long long kn = h ? (k + 1 < a.n_steps ? k + 1 : k) : k;
            const SlotStream nxt = load_slot(a, ((kn * 2 + (1 - h)) * a.E + e) * nh + (h ? t0 : t1));
            const bool live = ens_live && slot < (h ? a.W / 2 : nh);
            const int i = cur.active - (int)base, p = cur.partner - (int)base;
            double row[NDIM], lp_row;
            // every lane of a slot reads the rows before lane 0 writes (same wave, program order)
            const bool acc = stretch_move<STAGED ? 2 : 1>(xs + (long long)i * NDIM, xs + (long long)p * NDIM, ls[i], cur.z,
                                                          cur.factor, cur.logu, lp, (int)(base + i), g, a.status, row,
                                                          lp_row, &loc, recs);
            if (live && g == 0) {
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
            cur = nxt;
            lds_barrier();     // the other half reads the rows just written
        }
        if (store) { ++srow; until_store = a.thin_by; }
    }
    if (ens_live) {
        for (long long i = lane; i < a.W * NDIM; i += a.lanes_per_ens) a.coords[base * NDIM + i] = xs[i];
        for (long long i = lane; i < a.W; i += a.lanes_per_ens) a.logp[base + i] = ls[i];
    }
}

// ---------------------------------------------------------------------------------
// Persistent sampler for ONE ensemble too big for a workgroup (1,024 < W <= 8,192 walkers; BASELINE config 2's
// 4,096 sit here): G workgroups run every iteration of a chunk inside one launch and meet at a barrier of their
// own after every half-step.  A half-step of such an ensemble is 0.1-0.5 us of arithmetic; as a launch of its own
// it costs 5-10 us (launch gap, ramp, two dependent trips to memory).
//
// State: one 64-byte row per walker in memory -- theta[0..NDIM), padding, the log-probability in [7] -- so that a
// slot fetches its walker and its partner with two independent line reads and commits with one line write.
// Slot t of a half-step belongs to L adjacent lanes of workgroup t / (256 / L), as in k_stretch_half; the random
// stream is the pre-drawn one of every other driver; the chain is bit-identical to theirs.
//
// The barrier and the hand-off (benchmarks/micro/xcd_barrier.hip measured both; MI355X_MICROARCH.md, "inter-workgroup
// visibility"): the group is the workgroups with blockIdx.x % 8 == 0 of a grid of 8 G -- the dispatcher deals
// blocks round-robin over the 8 XCDs, so the group USUALLY shares one XCD and one L2.  That is observed, not
// promised, so it is checked, per launch: every workgroup reports its HW_REG_XCC_ID before the first barrier.
//   * placement-independent protocol (always correct; the first barrier, and every barrier unless all ids agree):
//     rows stored write-through (sc1) and drained, workgroup barrier, ONE lane adds to an agent-scope counter and
//     polls it with sc1 loads, workgroup barrier, rows loaded with sc1 (past the L1).
//   * one-XCD protocol (all ids equal): plain row stores -- the lines stay in the XCD's L2 -- drained, the counter
//     added to at workgroup scope (executes in that L2), the same sc1 poll and sc1 row loads: 1.35 us against 2.4
//     per half-step at 2,048 slots.
// Every poll loop is bounded (spin_limit): if the workgroups cannot all be resident (a GPU shared with another
// process) each gives up, sets status bit 2 and ends; the host then refuses the chunk.
// ---------------------------------------------------------------------------------
constexpr int GROUP_ROW = 8;            // doubles per state row: one 64-byte line
constexpr int GROUP_SYNC_WORDS = 576;   // the group's synchronisation words (see the kernel), ahead of its state rows
constexpr int GROUP_BLK = 256;

__device__ __forceinline__ void group_store_row(double *p, const dbl2 (&v)[4], bool through)
{
    if (through) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
                     "global_store_dwordx4 %0, %3, off offset:32 sc1\n\tglobal_store_dwordx4 %0, %4, off offset:48 sc1"
                     :: "v"(p), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory");
    } else {
        asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\t"
                     "global_store_dwordx4 %0, %3, off offset:32\n\tglobal_store_dwordx4 %0, %4, off offset:48"
                     :: "v"(p), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory");
    }
}

// two rows (a slot's walker and its partner), all eight loads in flight, past the L1
__device__ __forceinline__ void group_load_rows(const double *ps, const double *pc, dbl2 (&s)[4], dbl2 (&c)[4])
{
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %9, off sc1\n\tglobal_load_dwordx4 %5, %9, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %6, %9, off offset:32 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:48 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                 : "v"(ps), "v"(pc) : "memory");
}

// all workgroups of the group have arrived `target` times in all; false: gave up waiting
__device__ __forceinline__ bool group_barrier(unsigned *counter, unsigned target, bool one_xcd, unsigned limit, int *flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's state rows have left
    __syncthreads();
    if (threadIdx.x == 0) {
        if (one_xcd) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > limit) { ok = 0; break; }
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}

// The same barrier for a group spread over several XCDs, in two levels: a workgroup arrives at ITS XCD's counter
// (workgroup scope: the read-modify-write executes in that XCD's L2, which only its own workgroups touch), the last
// of the XCD's `mine` workgroups to arrive passes the round on to the group's counter (agent scope), and everybody
// polls that one.  One counter for all serialises every arrival at the memory side: 64 workgroups 1.1 us, 256 (cfg4:
// 32,768 walkers, one wave per compute unit) 4-5 us; two levels: 8 arrivals there, whatever the group's size.
// The counters only grow: round r is over when the group's counter reads r * xcds.
__device__ __forceinline__ bool group_barrier_two_level(unsigned *local, unsigned mine, unsigned *global, unsigned xcds,
                                                        unsigned round, unsigned limit, int *flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's state rows have left (write-through)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned before = __hip_atomic_fetch_add(local, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (before + 1u == round * mine) __hip_atomic_fetch_add(global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = round * xcds;
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(global, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > limit) { ok = 0; break; }
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}

__device__ __forceinline__ int group_xcc_id()
{
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}

// BLK lanes per workgroup: 256 where a slot has several lanes; 64 (one wave) where it has ONE -- the reduced
// PolynomialDecomposition kernels: 2,048 slots are then 32 workgroups on 32 compute units instead of 8 on 8, and the
// gather of their rows, which each compute unit's memory pipeline serialises, takes 0.5 us instead of 1.5 (in-kernel
// timers, 4,096 walkers: gather 1.48, evaluation 0.52, commit 0.14, barrier 1.41 with 8 workgroups of 256 lanes).
template <class LP, bool STAGED, int BLK = GROUP_BLK>
__global__ __launch_bounds__(BLK) void k_stretch_group(const PersistArgs a, const LP lp)
{
    constexpr int NDIM = LP::NDIM;
    constexpr int L = LP::L;
    static_assert(NDIM < GROUP_ROW, "a state row holds theta and the log-probability in 8 doubles");
    if (!a.spread && blockIdx.x % 8 != 0) return;         // the group: workgroups 0, 8, 16, ... of the grid
    extern __shared__ __attribute__((aligned(16))) double lds_records[];
    __shared__ int flag;
    __shared__ unsigned xcd_mine, xcd_count;
    const int member = a.spread ? blockIdx.x : blockIdx.x / 8;
    const int tid = threadIdx.x;
    // sync words (one 64-byte line each): [0] first barrier, [16] the half-steps' barriers, [32] min XCC id, [33] max,
    // [64 + 16 x] workgroups of the group on XCD x, [320 + 16 x] that XCD's arrivals (GROUP_SYNC_WORDS in all)
    unsigned *first = a.gsync, *rounds = a.gsync + 16, *xmin = a.gsync + 32, *xmax = a.gsync + 33;
    unsigned *members = a.gsync + 64, *arrivals = a.gsync + 320;
    const unsigned my_xcd = (unsigned)group_xcc_id();
    const double *recs = nullptr;
    if constexpr (STAGED) {
        if constexpr (StagesFromArgs<LP>::value) lp.stage_from_args(lds_records, tid);
        else {
            const double *__restrict__ src = lp.records(0);
            const int n = lp.n_freq() * LP::REC_DOUBLES;
            for (int i = tid; i < n; i += BLK) lds_records[i] = src[i];
        }
        recs = lds_records;
    }
    // the state into its rows (write-through: any placement may read them), then the first barrier
    for (long long w = (long long)member * BLK + tid; w < a.W; w += (long long)a.G * BLK) {
        dbl2 v[4];
        double r[GROUP_ROW];
#pragma unroll
        for (int q = 0; q < GROUP_ROW; ++q) r[q] = q < NDIM ? a.coords[w * NDIM + q] : 0.0;
        r[GROUP_ROW - 1] = a.logp[w];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q].x = r[2 * q]; v[q].y = r[2 * q + 1]; }
        group_store_row(a.gstate + w * GROUP_ROW, v, true);
    }
    if (tid == 0) {
        __hip_atomic_fetch_min(xmin, my_xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(xmax, my_xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(members + 16 * my_xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool ok = group_barrier(first, (unsigned)a.G, false, a.spin_limit, &flag);
    const bool one_xcd = __hip_atomic_load(xmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                         __hip_atomic_load(xmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {      // who shares my XCD, and how many XCDs hold a part of the group (the two-level barrier)
        unsigned n = 0;
        for (int x = 0; x < 16; ++x) n += __hip_atomic_load(members + 16 * x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        xcd_count = n;
        xcd_mine = __hip_atomic_load(members + 16 * my_xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned mine = xcd_mine, xcds = xcd_count;
    const long long nh = (a.W + 1) / 2;
    constexpr int SPW = BLK / L;                    // slots per workgroup
    const long long slot = (long long)member * SPW + tid / L;
    const int g = tid % L;
    const long long t0 = slot < nh ? slot : nh - 1;
    const long long t1 = slot < a.W / 2 ? slot : a.W / 2 - 1;
    const typename LP::Local loc = lp.local(0);
    SlotStream cur = load_slot(a, t0);
    unsigned round = 0;
    long long srow = 0, until_store = a.thin_by;
    for (long long k = 0; ok && k < a.n_steps; ++k) {
        const bool store = --until_store == 0;
        for (int h = 0; h < 2; ++h) {
            const bool live = slot < (h ? a.W / 2 : nh);
            const int i = cur.active, p = cur.partner;
            dbl2 sv[4], cv[4];
            group_load_rows(a.gstate + (long long)i * GROUP_ROW, a.gstate + (long long)p * GROUP_ROW, sv, cv);
            // the next half-step's entries travel while this one is evaluated
            const long long kn = h ? (k + 1 < a.n_steps ? k + 1 : k) : k;
            const SlotStream nxt = load_slot(a, (kn * 2 + (1 - h)) * nh + (h ? t0 : t1));
            double s_row[GROUP_ROW], c_row[GROUP_ROW];
#pragma unroll
            for (int q = 0; q < 4; ++q) { s_row[2 * q] = sv[q].x; s_row[2 * q + 1] = sv[q].y; c_row[2 * q] = cv[q].x; c_row[2 * q + 1] = cv[q].y; }
            double row[NDIM], lp_row;
            const bool acc = stretch_move<STAGED ? 2 : 1>(s_row, c_row, s_row[GROUP_ROW - 1], cur.z, cur.factor, cur.logu, lp, i, g,
                                                          a.status, row, lp_row, &loc, recs);
            if (live && g == 0) {
                if (acc) {
                    dbl2 v[4];
                    double r[GROUP_ROW];
#pragma unroll
                    for (int q = 0; q < GROUP_ROW; ++q) r[q] = q < NDIM ? row[q] : 0.0;
                    r[GROUP_ROW - 1] = lp_row;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q].x = r[2 * q]; v[q].y = r[2 * q + 1]; }
                    group_store_row(a.gstate + (long long)i * GROUP_ROW, v, !one_xcd);
                    if (a.naccept) atomicAdd(a.naccept + i, 1);
                }
                if (store && a.chain) {
                    double *cr = a.chain + (srow * a.W + i) * NDIM;
#pragma unroll
                    for (int q = 0; q < NDIM; ++q) __builtin_nontemporal_store(row[q], cr + q);
                }
                if (store && a.logp_chain) __builtin_nontemporal_store(lp_row, a.logp_chain + srow * a.W + i);
            }
            cur = nxt;
            ++round;
            ok = one_xcd ? group_barrier(rounds, round * (unsigned)a.G, true, a.spin_limit, &flag)
                         : group_barrier_two_level(arrivals + 16 * my_xcd, mine, rounds, xcds, round, a.spin_limit, &flag);
            if (!ok) break;
        }
        if (store) { ++srow; until_store = a.thin_by; }
    }
    if (!ok) {
        if (tid == 0) atomicOr(a.status, 4);
        return;
    }
    for (long long w = (long long)member * BLK + tid; w < a.W; w += (long long)a.G * BLK) {
        dbl2 v[4], dummy[4];
        group_load_rows(a.gstate + w * GROUP_ROW, a.gstate + w * GROUP_ROW, v, dummy);
        double r[GROUP_ROW];
#pragma unroll
        for (int q = 0; q < 4; ++q) { r[2 * q] = v[q].x; r[2 * q + 1] = v[q].y; }
#pragma unroll
        for (int q = 0; q < NDIM; ++q) a.coords[w * NDIM + q] = r[q];
        a.logp[w] = r[GROUP_ROW - 1];
    }
}

}  // namespace bisip

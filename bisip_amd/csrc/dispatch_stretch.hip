// dispatch_stretch.hip -- stretch-move launches: half-step / eval / apply / persistent.
#include "host.h"

#include <cstdlib>

using namespace bisip;
using namespace bisip::host;

namespace {

// Persistent kernel: workgroup shape and LDS plan (see k_stretch_persistent).
template <class LP>
int launch_persistent(const PersistArgs &p0, const LP &lp, hipStream_t st)
{
    PersistArgs p = p0;
    const long long nh = (p.W + 1) / 2;
    p.lanes_per_ens = (int)(((nh * LP::L + 63) / 64) * 64);          // <= 512: stretch_lanes()
    const size_t state_bytes = (size_t)p.W * (LP::NDIM + 1) * sizeof(double);
    size_t rec_bytes = 0;
    if constexpr (LP::CAN_STAGE) {
        static const bool off = std::getenv("BISIP_NO_LDS_STAGING") != nullptr;   // A/B runs
        rec_bytes = off ? 0 : (((size_t)lp.n_freq() * LP::REC_DOUBLES + 1) & ~(size_t)1) * sizeof(double);
        if (state_bytes + rec_bytes + 16 > 65536) rec_bytes = 0;                  // does not fit: scalar-cache path
    }
    p.rec_stride = (long long)(rec_bytes / sizeof(double));
    // ensembles per workgroup: up to 256 lanes per workgroup when there are many ensembles
    long long epw = 1;
    if (p.E > 1) {
        epw = 256 / p.lanes_per_ens;                     // (and so never above the 512-lane bound)
        if (epw < 1) epw = 1;
        while (epw > 1 && epw * (state_bytes + rec_bytes) + 16 > 65536) --epw;
        if (epw > p.E) epw = p.E;
    }
    p.epw = (int)epw;
    const unsigned threads = (unsigned)(epw * p.lanes_per_ens);
    const unsigned grid = (unsigned)((p.E + epw - 1) / epw);
    const size_t lds = ((size_t)epw * state_bytes + 15) / 16 * 16 + (size_t)epw * rec_bytes;
    if constexpr (LP::CAN_STAGE) {
        if (rec_bytes) {
            hipLaunchKernelGGL((k_stretch_persistent<LP, true>), dim3(grid), dim3(threads), lds, st, p, lp);
            HIP_TRY(hipGetLastError());
            return BISIP_OK;
        }
    }
    hipLaunchKernelGGL((k_stretch_persistent<LP, false>), dim3(grid), dim3(threads), lds, st, p, lp);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

template <class LP>
int launch_stretch(const StretchWork &work, const LP &lp, hipStream_t st)
{
    const StretchKind kind = work.kind;
    if (kind == STRETCH_PERSIST) return launch_persistent(*work.persist, lp, st);
    const StretchArgs &a = *work.half;
    if (kind == STRETCH_HALF) {
        // a launch that fills the chip (>= one wave per SIMD) goes out as four-wave workgroups, one
        // per CU; smaller ones as single waves so that they spread over as many CUs as possible
        if (a.n_slots * LP::L >= 65536) {
            const unsigned grid = (unsigned)((a.n_slots * LP::L + 255) / 256);
            hipLaunchKernelGGL((k_stretch_half<LP, 256>), dim3(grid), dim3(256), 0, st, a, lp);
        } else {
            const unsigned grid = (unsigned)((a.n_slots * LP::L + 63) / 64);
            hipLaunchKernelGGL((k_stretch_half<LP, 64>), dim3(grid), dim3(64), 0, st, a, lp);
        }
    } else {
        const unsigned grid = (unsigned)(((a.slot_hi - a.slot_lo) * LP::L + 63) / 64);
        hipLaunchKernelGGL((k_stretch_eval<LP>), dim3(grid), dim3(64), 0, st, a, lp);
    }
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

// lanes per slot of a stretch dispatch: as many as lanes_per_walker() grants for the number of
// slots evaluated at once (all ensembles' for the persistent kernel, whose workgroups run
// concurrently), capped there by the 512-lane workgroup that holds one ensemble's half.
int stretch_lanes(const StretchWork &w)
{
    // tuning knob for benchmarks/: BISIP_STRETCH_LANES=1|2|4 overrides the rule below (the value
    // never changes a result -- logprob_row is bit-identical for every L -- only the wave count)
    if (const char *env = std::getenv("BISIP_STRETCH_LANES")) {
        const int v = std::atoi(env);
        if (v == 1 || v == 2 || v == 4) {
            if (w.kind != STRETCH_PERSIST) return v;
            const long long nh = (w.persist->W + 1) / 2;
            return nh * v <= 512 ? v : (nh * 2 <= 512 ? 2 : 1);
        }
    }
    if (w.kind == STRETCH_PERSIST) {
        const long long nh = (w.persist->W + 1) / 2;
        const int fit = nh * 4 <= 512 ? 4 : (nh * 2 <= 512 ? 2 : 1);
        const int want = lanes_per_walker(nh * w.persist->E);
        return want < fit ? want : fit;
    }
    return lanes_per_walker(w.kind == STRETCH_HALF ? w.half->n_slots : w.half->slot_hi - w.half->slot_lo);
}

template <class M>
int stretch_generic(const bisip_ctx *c, const StretchWork &a, hipStream_t st)
{
    const ModelOperands o{c->d_cb_lp ? c->d_cb_lp : c->d_cb, c->N, c->lconst};
    switch (CoopLimit<M>::value > 0 ? stretch_lanes(a) : 1) {
    case 4: { GenericLP<M, 4> lp; lp.o = o; lp.b = c->bounds; return launch_stretch(a, lp, st); }
    case 2: { GenericLP<M, 2> lp; lp.o = o; lp.b = c->bounds; return launch_stretch(a, lp, st); }
    default: { GenericLP<M, 1> lp; lp.o = o; lp.b = c->bounds; return launch_stretch(a, lp, st); }
    }
}

template <int P, bool COMP>
int stretch_reduced(const bisip_ctx *c, const StretchWork &a, hipStream_t st)
{
    ReducedLP<P, COMP> lp;
    fill_reduced<P>(c, COMP, lp.r);
    lp.lconst = c->lconst;
    lp.b = c->bounds;
    return launch_stretch(a, lp, st);
}

template <class M, bool U, int L>
int stretch_generic_batch_l(const bisip_ctx *c, const StretchWork &a, long long Wp, hipStream_t st)
{
    BatchGenericLP<M, U, L> lp;
    lp.cb = c->d_cb_lp ? c->d_cb_lp : c->d_cb; lp.cb_stride = c->cb_stride; lp.Wp = Wp; lp.lconst = c->d_lconst; lp.N = c->N;
    lp.b = c->bounds;
    return launch_stretch(a, lp, st);
}

template <class M, bool U>
int stretch_generic_batch(const bisip_ctx *c, const StretchWork &a, long long Wp, hipStream_t st)
{
    // a wave of 64/L slots must stay inside one spectrum for the uniform (scalar) operand path
    // (the persistent kernel's workgroup is one ensemble: always inside one spectrum)
    const int want = CoopLimit<M>::value > 0 ? stretch_lanes(a) : 1;
    const bool whole = !U || a.kind == STRETCH_PERSIST;
    if (want == 4 && (whole || (Wp / 2) % 16 == 0)) return stretch_generic_batch_l<M, U, 4>(c, a, Wp, st);
    if (want >= 2 && (whole || (Wp / 2) % 32 == 0)) return stretch_generic_batch_l<M, U, 2>(c, a, Wp, st);
    return stretch_generic_batch_l<M, U, 1>(c, a, Wp, st);
}

template <int P, bool U, bool COMP>
int stretch_reduced_batch(const bisip_ctx *c, const StretchWork &a, long long Wp, hipStream_t st)
{
    BatchReducedLP<P, U, COMP> lp;
    lp.red = reinterpret_cast<const ReducedArgs<P> *>(c->red[COMP ? 1 : 0].d_red); lp.Wp = Wp; lp.lconst = c->d_lconst;
    lp.b = c->bounds;
    return launch_stretch(a, lp, st);
}

// batch of spectra: Wp walkers per spectrum; a wave of 64 slots stays inside one spectrum
// iff (Wp/2) % 64 == 0
int dispatch_stretch_batch(const bisip_ctx *c, const StretchWork &a, long long Wp, hipStream_t st)
{
    // in the persistent kernel a workgroup IS one ensemble, so the spectrum is always uniform
    const bool u = (Wp % 128) == 0 || a.kind == STRETCH_PERSIST;
#define GEN(M) return u ? stretch_generic_batch<M, true>(c, a, Wp, st) : stretch_generic_batch<M, false>(c, a, Wp, st);
#define RED(p, comp) return u ? stretch_reduced_batch<p, true, comp>(c, a, Wp, st) : stretch_reduced_batch<p, false, comp>(c, a, Wp, st);
    switch (c->model_id) {
    case BISIP_MODEL_POLYDECOMP:
        if (effective_variant(c) == BISIP_VARIANT_REDUCED) {
            switch (c->P) {
#define X(p) case p: RED(p, false)
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#undef X
            }
        } else if (effective_variant(c) == BISIP_VARIANT_REDUCED_COMP) {
            switch (c->P) {
#define X(p) case p: RED(p, true)
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#undef X
            }
        } else {
            switch (c->P) {
#define X(p) case p: GEN(PDCollapsed<p>)
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#undef X
            }
        }
        break;
    case BISIP_MODEL_COLECOLE:
        switch (c->D) {
#define X(d) case d: GEN(ColeCole<d>)
            X(1) X(2) X(3) X(4) X(5)
#undef X
        }
        break;
    case BISIP_MODEL_DIAS2000: GEN(Dias)
    case BISIP_MODEL_SHIN2015: GEN(Shin)
    }
#undef GEN
#undef RED
    return fail(BISIP_EUNSUPPORTED, "no batch stretch kernel for this model shape");
}

}  // namespace

namespace bisip {
namespace host {

StretchArgs to_device_args(const bisip_stretch_args *u)
{
    StretchArgs a;
    a.coords = u->coords; a.logp = u->logp;
    a.active = u->active; a.partner = u->partner;
    a.zz = u->zz; a.factor = u->factor; a.logu = u->logu;
    a.n_slots = u->n_slots; a.slot_lo = u->slot_lo; a.slot_hi = u->slot_hi;
    a.block = u->block; a.chain_row = u->chain_row; a.logp_row = u->logp_row;
    a.naccept = u->naccept; a.status = u->status;
    a.pad = u->pad;
    const long long world = u->world > 0 ? u->world : 1;
    a.base = u->n_slots / world;
    a.extra = u->n_slots % world;
    return a;
}

int dispatch_stretch(const bisip_ctx *c, const StretchWork &a, long long Wp, hipStream_t st)
{
    if (c->E > 1) {
        if (Wp < 2 || (Wp & 1)) return fail(BISIP_EINVAL, "batch context: walkers_per_spectrum must be even and >= 2, got %lld", Wp);
        return dispatch_stretch_batch(c, a, Wp, st);
    }
    switch (c->model_id) {
    case BISIP_MODEL_POLYDECOMP:
        // the sampler kernels exist for the reduced and the collapsed formulation only; running
        // another one here would store log-probabilities that bisip_logprob does not reproduce
        // bit for bit, so say so and let the caller drive the move from the host
        if (effective_variant(c) == BISIP_VARIANT_FAITHFUL || effective_variant(c) == BISIP_VARIANT_WAVE)
            return fail(BISIP_EUNSUPPORTED, "the device stretch move has no kernel for the faithful / wave "
                        "formulation: use variant auto, reduced or collapsed, or the host-loop sampler");
        if (effective_variant(c) == BISIP_VARIANT_REDUCED) {
            switch (c->P) {
#define X(p) case p: return stretch_reduced<p, false>(c, a, st);
                PD_CASES(X)
#undef X
            }
        } else if (effective_variant(c) == BISIP_VARIANT_REDUCED_COMP) {
            switch (c->P) {
#define X(p) case p: return stretch_reduced<p, true>(c, a, st);
                PD_CASES(X)
#undef X
            }
        } else {
            switch (c->P) {
#define X(p) case p: return stretch_generic<PDCollapsed<p>>(c, a, st);
                PD_CASES(X)
#undef X
            }
        }
        break;
    case BISIP_MODEL_COLECOLE:
        switch (c->D) {
#define X(d) case d: return stretch_generic<ColeCole<d>>(c, a, st);
            CC_CASES(X)
#undef X
        }
        break;
    case BISIP_MODEL_DIAS2000: return stretch_generic<Dias>(c, a, st);
    case BISIP_MODEL_SHIN2015: return stretch_generic<Shin>(c, a, st);
    }
    return fail(BISIP_EUNSUPPORTED, "no stretch kernel for this model shape");
}

int dispatch_apply(const bisip_ctx *c, const StretchArgs &a, hipStream_t st)
{
    const unsigned grid = (unsigned)((a.n_slots + 63) / 64);
    switch (c->ndim) {
#define X(n) case n: hipLaunchKernelGGL((k_stretch_apply<n>), dim3(grid), dim3(64), 0, st, a); break;
        X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#undef X
    default: return fail(BISIP_EUNSUPPORTED, "ndim=%d", c->ndim);
    }
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

}  // namespace host
}  // namespace bisip

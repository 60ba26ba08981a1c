// dispatch_stretch.hip -- stretch-move launches: half-step / eval / apply / persistent.
#include "host.h"

using namespace bisip;
using namespace bisip::host;

namespace {

template <class LP>
int launch_stretch(const StretchWork &work, const LP &lp, hipStream_t st)
{
    const StretchKind kind = work.kind;
    if (kind == STRETCH_PERSIST) {
        const PersistArgs &p = *work.persist;
        const long long nh = (p.W + 1) / 2;
        const unsigned threads = (unsigned)(((nh * LP::L + 63) / 64) * 64);   // <= 1024: stretch_lanes()
        const size_t lds = (size_t)p.W * (LP::NDIM + 1) * sizeof(double);
        hipLaunchKernelGGL((k_stretch_persistent<LP>), dim3((unsigned)p.E), dim3(threads), lds, st, p, lp);
        HIP_TRY(hipGetLastError());
        return BISIP_OK;
    }
    const StretchArgs &a = *work.half;
    if (kind == STRETCH_HALF) {
        const unsigned grid = (unsigned)((a.n_slots * LP::L + 63) / 64);
        hipLaunchKernelGGL((k_stretch_half<LP>), dim3(grid), dim3(64), 0, st, a, lp);
    } else {
        const unsigned grid = (unsigned)(((a.slot_hi - a.slot_lo) * LP::L + 63) / 64);
        hipLaunchKernelGGL((k_stretch_eval<LP>), dim3(grid), dim3(64), 0, st, a, lp);
    }
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

// lanes per slot of a stretch dispatch: as many as lanes_per_walker() grants for the number of
// slots evaluated at once (all ensembles' for the persistent kernel, whose workgroups run
// concurrently), capped there by the 1024-lane workgroup that holds one ensemble's half.
int stretch_lanes(const StretchWork &w)
{
    if (w.kind == STRETCH_PERSIST) {
        const long long nh = (w.persist->W + 1) / 2;
        const int fit = nh * 4 <= 1024 ? 4 : (nh * 2 <= 1024 ? 2 : 1);
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

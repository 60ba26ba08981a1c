// dispatch_stretch.hip -- stretch-move launches, single spectrum: half-step / eval / apply /
// persistent (the batch-of-spectra instantiations live in dispatch_stretch_batch.hip).
#include "stretch_launch.h"

using namespace bisip;
using namespace bisip::host;

namespace {

template <class M>
int stretch_generic(const bisip_ctx *c, const StretchWork &a, hipStream_t st)
{
    const ModelOperands o{c->d_cb_lp ? c->d_cb_lp : c->d_cb, c->N, c->lconst};
    if constexpr (CoopLimit<M>::value > 0) {      // (else one lane per slot: the other kernels are never instantiated)
        switch (stretch_lanes(a)) {
        case 8: { GenericLP<M, 8> lp; lp.o = o; lp.b = c->bounds; return launch_stretch(a, lp, st); }     // (the multi-workgroup sampler only)
        case 4: { GenericLP<M, 4> lp; lp.o = o; lp.b = c->bounds; return launch_stretch(a, lp, st); }
        case 2: { GenericLP<M, 2> lp; lp.o = o; lp.b = c->bounds; return launch_stretch(a, lp, st); }
        default: break;
        }
    }
    GenericLP<M, 1> lp; lp.o = o; lp.b = c->bounds;
    return launch_stretch(a, lp, st);
}

template <int P, bool COMP>
int stretch_reduced(const bisip_ctx *c, const StretchWork &a, hipStream_t st)
{
    ReducedLP<P, COMP> lp;
    fill_reduced<P, COMP>(c, lp.r);
    lp.lconst = c->lconst;
    lp.b = c->bounds;
    return launch_stretch(a, lp, st);
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
    a.packed = nullptr;
    a.perm = nullptr; a.draw_W = 0; a.draw_a = 0.0; a.draw_ndim_m1 = 0.0;
    a.seed_lo = a.seed_hi = a.draw_step = a.draw_e = 0u; a.draw_h = 0;
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

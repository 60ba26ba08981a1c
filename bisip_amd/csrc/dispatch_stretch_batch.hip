// dispatch_stretch_batch.hip -- stretch-move launches for a batch of spectra (one context, E
// ensembles): half-step and persistent kernels over BatchGenericLP / BatchReducedLP.
#include "stretch_launch.h"

using namespace bisip;
using namespace bisip::host;

namespace {

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
    if constexpr (CoopLimit<M>::value > 0) {      // (models whose frequency loop is too cheap to split have one lane per slot: no other kernels exist)
        const int want = stretch_lanes(a);
        const bool whole = !U || a.kind == STRETCH_PERSIST;
        if (want == 4 && (whole || (Wp / 2) % 16 == 0)) return stretch_generic_batch_l<M, U, 4>(c, a, Wp, st);
        if (want >= 2 && (whole || (Wp / 2) % 32 == 0)) return stretch_generic_batch_l<M, U, 2>(c, a, Wp, st);
    }
    return stretch_generic_batch_l<M, U, 1>(c, a, Wp, st);
}

template <int P, bool U, bool COMP>
int stretch_reduced_batch(const bisip_ctx *c, const StretchWork &a, long long Wp, hipStream_t st)
{
    BatchReducedLP<P, U, COMP> lp;
    lp.red = reinterpret_cast<const ReducedArgs<P, COMP> *>(c->red[COMP ? 1 : 0].d_red); lp.Wp = Wp; lp.lconst = c->d_lconst;
    lp.red_plain = reinterpret_cast<const ReducedArgs<P, false> *>(c->red[0].d_red);
    lp.tier = COMP && c->mixed ? c->d_tier : nullptr;
    lp.b = c->bounds;
    return launch_stretch(a, lp, st);
}

// batch of spectra: Wp walkers per spectrum; a wave of 64 slots stays inside one spectrum
// iff (Wp/2) % 64 == 0
int dispatch_stretch_batch_impl(const bisip_ctx *c, const StretchWork &a, long long Wp, hipStream_t st)
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

int dispatch_stretch_batch(const bisip_ctx *c, const StretchWork &a, long long Wp, hipStream_t st)
{
    return dispatch_stretch_batch_impl(c, a, Wp, st);
}

}  // namespace host
}  // namespace bisip

// dispatch_logprob.hip -- log-probability launches (single spectrum and batch of spectra).
#include "host.h"

using namespace bisip;
using namespace bisip::host;

namespace {


template <class M, int L>
int launch_logprob_small(const LaunchArgs &a, bool vec, hipStream_t st)
{
    const unsigned grid = (unsigned)((a.W * L + BLK_SMALL - 1) / BLK_SMALL);
    if (vec) hipLaunchKernelGGL((k_logprob<M, BLK_SMALL, true, L>), dim3(grid), dim3(BLK_SMALL), 0, st, a);
    else hipLaunchKernelGGL((k_logprob<M, BLK_SMALL, false, L>), dim3(grid), dim3(BLK_SMALL), 0, st, a);
    return BISIP_OK;
}

template <class M>
int launch_logprob(const bisip_ctx *c, const double *theta, int64_t W, double *out, hipStream_t st)
{
    const LaunchArgs a = make_args(c, theta, out, W, c->d_cb_lp ? c->d_cb_lp : c->d_cb);
    const bool vec = ((uintptr_t)theta % 16) == 0;
    if (W < SMALL_W) {
        // few walkers: several lanes per walker so the launch still covers the chip
        switch (W <= CoopLimit<M>::value ? lanes_per_walker(W) : 1) {
        case 4: launch_logprob_small<M, 4>(a, vec, st); break;
        case 2: launch_logprob_small<M, 2>(a, vec, st); break;
        default: launch_logprob_small<M, 1>(a, vec, st); break;
        }
    } else {
        const unsigned grid = (unsigned)((W + BLK_LARGE - 1) / BLK_LARGE);
        if (vec) hipLaunchKernelGGL((k_logprob<M, BLK_LARGE, true>), dim3(grid), dim3(BLK_LARGE), 0, st, a);
        else hipLaunchKernelGGL((k_logprob<M, BLK_LARGE, false>), dim3(grid), dim3(BLK_LARGE), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

// PolynomialDecomposition collapsed, many walkers: two rows per lane (see k_logprob_x2)
template <int P>
int launch_collapsed(const bisip_ctx *c, const double *theta, int64_t W, double *out, hipStream_t st)
{
    if (W < SMALL_W) return launch_logprob<PDCollapsed<P>>(c, theta, W, out, st);
    const LaunchArgs a = make_args(c, theta, out, W, c->d_cb_lp);
    const bool vec = ((uintptr_t)theta % 16) == 0;
    // 256-lane workgroups: 5 % faster than 128 here (benchmarks/micro/collapsed_variants.hip)
    const unsigned grid = (unsigned)((W + 2 * BLK_LARGE - 1) / (2 * BLK_LARGE));
    if (vec) hipLaunchKernelGGL((k_logprob_x2<PDCollapsed<P>, BLK_LARGE, true>), dim3(grid), dim3(BLK_LARGE), 0, st, a);
    else hipLaunchKernelGGL((k_logprob_x2<PDCollapsed<P>, BLK_LARGE, false>), dim3(grid), dim3(BLK_LARGE), 0, st, a);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

template <int P, bool COMP>
int launch_reduced(const bisip_ctx *c, const double *theta, int64_t W, double *out, hipStream_t st)
{
    const bool vec = ((uintptr_t)theta % 16) == 0;
    if constexpr (COMP && P >= REDUCED_COMP_MEMORY_OPERANDS_FROM) {
        // operands from memory (host.h: REDUCED_COMP_MEMORY_OPERANDS_FROM): the batch kernel, one spectrum
        if (W >= SMALL_W && c->red[1].d_red && c->d_lconst) {
            BatchArgs b = make_batch_args(c, theta, out, W);
            b.Wp = (W + BLK_LARGE - 1) / BLK_LARGE * BLK_LARGE;    // every workgroup: spectrum 0
            b.red = c->red[1].d_red;
            b.tier = nullptr;
            const unsigned grid = (unsigned)((W + BLK_LARGE - 1) / BLK_LARGE);
            if (vec) hipLaunchKernelGGL((k_logprob_batch_reduced_stream<P, BLK_LARGE, true, true>), dim3(grid), dim3(BLK_LARGE), 0, st, b);
            else hipLaunchKernelGGL((k_logprob_batch_reduced_stream<P, BLK_LARGE, false, true>), dim3(grid), dim3(BLK_LARGE), 0, st, b);
            HIP_TRY(hipGetLastError());
            return BISIP_OK;
        }
    }
    const LaunchArgs a = make_args(c, theta, out, W, nullptr);
    ReducedArgs<P, COMP> r;
    fill_reduced<P, COMP>(c, r);
    if (W < SMALL_W) {
        const unsigned grid = (unsigned)((W + BLK_SMALL - 1) / BLK_SMALL);
        if (vec) hipLaunchKernelGGL((k_logprob_pd_reduced<P, BLK_SMALL, true, COMP>), dim3(grid), dim3(BLK_SMALL), 0, st, a, r);
        else hipLaunchKernelGGL((k_logprob_pd_reduced<P, BLK_SMALL, false, COMP>), dim3(grid), dim3(BLK_SMALL), 0, st, a, r);
    } else {
        // the plain form is HBM-bound: 128-lane workgroups stream best; the compensated one is
        // VALU-bound (~11 flops per matrix entry): 256 lanes, like the other compute-bound kernels
        constexpr int BLK = COMP ? BLK_LARGE : BLK_STREAM;
        const unsigned grid = (unsigned)((W + BLK - 1) / BLK);
        if (vec) hipLaunchKernelGGL((k_logprob_pd_reduced<P, BLK, true, COMP>), dim3(grid), dim3(BLK), 0, st, a, r);
        else hipLaunchKernelGGL((k_logprob_pd_reduced<P, BLK, false, COMP>), dim3(grid), dim3(BLK), 0, st, a, r);
    }
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

template <int P>
int launch_faithful(const bisip_ctx *c, const double *theta, int64_t W, double *out, hipStream_t st)
{
    const LaunchArgs a = make_args(c, theta, out, W, c->d_cb_faithful);
    const bool vec = ((uintptr_t)theta % 16) == 0;
    const unsigned grid = (unsigned)((W + BLK_SMALL - 1) / BLK_SMALL);
    if (vec) hipLaunchKernelGGL((k_logprob_pd_faithful<P, BLK_SMALL, true>), dim3(grid), dim3(BLK_SMALL), 0, st, a, c->S);
    else hipLaunchKernelGGL((k_logprob_pd_faithful<P, BLK_SMALL, false>), dim3(grid), dim3(BLK_SMALL), 0, st, a, c->S);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

template <int P>
int launch_wave(const bisip_ctx *c, const double *theta, int64_t W, double *out, hipStream_t st)
{
    const LaunchArgs a = make_args(c, theta, out, W, c->d_cb_lp);
    const size_t lds = (size_t)c->N * (4 + 2 * (P + 1)) * sizeof(double);
    // persistent waves: 8 workgroups of 4 waves per CU, fewer when there are fewer walkers
    long long blocks = (W + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (2 * c->N <= 64) hipLaunchKernelGGL((k_logprob_pd_wave<P, 1>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_logprob_pd_wave<P, 2>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

template <class M>
int launch_logprob_batch(const bisip_ctx *c, const double *theta, int64_t W, double *out, hipStream_t st)
{
    BatchArgs a = make_batch_args(c, theta, out, W);
    if (c->d_cb_lp) a.cb = c->d_cb_lp;
    const unsigned grid = (unsigned)((W + 63) / 64);
    if (a.Wp % 64 == 0) hipLaunchKernelGGL((k_logprob_batch<M, true>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_logprob_batch<M, false>), dim3(grid), dim3(64), 0, st, a);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

template <int P, bool COMP>
int launch_reduced_batch(const bisip_ctx *c, const double *theta, int64_t W, double *out, hipStream_t st)
{
    const BatchArgs a = make_batch_args(c, theta, out, W);
    constexpr int BLK = COMP ? BLK_LARGE : BLK_STREAM;
    if (W >= SMALL_W && a.Wp % BLK == 0) {      // bulk evaluation: stream like the single-spectrum kernel
        const unsigned sgrid = (unsigned)((W + BLK - 1) / BLK);
        if (((uintptr_t)theta % 16) == 0) hipLaunchKernelGGL((k_logprob_batch_reduced_stream<P, BLK, true, COMP>), dim3(sgrid), dim3(BLK), 0, st, a);
        else hipLaunchKernelGGL((k_logprob_batch_reduced_stream<P, BLK, false, COMP>), dim3(sgrid), dim3(BLK), 0, st, a);
        HIP_TRY(hipGetLastError());
        return BISIP_OK;
    }
    const unsigned grid = (unsigned)((W + 63) / 64);
    if (a.Wp % 64 == 0) hipLaunchKernelGGL((k_logprob_batch_reduced<P, true, COMP>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_logprob_batch_reduced<P, false, COMP>), dim3(grid), dim3(64), 0, st, a);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

int dispatch_logprob_batch(const bisip_ctx *c, const double *theta, int64_t W, double *out, hipStream_t st)
{
    switch (c->model_id) {
    case BISIP_MODEL_POLYDECOMP:
        if (effective_variant(c) == BISIP_VARIANT_REDUCED) {
            switch (c->P) {
#define X(p) case p: return launch_reduced_batch<p, false>(c, theta, W, out, st);
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#undef X
            }
        } else if (effective_variant(c) == BISIP_VARIANT_REDUCED_COMP) {
            switch (c->P) {
#define X(p) case p: return launch_reduced_batch<p, true>(c, theta, W, out, st);
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#undef X
            }
        } else {
            switch (c->P) {
#define X(p) case p: return launch_logprob_batch<PDCollapsed<p>>(c, theta, W, out, st);
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#undef X
            }
        }
        break;
    case BISIP_MODEL_COLECOLE:
        switch (c->D) {
#define X(d) case d: return launch_logprob_batch<ColeCole<d>>(c, theta, W, out, st);
            X(1) X(2) X(3) X(4) X(5)
#undef X
        }
        break;
    case BISIP_MODEL_DIAS2000: return launch_logprob_batch<Dias>(c, theta, W, out, st);
    case BISIP_MODEL_SHIN2015: return launch_logprob_batch<Shin>(c, theta, W, out, st);
    }
    return fail(BISIP_EUNSUPPORTED, "no batch kernel for this model shape");
}

}  // namespace

namespace bisip {
namespace host {

// Lanes per walker for launches that cannot fill the chip with one lane per walker: as many
// as keep the launch within one wave per SIMD (1024 SIMDs x 64 lanes = 65536 lanes) -- 4 up to
// 16 Ki walkers, 2 up to 32 Ki, else 1.  The value never changes a result (logprob_row is
// bit-identical for every L), only the wave count.
int lanes_per_walker(long long walkers)
{
    return walkers <= 16384 ? 4 : (walkers <= 32768 ? 2 : 1);
}

int dispatch_logprob(const bisip_ctx *c, const double *theta, int64_t W, double *out, hipStream_t st)
{
    if (W == 0) return BISIP_OK;
    if ((W + BLK_SMALL - 1) / BLK_SMALL > 0x7fffffffLL)
        return fail(BISIP_EINVAL, "W=%lld exceeds the launch grid limit", (long long)W);
    if (c->E > 1) {
        if (W % c->E) return fail(BISIP_EINVAL, "W=%lld is not a multiple of n_spectra=%d", (long long)W, c->E);
        if (W > 0xffffffffLL) return fail(BISIP_EINVAL, "W=%lld: a batch context takes fewer than 2^32 rows per call", (long long)W);
        return dispatch_logprob_batch(c, theta, W, out, st);
    }
    switch (c->model_id) {
    case BISIP_MODEL_POLYDECOMP: {
        const int v = effective_variant(c);
        if (v == BISIP_VARIANT_REDUCED) {
            switch (c->P) {
#define X(p) case p: return launch_reduced<p, false>(c, theta, W, out, st);
                PD_CASES(X)
#undef X
            }
        } else if (v == BISIP_VARIANT_REDUCED_COMP) {
            switch (c->P) {
#define X(p) case p: return launch_reduced<p, true>(c, theta, W, out, st);
                PD_CASES(X)
#undef X
            }
        } else if (v == BISIP_VARIANT_COLLAPSED) {
            switch (c->P) {
#define X(p) case p: return launch_collapsed<p>(c, theta, W, out, st);
                PD_CASES(X)
#undef X
            }
        } else if (v == BISIP_VARIANT_WAVE) {
            switch (c->P) {
#define X(p) case p: return launch_wave<p>(c, theta, W, out, st);
                PD_CASES(X)
#undef X
            }
        } else if (v == BISIP_VARIANT_FAITHFUL) {
            if (!c->d_cb_faithful) return fail(BISIP_EUNSUPPORTED, "the faithful formulation is not available for this context");
            switch (c->P) {
#define X(p) case p: return launch_faithful<p>(c, theta, W, out, st);
                PD_CASES(X)
#undef X
            }
        }
        return fail(BISIP_EUNSUPPORTED, "no kernel for poly_deg=%d variant=%d", c->P, v);
    }
    case BISIP_MODEL_COLECOLE:
        switch (c->D) {
#define X(d) case d: return launch_logprob<ColeCole<d>>(c, theta, W, out, st);
            CC_CASES(X)
#undef X
        }
        return fail(BISIP_EUNSUPPORTED, "no kernel for n_modes=%d", c->D);
    case BISIP_MODEL_DIAS2000: return launch_logprob<Dias>(c, theta, W, out, st);
    case BISIP_MODEL_SHIN2015: return launch_logprob<Shin>(c, theta, W, out, st);
    }
    return fail(BISIP_EINVAL, "bad model_id %d", c->model_id);
}

}  // namespace host
}  // namespace bisip

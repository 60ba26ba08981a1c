// dispatch_forward.hip -- batched forward() launches.
#include "host.h"
#include <type_traits>

using namespace bisip;
using namespace bisip::host;

namespace {

// PolynomialDecomposition's forward is 17 instructions per frequency: whole rows through LDS (k_forward_rows) beat
// the 16-frequency tiles up to N = 32 (4.95 -> 5.3 TB/s of Z at N = 32).  The transcendental models keep the tiles
// there: all of a row's arithmetic in front of all of its stores costs them more than the longer runs gain (4.9 ->
// 4.1 TB/s, ColeCole<2>; 4.5 -> 3.8, Shin).
template <class M> struct cheap_eval : std::false_type {};
template <int P> struct cheap_eval<PDCollapsed<P>> : std::true_type {};

// spectrum >= 0: the rows belong to `count` consecutive spectra of a batch context, starting with that one,
// W / count rows each (bisip_forward_spectra_dev)
template <class M>
int launch_forward(const bisip_ctx *c, const double *theta, int64_t W, double *Z, hipStream_t st, long long spectrum = -1,
                   long long count = 1)
{
    LaunchArgs a = make_args(c, theta, Z, W, c->d_cb + (spectrum >= 0 ? spectrum * c->cb_stride : 0));
    if (c->E > 1 && spectrum < 0) { a.Wp = W / c->E; a.cb_stride = c->cb_stride; }   // batch: caller checked Wp % 64 == 0
    if (spectrum >= 0 && count > 1) { a.Wp = W / count; a.cb_stride = c->cb_stride; }   // caller checked (W / count) % 64 == 0
    // one single-wave workgroup per 64 walkers; the grid is not capped at the resident count
    // (a capped, looping grid measured 0-25 % slower depending on the box: benchmarks/micro/forward_variants.hip)
    const unsigned grid = (unsigned)((W + 63) / 64);
    const int N = c->N;
    const bool wide = (N % 2 == 0) && ((uintptr_t)Z % 16) == 0;   // 16-byte store pieces
    if ((N % 16 != 0 || cheap_eval<M>::value) && N <= 24) {
        if (wide) hipLaunchKernelGGL((k_forward_rows<M, 24, true>), dim3(grid), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_forward_rows<M, 24, false>), dim3(grid), dim3(64), 0, st, a);
    } else if ((N % 16 != 0 || cheap_eval<M>::value) && N <= 32) {
        if (wide) hipLaunchKernelGGL((k_forward_rows<M, 32, true>), dim3(grid), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_forward_rows<M, 32, false>), dim3(grid), dim3(64), 0, st, a);
    } else if (N % 16 == 0 && wide) hipLaunchKernelGGL((k_forward_tiled16<M>), dim3(grid), dim3(64), 0, st, a);
    else if (((uintptr_t)theta % 16) == 0) hipLaunchKernelGGL((k_forward_tiled<M, true>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_forward_tiled<M, false>), dim3(grid), dim3(64), 0, st, a);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

// column-major output for `count` consecutive spectra starting with `spectrum`, W / count rows each
template <class M>
int launch_forward_columns(const bisip_ctx *c, const double *theta, int64_t W, double *cols, hipStream_t st, long long spectrum,
                           long long count)
{
    LaunchArgs a = make_args(c, theta, cols, W, c->d_cb + spectrum * c->cb_stride);
    a.Wp = W / count; a.cb_stride = c->cb_stride;
    hipLaunchKernelGGL((k_forward_columns<M>), dim3((unsigned)((W + 63) / 64)), dim3(64), 0, st, a);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

template <class M>
int launch_forward_batch(const bisip_ctx *c, const double *theta, int64_t W, double *Z, hipStream_t st)
{
    // 64-walker blocks never straddle two spectra: the tiled / whole-row kernels apply
    if ((W / c->E) % 64 == 0) return launch_forward<M>(c, theta, W, Z, st);
    const BatchArgs a = make_batch_args(c, theta, Z, W);
    const long long total = (long long)W * c->N;
    hipLaunchKernelGGL((k_forward_batch<M>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

int dispatch_forward_batch(const bisip_ctx *c, const double *theta, int64_t W, double *Z, hipStream_t st)
{
    switch (c->model_id) {
    case BISIP_MODEL_POLYDECOMP:
        switch (c->P) {
#define X(p) case p: return launch_forward_batch<PDCollapsed<p>>(c, theta, W, Z, st);
            X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#undef X
        }
        break;
    case BISIP_MODEL_COLECOLE:
        switch (c->D) {
#define X(d) case d: return launch_forward_batch<ColeCole<d>>(c, theta, W, Z, st);
            X(1) X(2) X(3) X(4) X(5)
#undef X
        }
        break;
    case BISIP_MODEL_DIAS2000: return launch_forward_batch<Dias>(c, theta, W, Z, st);
    case BISIP_MODEL_SHIN2015: return launch_forward_batch<Shin>(c, theta, W, Z, st);
    }
    return fail(BISIP_EUNSUPPORTED, "no batch forward kernel for this model shape");
}

}  // namespace

namespace bisip {
namespace host {

int dispatch_forward_columns(const bisip_ctx *c, const double *theta, int64_t W, double *cols, hipStream_t st, long long spectrum,
                             long long count)
{
    if (W == 0) return BISIP_OK;
    if (spectrum < 0 || count < 1 || spectrum + count > c->E) return fail(BISIP_EINVAL, "spectra [%lld, %lld) of %d", spectrum, spectrum + count, c->E);
    if (W % count) return fail(BISIP_EINVAL, "W=%lld rows do not divide over %lld spectra", (long long)W, count);
    if ((W + 63) / 64 > 0x7fffffffLL) return fail(BISIP_EINVAL, "W=%lld exceeds the launch grid limit", (long long)W);
    if (((uintptr_t)theta % 8) || ((uintptr_t)cols % 8)) return fail(BISIP_EINVAL, "buffers must be 8-byte aligned");
    switch (c->model_id) {
    case BISIP_MODEL_POLYDECOMP:
        switch (c->P) {
#define X(p) case p: return launch_forward_columns<PDCollapsed<p>>(c, theta, W, cols, st, spectrum, count);
            PD_CASES(X)
#undef X
        }
        break;
    case BISIP_MODEL_COLECOLE:
        switch (c->D) {
#define X(d) case d: return launch_forward_columns<ColeCole<d>>(c, theta, W, cols, st, spectrum, count);
            CC_CASES(X)
#undef X
        }
        break;
    case BISIP_MODEL_DIAS2000: return launch_forward_columns<Dias>(c, theta, W, cols, st, spectrum, count);
    case BISIP_MODEL_SHIN2015: return launch_forward_columns<Shin>(c, theta, W, cols, st, spectrum, count);
    }
    return fail(BISIP_EUNSUPPORTED, "no forward kernel for this model shape");
}

int dispatch_forward(const bisip_ctx *c, const double *theta, int64_t W, double *Z, hipStream_t st, long long spectrum,
                     long long count)
{
    if (W == 0) return BISIP_OK;
    if (spectrum >= 0 && (count < 1 || spectrum + count > c->E)) return fail(BISIP_EINVAL, "spectra [%lld, %lld) of %d", spectrum, spectrum + count, c->E);
    if (spectrum >= 0 && count > 1 && (W % count || (W / count) % 64))
        return fail(BISIP_EINVAL, "W=%lld rows over %lld spectra: each needs a multiple of 64 rows", (long long)W, count);
    if (((long long)W * c->N + 255) / 256 > 0x7fffffffLL)
        return fail(BISIP_EINVAL, "W=%lld exceeds the launch grid limit", (long long)W);
    if (((uintptr_t)theta % 8) || ((uintptr_t)Z % 8)) return fail(BISIP_EINVAL, "buffers must be 8-byte aligned");
    if (c->E > 1 && spectrum < 0) {
        if (W % c->E) return fail(BISIP_EINVAL, "W=%lld is not a multiple of n_spectra=%d", (long long)W, c->E);
        return dispatch_forward_batch(c, theta, W, Z, st);
    }
    switch (c->model_id) {
    case BISIP_MODEL_POLYDECOMP:
        switch (c->P) {
#define X(p) case p: return launch_forward<PDCollapsed<p>>(c, theta, W, Z, st, spectrum, count);
            PD_CASES(X)
#undef X
        }
        break;
    case BISIP_MODEL_COLECOLE:
        switch (c->D) {
#define X(d) case d: return launch_forward<ColeCole<d>>(c, theta, W, Z, st, spectrum, count);
            CC_CASES(X)
#undef X
        }
        break;
    case BISIP_MODEL_DIAS2000: return launch_forward<Dias>(c, theta, W, Z, st, spectrum, count);
    case BISIP_MODEL_SHIN2015: return launch_forward<Shin>(c, theta, W, Z, st, spectrum, count);
    }
    return fail(BISIP_EUNSUPPORTED, "no forward kernel for this model shape");
}

}  // namespace host
}  // namespace bisip

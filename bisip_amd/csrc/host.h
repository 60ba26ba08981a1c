// host.h -- what the translation units of libbisip_hip.so share: the context, the error
// plumbing and the dispatch entry points.  The kernels are templates, so every dispatch_*.hip
// instantiates only its own family and the four units compile in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/bisip_hip.h"
#include "host_precompute.h"
#include "kernels.h"
#include "sampler_kernels.h"

namespace bisip {
namespace host {

int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return ::bisip::host::fail(BISIP_EHIP, "%s failed: %s (%s:%d)", #expr,     \
                                       hipGetErrorString(e_), __FILE__, __LINE__);     \
    } while (0)

constexpr int BLK_SMALL = 64;    // few walkers: spread them over more CUs
constexpr int BLK_LARGE = 256;
constexpr int BLK_STREAM = 128;  // HBM-bound reduced kernel: 128-lane workgroups stream ~3 % faster
                                 // than 256 (interleaved A/B, benchmarks/micro/reduced_variants.hip)
constexpr long long SMALL_W = 256LL * 256 * 2;  // below this, 64-lane workgroups
// From this polynomial degree on, the bulk launch of a lone spectrum's COMPENSATED reduced kernel reads its operands
// from memory through the scalar path, as a batch does, instead of taking them as kernel arguments: the triangle and
// its low words (99 scalars at degree 6, 171 at degree 9) no longer fit the scalar registers, and where kernel
// arguments are spilled into vector lanes (162 at degree 7), operands in memory are simply loaded again
// (measured at 2^23 walkers: +15 % at degree 6, +13 % at 7, +22 % at 8-9, +20 % at 10; equal at degree 5).
constexpr int REDUCED_COMP_MEMORY_OPERANDS_FROM = 6;
constexpr double BISIP_REDUCED_ERR_MAX = 1e-12;  // AUTO keeps the QR-reduced form below this estimate

#define PD_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#define CC_CASES(X) X(1) X(2) X(3) X(4) X(5)

}  // namespace host
}  // namespace bisip

struct bisip_ctx {
    int device = 0, model_id = 0, N = 0, ndim = 0, variant = BISIP_VARIANT_AUTO;
    int P = 0, D = 0, S = 0;
    double c_exp = 1.0, lconst = 0.0;
    double auto_err_max = bisip::host::BISIP_REDUCED_ERR_MAX;   // AUTO keeps a QR-reduced tier whose estimate is below this
    bisip::Bounds bounds{};
    double lnw_min = 0.0, lnw_max = 0.0;   // over every frequency of every spectrum (bound_flags)
    bool grid_ok = false;                  // some spectrum's frequencies lie on a geometric grid (grid_step; BOUNDS_GRID; the loop is chosen per spectrum)
    int E = 1;                      // spectra in the context (batch of spectra: E > 1)
    double *d_cb = nullptr;        // records for k_forward (and k_logprob of CC/Dias/Shin): (E, N, REC)
    double *d_cb_lp = nullptr;     // PolynomialDecomposition: 1/sigma-weighted log-prob records (E, N, REC)
    double *d_cb_faithful = nullptr;
    double *d_lconst = nullptr;    // (E,)  batch; PolynomialDecomposition from degree 6 on
    long long cb_stride = 0;
    // QR-reduced form, two arithmetic tiers: [0] plain, [1] compensated (kernels.h:
    // logprob_row_reduced<P, COMP>); each has its own expansion point, and the compensated tier its own
    // operands (triangle, rest) from the QR carried out in binary128 (host_precompute.h: ReducedProblem)
    struct ReducedTier {
        std::vector<double> bhat, evec, elo;   // spectrum 0, for the kernarg segment
        std::vector<double> Rpacked;            // spectrum 0, packed upper triangle as this tier's kernel holds it
        std::vector<float> Rlo_packed;          // tier 1: its low word (even count)
        double rest = 0.0;                      // spectrum 0
        double err = INFINITY;                  // worst estimated relative log-prob error over the spectra that RUN this tier; INFINITY: not estimated
        bool valid = false;                     // estimated for the current prior box (every spectrum that needs it)
        void *d_red = nullptr;                  // (E,) ReducedArgs<P>: a batch's tiers; a lone spectrum's compensated tier from degree 6 on
        std::vector<double> image;              // host copy of d_red (entries are rewritten spectrum by spectrum)
        std::vector<double> est;                // the estimate of every spectrum (tier 1: 0 where the spectrum does not run it)
        std::vector<unsigned char> done;        // per spectrum: estimated for the current box
    };
    ReducedTier red[2];
    // bisip_logprob measures the reduced kernel it ran on rows of its own batches (first call, then
    // every 2^n-th); a tier found wanting is closed to BISIP_VARIANT_AUTO until the box changes
    bool demoted[2] = {false, false};
    // A batch on BISIP_VARIANT_AUTO whose spectra do not all pass the plain tier launches the compensated
    // kernels with a tier per spectrum (BatchArgs::tier): tier_of[e] = 0 where the plain estimate of spectrum e
    // passes -- every spectrum runs what a context of its own would run, and only the others pay for the
    // compensated tier's operands and estimate.  mix_off: the guard closed the mix.
    std::vector<unsigned char> tier_of;
    unsigned char *d_tier = nullptr;
    bool mixed = false, mix_off = false;
    bool guard_on = true;
    int64_t guard_calls = 0, guard_checks = 0;
    double guard_worst = 0.0;
    int guard_escalations = 0;
    // per spectrum, for re-centring the reduced form when the prior box changes
    std::vector<bisip::ReducedProblem> reduced;
    // PolynomialDecomposition: host copies of what the compensated tier's binary128 operands are built from, on
    // demand (a spectrum that passes the plain tier never needs them)
    std::vector<double> h_w, h_zn, h_err, h_taus, h_log_taus;
    // workspace of the host-pointer entry points
    double *d_ws = nullptr;
    size_t ws_bytes = 0;
    double *d_gather = nullptr;    // sharded sampler: world slabs of ceil(slots/world) x (ndim+2)
    size_t gather_bytes = 0;
    double *d_packed = nullptr;    // big single ensembles: the chunk's state as one 64-byte row per walker (bisip_stretch_run_dev)
    size_t packed_bytes = 0;
    char *d_group = nullptr;       // multi-workgroup persistent sampler: 256 B of synchronisation words, then W rows of 64 B
    size_t group_bytes = 0;
    int64_t spectrum_offset = 0;   // batch: survey index of spectrum 0 (keys the Philox stream)
    // big host-buffer calls (bisip_logprob / bisip_forward): pinned double buffers, a second stream for the way
    // back, events per buffer (host_pipeline in bisip_hip.hip)
    char *h_pipe = nullptr;
    size_t pipe_bytes = 0;
    hipStream_t stream_back = nullptr;
    hipEvent_t pipe_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    char *h_pin = nullptr;         // pinned, device-mapped staging for small host-buffer calls
    char *d_pin = nullptr;         // its device-side address
    static constexpr size_t PIN_BYTES = 1 << 20;
    static constexpr size_t ZEROCOPY_BYTES = 64 << 10;
    hipStream_t stream = nullptr;
    const char *kernel_name = "";
};

namespace bisip {
namespace host {

int effective_variant(const bisip_ctx *c);
inline bool is_reduced(int v) { return v == BISIP_VARIANT_REDUCED || v == BISIP_VARIANT_REDUCED_COMP; }

// kernarg image of spectrum 0's reduced operands for the tier a variant runs
template <int P, bool COMP>
inline void fill_reduced(const bisip_ctx *c, ReducedArgs<P, COMP> &r)
{
    const bisip_ctx::ReducedTier &t = c->red[COMP ? 1 : 0];
    if constexpr (COMP) std::memcpy(r.Rlo, t.Rlo_packed.data(), sizeof(float) * t.Rlo_packed.size());
    std::memcpy(r.R, t.Rpacked.data(), sizeof(r.R));
    std::memcpy(r.bhat, t.bhat.data(), sizeof(r.bhat));
    std::memcpy(r.e, t.evec.data(), sizeof(r.e));
    std::memcpy(r.elo, t.elo.data(), sizeof(r.elo));
    r.rest = t.rest;
}
LaunchArgs make_args(const bisip_ctx *c, const double *theta, double *out, int64_t W, const double *cb);
BatchArgs make_batch_args(const bisip_ctx *c, const double *theta, double *out, int64_t W);

// Lanes per walker for launches that cannot fill the chip with one lane per walker
// (dispatch_logprob.hip); the value never changes a result, only the wave count.
int lanes_per_walker(long long walkers);
// models whose frequency loop is too cheap to split over lanes
template <class M> struct CoopLimit { static constexpr long long value = 1LL << 40; };
// PDCollapsed spends 15 FMAs per frequency: passing the running sums between lanes costs as
// much as the residual it parallelises (measured 5.7 vs 5.2 us at 4096 walkers), so one lane.
template <int P> struct CoopLimit<PDCollapsed<P>> { static constexpr long long value = 0; };

// dispatch_logprob.hip / dispatch_forward.hip
int dispatch_logprob(const bisip_ctx *c, const double *theta, int64_t W, double *out, hipStream_t st);
int dispatch_forward(const bisip_ctx *c, const double *theta, int64_t W, double *Z, hipStream_t st, long long spectrum = -1,
                     long long count = 1);
int dispatch_forward_columns(const bisip_ctx *c, const double *theta, int64_t W, double *cols, hipStream_t st, long long spectrum,
                             long long count);

// dispatch_stretch.hip
enum StretchKind { STRETCH_HALF, STRETCH_EVAL, STRETCH_PERSIST, STRETCH_GROUP };

// what one stretch dispatch launches: a half-step / eval kernel over StretchArgs, or the
// persistent kernel over PersistArgs
struct StretchWork {
    StretchKind kind;
    const StretchArgs *half;
    const PersistArgs *persist;
};

StretchArgs to_device_args(const bisip_stretch_args *u);
int dispatch_stretch(const bisip_ctx *c, const StretchWork &a, long long Wp, hipStream_t st);
int dispatch_stretch_batch(const bisip_ctx *c, const StretchWork &a, long long Wp, hipStream_t st);   // dispatch_stretch_batch.hip
int dispatch_apply(const bisip_ctx *c, const StretchArgs &a, hipStream_t st);

}  // namespace host
}  // namespace bisip

// bisip_hip.hip -- C ABI (include/bisip_hip.h) over the kernels in kernels.h.
// Host side: context = device copies of the walker-independent operands + the prior
// box; every call is one kernel launch on the caller's stream.
#include <cstdlib>

#include "host.h"
#include "host_precompute.h"
#include "chain_kernels.h"

#include <atomic>
#include <exception>

using namespace bisip;
using namespace bisip::host;

namespace {
thread_local char g_err[512] = "";
}

namespace bisip {
namespace host {

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int effective_variant(const bisip_ctx *c)
{
    if (c->model_id != BISIP_MODEL_POLYDECOMP) return BISIP_VARIANT_COLLAPSED;
    if (c->variant != BISIP_VARIANT_AUTO) return c->variant;
    // the QR-reduced form is the fast one, but only where it is accurate: with fewer data rows
    // (2N) than unknowns (P+2) there is no triangle; otherwise the plain kernel while the
    // host-side emulation of its arithmetic stays within 1e-12 of long double on the probe rows,
    // else the compensated kernel under the same test (ill-conditioned designs: high degree,
    // small exponent), else the per-frequency form, which mirrors the reference's sums
    if (2 * c->N < c->P + 2) return BISIP_VARIANT_COLLAPSED;
    if (c->red[0].err <= c->auto_err_max && !c->demoted[0]) return BISIP_VARIANT_REDUCED;
    if (c->red[1].err <= c->auto_err_max && !c->demoted[1]) return BISIP_VARIANT_REDUCED_COMP;
    return BISIP_VARIANT_COLLAPSED;
}

LaunchArgs make_args(const bisip_ctx *c, const double *theta, double *out, int64_t W,
                     const double *cb)
{
    LaunchArgs a;
    a.theta = theta;
    a.out = out;
    a.W = W;
    a.cb = cb;
    a.N = c->N;
    a.lconst = c->lconst;
    a.b = c->bounds;
    return a;
}

BatchArgs make_batch_args(const bisip_ctx *c, const double *theta, double *out, int64_t W)
{
    BatchArgs a;
    a.theta = theta; a.out = out; a.W = W; a.Wp = W / c->E;
    a.cb = c->d_cb; a.cb_stride = c->cb_stride; a.lconst = c->d_lconst;
    const bool comp = effective_variant(c) == BISIP_VARIANT_REDUCED_COMP;
    a.red = c->red[comp ? 1 : 0].d_red;
    a.red_plain = c->red[0].d_red;
    a.tier = comp && c->mixed ? c->d_tier : nullptr;
    a.N = c->N; a.b = c->bounds;
    return a;
}

}  // namespace host
}  // namespace bisip

namespace {

int upload(double **dst, const std::vector<double> &src)
{
    HIP_TRY(hipMalloc((void **)dst, src.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice));
    return BISIP_OK;
}

int check_stretch(const bisip_ctx *c, const bisip_stretch_args *u, bool need_block)
{
    if (!c || !u) return fail(BISIP_EINVAL, "null argument");
    if (u->n_slots < 0) return fail(BISIP_EINVAL, "n_slots < 0");
    if (u->n_slots == 0) return BISIP_OK;
    if (!u->coords || !u->logp || !u->active || !u->status)
        return fail(BISIP_EINVAL, "coords/logp/active/status must not be null");
    if (need_block && !u->block) return fail(BISIP_EINVAL, "block must not be null");
    return BISIP_OK;
}

int ensure_ws(bisip_ctx *c, size_t bytes)
{
    if (c->ws_bytes >= bytes) return BISIP_OK;
    if (c->d_ws) { (void)hipFree(c->d_ws); c->d_ws = nullptr; c->ws_bytes = 0; }
    hipError_t e = hipMalloc((void **)&c->d_ws, bytes);
    if (e != hipSuccess) return fail(BISIP_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    c->ws_bytes = bytes;
    return BISIP_OK;
}

const char *name_for(const bisip_ctx *c)
{
    switch (c->model_id) {
    case BISIP_MODEL_POLYDECOMP:
        switch (effective_variant(c)) {
        case BISIP_VARIANT_REDUCED: return "k_logprob_pd_reduced";
        case BISIP_VARIANT_REDUCED_COMP: return "k_logprob_pd_reduced_comp";
        case BISIP_VARIANT_FAITHFUL: return "k_logprob_pd_faithful";
        case BISIP_VARIANT_WAVE: return "k_logprob_pd_wave";
        default: return "k_logprob<PDCollapsed>";
        }
    case BISIP_MODEL_COLECOLE: return "k_logprob<ColeCole>";
    case BISIP_MODEL_DIAS2000: return "k_logprob<Dias>";
    default: return "k_logprob<Shin>";
    }
}

}  // namespace

// May ColeCole<D> / Shin / Dias run their FAST frequency loop -- ONE reciprocal per group of denominators
// (kernels.h: rcp_batch_n; rcp_joint for the up to four denominators of a PAIR of frequencies), no exponent
// clamp?  Only if, everywhere inside the prior box, every denominator of
// a frequency lies in [1, 2^225], so that a product of four is a normal number: exponents
// y = c log2e (ln w + log_tau) (ColeCole) or log2e (n ln w + log_Q) (Shin) bounded by 110, cos(c pi/2) >= 0
// (c or n within [0, 1]: the real part of each denominator term is then >= 1), and for Shin R <= 1 (1/R >= 1;
// its clamp at 2^110 bounds the other side).  The reference's default boxes pass with y <= 26; a user who
// widens a box past this gets the safe loop: one reciprocal per term, exponents clamped at 500 (where a
// squared magnitude would overflow).
static_assert(bisip::HOST_GRID_BLOCK == bisip::GRID_BLOCK, "the host's grid check and the kernels' stepped loops use one block length");

// The estimate below which BISIP_VARIANT_AUTO keeps a QR-reduced tier.  BISIP_AUTO_ERR_MAX (a test hook, read
// when a context is created: tests let a tier's estimate "pass" on a design where it would not, to see the
// guards catch it) overrides the 1e-12; validated ((0, 1]) and announced on stderr, never silent.
static double auto_err_max_from_env()
{
    const char *s = std::getenv("BISIP_AUTO_ERR_MAX");
    if (!s) return BISIP_REDUCED_ERR_MAX;
    char *end = nullptr;
    const double v = std::strtod(s, &end);
    const bool ok = end != s && v > 0.0 && v <= 1.0;
    std::fprintf(stderr, "bisip: BISIP_AUTO_ERR_MAX=%s %s (default %.0e): BISIP_VARIANT_AUTO keeps a QR-reduced tier below "
                         "this ESTIMATE -- a test hook\n", s, ok ? "overrides the threshold" : "ignored", BISIP_REDUCED_ERR_MAX);
    return ok ? v : BISIP_REDUCED_ERR_MAX;
}

static int bound_flags(const bisip_ctx *c)
{
    constexpr double LOG2E = 1.4426950408889634, YMAX = 110.0;
    const double *lo = c->bounds.lo, *hi = c->bounds.hi;
    const double lw = std::fmax(std::fabs(c->lnw_min), std::fabs(c->lnw_max));
    auto mag = [](double a, double b) { return std::fmax(std::fabs(a), std::fabs(b)); };   // inf for an open side
    bool ok = true;
    double ymax = 0.0;
    if (c->model_id == BISIP_MODEL_COLECOLE) {
        const int D = c->D;
        for (int i = 0; i < D; ++i) {
            const double llo = lo[1 + D + i], lhi = hi[1 + D + i], clo = lo[1 + 2 * D + i], chi = hi[1 + 2 * D + i];
            const double y = mag(clo, chi) * (lw + mag(llo, lhi)) * LOG2E;
            ymax = y <= ymax ? ymax : y;                  // NaN (a NaN bound) propagates as "unbounded"
            ok = ok && clo >= 0.0 && chi <= 1.0;
        }
    } else if (c->model_id == BISIP_MODEL_SHIN2015) {
        for (int i = 0; i < 2; ++i) {
            const double rlo = lo[i], rhi = hi[i], qlo = lo[2 + i], qhi = hi[2 + i], nlo = lo[4 + i], nhi = hi[4 + i];
            const double y = (mag(nlo, nhi) * lw + mag(qlo, qhi)) * LOG2E;
            ymax = y <= ymax ? ymax : y;
            ok = ok && rlo >= 0.0 && rhi <= 1.0 && nlo >= 0.0 && nhi <= 1.0;
        }
    } else if (c->model_id == BISIP_MODEL_DIAS2000) {
        // Dias shares ONE reciprocal between frequencies 2k and 2k+1 (kernels.h: Dias::residual2<true>): the
        // product of two D = X^2 + Y^2 must be a normal number with a normal reciprocal everywhere in the box.
        // Bounds of D over the box (parameters r0, m, log_tau, eta, delta; tau' >= 0 needs m, delta within
        // [0, 1] and is clamped at 1e50 in the kernel; u = sqrt(w)):
        //   D >= n2^2 >= (u tau)^4,   D <= (n2 + b g)^2 + (b (u n2 + teh))^2 at the upper ends.
        const double u_max = std::exp(0.5 * c->lnw_max), u_min = std::exp(0.5 * c->lnw_min);
        const double tau_max = std::exp(hi[2]), tau_min = std::exp(lo[2]);
        const double teh = tau_max * mag(lo[3], hi[3]) * 0.70710678118654752440;
        const double g = u_max * tau_max + teh, n2 = teh * teh + g * g, b = 1e50 * u_max;
        const double X = n2 + b * g, Y = b * (u_max * n2 + teh);
        const double d_max = X * X + Y * Y, d_min = std::pow(u_min * tau_min, 4.0);
        ok = lo[1] >= 0.0 && hi[1] <= 1.0 && lo[4] >= 0.0 && hi[4] <= 1.0 && mag(lo[0], hi[0]) <= 1e10 &&
             d_max <= 1e140 && d_min >= 1e-140;          // false for NaN / inf anywhere
        return ok ? BOUNDS_FAST : 0;
    } else {
        return 0;
    }
    if (!(ok && ymax <= YMAX)) return 0;
    const int terms = c->model_id == BISIP_MODEL_COLECOLE ? c->D : 2;
    return c->grid_ok && terms <= GRID_MAX_TERMS ? (BOUNDS_FAST | BOUNDS_GRID) : BOUNDS_FAST;
}

// PolynomialDecomposition, reduced form: (re)choose the expansion point bhat for the current
// prior box (host_precompute.h:reduced_center), record the kernel's estimated rounding error,
// refresh the kernarg copy of spectrum 0 and the device copies of a batch.
// the open-box prior of SURVEY.md §8 a2 on the host (rows outside it never reach a formulation)
static bool in_prior_host(const double *th, const bisip_ctx *c)
{
    for (int q = 0; q < c->ndim; ++q)
        if (!(c->bounds.lo[q] < th[q] && th[q] < c->bounds.hi[q])) return false;
    return true;
}

// Only what the current variant can run is (re)computed -- the probing is most of what a batch context costs
// to build.  Tier 0 (plain) for every spectrum under AUTO / REDUCED.  Tier 1 (compensated: operands from a QR
// in binary128, built on demand) for every spectrum under REDUCED_COMP, and under AUTO for exactly the spectra
// whose plain estimate fails (all of them once the guard has closed the plain tier or the mix): on
// well-conditioned designs -- every bundled spectrum, the headline shape -- the compensated tier is never looked
// at.  A spectrum's probe rows serve both tiers.  set_variant, set_bounds and the guard come back here.
// Which spectra of a batch take the plain tier inside a compensated launch (host.h: tier_of, mixed).
static int update_tiers(bisip_ctx *c)
{
    c->mixed = false;
    if (c->E <= 1 || c->variant != BISIP_VARIANT_AUTO || c->mix_off || c->demoted[0]) return BISIP_OK;
    if (!c->red[0].valid || !c->red[1].valid || !c->red[0].d_red || !c->red[1].d_red) return BISIP_OK;
    if (effective_variant(c) != BISIP_VARIANT_REDUCED_COMP) return BISIP_OK;
    const size_t E = (size_t)c->E;
    if (c->red[0].est.size() != E) return BISIP_OK;
    c->tier_of.assign(E, 1);
    size_t plain = 0;
    for (size_t e = 0; e < E; ++e)
        if (c->red[0].est[e] <= c->auto_err_max) { c->tier_of[e] = 0; ++plain; }
    if (plain == 0) return BISIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->d_tier) HIP_TRY(hipMalloc((void **)&c->d_tier, E));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(c->d_tier, c->tier_of.data(), E, hipMemcpyHostToDevice));
    c->mixed = true;
    return BISIP_OK;
}

// does spectrum e run the compensated tier under the context's present variant and history?
static bool needs_comp(const bisip_ctx *c, size_t e)
{
    if (2 * c->N < c->P + 2 && c->variant == BISIP_VARIANT_AUTO) return false;   // no triangle: AUTO runs the per-frequency form
    if (c->variant == BISIP_VARIANT_REDUCED_COMP) return true;
    if (c->variant != BISIP_VARIANT_AUTO) return false;
    if (c->demoted[0] || c->mix_off) return true;
    return c->red[0].done.size() > e && c->red[0].done[e] && !(c->red[0].est[e] <= c->auto_err_max);
}

// binary128 operands of the spectra in `need` that lack them: the kernel sums once per distinct frequency list
// (the spectra of a survey usually share one), then one QR per spectrum on the host threads
static void make_quad_operands(bisip_ctx *c, const std::vector<int64_t> &need)
{
    const int N = c->N, S = c->S, D = c->P + 1;
    std::vector<int64_t> todo;
    for (int64_t e : need)
        if (!c->reduced[(size_t)e].has_quad()) todo.push_back(e);
    if (todo.empty()) return;
    // group by frequency list
    std::vector<int64_t> rep;                       // a representative spectrum of every distinct list
    std::vector<int> group(todo.size());
    for (size_t i = 0; i < todo.size(); ++i) {
        const double *we = &c->h_w[(size_t)todo[i] * N];
        int g = -1;
        for (size_t k = rep.size(); k-- > 0 && g < 0;)      // the latest list first: consecutive spectra share theirs
            if (std::memcmp(&c->h_w[(size_t)rep[k] * N], we, sizeof(double) * (size_t)N) == 0) g = (int)k;
        if (g < 0) { g = (int)rep.size(); rep.push_back(todo[i]); }
        group[i] = g;
    }
    std::vector<std::shared_ptr<const QuadKernelSums>> ks(rep.size());
    if (rep.size() == 1)       // one frequency list (a lone spectrum, a survey): its rows over the host threads
        ks[0] = polydecomp_kernel_sums_quad(N, &c->h_w[(size_t)rep[0] * N], S, c->h_taus.data(), D, c->h_log_taus.data(), c->c_exp, true);
    else
        parallel_blocks((int64_t)rep.size(), 1, [&](int64_t lo, int64_t hi) {
            for (int64_t k = lo; k < hi; ++k)
                ks[(size_t)k] = polydecomp_kernel_sums_quad(N, &c->h_w[(size_t)rep[(size_t)k] * N], S, c->h_taus.data(), D,
                                                            c->h_log_taus.data(), c->c_exp);
        });
    parallel_blocks((int64_t)todo.size(), 1, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const size_t e = (size_t)todo[(size_t)i];
            reduced_make_quad(*ks[(size_t)group[(size_t)i]], &c->h_zn[e * 2 * N], &c->h_err[e * 2 * N], c->reduced[e]);
        }
    });
}

static int recenter_reduced(bisip_ctx *c)
{
    if (c->model_id != BISIP_MODEL_POLYDECOMP || c->reduced.empty()) return BISIP_OK;
    const int n = c->P + 2;
    const size_t tri = (size_t)n * (n + 1) / 2;
    const size_t E = c->reduced.size();
    const double w_shell = reduced_shell_weight();          // read once, on this thread
    // == sizeof(ReducedArgs<P, COMP>)/8: the compensated tier's image starts with the triangle's low word
    // as floats, an even number of them
    const size_t lo_floats = (tri + 1) & ~(size_t)1;
    const size_t red_doubles[2] = {tri + 3 * (size_t)n + 1, tri + 3 * (size_t)n + 1 + lo_floats / 2};
    // a device image of a tier's operands: every tier of a batch; of a lone spectrum its compensated tier from
    // degree 6 on, where the bulk kernel reads the operands from memory (dispatch_logprob.hip: launch_reduced)
    auto has_image = [&](int tier) { return c->E > 1 || (tier == 1 && c->P >= REDUCED_COMP_MEMORY_OPERANDS_FROM); };
    for (int tier = 0; tier < 2; ++tier) {
        bisip_ctx::ReducedTier &T = c->red[tier];
        if (T.done.size() != E) { T.done.assign(E, 0); T.est.assign(E, 0.0); }
        if (has_image(tier) && T.image.size() != red_doubles[tier] * E) T.image.assign(red_doubles[tier] * E, 0.0);
    }
    // one spectrum's entry of a tier: its image (batch) and, for spectrum 0, the kernarg copy
    auto store = [&](int tier, size_t e, const double *Rfull, const float *Rlo_full, double rest, const double *bh,
                     const double *ev, const double *el) {
        bisip_ctx::ReducedTier &T = c->red[tier];
        std::vector<double> Rp;
        std::vector<float> Rlo;
        for (int i = 0; i < n; ++i)
            for (int j = i; j < n; ++j) {
                Rp.push_back(Rfull[(size_t)i * n + j]);
                if (tier == 1) Rlo.push_back(Rlo_full[(size_t)i * n + j]);
            }
        if (tier == 1) Rlo.resize(lo_floats, 0.0f);
        if (e == 0) {
            T.Rpacked = Rp; T.Rlo_packed = Rlo; T.rest = rest;
            T.bhat.assign(bh, bh + n); T.evec.assign(ev, ev + n); T.elo.assign(el, el + n);
        }
        if (has_image(tier)) {  // ReducedArgs<P, COMP> image: [Rlo |] R | bhat | e | elo | rest
            double *dst = &T.image[red_doubles[tier] * e];
            if (tier == 1) { std::memcpy(dst, Rlo.data(), lo_floats * sizeof(float)); dst += lo_floats / 2; }
            dst = std::copy(Rp.begin(), Rp.end(), dst);
            dst = std::copy(bh, bh + n, dst);
            dst = std::copy(ev, ev + n, dst);
            dst = std::copy(el, el + n, dst);
            *dst = rest;
        }
    };
    bool changed[2] = {false, false};
    // ---- tier 0, and tier 1 of the spectra known to need it, in one pass over the spectra (shared probes)
    const bool want0 = (c->variant == BISIP_VARIANT_AUTO || c->variant == BISIP_VARIANT_REDUCED) && !c->red[0].valid;
    auto pass = [&](const std::vector<int64_t> &which, bool do0) {
        // the binary128 operands of the spectra already known to need tier 1 (under AUTO a spectrum is known
        // to once its plain estimate has failed: those come back in a second pass)
        std::vector<int64_t> need;
        for (int64_t e : which)
            if (needs_comp(c, (size_t)e) && !c->red[1].done[(size_t)e]) need.push_back(e);
        make_quad_operands(c, need);
        parallel_blocks((int64_t)which.size(), 4, [&](int64_t i_lo, int64_t i_hi) {
            std::vector<double> bh(n), ev(n), el(n);
            ReducedProbes probes;
            for (int64_t i = i_lo; i < i_hi; ++i) {
                const size_t e = (size_t)which[(size_t)i];
                bisip::ReducedProblem &rp = c->reduced[e];
                reduced_probes(rp, c->bounds.lo, c->bounds.hi, probes);
                if (do0 && !c->red[0].done[e]) {
                    c->red[0].est[e] = reduced_center_plain(rp, probes, c->bounds.lo, c->bounds.hi, w_shell, bh.data(), ev.data(), el.data());
                    store(0, e, rp.R.data(), nullptr, rp.rest, bh.data(), ev.data(), el.data());
                    c->red[0].done[e] = 1;
                }
                if (rp.has_quad() && needs_comp(c, e) && !c->red[1].done[e]) {
                    c->red[1].est[e] = reduced_center_comp(rp, probes, c->bounds.lo, c->bounds.hi, w_shell, bh.data(), ev.data(), el.data());
                    store(1, e, rp.Rc.data(), rp.Rc_lo.data(), rp.rest_c, bh.data(), ev.data(), el.data());
                    c->red[1].done[e] = 1;
                }
            }
        });
    };
    std::vector<int64_t> all(E);
    for (size_t e = 0; e < E; ++e) all[e] = (int64_t)e;
    if (want0) {
        pass(all, true);
        changed[0] = true;
        c->red[0].valid = true;
        c->red[0].err = 0.0;
        for (double v : c->red[0].est)
            if (!(v <= c->red[0].err)) c->red[0].err = v;
    }
    // ---- tier 1 for the spectra that turned out to need it (their plain estimate failed just now, or the
    // variant / the guard asks for it)
    bool any_comp = false;
    std::vector<int64_t> late;
    for (size_t e = 0; e < E; ++e)
        if (needs_comp(c, e)) {
            any_comp = true;
            if (!c->red[1].done[e]) late.push_back((int64_t)e);
        }
    if (!late.empty()) pass(late, false);
    if (any_comp) {
        changed[1] = changed[1] || !late.empty() || !c->red[1].valid;
        c->red[1].valid = true;
        c->red[1].err = 0.0;
        for (size_t e = 0; e < E; ++e)
            if (needs_comp(c, e) && !(c->red[1].est[e] <= c->red[1].err)) c->red[1].err = c->red[1].est[e];
        if (c->E == 1 && c->red[1].Rpacked.empty()) return fail(BISIP_EHIP, "internal: compensated operands missing");
    }
    // a batch's compensated image needs an entry for EVERY spectrum the kernel may index: spectra on the plain
    // tier of a mixed launch read their plain operands (BatchArgs::red_plain) and leave theirs at zero
    {
        HIP_TRY(hipSetDevice(c->device));
        for (int tier = 0; tier < 2; ++tier) {
            bisip_ctx::ReducedTier &T = c->red[tier];
            if (!changed[tier] || !has_image(tier) || T.image.empty()) continue;
            if (!T.d_red) HIP_TRY(hipMalloc(&T.d_red, T.image.size() * sizeof(double)));
            // set_bounds between launches: the copy is ordered after earlier work by the sync
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemcpy(T.d_red, T.image.data(), T.image.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    return update_tiers(c);
}

// no C++ exception may cross the C ABI: the host-side precompute allocates (std::vector)
template <class F>
static int guarded(F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(BISIP_ENOMEM, "out of host memory");
    } catch (const std::exception &ex) {
        return fail(BISIP_EINVAL, "internal error: %s", ex.what());
    }
}

namespace {
constexpr int FP64_PROBE_GROUPS = 256 * 8;     // 8 waves per SIMD on each of the 256 compute units
// one wave; sleeps (s_sleep: no issue slots taken from the kernel under measurement) until the
// 100 MHz counter has advanced by `ticks`, at most `max_polls` polls so that the wave always ends
__global__ __launch_bounds__(64) void k_clock_probe(long long *out, long long ticks, int max_polls)
{
    const long long t0 = clock64(), r0 = wall_clock64();
    long long r1 = r0;
    for (int i = 0; i < max_polls && r1 - r0 < ticks; ++i) {
        __builtin_amdgcn_s_sleep(64);
        r1 = wall_clock64();
    }
    const long long t1 = clock64();
    r1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t0; out[1] = t1; out[2] = r0; out[3] = r1; }
}

// A stream of independent fp64 FMAs, nothing to wait for, no memory: the ceiling the compute-bound kernels are held
// to (bisip_fp64_stream_probe_dev).  8 accumulators in fixed registers, d = d * s + v with one scalar source (the
// commonest form in kernels.h), operands with full mantissas -- a stream of small integers draws less power and
// holds a higher clock than any real kernel (benchmarks/micro/fp64_stream_ceiling.hip).
#define BISIP_FMA8(d0, d1, d2, d3, d4, d5, d6, d7)                                                            \
    "v_fma_f64 " d0 ", " d0 ", s[20:21], v[42:43]\n v_fma_f64 " d1 ", " d1 ", s[20:21], v[42:43]\n"            \
    "v_fma_f64 " d2 ", " d2 ", s[20:21], v[42:43]\n v_fma_f64 " d3 ", " d3 ", s[20:21], v[42:43]\n"            \
    "v_fma_f64 " d4 ", " d4 ", s[20:21], v[42:43]\n v_fma_f64 " d5 ", " d5 ", s[20:21], v[42:43]\n"            \
    "v_fma_f64 " d6 ", " d6 ", s[20:21], v[42:43]\n v_fma_f64 " d7 ", " d7 ", s[20:21], v[42:43]\n"
#define BISIP_FMA8_ROUND BISIP_FMA8("v[20:21]", "v[22:23]", "v[24:25]", "v[26:27]", "v[28:29]", "v[30:31]", "v[32:33]", "v[34:35]")
#define BISIP_FMA_CLOBBER "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35"
__global__ __launch_bounds__(256) void k_fp64_stream_probe(double *out, int rounds, double seed)
{
    asm volatile("v_mov_b64 v[40:41], %0\n v_mov_b64 v[42:43], %1\n s_mov_b64 s[20:21], %2\n"
                 "v_mov_b64 v[20:21], v[40:41]\n v_mov_b64 v[22:23], v[42:43]\n v_mov_b64 v[24:25], v[40:41]\n"
                 "v_mov_b64 v[26:27], v[42:43]\n v_mov_b64 v[28:29], v[40:41]\n v_mov_b64 v[30:31], v[42:43]\n"
                 "v_mov_b64 v[32:33], v[40:41]\n v_mov_b64 v[34:35], v[42:43]\n"
                 :: "v"(seed * (1.0 + 1e-9 * threadIdx.x)), "v"(0.7853981633974483 + 1e-7 * threadIdx.x), "s"(1.0000003141592653)
                 : BISIP_FMA_CLOBBER, "v40", "v41", "v42", "v43", "s20", "s21");
    for (int it = 0; it < rounds; ++it)
        asm volatile(BISIP_FMA8_ROUND BISIP_FMA8_ROUND BISIP_FMA8_ROUND BISIP_FMA8_ROUND ::: BISIP_FMA_CLOBBER);
    double r;
    asm volatile("v_add_f64 %0, v[20:21], v[34:35]" : "=v"(r));
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = r;
}
}  // namespace

// A very big host-buffer call as a pipeline over chunks of rows: the caller's (pageable) rows are copied into a
// pinned buffer by the host threads, go up on one stream, are evaluated there, and the results come back on a
// second stream into a pinned buffer from which the host threads copy them out -- chunk k+1 is being staged while
// chunk k travels and chunk k-1 returns; the device holds two chunks, not the whole batch (bisip_forward of 4M
// rows: 2.3 GB otherwise).  What it buys in time is modest and only beyond the host's last-level cache: measured
// (profiles/r05_micro_host_pipeline.txt) a call is bound by the host's copy out of pageable memory -- 54 GB/s over the
// link while the source sits in cache (64 MB), 37-41 GB/s from DRAM (256 MB) whichever way it is staged, overlapped
// or not -- so the pipeline starts at 192 MB (+4-6 % there) and everything smaller keeps the one-launch path.
// launch(d_in, rows, d_out, stream) enqueues the kernel for one chunk.  Same bits as one launch over everything: a
// row's value does not depend on where it sits in a batch.
constexpr size_t PIPE_CHUNK_BYTES = 64u << 20;      // rows in + results out of one chunk
constexpr size_t PIPE_FROM_BYTES = 192u << 20;      // smaller calls: the direct path (one synchronisation)

template <class Launch>
static int host_pipeline(bisip_ctx *c, const char *in, size_t in_row, char *out, size_t out_row, int64_t W, Launch &&launch)
{
    int64_t rows = (int64_t)(PIPE_CHUNK_BYTES / (in_row + out_row));
    rows = rows < 256 ? 256 : (rows / 256) * 256;
    const size_t in_al = ((size_t)rows * in_row + 255) & ~(size_t)255, out_al = ((size_t)rows * out_row + 255) & ~(size_t)255;
    const size_t need = 2 * (in_al + out_al);
    if (c->pipe_bytes < need) {
        if (c->h_pipe) { HIP_TRY(hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_pipe); c->h_pipe = nullptr; c->pipe_bytes = 0; }
        HIP_TRY(hipHostMalloc((void **)&c->h_pipe, need, hipHostMallocDefault));
        c->pipe_bytes = need;
    }
    if (!c->stream_back) HIP_TRY(hipStreamCreateWithFlags(&c->stream_back, hipStreamNonBlocking));
    for (hipEvent_t &ev : c->pipe_ev)
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int rc = ensure_ws(c, need);
    if (rc != BISIP_OK) return rc;
    char *h_in[2] = {c->h_pipe, c->h_pipe + in_al}, *h_out[2] = {c->h_pipe + 2 * in_al, c->h_pipe + 2 * in_al + out_al};
    char *d_in[2] = {(char *)c->d_ws, (char *)c->d_ws + in_al}, *d_out[2] = {(char *)c->d_ws + 2 * in_al, (char *)c->d_ws + 2 * in_al + out_al};
    hipEvent_t *up = c->pipe_ev, *done = c->pipe_ev + 2, *back = c->pipe_ev + 4;     // per buffer: rows are up / evaluated / results are back
    const int64_t chunks = (W + rows - 1) / rows;
    auto copy = [](char *dst, const char *src, size_t bytes) {
        parallel_blocks((int64_t)bytes, 1 << 20, [&](int64_t lo, int64_t hi) { std::memcpy(dst + lo, src + lo, (size_t)(hi - lo)); });
    };
    auto collect = [&](int64_t k) -> int {         // chunk k's results, once they are back
        const int b = (int)(k & 1);
        const int64_t r0 = k * rows, n = (W - r0 < rows) ? W - r0 : rows;
        HIP_TRY(hipEventSynchronize(back[b]));
        copy(out + (size_t)r0 * out_row, h_out[b], (size_t)n * out_row);
        return BISIP_OK;
    };
    for (int64_t k = 0; k < chunks; ++k) {
        const int b = (int)(k & 1);
        const int64_t r0 = k * rows, n = (W - r0 < rows) ? W - r0 : rows;
        if (k >= 2) HIP_TRY(hipEventSynchronize(up[b]));                 // chunk k-2 has left this staging buffer
        copy(h_in[b], in + (size_t)r0 * in_row, (size_t)n * in_row);
        HIP_TRY(hipMemcpyAsync(d_in[b], h_in[b], (size_t)n * in_row, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(up[b], c->stream));
        if (k >= 2) HIP_TRY(hipStreamWaitEvent(c->stream, back[b], 0));  // chunk k-2's results have left d_out[b]
        rc = launch(d_in[b], n, d_out[b], c->stream);
        if (rc != BISIP_OK) { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->stream_back); return rc; }
        HIP_TRY(hipEventRecord(done[b], c->stream));
        if (k >= 1) { rc = collect(k - 1); if (rc != BISIP_OK) return rc; }   // (h_out[b] was emptied by collect(k - 2) one round ago)
        HIP_TRY(hipStreamWaitEvent(c->stream_back, done[b], 0));
        HIP_TRY(hipMemcpyAsync(h_out[b], d_out[b], (size_t)n * out_row, hipMemcpyDeviceToHost, c->stream_back));
        HIP_TRY(hipEventRecord(back[b], c->stream_back));
    }
    return collect(chunks - 1);
}

extern "C" {

int bisip_abi_version(void) { return BISIP_ABI_VERSION; }

int bisip_clock_probe_dev(int64_t *d_out, double window_us, void *stream)
{
    if (!d_out) return fail(BISIP_EINVAL, "bisip_clock_probe_dev: null output");
    if (!(window_us > 0.0) || window_us > 1e5) return fail(BISIP_EINVAL, "bisip_clock_probe_dev: window_us out of (0, 1e5]");
    const long long ticks = (long long)(window_us * 100.0);
    // one poll sleeps 64*64 = 4096 cycles (~2 us): bound the loop at 4x the polls the window needs
    const int max_polls = (int)(window_us * 2.0) + 64;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long *)d_out, ticks, max_polls);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

int64_t bisip_fp64_stream_probe_lanes(void) { return (int64_t)FP64_PROBE_GROUPS * 256; }

int bisip_fp64_stream_probe_dev(double *d_out, int rounds, int64_t *wave_instructions, void *stream)
{
    if (!d_out) return fail(BISIP_EINVAL, "bisip_fp64_stream_probe_dev: null output");
    if (rounds < 1 || rounds > (1 << 20)) return fail(BISIP_EINVAL, "bisip_fp64_stream_probe_dev: rounds=%d not in [1, 2^20]", rounds);
    hipLaunchKernelGGL(k_fp64_stream_probe, dim3(FP64_PROBE_GROUPS), dim3(256), 0, (hipStream_t)stream, d_out, rounds, 1.2345678901234567);
    HIP_TRY(hipGetLastError());
    if (wave_instructions) *wave_instructions = (int64_t)FP64_PROBE_GROUPS * 4 * 32 * rounds;
    return BISIP_OK;
}

const char *bisip_last_error(void) { return g_err; }

int bisip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bisip_polydecomp_operands(int N, const double *w, const double *zn, const double *zn_err,
                              const bisip_model_desc *desc, double *G_re, double *G_im,
                              double *R, double *bhat, double *e, double *rest, double *lconst)
{
    if (!w || !zn || !zn_err || !desc || !desc->taus || !desc->log_taus || !G_re || !G_im || !R ||
        !bhat || !e || !rest || !lconst)
        return fail(BISIP_EINVAL, "null argument");
    if (N < 1 || desc->poly_deg < 0 || desc->poly_deg > BISIP_MAX_POLY_DEG || desc->n_taus < 1)
        return fail(BISIP_EINVAL, "bad shape");
    return guarded([&] {
        PolyDecompOperands o;
        const int D = desc->poly_deg + 1, n = D + 1;
        polydecomp_operands(N, w, desc->n_taus, desc->taus, D, desc->log_taus, desc->c_exp, zn, zn_err, o);
        std::memcpy(G_re, o.G_re.data(), sizeof(double) * (size_t)N * D);
        std::memcpy(G_im, o.G_im.data(), sizeof(double) * (size_t)N * D);
        std::memcpy(R, o.R.data(), sizeof(double) * (size_t)n * n);
        std::memcpy(bhat, o.bhat.data(), sizeof(double) * n);
        std::memcpy(e, o.e.data(), sizeof(double) * n);
        *rest = o.rest;
        *lconst = loglike_const(2 * N, zn_err);
        return (int)BISIP_OK;
    });
}

int bisip_polydecomp_reduced_estimates(int N, const double *w, const double *zn, const double *zn_err,
                                       const bisip_model_desc *desc, const double *lo, const double *hi,
                                       double *est)
{
    if (!w || !zn || !zn_err || !desc || !desc->taus || !desc->log_taus || !lo || !hi || !est)
        return fail(BISIP_EINVAL, "null argument");
    if (N < 1 || desc->poly_deg < 0 || desc->poly_deg > BISIP_MAX_POLY_DEG || desc->n_taus < 1)
        return fail(BISIP_EINVAL, "bad shape");
    return guarded([&] {
        PolyDecompOperands o;
        const int D = desc->poly_deg + 1, n = D + 1;
        polydecomp_operands(N, w, desc->n_taus, desc->taus, D, desc->log_taus, desc->c_exp, zn, zn_err, o);
        ReducedProblem p;
        reduced_from_operands(o, loglike_const(2 * N, zn_err), p);
        if (2 * N >= n)     // a design with a triangle: the compensated tier's operands in binary128
            reduced_make_quad(*polydecomp_kernel_sums_quad(N, w, desc->n_taus, desc->taus, D, desc->log_taus, desc->c_exp, true), zn, zn_err, p);
        ReducedProbes probes;
        reduced_probes(p, lo, hi, probes);
        const double w_shell = reduced_shell_weight();
        std::vector<double> bh(n), ev(n), el(n);
        est[0] = reduced_center_plain(p, probes, lo, hi, w_shell, bh.data(), ev.data(), el.data());
        est[1] = reduced_center_comp(p, probes, lo, hi, w_shell, bh.data(), ev.data(), el.data());
        return (int)BISIP_OK;
    });
}

int bisip_polydecomp_reduced_reference(int N, const double *w, const double *zn, const double *zn_err,
                                       const bisip_model_desc *desc, const double *theta, int64_t W, double *logp)
{
    if (!w || !zn || !zn_err || !desc || !desc->taus || !desc->log_taus || (W > 0 && (!theta || !logp)))
        return fail(BISIP_EINVAL, "null argument");
    if (N < 1 || W < 0 || desc->poly_deg < 0 || desc->poly_deg > BISIP_MAX_POLY_DEG || desc->n_taus < 1)
        return fail(BISIP_EINVAL, "bad shape");
    if (2 * N < desc->poly_deg + 2) return fail(BISIP_EUNSUPPORTED, "2N < poly_deg + 2: the design has no triangle");
    return guarded([&] {
        // the reduced form with operands from a QR in binary128, evaluated in binary128
        const int D = desc->poly_deg + 1;
        ReducedProblem p;
        p.n = D + 1;
        p.lconst = loglike_const(2 * N, zn_err);
        reduced_make_quad(*polydecomp_kernel_sums_quad(N, w, desc->n_taus, desc->taus, D, desc->log_taus, desc->c_exp, true), zn, zn_err, p);
        parallel_blocks(W, 256, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) logp[i] = reduced_logp_reference(p, theta + i * p.n);
        });
        return (int)BISIP_OK;
    });
}

static int build_context(bisip_ctx **out, int device, int model_id, int E, int N, const double *w,
                         const double *zn, const double *zn_err, int ndim, const double *lo,
                         const double *hi, const bisip_model_desc *desc)
{
    if (!out || !w || !zn || !zn_err || !lo || !hi) return fail(BISIP_EINVAL, "null argument");
    *out = nullptr;
    if (E < 1 || E > (1 << 24)) return fail(BISIP_EINVAL, "n_spectra=%d out of range", E);
    if (N < 1 || N > 4096) return fail(BISIP_EINVAL, "N=%d out of range [1,4096]", N);
    if (ndim < 1 || ndim > BISIP_MAX_NDIM) return fail(BISIP_EINVAL, "ndim=%d out of range", ndim);
    for (long long i = 0; i < 2LL * N * E; ++i)
        if (!(zn_err[i] > 0.0) || !std::isfinite(zn_err[i]) || !std::isfinite(zn[i]))
            return fail(BISIP_EINVAL, "zn/zn_err[%lld] must be finite and zn_err > 0", i);
    for (long long j = 0; j < (long long)N * E; ++j)
        if (!(w[j] > 0.0) || !std::isfinite(w[j])) return fail(BISIP_EINVAL, "w[%lld] must be finite and > 0", j);

    int P = 0, D = 0, S = 0;
    switch (model_id) {
    case BISIP_MODEL_POLYDECOMP:
        if (!desc || !desc->taus || !desc->log_taus) return fail(BISIP_EINVAL, "PolynomialDecomposition needs taus/log_taus");
        P = desc->poly_deg; S = desc->n_taus;
        if (P < 0 || P > BISIP_MAX_POLY_DEG) return fail(BISIP_EUNSUPPORTED, "poly_deg=%d not in [0,%d]", P, BISIP_MAX_POLY_DEG);
        if (S < 1 || S > 8192) return fail(BISIP_EINVAL, "n_taus=%d out of range", S);
        if (ndim != P + 2) return fail(BISIP_EINVAL, "ndim=%d but poly_deg+2=%d", ndim, P + 2);
        break;
    case BISIP_MODEL_COLECOLE:
        if (!desc) return fail(BISIP_EINVAL, "PeltonColeCole needs n_modes");
        D = desc->n_modes;
        if (D < 1 || D > BISIP_MAX_MODES) return fail(BISIP_EUNSUPPORTED, "n_modes=%d not in [1,%d]", D, BISIP_MAX_MODES);
        if (ndim != 1 + 3 * D) return fail(BISIP_EINVAL, "ndim=%d but 1+3*n_modes=%d", ndim, 1 + 3 * D);
        break;
    case BISIP_MODEL_DIAS2000:
        if (ndim != 5) return fail(BISIP_EINVAL, "Dias2000 has ndim 5, got %d", ndim);
        break;
    case BISIP_MODEL_SHIN2015:
        if (ndim != 6) return fail(BISIP_EINVAL, "Shin2015 has ndim 6, got %d", ndim);
        break;
    default: return fail(BISIP_EINVAL, "bad model_id %d", model_id);
    }

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(BISIP_EINVAL, "device %d not in [0,%d)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    bisip_ctx *c = new (std::nothrow) bisip_ctx;
    if (!c) return fail(BISIP_ENOMEM, "out of host memory");
    c->device = device; c->model_id = model_id; c->N = N; c->ndim = ndim; c->E = E;
    c->P = P; c->D = D; c->S = S;
    if (model_id == BISIP_MODEL_POLYDECOMP) c->auto_err_max = auto_err_max_from_env();
    for (int q = 0; q < MAXD; ++q) { c->bounds.lo[q] = 0.0; c->bounds.hi[q] = 0.0; }
    for (int q = 0; q < ndim; ++q) { c->bounds.lo[q] = lo[q]; c->bounds.hi[q] = hi[q]; }
    c->lnw_min = INFINITY; c->lnw_max = -INFINITY;
    for (long long j = 0; j < (long long)N * E; ++j) {
        const double l = std::log(w[j]);
        c->lnw_min = std::fmin(c->lnw_min, l);
        c->lnw_max = std::fmax(c->lnw_max, l);
    }
    c->bounds.flags = bound_flags(c);

    const int rec = model_id == BISIP_MODEL_POLYDECOMP ? 4 + 2 * (P + 1) : 8;
    c->cb_stride = (long long)N * rec;
    std::vector<double> cb((size_t)E * N * rec, 0.0), lconsts(E), cb_lp;
    if (model_id == BISIP_MODEL_POLYDECOMP) cb_lp.assign((size_t)E * N * rec, 0.0);
    int rc = BISIP_OK;
    if (model_id == BISIP_MODEL_POLYDECOMP) {
        c->c_exp = desc->c_exp;
        c->reduced.resize((size_t)E);
        // what the compensated tier's binary128 operands are built from, if a spectrum ever needs them
        c->h_w.assign(w, w + (size_t)E * N);
        c->h_zn.assign(zn, zn + (size_t)E * 2 * N);
        c->h_err.assign(zn_err, zn_err + (size_t)E * 2 * N);
        c->h_taus.assign(desc->taus, desc->taus + S);
        c->h_log_taus.assign(desc->log_taus, desc->log_taus + (size_t)(P + 1) * S);
    }
    std::vector<double> fb;       // E == 1: loop-faithful records (see k_logprob_pd_faithful)
    // per-spectrum operands, blocks of spectra on host threads (each writes its own slots).  The
    // kernel sums K, G depend on the frequencies only: a block reuses them while consecutive
    // spectra share their frequency list, as the spectra of a survey usually do.
    std::atomic<int> on_grid{0};
    parallel_blocks(E, 8, [&](int64_t e_lo, int64_t e_hi) {
        PolyDecompOperands o;
        const double *w_of_o = nullptr;
        for (int64_t e = e_lo; e < e_hi; ++e) {
            const double *we = w + (size_t)e * N, *zne = zn + (size_t)e * 2 * N, *erre = zn_err + (size_t)e * 2 * N;
            lconsts[e] = loglike_const(2 * N, erre);
            std::vector<double> lnw, iv;
            common_operands(N, we, erre, lnw, iv);
            double *base = &cb[(size_t)e * N * rec];
            for (int j = 0; j < N; ++j) {
                double *r = base + (size_t)j * rec;
                r[0] = zne[j]; r[1] = zne[N + j]; r[2] = iv[j]; r[3] = iv[N + j];
                if (model_id != BISIP_MODEL_POLYDECOMP) { r[4] = we[j]; r[5] = lnw[j]; r[6] = (double)sqrtl((long double)we[j]); }
            }
            if (model_id != BISIP_MODEL_POLYDECOMP) {
                double dlnw = 0.0;
                if (grid_step(N, we, lnw.data(), &dlnw)) on_grid.store(1);
                for (int j = 0; j < N; ++j) base[(size_t)j * rec + 7] = dlnw;
            }
            if (model_id != BISIP_MODEL_POLYDECOMP) continue;
            if (!w_of_o || std::memcmp(w_of_o, we, sizeof(double) * (size_t)N) != 0) {
                polydecomp_kernel_sums(N, we, S, desc->taus, P + 1, desc->log_taus, desc->c_exp, o);
                w_of_o = we;
            }
            polydecomp_reduce(zne, erre, o);
            for (int j = 0; j < N; ++j) {
                double *r = base + (size_t)j * rec;
                for (int p = 0; p <= P; ++p) {
                    r[4 + p] = o.G_re[(size_t)j * (P + 1) + p];
                    r[4 + P + 1 + p] = o.G_im[(size_t)j * (P + 1) + p];
                }
            }
            for (int j = 0; j < N; ++j) {  // 1/sigma-weighted rows for the collapsed log-prob kernel
                double *r = &cb_lp[(size_t)e * N * rec + (size_t)j * rec];
                const long double sr = 1.0L / (long double)erre[j], si = 1.0L / (long double)erre[N + j];
                r[0] = (double)((long double)zne[j] * sr); r[1] = (double)((long double)zne[N + j] * si);
                r[2] = (double)(-sr); r[3] = 0.0;
                for (int p = 0; p <= P; ++p) {
                    r[4 + p] = (double)(sr * (long double)o.G_re[(size_t)j * (P + 1) + p]);
                    r[4 + P + 1 + p] = (double)(si * (long double)o.G_im[(size_t)j * (P + 1) + p]);
                }
            }
            reduced_from_operands(o, lconsts[e], c->reduced[(size_t)e]);
            if (E == 1) {
                const int JB = 16, nb = (N + JB - 1) / JB;
                const size_t blk_stride = 4 * (size_t)JB + (size_t)S * 2 * JB;
                fb.assign((size_t)S * 8 + (size_t)nb * blk_stride, 0.0);
                for (int k = 0; k < S; ++k)
                    for (int p = 0; p <= P && p < 8; ++p) fb[(size_t)k * 8 + p] = desc->log_taus[(size_t)p * S + k];
                for (int j = 0; j < N; ++j) {
                    double *blk = &fb[(size_t)S * 8 + (size_t)(j / JB) * blk_stride];
                    const int jj = j % JB;
                    blk[jj] = zne[j]; blk[JB + jj] = zne[N + j];
                    blk[2 * JB + jj] = iv[j]; blk[3 * JB + jj] = iv[N + j];
                    for (int k = 0; k < S; ++k) {
                        blk[4 * JB + (size_t)k * 2 * JB + jj] = o.K_re[(size_t)j * S + k];
                        blk[4 * JB + (size_t)k * 2 * JB + JB + jj] = o.K_im[(size_t)j * S + k];
                    }
                }
            }
        }
    });
    // the stepped loop is chosen per spectrum (rec[7] = its step, 0 off any grid): the flag says that some
    // spectrum of the context is on a grid.  (BOUNDS_FAST stays a property of the context: the box against the
    // frequency range of ALL its spectra.)
    c->grid_ok = model_id != BISIP_MODEL_POLYDECOMP && on_grid.load() != 0 && std::getenv("BISIP_NO_GRID") == nullptr;
    c->bounds.flags = bound_flags(c);
    if (!fb.empty() && P < 8) rc = upload(&c->d_cb_faithful, fb);
    c->lconst = lconsts[0];
    if (rc == BISIP_OK) rc = upload(&c->d_cb, cb);
    if (rc == BISIP_OK && !cb_lp.empty()) rc = upload(&c->d_cb_lp, cb_lp);
    if (rc == BISIP_OK && (E > 1 || (model_id == BISIP_MODEL_POLYDECOMP && P >= REDUCED_COMP_MEMORY_OPERANDS_FROM))) rc = upload(&c->d_lconst, lconsts);
    if (rc == BISIP_OK) rc = recenter_reduced(c);
    if (rc == BISIP_OK) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) rc = fail(BISIP_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    if (rc != BISIP_OK) { bisip_ctx_destroy(c); return rc; }
    c->kernel_name = name_for(c);
    *out = c;
    return BISIP_OK;
}

int bisip_ctx_create(bisip_ctx **out, int device, int model_id, int N, const double *w,
                     const double *zn, const double *zn_err, int ndim, const double *lo,
                     const double *hi, const bisip_model_desc *desc)
{
    return guarded([&] { return build_context(out, device, model_id, 1, N, w, zn, zn_err, ndim, lo, hi, desc); });
}

int bisip_batch_create(bisip_ctx **out, int device, int model_id, int n_spectra, int N,
                       const double *w, const double *zn, const double *zn_err, int ndim,
                       const double *lo, const double *hi, const bisip_model_desc *desc)
{
    return guarded([&] { return build_context(out, device, model_id, n_spectra, N, w, zn, zn_err, ndim, lo, hi, desc); });
}

int bisip_ctx_nspectra(const bisip_ctx *c) { return c ? c->E : BISIP_EINVAL; }

void bisip_ctx_destroy(bisip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_cb) (void)hipFree(c->d_cb);
    if (c->d_cb_lp) (void)hipFree(c->d_cb_lp);
    if (c->d_cb_faithful) (void)hipFree(c->d_cb_faithful);
    if (c->d_lconst) (void)hipFree(c->d_lconst);
    for (auto &t : c->red) if (t.d_red) (void)hipFree(t.d_red);
    if (c->d_tier) (void)hipFree(c->d_tier);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_gather) (void)hipFree(c->d_gather);
    if (c->d_group) (void)hipFree(c->d_group);
    if (c->d_packed) (void)hipFree(c->d_packed);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_pipe) (void)hipHostFree(c->h_pipe);
    for (hipEvent_t ev : c->pipe_ev) if (ev) (void)hipEventDestroy(ev);
    if (c->stream_back) (void)hipStreamDestroy(c->stream_back);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bisip_ctx_set_bounds(bisip_ctx *c, const double *lo, const double *hi)
{
    if (!c || !lo || !hi) return fail(BISIP_EINVAL, "null argument");
    bool same = true;
    for (int q = 0; q < c->ndim; ++q) same = same && c->bounds.lo[q] == lo[q] && c->bounds.hi[q] == hi[q];
    if (same) return BISIP_OK;
    for (int q = 0; q < c->ndim; ++q) { c->bounds.lo[q] = lo[q]; c->bounds.hi[q] = hi[q]; }
    c->bounds.flags = bound_flags(c);
    c->demoted[0] = c->demoted[1] = false;                            // observed on the old box
    c->mix_off = false;
    for (auto &t : c->red) { t.valid = false; t.err = INFINITY; t.done.assign(t.done.size(), 0); }     // estimates and expansion points belong to the old box
    const int rc = guarded([&] { return recenter_reduced(c); });   // the reduced form expands about a point of the box
    c->kernel_name = name_for(c);                                  // AUTO may change formulation with the box
    return rc;
}

int bisip_ctx_set_variant(bisip_ctx *c, int variant)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (variant < BISIP_VARIANT_AUTO || variant > BISIP_VARIANT_REDUCED_COMP)
        return fail(BISIP_EINVAL, "bad variant %d", variant);
    if (variant == BISIP_VARIANT_WAVE && (c->model_id != BISIP_MODEL_POLYDECOMP || c->E > 1 || c->N > 64))
        return fail(BISIP_EUNSUPPORTED, "the wave-per-walker formulation needs a single-spectrum PolynomialDecomposition with N <= 64");
    if (c->model_id != BISIP_MODEL_POLYDECOMP && variant != BISIP_VARIANT_AUTO &&
        variant != BISIP_VARIANT_COLLAPSED)
        return fail(BISIP_EUNSUPPORTED, "this model has a single formulation");
    if (variant == BISIP_VARIANT_FAITHFUL && c->E > 1)
        return fail(BISIP_EUNSUPPORTED, "the faithful formulation has no batch-of-spectra kernel");
    if (variant == BISIP_VARIANT_FAITHFUL && !c->d_cb_faithful)
        return fail(BISIP_EUNSUPPORTED, "faithful variant needs poly_deg <= 7");
    c->variant = variant;
    const int rc = guarded([&] { return recenter_reduced(c); });   // a tier this variant needs may not have been estimated yet
    c->kernel_name = name_for(c);
    return rc;
}

int bisip_ctx_get_variant(const bisip_ctx *c) { return c ? effective_variant(c) : BISIP_EINVAL; }
int bisip_ctx_ndim(const bisip_ctx *c) { return c ? c->ndim : BISIP_EINVAL; }
int bisip_ctx_nfreq(const bisip_ctx *c) { return c ? c->N : BISIP_EINVAL; }
int bisip_ctx_device(const bisip_ctx *c) { return c ? c->device : BISIP_EINVAL; }
int bisip_ctx_loop_flags(const bisip_ctx *c) { return c ? c->bounds.flags : BISIP_EINVAL; }

int bisip_frequency_grid_step(int N, const double *w, double *step)
{
    if (!w || !step || N < 0) return fail(BISIP_EINVAL, "bad argument");
    return guarded([&] {
        std::vector<double> lnw((size_t)N);
        for (int j = 0; j < N; ++j) lnw[(size_t)j] = (double)logl((long double)w[j]);
        return grid_step(N, w, lnw.data(), step) ? 1 : 0;
    });
}
double bisip_ctx_loglike_const(const bisip_ctx *c) { return c ? c->lconst : NAN; }
const char *bisip_ctx_kernel_name(const bisip_ctx *c) { return c ? c->kernel_name : ""; }
double bisip_ctx_reduced_error(const bisip_ctx *c)
{
    if (!c) return NAN;
    const int v = effective_variant(c);
    if (v == BISIP_VARIANT_REDUCED) return c->red[0].err;
    if (v == BISIP_VARIANT_REDUCED_COMP && c->mixed) {           // each spectrum under the tier it runs
        double worst = 0.0;
        for (size_t e = 0; e < c->tier_of.size(); ++e) {
            const double x = c->red[c->tier_of[e]].est[e];
            if (!(x <= worst)) worst = x;
        }
        return worst;
    }
    if (v == BISIP_VARIANT_REDUCED_COMP) return c->red[1].err;
    return c->red[0].err <= c->red[1].err ? c->red[0].err : c->red[1].err;
}

int bisip_ctx_reduced_tiers(const bisip_ctx *c, int64_t *n_plain, int64_t *n_comp)
{
    if (!c || !n_plain || !n_comp) return fail(BISIP_EINVAL, "null argument");
    *n_plain = *n_comp = 0;
    const int v = effective_variant(c);
    if (c->model_id != BISIP_MODEL_POLYDECOMP) return BISIP_OK;
    if (v == BISIP_VARIANT_REDUCED) *n_plain = c->E;
    else if (v == BISIP_VARIANT_REDUCED_COMP) {
        if (c->mixed)
            for (unsigned char t : c->tier_of) ++*(t ? n_comp : n_plain);
        else *n_comp = c->E;
    }
    return BISIP_OK;
}

int bisip_logprob_dev(bisip_ctx *c, const double *d_theta, int64_t W, double *d_logp, void *stream)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (W < 0) return fail(BISIP_EINVAL, "W=%lld < 0", (long long)W);
    if (W > 0 && (!d_theta || !d_logp)) return fail(BISIP_EINVAL, "null buffer");
    if (((uintptr_t)d_theta % 8) || ((uintptr_t)d_logp % 8)) return fail(BISIP_EINVAL, "buffers must be 8-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    return dispatch_logprob(c, d_theta, W, d_logp, (hipStream_t)stream);
}

int bisip_forward_dev(bisip_ctx *c, const double *d_theta, int64_t W, double *d_Z, void *stream)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (W < 0) return fail(BISIP_EINVAL, "W=%lld < 0", (long long)W);
    if (W > 0 && (!d_theta || !d_Z)) return fail(BISIP_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(c->device));
    return dispatch_forward(c, d_theta, W, d_Z, (hipStream_t)stream);
}

int bisip_forward_spectra_dev(bisip_ctx *c, int64_t first_spectrum, int64_t n_spectra, const double *d_theta, int64_t W,
                              double *d_Z, void *stream)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (first_spectrum < 0 || n_spectra < 1 || first_spectrum + n_spectra > c->E)
        return fail(BISIP_EINVAL, "spectra [%lld, %lld) not in [0,%d)", (long long)first_spectrum, (long long)(first_spectrum + n_spectra), c->E);
    if (W < 0) return fail(BISIP_EINVAL, "W=%lld < 0", (long long)W);
    if (W > 0 && (!d_theta || !d_Z)) return fail(BISIP_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(c->device));
    return dispatch_forward(c, d_theta, W, d_Z, (hipStream_t)stream, first_spectrum, n_spectra);
}

int bisip_forward_columns_dev(bisip_ctx *c, int64_t first_spectrum, int64_t n_spectra, const double *d_theta, int64_t W,
                              double *d_cols, void *stream)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (W < 0) return fail(BISIP_EINVAL, "W=%lld < 0", (long long)W);
    if (W > 0 && (!d_theta || !d_cols)) return fail(BISIP_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(c->device));
    return dispatch_forward_columns(c, d_theta, W, d_cols, (hipStream_t)stream, first_spectrum, n_spectra);
}

int bisip_forward_spectrum_dev(bisip_ctx *c, int64_t spectrum, const double *d_theta, int64_t W, double *d_Z, void *stream)
{
    return bisip_forward_spectra_dev(c, spectrum, 1, d_theta, W, d_Z, stream);
}

int bisip_loglike_z_dev(bisip_ctx *c, const double *d_Z, int64_t W, double *d_out, void *stream)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (c->E > 1) return fail(BISIP_EUNSUPPORTED, "bisip_loglike_z takes a single-spectrum context");
    if (W < 0) return fail(BISIP_EINVAL, "W=%lld < 0", (long long)W);
    if (W == 0) return BISIP_OK;
    if (!d_Z || !d_out) return fail(BISIP_EINVAL, "null buffer");
    if ((W + 3) / 4 > 0x7fffffffLL) return fail(BISIP_EINVAL, "W=%lld exceeds the launch grid limit", (long long)W);
    HIP_TRY(hipSetDevice(c->device));
    const int rec = (int)(c->cb_stride / c->N);
    hipLaunchKernelGGL(k_loglike_z, dim3((unsigned)((W + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_Z, d_out,
                       (long long)W, c->d_cb, rec, c->N, c->lconst);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

int bisip_loglike_z(bisip_ctx *c, const double *Z, int64_t W, double *out)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (W < 0) return fail(BISIP_EINVAL, "W=%lld < 0", (long long)W);
    if (W == 0) return BISIP_OK;
    if (!Z || !out) return fail(BISIP_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(c->device));
    const size_t zb = (size_t)W * 2 * c->N * sizeof(double), ob = (size_t)W * sizeof(double);
    const size_t zb_al = (zb + 255) & ~(size_t)255;
    int rc = ensure_ws(c, zb_al + ob);
    if (rc != BISIP_OK) return rc;
    double *d_Z = c->d_ws;
    double *d_out = (double *)((char *)c->d_ws + zb_al);
    HIP_TRY(hipMemcpyAsync(d_Z, Z, zb, hipMemcpyHostToDevice, c->stream));
    rc = bisip_loglike_z_dev(c, d_Z, W, d_out, c->stream);
    if (rc != BISIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_out, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BISIP_OK;
}

int bisip_stretch_half_dev(bisip_ctx *c, const bisip_stretch_args *u, void *stream)
{
    int rc = check_stretch(c, u, false);
    if (rc != BISIP_OK || u->n_slots == 0) return rc;
    if (!u->partner || !u->zz || !u->factor || !u->logu) return fail(BISIP_EINVAL, "null RNG stream");
    HIP_TRY(hipSetDevice(c->device));
    const StretchArgs a = to_device_args(u);
    return dispatch_stretch(c, StretchWork{STRETCH_HALF, &a, nullptr}, u->walkers_per_spectrum, (hipStream_t)stream);
}

int bisip_stretch_eval_dev(bisip_ctx *c, const bisip_stretch_args *u, void *stream)
{
    int rc = check_stretch(c, u, true);
    if (rc != BISIP_OK || u->n_slots == 0) return rc;
    if (!u->partner || !u->zz || !u->factor || !u->logu) return fail(BISIP_EINVAL, "null RNG stream");
    if (u->slot_lo < 0 || u->slot_hi > u->n_slots || u->slot_lo > u->slot_hi)
        return fail(BISIP_EINVAL, "bad slot range [%lld,%lld) of %lld", (long long)u->slot_lo,
                    (long long)u->slot_hi, (long long)u->n_slots);
    if (u->slot_lo == u->slot_hi) return BISIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const StretchArgs a = to_device_args(u);
    return dispatch_stretch(c, StretchWork{STRETCH_EVAL, &a, nullptr}, u->walkers_per_spectrum, (hipStream_t)stream);
}

int bisip_stretch_apply_dev(bisip_ctx *c, const bisip_stretch_args *u, void *stream)
{
    int rc = check_stretch(c, u, true);
    if (rc != BISIP_OK || u->n_slots == 0) return rc;
    if (u->world < 1 || u->pad * u->world < u->n_slots)
        return fail(BISIP_EINVAL, "gathered block too small: world=%d pad=%lld n_slots=%lld", u->world,
                    (long long)u->pad, (long long)u->n_slots);
    HIP_TRY(hipSetDevice(c->device));
    return dispatch_apply(c, to_device_args(u), (hipStream_t)stream);
}

// does a chunk of this context's ensemble sample on a packed state (k_stretch_half_packed)?
static bool packed_state(const bisip_ctx *c, int64_t W, int64_t n_steps)
{
    // BISIP_NO_PACKED_STATE (read per call) keeps the plain layout: A/B runs and the test of their equality.
    return c->E == 1 && c->ndim < PACKED_ROW && W / 2 >= 65536 && lanes_per_walker(W / 2) == 1 && n_steps > 0 &&
           std::getenv("BISIP_NO_PACKED_STATE") == nullptr;
}

// what bisip_stretch_run_philox_dev adds to a chunk: the philox contract's parameters instead of its arrays
struct PhiloxStream {
    double a;
    uint64_t seed;
    int64_t step0;
    const int32_t *d_perm;   // (n_steps, 3): A, Ainv, B per iteration
};

static int stretch_run(bisip_ctx *c, const bisip_stretch_args *first, int64_t W, int64_t n_steps, int64_t thin_by,
                       const PhiloxStream *draw, void *stream)
{
    const int64_t nh = (W + 1) / 2;
    bisip_stretch_args u = *first;
    // A single ensemble that fills the chip with one lane per slot samples this chunk on a packed state: one aligned
    // 64-byte row per walker (k_stretch_half_packed), packed here, unpacked after the last half-step.
    const bool packed = packed_state(c, W, n_steps);
    if (draw && !packed) return fail(BISIP_EUNSUPPORTED, "the stream is drawn in place only by the packed-state half-step (bisip_stretch_philox_inline)");
    if (packed) {
        const size_t need = (size_t)W * PACKED_ROW * sizeof(double);
        if (c->packed_bytes < need) {
            if (c->d_packed) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(c->d_packed); c->d_packed = nullptr; c->packed_bytes = 0; }
            hipError_t e = hipMalloc((void **)&c->d_packed, need);
            if (e != hipSuccess) return fail(BISIP_ENOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
            c->packed_bytes = need;
        }
        hipLaunchKernelGGL(k_state_repack<true>, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           first->coords, first->logp, c->d_packed, (long long)W, c->ndim);
        HIP_TRY(hipGetLastError());
    }
    for (int64_t k = 0; k < n_steps; ++k) {
        for (int h = 0; h < 2; ++h) {
            const int64_t off = (k * 2 + h) * nh;
            if (!draw) {
                u.active = first->active + off; u.partner = first->partner + off;
                u.zz = first->zz + off; u.factor = first->factor + off; u.logu = first->logu + off;
            }
            u.n_slots = h ? W / 2 : nh;
            const bool store = ((k + 1) % thin_by) == 0;   // the walkers of BOTH halves of a stored step
            const int64_t srow = k / thin_by;
            u.chain_row = (store && first->chain_row) ? first->chain_row + srow * W * c->ndim : nullptr;
            u.logp_row = (store && first->logp_row) ? first->logp_row + srow * W : nullptr;
            StretchArgs a = to_device_args(&u);
            if (packed) a.packed = c->d_packed;
            if (draw) {
                a.perm = draw->d_perm + 3 * k;
                a.draw_W = W; a.draw_a = draw->a; a.draw_ndim_m1 = (double)(c->ndim - 1);
                a.seed_lo = (unsigned int)(draw->seed & 0xffffffffu); a.seed_hi = (unsigned int)(draw->seed >> 32);
                a.draw_step = (unsigned int)(draw->step0 + k); a.draw_e = (unsigned int)c->spectrum_offset; a.draw_h = h;
            }
            int rc = dispatch_stretch(c, StretchWork{STRETCH_HALF, &a, nullptr}, u.walkers_per_spectrum, (hipStream_t)stream);
            if (rc != BISIP_OK) return rc;
        }
    }
    if (packed) {
        hipLaunchKernelGGL(k_state_repack<false>, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           first->coords, first->logp, c->d_packed, (long long)W, c->ndim);
        HIP_TRY(hipGetLastError());
    }
    return BISIP_OK;
}

int bisip_stretch_run_dev(bisip_ctx *c, const bisip_stretch_args *first, int64_t W, int64_t n_steps,
                          int64_t thin_by, void *stream)
{
    if (!c || !first) return fail(BISIP_EINVAL, "null argument");
    if (thin_by < 1 || n_steps % thin_by) return fail(BISIP_EINVAL, "n_steps=%lld must be a multiple of thin_by=%lld", (long long)n_steps, (long long)thin_by);
    if (W < 2 || n_steps < 0) return fail(BISIP_EINVAL, "bad W=%lld or n_steps=%lld", (long long)W, (long long)n_steps);
    if (!first->coords || !first->logp || !first->active || !first->partner || !first->zz ||
        !first->factor || !first->logu || !first->status)
        return fail(BISIP_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(c->device));
    return stretch_run(c, first, W, n_steps, thin_by, nullptr, stream);
}

int bisip_stretch_philox_inline(const bisip_ctx *c, int64_t W)
{
    return c && W >= 2 && W <= 0x7fffffffLL && packed_state(c, W, 1) ? 1 : 0;
}

int bisip_stretch_run_philox_dev(bisip_ctx *c, const bisip_stretch_args *first, int64_t W, int64_t n_steps,
                                 int64_t thin_by, double a, uint64_t seed, int64_t step0, const int32_t *d_perm,
                                 void *stream)
{
    if (!c || !first || !d_perm) return fail(BISIP_EINVAL, "null argument");
    if (thin_by < 1 || n_steps % thin_by) return fail(BISIP_EINVAL, "n_steps=%lld must be a multiple of thin_by=%lld", (long long)n_steps, (long long)thin_by);
    if (W < 2 || W > 0x7fffffffLL || n_steps < 0 || step0 < 0 || step0 + n_steps > 0xffffffffLL)
        return fail(BISIP_EINVAL, "bad W=%lld or step range [%lld, +%lld)", (long long)W, (long long)step0, (long long)n_steps);
    if (!(a > 0.0)) return fail(BISIP_EINVAL, "stretch scale a=%g must be positive", a);
    if (!first->coords || !first->logp || !first->status) return fail(BISIP_EINVAL, "null buffer");
    if (n_steps == 0) return BISIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    const PhiloxStream draw{a, seed, step0, d_perm};
    return stretch_run(c, first, W, n_steps, thin_by, &draw, stream);
}

// rows lo, lo + stride, ... < hi of a (W, ndim) batch against the yardstick of the spectrum each belongs to
// (reduced_logp_reference_rows), in blocks of up to 64 rows of one spectrum
static double reduced_check_rows(const bisip_ctx *c, const double *theta, int64_t W, const double *logp,
                                 int64_t lo, int64_t hi, int64_t stride)
{
    constexpr int BLOCK = 64;
    const int64_t per = c->E > 1 ? W / c->E : W;
    const int n = c->ndim;
    double rows[BLOCK * MAXD], want[BLOCK], got[BLOCK];
    int held = 0;
    int64_t spectrum = -1;
    double w = 0.0;
    auto flush = [&] {
        if (!held) return;
        reduced_logp_reference_rows(c->reduced[(size_t)spectrum], rows, held, want);
        for (int r = 0; r < held; ++r) {
            const double scale = std::fabs(want[r]) > 1.0 ? std::fabs(want[r]) : 1.0;
            const double rel = std::fabs(got[r] - want[r]) / scale;
            if (!(rel <= w)) w = rel;                          // NaN counts as worst
        }
        held = 0;
    };
    for (int64_t i = lo; i < hi; i += stride) {
        const double *th = theta + i * n;
        if (!in_prior_host(th, c)) continue;                  // the prior decides those rows, exactly
        const int64_t e = per ? i / per : 0;
        if (e != spectrum || held == BLOCK) { flush(); spectrum = e; }
        std::memcpy(rows + held * n, th, sizeof(double) * (size_t)n);
        got[held++] = logp[i];
    }
    flush();
    return w;
}

int bisip_ctx_reduced_check(bisip_ctx *c, const double *theta, int64_t W, const double *logp, double *worst_rel)
{
    if (!c || !theta || !logp || !worst_rel) return fail(BISIP_EINVAL, "null argument");
    if (c->model_id != BISIP_MODEL_POLYDECOMP || c->reduced.empty())
        return fail(BISIP_EUNSUPPORTED, "only PolynomialDecomposition contexts have a QR-reduced form");
    if (W < 0 || (c->E > 1 && W % c->E)) return fail(BISIP_EINVAL, "W=%lld is not a multiple of the %d spectra", (long long)W, c->E);
    std::vector<double> worst((size_t)host_threads() + 1, 0.0);
    return guarded([&] {
        std::atomic<int> slot{0};
        parallel_blocks(W, 2048, [&](int64_t lo, int64_t hi) {
            worst[(size_t)slot.fetch_add(1)] = reduced_check_rows(c, theta, W, logp, lo, hi, 1);
        });
        double all = 0.0;
        for (double w : worst)
            if (!(w <= all)) all = w;
        *worst_rel = all;
        return (int)BISIP_OK;
    });
}

int bisip_ctx_reduced_guard(bisip_ctx *c, int enable, int64_t *n_checks, double *worst_rel, int *escalations)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (enable == 0 || enable == 1) c->guard_on = enable == 1;
    if (n_checks) *n_checks = c->guard_checks;
    if (worst_rel) *worst_rel = c->guard_worst;
    if (escalations) *escalations = c->guard_escalations;
    return BISIP_OK;
}

int bisip_read_tables(const char *const *paths, int64_t n_files, int headers, int64_t n_rows,
                      double *tables, int32_t *status, int threads)
{
    if (!paths || !tables || !status) return fail(BISIP_EINVAL, "null argument");
    if (n_files < 0 || n_rows < 1 || headers < 0)
        return fail(BISIP_EINVAL, "bad n_files=%lld, n_rows=%lld or headers=%d", (long long)n_files, (long long)n_rows, headers);
    return guarded([&] { read_tables(paths, n_files, headers, n_rows, tables, status, threads); return (int)BISIP_OK; });
}

int bisip_ctx_set_spectrum_offset(bisip_ctx *c, int64_t first_spectrum)
{
    if (!c) return fail(BISIP_EINVAL, "null argument");
    if (first_spectrum < 0 || first_spectrum + c->E > 0x7fffffffLL)     // 31 bits of the Philox counter
        return fail(BISIP_EINVAL, "first_spectrum=%lld out of range", (long long)first_spectrum);
    c->spectrum_offset = first_spectrum;
    return BISIP_OK;
}

int bisip_stretch_draw_dev(bisip_ctx *c, int64_t W, double a, uint64_t seed, int64_t step0,
                           int64_t n_steps, const int32_t *d_perm, int32_t *d_active,
                           int32_t *d_partner, double *d_zz, double *d_factor, double *d_logu,
                           void *stream)
{
    if (!c || !d_perm || !d_active || !d_partner || !d_zz || !d_factor || !d_logu)
        return fail(BISIP_EINVAL, "null argument");
    if (W < 2 || W > 0x7fffffffLL || n_steps < 0 || step0 < 0 || step0 + n_steps > 0xffffffffLL)
        return fail(BISIP_EINVAL, "bad W/step range");
    if (n_steps == 0) return BISIP_OK;
    HIP_TRY(hipSetDevice(c->device));
    DrawArgs d;
    d.W = W; d.nh = (W + 1) / 2; d.n_steps = n_steps; d.step0 = step0; d.E = c->E; d.e0 = c->spectrum_offset;
    if (c->E > 1 && (W & 1)) return fail(BISIP_EINVAL, "batch context: walkers per spectrum must be even");
    d.a = a; d.ndim_m1 = (double)(c->ndim - 1);
    d.seed_lo = (unsigned int)(seed & 0xffffffffu); d.seed_hi = (unsigned int)(seed >> 32);
    d.perm = d_perm; d.active = d_active; d.partner = d_partner;
    d.zz = d_zz; d.factor = d_factor; d.logu = d_logu;
    const long long per_half = d.E * d.nh, halves = 2 * n_steps;
    d.flat = per_half < 256 && per_half * halves + 256 <= 0xffffffffLL;
    if (d.flat) {
        hipLaunchKernelGGL(k_stretch_draw, dim3((unsigned)((per_half * halves + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, d);
        HIP_TRY(hipGetLastError());
        return BISIP_OK;
    }
    if ((per_half + 255) / 256 > 0x7fffffffLL) return fail(BISIP_EINVAL, "too many slots per half-step");
    const unsigned gy = (unsigned)(halves < 32768 ? halves : 32768);
    const long long gz = (halves + gy - 1) / gy;
    if (gz > 65535) return fail(BISIP_EINVAL, "n_steps=%lld too large for one draw (chunk the run)", (long long)n_steps);
    hipLaunchKernelGGL(k_stretch_draw, dim3((unsigned)((per_half + 255) / 256), gy, (unsigned)gz), dim3(256), 0,
                       (hipStream_t)stream, d);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

static int moment_splits(int64_t n_samples, int64_t E)
{
    // enough workgroups to fill the chip (256 CUs x 16), at least four samples each
    int64_t s = (4096 + E - 1) / E;
    if (s > n_samples / 4) s = n_samples / 4;
    return (int)(s < 1 ? 1 : s);
}

int64_t bisip_chain_moments_workspace(int64_t n_samples, int64_t n_ensembles, int ndim)
{
    if (n_samples < 1 || n_ensembles < 1 || ndim < 1) return 0;
    return n_ensembles * moment_splits(n_samples, n_ensembles) * (int64_t)ndim;
}

int bisip_chain_moments_dev(const double *d_chain, int64_t n_samples, int64_t sample_stride,
                            int64_t n_ensembles, int64_t walkers_per_ensemble, int ndim,
                            double *d_mean, double *d_std, double *d_work, void *stream)
{
    if (!d_chain || !d_mean || !d_std || !d_work) return fail(BISIP_EINVAL, "null argument");
    if (ndim < 1 || ndim > BISIP_MAX_NDIM) return fail(BISIP_EINVAL, "ndim=%d out of range", ndim);
    if (n_samples < 1 || n_ensembles < 1 || n_ensembles > 0x7fffffffLL || walkers_per_ensemble < 1)
        return fail(BISIP_EINVAL, "bad chain shape");
    if (sample_stride < n_ensembles * walkers_per_ensemble * ndim)
        return fail(BISIP_EINVAL, "sample_stride smaller than one sample");
    MomentArgs a;
    a.chain = d_chain; a.n_samples = n_samples; a.sample_stride = sample_stride;
    a.E = n_ensembles; a.Wp = walkers_per_ensemble; a.ndim = ndim;
    a.splits = moment_splits(n_samples, n_ensembles);
    a.mean = d_mean; a.std = d_std; a.partial = d_work;
    const dim3 grid((unsigned)n_ensembles, (unsigned)a.splits);
    const dim3 fin((unsigned)((n_ensembles * ndim + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_moments_partial<0>, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_moments_finish<0>, fin, dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_moments_partial<1>, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_moments_finish<1>, fin, dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

int bisip_stretch_persistent_dev(bisip_ctx *c, const bisip_persist_args *u, void *stream)
{
    if (!c || !u) return fail(BISIP_EINVAL, "null argument");
    if (!u->coords || !u->logp || !u->status) return fail(BISIP_EINVAL, "null buffer");
    if (!u->active || !u->partner || !u->zz || !u->factor || !u->logu) return fail(BISIP_EINVAL, "null RNG stream");
    const int64_t Wp = u->walkers_per_ensemble;
    if (Wp < 2 || u->n_walkers < Wp || u->n_walkers % Wp) return fail(BISIP_EINVAL, "bad walker counts");
    const int64_t E = u->n_walkers / Wp;
    if (E != c->E) return fail(BISIP_EINVAL, "n_walkers/walkers_per_ensemble=%lld but the context holds %d spectra", (long long)E, c->E);
    if (c->E > 1 && (Wp & 1)) return fail(BISIP_EINVAL, "batch context: walkers per spectrum must be even");
    if (u->thin_by < 1 || u->n_steps < 0 || u->n_steps % u->thin_by) return fail(BISIP_EINVAL, "n_steps must be a multiple of thin_by");
    const bool one_workgroup = !((size_t)Wp * (c->ndim + 1) * sizeof(double) > 65536 || (Wp + 1) / 2 > 512);
    // a single ensemble beyond one workgroup: several workgroups and a barrier of their own per half-step
    // (k_stretch_group), up to 32,768 walkers (256 workgroups at most: launch_group) of at most 7 parameters
    const bool group = !one_workgroup && c->E == 1 && Wp <= 32768 && c->ndim <= 7;
    if (!one_workgroup && !group)
        return fail(BISIP_EUNSUPPORTED, "ensemble of %lld walkers does not fit one workgroup%s", (long long)Wp,
                    c->E == 1 && Wp <= 32768 ? " and its rows do not fit 64 bytes (ndim > 7)" : "");
    if (u->n_steps == 0) return BISIP_OK;
    if (u->n_steps * 2 * 64 > 0x7fffffffLL) return fail(BISIP_EINVAL, "n_steps=%lld: chunk the run", (long long)u->n_steps);
    HIP_TRY(hipSetDevice(c->device));
    PersistArgs p;
    p.gstate = nullptr; p.gsync = nullptr; p.G = 0; p.spin_limit = 0; p.spread = 0;
    if (group) {
        constexpr size_t SYNC = 4096;        // >= GROUP_SYNC_WORDS unsigned, and the rows stay 64-byte aligned
        static_assert(SYNC >= GROUP_SYNC_WORDS * sizeof(unsigned) && SYNC % 64 == 0, "the group's sync words");
        const size_t need = SYNC + (size_t)Wp * 64;
        if (c->group_bytes < need) {
            if (c->d_group) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(c->d_group); c->d_group = nullptr; c->group_bytes = 0; }
            hipError_t e = hipMalloc((void **)&c->d_group, need);
            if (e != hipSuccess) return fail(BISIP_ENOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
            c->group_bytes = need;
        }
        // counters 0, the smallest XCC id seen so far "none" (0xffffffff); ordered before the kernel on its stream
        HIP_TRY(hipMemsetAsync(c->d_group, 0, SYNC, (hipStream_t)stream));
        HIP_TRY(hipMemsetAsync(c->d_group + 32 * sizeof(unsigned), 0xff, sizeof(unsigned), (hipStream_t)stream));
        p.gsync = (unsigned *)c->d_group;
        p.gstate = (double *)(c->d_group + SYNC);
        p.spin_limit = 1u << 22;        // ~0.3 s of polling: only workgroups that never become resident together get there
    }
    p.coords = u->coords; p.logp = u->logp; p.W = Wp; p.E = E; p.n_steps = u->n_steps;
    p.thin_by = u->thin_by;
    p.active = u->active; p.partner = u->partner; p.zz = u->zz; p.factor = u->factor; p.logu = u->logu;
    p.chain = u->chain; p.logp_chain = u->logp_chain; p.naccept = u->naccept; p.status = u->status;
    return dispatch_stretch(c, StretchWork{group ? STRETCH_GROUP : STRETCH_PERSIST, nullptr, &p}, Wp, (hipStream_t)stream);
}

void bisip_philox4x32(const uint32_t *counter, const uint32_t *key, uint32_t *out)
{
    const Philox4 r = philox4x32_10(counter[0], counter[1], counter[2], counter[3], key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}

static int logprob_host_once(bisip_ctx *c, const double *theta, int64_t W, double *logp);

// a measured tier past this (a fifth of the parity tolerance) is closed to BISIP_VARIANT_AUTO
constexpr double GUARD_TOL = 2e-11;

// Close the tier a context on BISIP_VARIANT_AUTO runs (it was just MEASURED past GUARD_TOL) and move to the next
// formulation: a mixed batch first to "every spectrum compensated", else plain -> compensated -> per-frequency.
// The choice holds until the prior box changes (bisip_ctx_set_bounds).
static int guard_escalate(bisip_ctx *c, int v)
{
    if (v == BISIP_VARIANT_REDUCED_COMP && c->mixed) c->mix_off = true;   // first every spectrum compensated
    else c->demoted[v == BISIP_VARIANT_REDUCED ? 0 : 1] = true;
    ++c->guard_escalations;
    const int rc = guarded([&] { return recenter_reduced(c); });    // the next tier may not have been estimated yet
    c->kernel_name = name_for(c);
    return rc;
}

// The QR-reduced kernels were chosen from an error ESTIMATE on probe rows (recenter_reduced); here the
// kernel that just ran is MEASURED on up to 256 rows of the caller's own batch against the reduced form
// in long double -- on the first call of a context and on every 2^n-th after it, so a long emcee run pays
// a few dozen checks of ~50 us.  Past GUARD_TOL a context on BISIP_VARIANT_AUTO closes that tier, moves to the
// next formulation (compensated, then per-frequency), and the batch is evaluated again with it; a
// caller-forced variant is only recorded (bisip_ctx_reduced_guard reports both).
static int guard_after_logprob(bisip_ctx *c, const double *theta, int64_t W, double *logp)
{
    if (!c->guard_on || c->model_id != BISIP_MODEL_POLYDECOMP || c->reduced.empty()) return BISIP_OK;
    if (c->E > 1 && W % c->E) return BISIP_OK;
    const int64_t call = ++c->guard_calls;
    if (call & (call - 1)) return BISIP_OK;                       // 1, 2, 4, 8, ...
    for (int round = 0; round < 3; ++round) {
        const int v = effective_variant(c);
        if (v != BISIP_VARIANT_REDUCED && v != BISIP_VARIANT_REDUCED_COMP) return BISIP_OK;
        const int64_t stride = W > 256 ? W / 256 : 1;
        const double worst = reduced_check_rows(c, theta, W, logp, 0, W, stride);
        ++c->guard_checks;
        if (!(worst <= c->guard_worst)) c->guard_worst = worst;
        if (worst <= GUARD_TOL || c->variant != BISIP_VARIANT_AUTO) return BISIP_OK;
        int rc = guard_escalate(c, v);
        if (rc != BISIP_OK) return rc;
        rc = logprob_host_once(c, theta, W, logp);
        if (rc != BISIP_OK) return rc;
    }
    return BISIP_OK;
}

// The same measurement for callers that hold their rows on the device and bring a few of them to the host: the
// device sampler's own guard (rows of the initial ensemble and, chunk by chunk, the stored samples nearest to
// the shell logp = 0; bisip_chain_shell_rows_dev picks them).  Does not evaluate anything: the caller re-runs.
int bisip_ctx_reduced_guard_rows(bisip_ctx *c, const double *theta, int64_t W, const double *logp, double *worst_rel,
                                 int *escalated)
{
    if (!c || !worst_rel || !escalated || (W > 0 && (!theta || !logp))) return fail(BISIP_EINVAL, "null argument");
    *worst_rel = 0.0;
    *escalated = 0;
    if (c->model_id != BISIP_MODEL_POLYDECOMP || c->reduced.empty())
        return fail(BISIP_EUNSUPPORTED, "only PolynomialDecomposition contexts have a QR-reduced form");
    if (W < 0 || (c->E > 1 && W % c->E)) return fail(BISIP_EINVAL, "W=%lld is not a multiple of the %d spectra", (long long)W, c->E);
    const int v = effective_variant(c);
    if (v != BISIP_VARIANT_REDUCED && v != BISIP_VARIANT_REDUCED_COMP) return BISIP_OK;      // nothing estimated runs
    double worst = 0.0;
    int rc = bisip_ctx_reduced_check(c, theta, W, logp, &worst);
    if (rc != BISIP_OK) return rc;
    ++c->guard_checks;
    if (!(worst <= c->guard_worst)) c->guard_worst = worst;
    *worst_rel = worst;
    if (worst <= GUARD_TOL || c->variant != BISIP_VARIANT_AUTO || !c->guard_on) return BISIP_OK;
    rc = guard_escalate(c, v);
    if (rc != BISIP_OK) return rc;
    *escalated = 1;
    return BISIP_OK;
}

int bisip_logprob(bisip_ctx *c, const double *theta, int64_t W, double *logp)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (W < 0) return fail(BISIP_EINVAL, "W=%lld < 0", (long long)W);
    if (W == 0) return BISIP_OK;
    if (!theta || !logp) return fail(BISIP_EINVAL, "null buffer");
    const int rc = logprob_host_once(c, theta, W, logp);
    return rc != BISIP_OK ? rc : guard_after_logprob(c, theta, W, logp);
}

static int logprob_host_once(bisip_ctx *c, const double *theta, int64_t W, double *logp)
{
    HIP_TRY(hipSetDevice(c->device));
    const size_t tb = (size_t)W * c->ndim * sizeof(double), ob = (size_t)W * sizeof(double);
    const size_t tb_al = (tb + 255) & ~(size_t)255;
    if (c->E == 1 && tb + ob >= PIPE_FROM_BYTES && std::getenv("BISIP_NO_HOST_PIPELINE") == nullptr)
        return guarded([&] {
            return host_pipeline(c, (const char *)theta, (size_t)c->ndim * sizeof(double), (char *)logp, sizeof(double), W,
                                 [&](char *d_in, int64_t n, char *d_res, hipStream_t st) {
                                     return dispatch_logprob(c, (const double *)d_in, n, (double *)d_res, st);
                                 });
        });
    int rc = ensure_ws(c, tb_al + ob);
    if (rc != BISIP_OK) return rc;
    double *d_theta = c->d_ws;
    double *d_out = (double *)((char *)c->d_ws + tb_al);
    if (tb_al + ob <= bisip_ctx::PIN_BYTES) {
        // small batch: stage through pinned memory so both copies are truly asynchronous and
        // the call pays one synchronisation (29 -> ~20 us per call)
        if (!c->h_pin) {
            HIP_TRY(hipHostMalloc((void **)&c->h_pin, bisip_ctx::PIN_BYTES, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void **)&c->d_pin, c->h_pin, 0));
        }
        std::memcpy(c->h_pin, theta, tb);
        if (tb + ob <= bisip_ctx::ZEROCOPY_BYTES) {
            // tiny batch: the kernel reads theta from and writes logp to the mapped host
            // buffer directly over PCIe -- no DMA copies at all
            rc = dispatch_logprob(c, (const double *)c->d_pin, W, (double *)(c->d_pin + tb_al), c->stream);
            if (rc != BISIP_OK) return rc;
            HIP_TRY(hipStreamSynchronize(c->stream));
            std::memcpy(logp, c->h_pin + tb_al, ob);
            return BISIP_OK;
        }
        HIP_TRY(hipMemcpyAsync(d_theta, c->h_pin, tb, hipMemcpyHostToDevice, c->stream));
        rc = dispatch_logprob(c, d_theta, W, d_out, c->stream);
        if (rc != BISIP_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->h_pin + tb, d_out, ob, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::memcpy(logp, c->h_pin + tb, ob);
        return BISIP_OK;
    }
    HIP_TRY(hipMemcpyAsync(d_theta, theta, tb, hipMemcpyHostToDevice, c->stream));
    rc = dispatch_logprob(c, d_theta, W, d_out, c->stream);
    if (rc != BISIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(logp, d_out, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BISIP_OK;
}

int bisip_forward(bisip_ctx *c, const double *theta, int64_t W, double *Z)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (W < 0) return fail(BISIP_EINVAL, "W=%lld < 0", (long long)W);
    if (W == 0) return BISIP_OK;
    if (!theta || !Z) return fail(BISIP_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(c->device));
    const size_t tb = (size_t)W * c->ndim * sizeof(double), zb = (size_t)W * 2 * c->N * sizeof(double);
    const size_t tb_al = (tb + 255) & ~(size_t)255;
    if (c->E == 1 && tb + zb >= PIPE_FROM_BYTES && std::getenv("BISIP_NO_HOST_PIPELINE") == nullptr)
        return guarded([&] {
            return host_pipeline(c, (const char *)theta, (size_t)c->ndim * sizeof(double), (char *)Z, (size_t)2 * c->N * sizeof(double), W,
                                 [&](char *d_in, int64_t n, char *d_res, hipStream_t st) {
                                     return dispatch_forward(c, (const double *)d_in, n, (double *)d_res, st);
                                 });
        });
    int rc = ensure_ws(c, tb_al + zb);
    if (rc != BISIP_OK) return rc;
    double *d_theta = c->d_ws;
    double *d_Z = (double *)((char *)c->d_ws + tb_al);
    HIP_TRY(hipMemcpyAsync(d_theta, theta, tb, hipMemcpyHostToDevice, c->stream));
    rc = dispatch_forward(c, d_theta, W, d_Z, c->stream);
    if (rc != BISIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(Z, d_Z, zb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BISIP_OK;
}

int bisip_forward_percentiles(bisip_ctx *c, const double *theta, int64_t W, const double *percentiles,
                              int n_percentiles, double *out)
{
    if (!c) return fail(BISIP_EINVAL, "null context");
    if (c->E > 1) return fail(BISIP_EUNSUPPORTED, "bisip_forward_percentiles takes a single-spectrum context");
    if (W < 1 || !theta || !percentiles || !out || n_percentiles < 1) return fail(BISIP_EINVAL, "bad argument");
    const int ncols = 2 * c->N;
    HIP_TRY(hipSetDevice(c->device));
    // the responses are written column by column (k_forward_columns) and the order statistics selected from
    // the columns: no transposition, no sort, no 2^31 limit
    const size_t tb = ((size_t)W * c->ndim * sizeof(double) + 255) & ~(size_t)255;
    const size_t zb = ((size_t)W * ncols * sizeof(double) + 255) & ~(size_t)255;
    const size_t ob = ((size_t)n_percentiles * ncols * sizeof(double) + 255) & ~(size_t)255;
    int rc = ensure_ws(c, tb + zb + ob);
    if (rc != BISIP_OK) return rc;
    char *base = (char *)c->d_ws;
    double *d_theta = (double *)base, *d_cols = (double *)(base + tb), *d_out = (double *)(base + tb + zb);
    HIP_TRY(hipMemcpyAsync(d_theta, theta, (size_t)W * c->ndim * sizeof(double), hipMemcpyHostToDevice, c->stream));
    rc = dispatch_forward_columns(c, d_theta, W, d_cols, c->stream, 0, 1);
    if (rc != BISIP_OK) return rc;
    rc = bisip_columns_percentiles_dev(d_cols, ncols, W, percentiles, n_percentiles, d_out, c->stream);
    if (rc != BISIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)n_percentiles * ncols * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BISIP_OK;
}

}  // extern "C"


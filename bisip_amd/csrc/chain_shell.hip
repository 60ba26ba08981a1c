// chain_shell.hip -- the stored samples nearest to the shell logp = 0 (bisip_chain_shell_rows_dev).
//
// Why: the parity tolerance is relative to max(1, |logp|), so a kernel's ABSOLUTE error counts where a
// walker crosses logp = 0 on its way in, not around the mode.  The QR-reduced PolynomialDecomposition kernels
// are chosen from an error ESTIMATE (bisip_ctx_reduced_error); the device sampler therefore measures the tier
// it runs on rows of its own chain -- per ensemble the k stored samples of smallest |logp| -- against the
// host's binary128 / long-double yardstick while the next chunk runs (bisip_ctx_reduced_guard_rows,
// bisip_amd/sampler.py).  This unit finds those rows where the chain lies: a radix selection on the 63-bit
// image of |logp| (two 12-bit digits: exponent and 13 bits of mantissa -- ties beyond that resolution, 1e-4
// relative, are as near to the shell as each other), then a compaction of the rows and their log-probabilities
// into one small block that goes to the host.  Reference counterpart: none (the reference never looks at its
// own rounding; src/bisip/models.py:111-118 is the run this guards).
//
// Mapping.  Many ensembles (a batch of spectra): ONE workgroup per ensemble does the whole selection with its
// histogram in LDS (three sweeps over the ensemble's ns x Wp log-probabilities).  Few ensembles: the sweeps are
// split over enough workgroups to fill the chip, histograms meet in global memory, and the levels are separate
// launches (hist, threshold, hist, threshold, compact).  HBM-bound: 8 B per stored sample and sweep.
#include "host.h"

using namespace bisip;
using namespace bisip::host;

namespace {

constexpr int NB = 4096;        // bins of one 12-bit digit
constexpr int LEVELS = 2;
constexpr int TPB = 256;

struct ShellState {             // per ensemble, in device memory (SPLIT) or LDS (fused)
    unsigned long long prefix;  // the digits fixed so far; ~0: fewer finite samples than k, all are taken
    unsigned shift;             // a sample's key is compared as (bits >> shift)
    unsigned below;             // samples whose key is smaller than `prefix`: all selected
    unsigned need;              // how many of the samples AT `prefix` are selected
    unsigned cnt_lo, cnt_tie;   // compaction cursors
    unsigned pad[2];
};

struct ShellArgs {
    const double *logp;         // (n_samples, E*Wp)
    const double *chain;        // (n_samples, E*Wp, ndim)
    long long n_samples, E, Wp;
    int ndim, k, n_stride;
    int ties;                   // 0: samples AT the threshold key are left out (the selected SET is then reproducible)
    int w_splits, s_splits;     // workgroups per ensemble = w_splits * s_splits
    unsigned *hist;             // (E, NB)      SPLIT only
    ShellState *state;          // (E,)         SPLIT only
    double *out;                // (E, k + n_stride, ndim + 1)
};

__device__ __forceinline__ unsigned long long abs_bits(double x)
{
    return (unsigned long long)__double_as_longlong(x) & 0x7fffffffffffffffull;
}
__device__ __forceinline__ bool finite_bits(unsigned long long u) { return (u >> 52) != 0x7ffull; }

// this workgroup's share of ensemble e: samples [s0, s1) x walkers [w0, w1)
struct Share { long long s0, s1, w0, w1; };
__device__ __forceinline__ Share share_of(const ShellArgs &a, int split)
{
    const int ws = split % a.w_splits, ss = split / a.w_splits;
    Share r;
    r.w0 = a.Wp * ws / a.w_splits; r.w1 = a.Wp * (ws + 1) / a.w_splits;
    r.s0 = a.n_samples * ss / a.s_splits; r.s1 = a.n_samples * (ss + 1) / a.s_splits;
    return r;
}

// One count per candidate lane into an LDS histogram.  A posterior sample's log-probabilities cluster (a whole
// chain within a few units of its maximum): at the first level most lanes of a wave name the SAME bin and 64
// atomics on one LDS word take turns.  One round of wave aggregation on the first candidate's bin takes the
// cluster in a single atomic; the rest add for themselves.
__device__ __forceinline__ void hist_add(unsigned *hist, bool candidate, unsigned bin)
{
    const unsigned long long act = __ballot(candidate);
    if (!act) return;
    const int leader = __ffsll((long long)act) - 1;
    const unsigned b0 = (unsigned)__shfl((int)bin, leader);
    const unsigned long long same = __ballot(candidate && bin == b0);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b0], (unsigned)__popcll(same));
    else if (candidate && bin != b0) atomicAdd(&hist[bin], 1u);
}

// histogram of the next digit over the samples that are still candidates (finite, and at `prefix` so far)
__device__ __forceinline__ void sweep_hist(const ShellArgs &a, long long e, const Share &r, int level,
                                           unsigned long long prefix, unsigned *hist)
{
    const long long EW = a.E * a.Wp;
    const int shift = 63 - 12 * (level + 1);
    for (long long s = r.s0; s < r.s1; ++s) {
        const double *row = a.logp + s * EW + e * a.Wp;
        for (long long w = r.w0 + threadIdx.x; w < r.w1; w += TPB) {
            const unsigned long long u = abs_bits(__builtin_nontemporal_load(row + w));
            const bool candidate = finite_bits(u) && (level == 0 || (u >> (shift + 12)) == prefix);
            hist_add(hist, candidate, (unsigned)((u >> shift) & (NB - 1)));
        }
    }
}

// the digit at which the running count reaches `need`: all TPB lanes call it; hist in LDS or global memory.
// Returns through LDS scratch: digit (NB when the histogram holds fewer than need), count below it, total.
__device__ __forceinline__ void find_digit(const unsigned *hist, unsigned need, unsigned *scratch /* TPB + 4 */,
                                           unsigned &digit, unsigned &below, unsigned &total)
{
    constexpr int PER = NB / TPB;
    unsigned sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) sum += hist[threadIdx.x * PER + i];
    scratch[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0, d = NB, b = 0;
        int p = 0;
        for (; p < TPB; ++p) {
            if (run + scratch[p] >= need) break;
            run += scratch[p];
        }
        if (p < TPB) {
            for (int i = 0; i < PER; ++i) {
                const unsigned h = hist[p * PER + i];
                if (run + h >= need) { d = (unsigned)(p * PER + i); b = run; break; }
                run += h;
            }
        }
        unsigned tot = 0;
        for (int q = 0; q < TPB; ++q) tot += scratch[q];
        scratch[TPB] = d; scratch[TPB + 1] = b; scratch[TPB + 2] = tot;
    }
    __syncthreads();
    digit = scratch[TPB]; below = scratch[TPB + 1]; total = scratch[TPB + 2];
    __syncthreads();
}

// one level's outcome folded into the state (one lane)
__device__ __forceinline__ void advance(ShellState &st, int level, unsigned digit, unsigned below, unsigned total)
{
    if (digit == NB) {                  // fewer candidates than needed: only at level 0 (fewer finite samples than k)
        st.prefix = ~0ull; st.shift = 63; st.below = total; st.need = 0;
        return;
    }
    st.prefix = (st.prefix << 12) | digit;
    st.shift = (unsigned)(63 - 12 * (level + 1));
    st.below += below;
    st.need -= below;
}

__device__ __forceinline__ void copy_row(const ShellArgs &a, long long flat, long long e, unsigned slot)
{
    double *dst = a.out + ((long long)e * (a.k + a.n_stride) + slot) * (a.ndim + 1);
    const double *src = a.chain + flat * a.ndim;
    for (int q = 0; q < a.ndim; ++q) dst[q] = src[q];
    dst[a.ndim] = a.logp[flat];
}

// selected samples -> their slots.  Everything below the threshold key is taken (a set that does not depend on
// scheduling); of the samples AT the threshold key the first `need` to arrive.
__device__ __forceinline__ void sweep_compact(const ShellArgs &a, long long e, const Share &r, const ShellState &st,
                                              unsigned *cnt_lo, unsigned *cnt_tie)
{
    const long long EW = a.E * a.Wp;
    for (long long s = r.s0; s < r.s1; ++s) {
        const long long base = s * EW + e * a.Wp;
        for (long long w = r.w0 + threadIdx.x; w < r.w1; w += TPB) {
            const unsigned long long u = abs_bits(a.logp[base + w]);
            if (!finite_bits(u)) continue;
            const unsigned long long top = u >> st.shift;
            if (top < st.prefix) copy_row(a, base + w, e, atomicAdd(cnt_lo, 1u));
            else if (top == st.prefix && st.need && a.ties) {
                const unsigned j = atomicAdd(cnt_tie, 1u);
                if (j < st.need) copy_row(a, base + w, e, st.below + j);
            }
        }
    }
}

// slots no sample fills (fewer finite samples than k) hold NaN rows: the host's check skips them
__device__ __forceinline__ void fill_rest(const ShellArgs &a, long long e, const ShellState &st)
{
    const unsigned filled = st.below + (a.ties ? st.need : 0u);
    const int width = a.ndim + 1;
    double *dst = a.out + (long long)e * (a.k + a.n_stride) * width;
    for (long long i = (long long)filled * width + threadIdx.x; i < (long long)a.k * width; i += TPB)
        dst[i] = __builtin_nan("");
}

// n_stride evenly spaced walkers of the FIRST sample (the initial ensemble's spread over the prior box)
__device__ __forceinline__ void stride_rows(const ShellArgs &a, long long e)
{
    if ((int)threadIdx.x < a.n_stride) {
        const long long w = a.Wp * threadIdx.x / a.n_stride;
        copy_row(a, e * a.Wp + w, e, (unsigned)(a.k + threadIdx.x));
    }
}

// ---- many ensembles: the whole selection of ensemble blockIdx.x in one workgroup
__global__ __launch_bounds__(TPB) void k_shell_fused(const ShellArgs a)
{
    __shared__ unsigned hist[NB];
    __shared__ unsigned scratch[TPB + 4];
    __shared__ ShellState st;
    __shared__ unsigned cur[2];
    const long long e = blockIdx.x;
    const Share all{0, a.n_samples, 0, a.Wp};
    if (threadIdx.x == 0) { st.prefix = 0; st.shift = 63; st.below = 0; st.need = (unsigned)a.k; cur[0] = 0; cur[1] = 0; }
    for (int level = 0; level < LEVELS; ++level) {
        for (int b = threadIdx.x; b < NB; b += TPB) hist[b] = 0;
        __syncthreads();
        if (st.prefix != ~0ull) sweep_hist(a, e, all, level, st.prefix, hist);
        __syncthreads();
        unsigned digit, below, total;
        const bool live = st.prefix != ~0ull;
        find_digit(hist, st.need, scratch, digit, below, total);
        if (threadIdx.x == 0 && live) advance(st, level, digit, below, total);
        __syncthreads();
    }
    sweep_compact(a, e, all, st, &cur[0], &cur[1]);
    fill_rest(a, e, st);
    stride_rows(a, e);
}

// ---- few ensembles: one launch per step, workgroups (split, ensemble)
__global__ __launch_bounds__(TPB) void k_shell_hist(const ShellArgs a, const int level)
{
    __shared__ unsigned hist[NB];
    const long long e = blockIdx.y;
    const unsigned long long prefix = level ? a.state[e].prefix : 0ull;     // (level 0 starts from nothing: the state is written by its threshold step)
    if (prefix == ~0ull) return;
    for (int b = threadIdx.x; b < NB; b += TPB) hist[b] = 0;
    __syncthreads();
    sweep_hist(a, e, share_of(a, blockIdx.x), level, prefix, hist);
    __syncthreads();
    unsigned *g = a.hist + e * NB;
    for (int b = threadIdx.x; b < NB; b += TPB)
        if (hist[b]) atomicAdd(g + b, hist[b]);
}

__global__ __launch_bounds__(TPB) void k_shell_threshold(const ShellArgs a, const int level)
{
    __shared__ unsigned scratch[TPB + 4];
    const long long e = blockIdx.x;
    unsigned *g = a.hist + e * NB;
    ShellState st;
    if (level == 0) { st.prefix = 0; st.shift = 63; st.below = 0; st.need = (unsigned)a.k; st.cnt_lo = st.cnt_tie = 0; st.pad[0] = st.pad[1] = 0; }
    else st = a.state[e];
    const bool live = st.prefix != ~0ull;
    unsigned digit, below, total;
    find_digit(g, st.need, scratch, digit, below, total);
    if (live) advance(st, level, digit, below, total);
    for (int b = threadIdx.x; b < NB; b += TPB) g[b] = 0;      // the next level's histogram starts empty
    if (threadIdx.x == 0) a.state[e] = st;
    if (level == LEVELS - 1) { fill_rest(a, e, st); stride_rows(a, e); }
}

__global__ __launch_bounds__(TPB) void k_shell_compact(const ShellArgs a)
{
    const long long e = blockIdx.y;
    const ShellState st = a.state[e];
    sweep_compact(a, e, share_of(a, blockIdx.x), st, &a.state[e].cnt_lo, &a.state[e].cnt_tie);
}

// how the work of one ensemble is split (1 x 1: the fused kernel)
void plan(long long n_samples, long long E, long long Wp, int &w_splits, int &s_splits)
{
    w_splits = s_splits = 1;
    if (E >= 512) return;
    long long want = (2048 + E - 1) / E;                                   // workgroups per ensemble to fill the chip
    const long long most = (n_samples * Wp + 4095) / 4096;                 // ... of at least 4096 samples each
    if (want > most) want = most;
    if (want <= 1) return;
    long long ss = want < n_samples ? want : n_samples;                    // whole samples first, then walker ranges
    long long ws = (want + ss - 1) / ss;
    if (ws > (Wp + TPB - 1) / TPB) ws = (Wp + TPB - 1) / TPB;
    if (ws < 1) ws = 1;
    w_splits = (int)ws; s_splits = (int)ss;
}

}  // namespace

extern "C" {

int64_t bisip_chain_shell_rows_workspace(int64_t n_ensembles)
{
    if (n_ensembles < 1) return 0;
    return n_ensembles * (int64_t)(NB * sizeof(unsigned) + sizeof(ShellState));
}

int bisip_chain_shell_rows_dev(const double *d_chain, const double *d_logp, int64_t n_samples, int64_t n_ensembles,
                               int64_t walkers_per_ensemble, int ndim, int k, int n_stride, int ties, double *d_out,
                               void *d_work, void *stream)
{
    if (!d_chain || !d_logp || !d_out || !d_work) return fail(BISIP_EINVAL, "null argument");
    if (n_samples < 1 || n_ensembles < 1 || n_ensembles > 0x7fffffffLL || walkers_per_ensemble < 1)
        return fail(BISIP_EINVAL, "bad chain shape");
    if (ndim < 1 || ndim > BISIP_MAX_NDIM) return fail(BISIP_EINVAL, "ndim=%d out of range", ndim);
    if (k < 1 || k > (1 << 20) || n_stride < 0 || n_stride > TPB || n_stride > walkers_per_ensemble)
        return fail(BISIP_EINVAL, "k=%d / n_stride=%d out of range", k, n_stride);
    if (n_samples * walkers_per_ensemble > 0xffffffffLL) return fail(BISIP_EUNSUPPORTED, "more than 2^32 samples per ensemble");
    ShellArgs a;
    a.logp = d_logp; a.chain = d_chain;
    a.n_samples = n_samples; a.E = n_ensembles; a.Wp = walkers_per_ensemble;
    // one ensemble: its samples are one contiguous run of n_samples * W log-probabilities (and rows)
    if (n_ensembles == 1 && n_stride == 0) { a.n_samples = 1; a.Wp = n_samples * walkers_per_ensemble; }
    a.ndim = ndim; a.k = k; a.n_stride = n_stride; a.ties = ties ? 1 : 0;
    plan(a.n_samples, a.E, a.Wp, a.w_splits, a.s_splits);
    a.hist = (unsigned *)d_work;
    a.state = (ShellState *)((char *)d_work + (size_t)n_ensembles * NB * sizeof(unsigned));
    a.out = d_out;
    hipStream_t st = (hipStream_t)stream;
    const int splits = a.w_splits * a.s_splits;
    if (splits == 1) {
        hipLaunchKernelGGL(k_shell_fused, dim3((unsigned)a.E), dim3(TPB), 0, st, a);
        HIP_TRY(hipGetLastError());
        return BISIP_OK;
    }
    HIP_TRY(hipMemsetAsync(d_work, 0, (size_t)n_ensembles * NB * sizeof(unsigned), st));      // the level-0 histograms
    const dim3 grid((unsigned)splits, (unsigned)a.E);
    for (int level = 0; level < LEVELS; ++level) {
        hipLaunchKernelGGL(k_shell_hist, grid, dim3(TPB), 0, st, a, level);
        hipLaunchKernelGGL(k_shell_threshold, dim3((unsigned)a.E), dim3(TPB), 0, st, a, level);
    }
    hipLaunchKernelGGL(k_shell_compact, grid, dim3(TPB), 0, st, a);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

}  // extern "C"

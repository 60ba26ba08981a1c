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
// launches (hist, hist, compact; every workgroup derives the threshold digits from the finished histograms
// itself).  HBM-bound: 8 B per stored sample and sweep, four loads in flight per lane.
#include "host.h"

using namespace bisip;
using namespace bisip::host;

namespace {

constexpr int NB = 4096;        // bins of one 12-bit digit
constexpr int LEVELS = 2;
constexpr int TPB = 256;
constexpr int UNROLL = 8;       // log-probabilities a lane has in flight

struct ShellState {
    unsigned long long prefix;  // the digits fixed so far; ~0: fewer finite samples than k, all are taken
    unsigned shift;             // a sample's key is compared as (bits >> shift)
    unsigned below;             // samples whose key is smaller than `prefix`: all selected
    unsigned need;              // how many of the samples AT `prefix` are selected
};

// per ensemble in the workspace (few ensembles): a histogram per level and the compaction cursors, zero at the start
struct ShellWork {
    unsigned hist[LEVELS][NB];
    unsigned cnt_lo, cnt_tie, pad[2];
};

struct ShellArgs {
    const double *logp;         // (n_samples, E*Wp)
    const double *chain;        // (n_samples, E*Wp, ndim)
    long long n_samples, E, Wp;
    long long stride_width;     // walkers of one sample (the stride rows are spread over the first sample)
    int ndim, k, n_stride;
    int ties;                   // 0: samples AT the threshold key are left out (the selected SET is then reproducible)
    int w_splits, s_splits;     // workgroups per ensemble = w_splits * s_splits
    ShellWork *work;            // (E,)  few ensembles only
    double *out;                // (E, k + n_stride, ndim + 1)
};

__device__ __forceinline__ unsigned long long abs_bits(double x)
{
    return (unsigned long long)__double_as_longlong(x) & 0x7fffffffffffffffull;
}
__device__ __forceinline__ bool finite_bits(unsigned long long u) { return (u >> 52) != 0x7ffull; }
constexpr unsigned long long NOT_A_SAMPLE = 0x7ff0000000000000ull;    // what a lane past the end of its share holds: the bits of inf

// this workgroup's share of ensemble e: samples [s0, s1) x walkers [w0, w1), swept as ONE index space so that a
// lane has UNROLL independent loads in flight whatever the shape (a survey's ensembles are 256 walkers wide and
// hundreds of samples long: sample by sample every lane would wait for one load at a time)
struct Share {
    long long s0, w0;
    unsigned width, total;      // walkers per sample of the share; samples x walkers
};
__device__ __forceinline__ Share share_of(const ShellArgs &a, int split)
{
    const int ws = split % a.w_splits, ss = split / a.w_splits;
    const long long w0 = a.Wp * ws / a.w_splits, w1 = a.Wp * (ws + 1) / a.w_splits;
    const long long s0 = a.n_samples * ss / a.s_splits, s1 = a.n_samples * (ss + 1) / a.s_splits;
    return Share{s0, w0, (unsigned)(w1 - w0), (unsigned)((s1 - s0) * (w1 - w0))};
}
// flat position (sample-major, as the chain lies) of element i of a share of ensemble e
__device__ __forceinline__ long long flat_of(const ShellArgs &a, long long e, const Share &r, unsigned i)
{
    if (r.width == r.total) return r.s0 * (a.E * a.Wp) + e * a.Wp + r.w0 + i;     // one sample's run: no division
    const unsigned s = i / r.width, w = i - s * r.width;
    return (r.s0 + s) * (a.E * a.Wp) + e * a.Wp + r.w0 + w;
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
    const int shift = 63 - 12 * (level + 1);
    for (unsigned i0 = threadIdx.x; i0 < r.total; i0 += UNROLL * TPB) {     // (wave-uniform trip count but for the last round)
        unsigned long long u[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const unsigned i = i0 + j * TPB;
            u[j] = i < r.total ? abs_bits(__builtin_nontemporal_load(a.logp + flat_of(a, e, r, i))) : NOT_A_SAMPLE;
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const bool candidate = finite_bits(u[j]) && (level == 0 || (u[j] >> (shift + 12)) == prefix);
            hist_add(hist, candidate, (unsigned)((u[j] >> shift) & (NB - 1)));
        }
    }
}

// the digit at which the running count reaches `need`: all TPB lanes call it; hist in LDS or global memory.
// digit = NB when the histogram holds fewer than need (or need is 0); below = the count of the bins under it.
// A scan over the lanes' 16-bin sums (within each wave by shuffles, across the four waves through LDS), then the
// one lane whose bins hold the crossing looks through them: one lane walking 256 sums and 16 bins, every read
// waiting for the one before, took 10-25 us -- three quarters of a sweep that is otherwise bound by the HBM.
__device__ __forceinline__ void find_digit(const unsigned *hist, unsigned need, unsigned *scratch /* TPB + 4 */,
                                           unsigned &digit, unsigned &below, unsigned &total)
{
    constexpr int PER = NB / TPB;
    unsigned h[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { h[i] = hist[threadIdx.x * PER + i]; sum += h[i]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)inc, off);
        if (lane >= off) inc += v;
    }
    if (lane == 63) scratch[wave] = inc;
    if (threadIdx.x == 0) { scratch[TPB] = NB; scratch[TPB + 1] = 0; }
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) { const unsigned v = scratch[w]; tot += v; if (w < wave) base += v; }
    inc += base;
    const unsigned ex = inc - sum;
    __syncthreads();                    // (every lane has read the defaults' neighbours before the owner writes)
    if (need && ex < need && need <= inc) {
        unsigned run = ex;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (run < need && run + h[i] >= need) { scratch[TPB] = (unsigned)(threadIdx.x * PER + i); scratch[TPB + 1] = run; }
            run += h[i];
        }
    }
    __syncthreads();
    digit = scratch[TPB]; below = scratch[TPB + 1]; total = tot;
    __syncthreads();
}

// one level's outcome folded into the state
__device__ __forceinline__ void advance(ShellState &st, int level, unsigned digit, unsigned below, unsigned total)
{
    if (st.prefix == ~0ull) return;
    if (digit == NB) {                  // fewer candidates than needed: only at level 0 (fewer finite samples than k)
        st.prefix = ~0ull; st.shift = 63; st.below = total; st.need = 0;
        return;
    }
    st.prefix = (st.prefix << 12) | digit;
    st.shift = (unsigned)(63 - 12 * (level + 1));
    st.below += below;
    st.need -= below;
}

__device__ __forceinline__ ShellState fresh_state(int k)
{
    ShellState st;
    st.prefix = 0; st.shift = 63; st.below = 0; st.need = (unsigned)k;
    return st;
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
    for (unsigned i0 = threadIdx.x; i0 < r.total; i0 += UNROLL * TPB) {
        unsigned long long u[UNROLL];
        long long f[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const unsigned i = i0 + j * TPB;
            f[j] = i < r.total ? flat_of(a, e, r, i) : 0;
            u[j] = i < r.total ? abs_bits(__builtin_nontemporal_load(a.logp + f[j])) : NOT_A_SAMPLE;
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            if (!finite_bits(u[j])) continue;
            const unsigned long long top = u[j] >> st.shift;
            if (top < st.prefix) copy_row(a, f[j], e, atomicAdd(cnt_lo, 1u));
            else if (top == st.prefix && st.need && a.ties) {
                const unsigned slot = atomicAdd(cnt_tie, 1u);
                if (slot < st.need) copy_row(a, f[j], e, st.below + slot);
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
        const long long w = a.stride_width * threadIdx.x / a.n_stride;
        copy_row(a, e * a.Wp + w, e, (unsigned)(a.k + threadIdx.x));
    }
}

// ---- many ensembles: the whole selection of ensemble blockIdx.x in one workgroup, histogram in LDS
__global__ __launch_bounds__(TPB) void k_shell_fused(const ShellArgs a)
{
    __shared__ unsigned hist[NB];
    __shared__ unsigned scratch[TPB + 4];
    __shared__ unsigned cur[2];
    const long long e = blockIdx.x;
    const Share all{0, 0, (unsigned)a.Wp, (unsigned)(a.n_samples * a.Wp)};
    if (threadIdx.x == 0) { cur[0] = 0; cur[1] = 0; }
    ShellState st = fresh_state(a.k);                              // the same in every lane
    for (int level = 0; level < LEVELS; ++level) {
        for (int b = threadIdx.x; b < NB; b += TPB) hist[b] = 0;
        __syncthreads();
        if (st.prefix != ~0ull) sweep_hist(a, e, all, level, st.prefix, hist);
        __syncthreads();
        unsigned digit, below, total;
        find_digit(hist, st.need, scratch, digit, below, total);
        advance(st, level, digit, below, total);
    }
    sweep_compact(a, e, all, st, &cur[0], &cur[1]);
    fill_rest(a, e, st);
    stride_rows(a, e);
}

// ---- few ensembles: workgroups (split, ensemble); the histograms of all splits meet in global memory, and every
// workgroup of the NEXT launch finds the threshold digit for itself (4096 words from the L2: cheaper than a launch
// of one workgroup in between, and a launch boundary is the only grid-wide barrier there is)
__device__ __forceinline__ ShellState state_after(const ShellArgs &a, long long e, int levels, unsigned *scratch)
{
    ShellState st = fresh_state(a.k);
    for (int level = 0; level < levels; ++level) {
        unsigned digit, below, total;
        find_digit(a.work[e].hist[level], st.need, scratch, digit, below, total);
        advance(st, level, digit, below, total);
    }
    return st;
}

__global__ __launch_bounds__(TPB) void k_shell_hist(const ShellArgs a, const int level)
{
    __shared__ unsigned hist[NB];
    __shared__ unsigned scratch[TPB + 4];
    const long long e = blockIdx.y;
    const ShellState st = state_after(a, e, level, scratch);
    if (st.prefix == ~0ull) return;
    for (int b = threadIdx.x; b < NB; b += TPB) hist[b] = 0;
    __syncthreads();
    sweep_hist(a, e, share_of(a, blockIdx.x), level, st.prefix, hist);
    __syncthreads();
    // Only the bins up to THIS share's own crossing digit go to global memory: the ensemble's crossing digit cannot
    // lie above it (the counts of all shares up to it already reach `need`), and what lies above the crossing is
    // never looked at.  A share of a posterior sample fills all 4096 bins of the second digit; up to its crossing
    // it holds fewer than k samples -- 4M global atomics for a 6.5M-sample chunk become a few hundred per workgroup.
    unsigned mine, below, total;
    find_digit(hist, st.need, scratch, mine, below, total);
    unsigned *g = a.work[e].hist[level];
    for (unsigned b = threadIdx.x; b < NB && b <= mine; b += TPB)
        if (hist[b]) atomicAdd(g + b, hist[b]);
}

__global__ __launch_bounds__(TPB) void k_shell_compact(const ShellArgs a)
{
    __shared__ unsigned scratch[TPB + 4];
    const long long e = blockIdx.y;
    const ShellState st = state_after(a, e, LEVELS, scratch);
    sweep_compact(a, e, share_of(a, blockIdx.x), st, &a.work[e].cnt_lo, &a.work[e].cnt_tie);
    if (blockIdx.x == 0) { fill_rest(a, e, st); stride_rows(a, e); }
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
    return n_ensembles < 512 ? n_ensembles * (int64_t)sizeof(ShellWork) : 16;       // (many ensembles: histograms in LDS)
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
    if (n_samples * walkers_per_ensemble > 0x7fffffffLL) return fail(BISIP_EUNSUPPORTED, "more than 2^31 samples per ensemble");
    ShellArgs a;
    a.logp = d_logp; a.chain = d_chain;
    a.n_samples = n_samples; a.E = n_ensembles; a.Wp = walkers_per_ensemble;
    // one ensemble: its samples are ONE contiguous run of n_samples * W log-probabilities (and rows), split into
    // contiguous ranges (the stride rows name walkers of the first sample: the same rows in either view)
    a.stride_width = walkers_per_ensemble;
    if (n_ensembles == 1) { a.n_samples = 1; a.Wp = n_samples * walkers_per_ensemble; }
    a.ndim = ndim; a.k = k; a.n_stride = n_stride; a.ties = ties ? 1 : 0;
    plan(a.n_samples, a.E, a.Wp, a.w_splits, a.s_splits);
    a.work = (ShellWork *)d_work;
    a.out = d_out;
    hipStream_t st = (hipStream_t)stream;
    const int splits = a.w_splits * a.s_splits;
    if (splits == 1) {
        hipLaunchKernelGGL(k_shell_fused, dim3((unsigned)a.E), dim3(TPB), 0, st, a);
        HIP_TRY(hipGetLastError());
        return BISIP_OK;
    }
    HIP_TRY(hipMemsetAsync(d_work, 0, (size_t)n_ensembles * sizeof(ShellWork), st));
    const dim3 grid((unsigned)splits, (unsigned)a.E);
    for (int level = 0; level < LEVELS; ++level) hipLaunchKernelGGL(k_shell_hist, grid, dim3(TPB), 0, st, a, level);
    hipLaunchKernelGGL(k_shell_compact, grid, dim3(TPB), 0, st, a);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

}  // extern "C"

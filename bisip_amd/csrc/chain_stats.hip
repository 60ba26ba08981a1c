// chain_stats.hip -- percentiles of a device-resident chain (bisip_chain_percentiles_dev).
//
// The reference summarises a fit with np.percentile(chain, p, axis=0) over the flattened
// chain (src/bisip/utils.py:37-53, get_param_percentile; default p = [2.5, 50, 97.5]).  For
// a batch of spectra the chain lives in HBM, so the percentiles are taken there:
//   1. k_gather_columns: (sample, walker, parameter) -> one contiguous column per
//      (ensemble, parameter), coalesced on both sides;
//   2. the two order statistics each percentile needs: radix selection, one workgroup per column
//      (k_segmented_select) -- or, for more than 8 percentiles of many columns, a rocPRIM segmented
//      radix sort of the E*ndim columns (library sort: hipCUB header) followed by
//   3. k_percentile_lerp: NumPy's 'linear' rule between the two neighbouring order statistics
//      (indices and weights are computed on the host exactly as numpy does).
#include "host.h"

#include <hipcub/hipcub.hpp>

#include <cstdlib>

using namespace bisip;
using namespace bisip::host;

namespace {

struct GatherArgs {
    const double *chain;
    long long n_samples, sample_stride, E, Wp;
    int ndim;
    double *cols;   // (E*ndim, n_samples*Wp)
};

// one thread per (sample, ensemble, walker): reads its ndim-double row, writes ndim columns
__global__ __launch_bounds__(256) void k_gather_columns(const GatherArgs a)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per_sample = a.E * a.Wp;
    if (idx >= a.n_samples * per_sample) return;
    const long long s = idx / per_sample, r = idx - s * per_sample;   // r = e*Wp + w
    const long long e = r / a.Wp, w = r - e * a.Wp;
    const double *row = a.chain + s * a.sample_stride + r * a.ndim;
    const long long n = a.n_samples * a.Wp;
    double *dst = a.cols + (e * a.ndim) * n + s * a.Wp + w;
    for (int q = 0; q < a.ndim; ++q) dst[(long long)q * n] = row[q];
}

// The same through an LDS tile of 64 rows x 64 parameters, for rows of many doubles (a model response has 2N):
// read row-major, written column-major, both coalesced -- one thread per row reads 8 bytes out of every
// 8*ndim and moves eight times the data it needs.
__global__ __launch_bounds__(256) void k_gather_columns_tiled(const GatherArgs a)
{
    __shared__ double tile[64][65];
    const long long tiles_w = (a.Wp + 63) / 64;
    const int tiles_q = (a.ndim + 63) / 64;
    long long b = blockIdx.x;
    const int tq = (int)(b % tiles_q); b /= tiles_q;
    const long long tw = b % tiles_w; b /= tiles_w;
    const long long e = b % a.E, s = b / a.E;
    const long long w0 = tw * 64;
    const int q0 = tq * 64;
    const int nr = (int)(a.Wp - w0 < 64 ? a.Wp - w0 : 64), nc = a.ndim - q0 < 64 ? a.ndim - q0 : 64;
    const double *__restrict__ src = a.chain + s * a.sample_stride + (e * a.Wp + w0) * a.ndim + q0;
    for (int idx = threadIdx.x; idx < nr * nc; idx += 256) {
        const int r = idx / nc, q = idx - r * nc;
        tile[r][q] = src[(long long)r * a.ndim + q];
    }
    __syncthreads();
    const long long n = a.n_samples * a.Wp;
    double *__restrict__ dst = a.cols + (e * a.ndim + q0) * n + s * a.Wp + w0;
    for (int idx = threadIdx.x; idx < nr * nc; idx += 256) {
        const int q = idx / nr, r = idx - q * nr;
        dst[(long long)q * n + r] = tile[r][q];
    }
}

struct LerpArgs {
    const double *sorted;   // (E*ndim, n)
    long long n, columns;
    int n_p;
    const long long *lo;    // (n_p,) lower order statistic
    const double *t;        // (n_p,) weight of the upper one
    double *out;            // (n_p, columns)
};

__global__ __launch_bounds__(256) void k_percentile_lerp(const LerpArgs a)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.columns * a.n_p) return;
    const long long col = idx % a.columns;
    const int k = (int)(idx / a.columns);
    const double *c = a.sorted + col * a.n;
    const long long lo = a.lo[k], hi = lo + 1 < a.n ? lo + 1 : a.n - 1;
    const double x = c[lo], y = c[hi], t = a.t[k];
    const double d = y - x;
    // a NaN anywhere in the column makes every percentile of it NaN (np.percentile); the radix sort
    // puts NaNs at the ends (sign bit set: first, clear: last)
    const bool has_nan = c[0] != c[0] || c[a.n - 1] != c[a.n - 1];
    // numpy.lib._function_base_impl._lerp
    a.out[idx] = has_nan ? __builtin_nan("") : (t >= 0.5 ? y - d * (1.0 - t) : x + d * t);
}

// ---------------------------------------------------------------------------------
// Selection instead of a sort.  A percentile needs two order statistics of its column, not the
// column in order: one workgroup per column finds them on the order-preserving 64-bit image of the
// doubles (k_segmented_select below).  A 4096-spectrum survey's model-space bands (262,144 columns
// of 12,800 values) took 229 ms with the segmented sort.  Same order statistics, same interpolation
// arithmetic (numpy's _lerp) => the same doubles as the sort path.
// ---------------------------------------------------------------------------------
constexpr int SEL_MAX_P = 8;            // percentiles per call on this path
constexpr int SEL_R = 2 * SEL_MAX_P;    // order statistics

struct SelectArgs {
    const double *cols;     // (columns, n), each column contiguous
    long long n, columns;
    int n_p;
    long long lo[SEL_MAX_P];   // lower order statistic of each percentile (in the kernarg segment: no upload, no wait)
    double t[SEL_MAX_P];       // weight of the upper one
    double *out;               // (n_p, columns)
};

__device__ __forceinline__ unsigned long long select_key(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);      // ascending keys <=> ascending doubles
}

__device__ __forceinline__ double select_value(unsigned long long k)
{
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// One 1024-lane workgroup per column, narrowing a RANGE of keys and finishing by counting.  VPT > 0: the
// column is held in REGISTERS (VPT keys per lane, read from memory once); VPT = 0: longer columns are re-read
// from memory in each of the three or four sweeps (min/max, one or two histograms, the survivors).
//   * Every rank keeps [base, base + 2^s): the keys that can still be it.  It starts as [min, max] of the
//     column; a pass histograms (key - base) >> (s - 8) -- 256 equal slices of the range -- finds the slice
//     that holds the rank and makes it the new range.  Radix digits of the keys themselves would put a column
//     that straddles a power of two (0.97 ... 1.03) into two bins of the first useful byte, and a thousand
//     lanes adding to two LDS words take turns; slices of the occupied range spread any sample over the bins.
//   * Histogram passes run only while more than SEL_CAP keys remain in the ranks' ranges -- one or two for a
//     posterior sample -- then the survivors go to LDS and each rank is found by counting the smaller ones.
//   * Ranks with the same range share a histogram and a survivor list (the two neighbours of a percentile
//     usually do until the end).
// Register form: columns of up to 40,960 values -- the model-space band of a spectrum (samples x walkers), most
// parameter columns.
constexpr int SEL_CAP = 192;            // survivors that are finished by counting (all ranks together)

// (two workgroups per CU when the keys leave room: the phases of one hide behind the other's)
template <int VPT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(VPT <= 16 ? 8 : 4, VPT <= 16 ? 8 : 4)))
void k_segmented_select(const SelectArgs a)
{
    __shared__ unsigned hist[SEL_R][256];
    __shared__ unsigned long long base[SEL_R];       // low end of each rank's range of keys
    __shared__ long long rem[SEL_R];                 // its rank among the keys in the range
    __shared__ unsigned surv[SEL_R];                 // how many keys the range holds
    __shared__ int group[SEL_R];                     // ranks with equal ranges share a histogram / a survivor list
    __shared__ unsigned long long gbase[SEL_R];
    __shared__ unsigned gsize[SEL_R], goffset[SEL_R], gcount[SEL_R];
    __shared__ unsigned long long cand[SEL_CAP];
    __shared__ unsigned long long wmin[16], wmax[16];
    __shared__ int n_groups, finish;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long col = blockIdx.x;
    const double *__restrict__ c = a.cols + col * a.n;
    const int R = 2 * a.n_p;
    unsigned long long key[VPT > 0 ? VPT : 1];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const long long i = (long long)j * 1024 + tid;
        key[j] = i < a.n ? select_key(__builtin_nontemporal_load(c + i)) : 0ull;
    }
    const int mine = (int)((a.n - tid + 1023) / 1024);      // how many of them are real (<= VPT)
    // f(key) for every key of the column this lane is responsible for
    auto for_each_key = [&](auto &&f) {
        if constexpr (VPT > 0) {
#pragma unroll
            for (int j = 0; j < VPT; ++j)
                if (j < mine) f(key[j]);
        } else {
            long long i = tid;
            for (; i + 3 * 1024 < a.n; i += 4 * 1024) {       // four loads in flight per lane
                const double v0 = c[i], v1 = c[i + 1024], v2 = c[i + 2048], v3 = c[i + 3072];
                f(select_key(v0)); f(select_key(v1)); f(select_key(v2)); f(select_key(v3));
            }
            for (; i < a.n; i += 1024) f(select_key(c[i]));
        }
    };
    unsigned long long kmin = ~0ull, kmax = 0ull;
    for_each_key([&](unsigned long long k) { kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; });
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long lo_ = __shfl_xor(kmin, d, 64), hi_ = __shfl_xor(kmax, d, 64);
        kmin = lo_ < kmin ? lo_ : kmin; kmax = hi_ > kmax ? hi_ : kmax;
    }
    if (lane == 0) { wmin[wave] = kmin; wmax[wave] = kmax; }
    __syncthreads();
    kmin = wmin[0]; kmax = wmax[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) { kmin = wmin[w] < kmin ? wmin[w] : kmin; kmax = wmax[w] > kmax ? wmax[w] : kmax; }
    // A NaN in the column: every percentile of it is NaN, as np.percentile (and so the reference's
    // get_model_percentile) returns.  Keys of NaNs lie beyond those of the infinities at either end.
    // kmin / kmax are the same in every lane: the whole workgroup leaves here, before any later barrier.
    if (kmax > 0xfff0000000000000ull || kmin < 0x000fffffffffffffull) {
        if (tid < a.n_p) a.out[(long long)tid * a.columns + col] = __builtin_nan("");
        return;
    }
    int s = kmax == kmin ? 0 : 64 - __clzll((long long)(kmax - kmin));     // bits of (key - base) still open; the same for every rank
    if (tid < R) {
        const long long lo = a.lo[tid >> 1];
        rem[tid] = (tid & 1) ? (lo + 1 < a.n ? lo + 1 : a.n - 1) : lo;
        base[tid] = kmin;
        surv[tid] = (unsigned)a.n;
    }
    if (tid == 0) finish = a.n <= SEL_CAP;
    __syncthreads();

    auto regroup = [&]() {                // wave 0; at most 16 ranks
        if (wave == 0) {
            int leader = lane;
            if (lane < R)
                for (int q = lane - 1; q >= 0; --q)
                    if (base[q] == base[lane]) leader = q;
            const bool is_leader = lane < R && leader == lane;
            const unsigned long long leaders = __ballot(is_leader);
            if (lane < R) {
                const int g = __popcll(leaders & ((1ull << leader) - 1ull));
                group[lane] = g;
                if (is_leader) { gbase[g] = base[lane]; gsize[g] = surv[lane]; }
            }
            if (lane == 0) n_groups = __popcll(leaders);
        }
    };
    auto in_range = [&](unsigned long long d) { return s >= 64 || (d >> s) == 0ull; };

    while (!finish && s > 0) {
        const int shift = s > 8 ? s - 8 : 0;
        regroup();
        __syncthreads();
        const int G = n_groups;
        for (int i = tid; i < G * 256; i += 1024) (&hist[0][0])[i] = 0;
        __syncthreads();
        for_each_key([&](unsigned long long k) {
            for (int g = 0; g < G; ++g) {
                const unsigned long long d = k - gbase[g];
                if (in_range(d)) atomicAdd(&hist[g][(unsigned)(d >> shift)], 1u);
            }
        });
        __syncthreads();
        if (wave < R) {                   // wave w finds the slice of rank w: scan of the 256 counts, four per lane
            const unsigned *h = hist[group[wave]];
            const unsigned c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
            const long long mine4 = (long long)c0 + c1 + c2 + c3;
            long long incl = mine4;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const long long up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
            const long long want = rem[wave], excl = incl - mine4;
            if (excl <= want && want < incl) {       // exactly one lane
                long long before = excl;
                int b = 4 * lane;
                unsigned cnt = c0;
                if (want >= before + c0) { before += c0; ++b; cnt = c1;
                    if (want >= before + c1) { before += c1; ++b; cnt = c2;
                        if (want >= before + c2) { before += c2; ++b; cnt = c3; } } }
                rem[wave] = want - before;
                base[wave] += (unsigned long long)b << shift;
                surv[wave] = cnt;
            }
        }
        __syncthreads();
        s = shift;
        if (tid == 0) {
            unsigned long long total = 0;
            for (int r = 0; r < R; ++r) total += surv[r];      // ranks that share a range counted twice: an upper bound
            finish = total <= SEL_CAP;
        }
        __syncthreads();
    }

    if (s > 0) {
        // the keys still inside a rank's range, group by group, into LDS; then count
        regroup();
        __syncthreads();
        const int G = n_groups;
        if (tid == 0) {
            unsigned off = 0;
            for (int g = 0; g < G; ++g) { goffset[g] = off; off += gsize[g]; gcount[g] = 0; }
        }
        __syncthreads();
        for_each_key([&](unsigned long long k) {
            for (int g = 0; g < G; ++g)
                if (in_range(k - gbase[g])) cand[goffset[g] + atomicAdd(&gcount[g], 1u)] = k;
        });
        __syncthreads();
        if (wave < R) {                   // wave w: the survivor of its group that has exactly rem[w] smaller ones
            const int g = group[wave];
            const unsigned m = gsize[g];
            const unsigned long long *cg = cand + goffset[g];
            const long long want = rem[wave];
            for (unsigned i = lane; i < m; i += 64) {
                const unsigned long long ki = cg[i];
                long long less = 0;
                for (unsigned q = 0; q < m; ++q) {
                    const unsigned long long kq = cg[q];
                    less += (kq < ki) || (kq == ki && q < i);
                }
                if (less == want) base[wave] = ki;
            }
        }
        __syncthreads();
    }
    if (tid < a.n_p) {
        const double x = select_value(base[2 * tid]), y = select_value(base[2 * tid + 1]), t = a.t[tid];
        const double d = y - x;
        // numpy.lib._function_base_impl._lerp, as k_percentile_lerp
        a.out[(long long)tid * a.columns + col] = t >= 0.5 ? y - d * (1.0 - t) : x + d * t;
    }
}

struct SegmentOffset {
    long long n;
    __host__ __device__ int operator()(int i) const { return (int)(i * n); }
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int sort_temp_bytes(long long items, long long columns, long long n, size_t *bytes)
{
    using Counting = hipcub::CountingInputIterator<int>;
    using Offsets = hipcub::TransformInputIterator<int, SegmentOffset, Counting>;
    Offsets begin(Counting(0), SegmentOffset{n}), end(Counting(1), SegmentOffset{n});
    size_t temp = 0;
    hipError_t e = hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, temp, (const double *)nullptr, (double *)nullptr,
                                                              (int)items, (int)columns, begin, end);
    if (e != hipSuccess) return fail(BISIP_EHIP, "segmented sort sizing failed: %s", hipGetErrorString(e));
    *bytes = temp;
    return BISIP_OK;
}

}  // namespace

namespace {

int64_t percentiles_workspace(int64_t n_samples, int64_t n_ensembles, int64_t walkers_per_ensemble, int ndim, int n_percentiles)
{
    if (n_samples < 1 || n_ensembles < 1 || walkers_per_ensemble < 1 || ndim < 1 || n_percentiles < 1) return 0;
    const long long n = n_samples * walkers_per_ensemble, columns = n_ensembles * ndim, items = n * columns;
    if (items > 0x7fffffffLL) return 0;
    size_t temp = 0;
    if (sort_temp_bytes(items, columns, n, &temp) != BISIP_OK) return 0;
    return (int64_t)(2 * align256((size_t)items * 8) + align256(temp) + align256((size_t)n_percentiles * 16));
}

// numpy's virtual index for method='linear', evaluated as numpy does: its table of methods gives
// 'linear' the closed form (n - 1) * q, not the general n*q + (alpha + q*(1 - alpha - beta)) - 1
// (numpy/lib/_function_base_impl.py:_QuantileMethods) -- the two differ in the last bits of the
// weight, which shows as soon as neighbouring order statistics are far apart (integer data)
int percentile_ranks(long long n, const double *percentiles, int n_percentiles, std::vector<long long> &lo, std::vector<double> &t)
{
    for (int k = 0; k < n_percentiles; ++k) {
        if (!(percentiles[k] >= 0.0 && percentiles[k] <= 100.0)) return fail(BISIP_EINVAL, "percentiles must be in [0, 100]");
        const double q = percentiles[k] / 100.0;
        double v = (double)(n - 1) * q;
        if (v < 0) v = 0;
        if (v > (double)(n - 1)) v = (double)(n - 1);
        const double f = std::floor(v);
        lo[k] = (long long)f;
        t[k] = v - f;
    }
    return BISIP_OK;
}

// the order statistics of `columns` contiguous columns of n values, SEL_MAX_P percentiles per launch
int select_columns(const double *cols, long long n, long long columns, int n_percentiles, const std::vector<long long> &lo,
                   const std::vector<double> &t, double *d_out, hipStream_t st)
{
    for (int k0 = 0; k0 < n_percentiles; k0 += SEL_MAX_P) {
        SelectArgs sa{};
        sa.cols = cols; sa.n = n; sa.columns = columns; sa.out = d_out + (long long)k0 * columns;
        sa.n_p = n_percentiles - k0 < SEL_MAX_P ? n_percentiles - k0 : SEL_MAX_P;
        for (int k = 0; k < sa.n_p; ++k) { sa.lo[k] = lo[k0 + k]; sa.t[k] = t[k0 + k]; }
        if (n <= 1024 * 8) hipLaunchKernelGGL(k_segmented_select<8>, dim3((unsigned)columns), dim3(1024), 0, st, sa);
        else if (n <= 1024 * 16) hipLaunchKernelGGL(k_segmented_select<16>, dim3((unsigned)columns), dim3(1024), 0, st, sa);
        else if (n <= 1024 * 40) hipLaunchKernelGGL(k_segmented_select<40>, dim3((unsigned)columns), dim3(1024), 0, st, sa);
        else hipLaunchKernelGGL(k_segmented_select<0>, dim3((unsigned)columns), dim3(1024), 0, st, sa);
        HIP_TRY(hipGetLastError());
    }
    return BISIP_OK;
}

int percentiles_impl(const double *d_chain, int64_t n_samples, int64_t sample_stride, int64_t n_ensembles,
                     int64_t walkers_per_ensemble, int ndim, const double *percentiles, int n_percentiles,
                     double *d_out, void *d_work, int64_t work_bytes, void *stream);

}  // namespace

extern "C" {

int64_t bisip_chain_percentiles_workspace(int64_t n_samples, int64_t n_ensembles,
                                          int64_t walkers_per_ensemble, int ndim, int n_percentiles)
{
    return percentiles_workspace(n_samples, n_ensembles, walkers_per_ensemble, ndim, n_percentiles);
}

int64_t bisip_column_percentiles_workspace(int64_t n_rows, int n_cols, int n_percentiles)
{
    return percentiles_workspace(1, 1, n_rows, n_cols, n_percentiles);
}

int bisip_column_percentiles_dev(const double *d_rows, int64_t n_rows, int n_cols, const double *percentiles,
                                 int n_percentiles, double *d_out, void *d_work, int64_t work_bytes, void *stream)
{
    if (n_cols < 1 || n_cols > 65536) return fail(BISIP_EINVAL, "n_cols=%d out of range", n_cols);
    // rows (n_rows, n_cols) = one sample of one ensemble of n_rows walkers with n_cols parameters
    return percentiles_impl(d_rows, 1, n_rows * (int64_t)n_cols, 1, n_rows, n_cols, percentiles, n_percentiles, d_out,
                            d_work, work_bytes, stream);
}

int64_t bisip_grouped_percentiles_workspace(int64_t n_groups, int64_t n_rows, int n_cols, int n_percentiles)
{
    return percentiles_workspace(1, n_groups, n_rows, n_cols, n_percentiles);
}

int bisip_grouped_percentiles_dev(const double *d_rows, int64_t n_groups, int64_t n_rows, int n_cols,
                                  const double *percentiles, int n_percentiles, double *d_out, void *d_work,
                                  int64_t work_bytes, void *stream)
{
    if (n_cols < 1 || n_cols > 65536) return fail(BISIP_EINVAL, "n_cols=%d out of range", n_cols);
    if (n_groups < 1) return fail(BISIP_EINVAL, "n_groups=%lld", (long long)n_groups);
    // (n_groups, n_rows, n_cols) = one sample of n_groups ensembles of n_rows walkers with n_cols parameters
    return percentiles_impl(d_rows, 1, n_groups * n_rows * (int64_t)n_cols, n_groups, n_rows, n_cols, percentiles,
                            n_percentiles, d_out, d_work, work_bytes, stream);
}

int bisip_columns_percentiles_dev(const double *d_cols, int64_t n_columns, int64_t n, const double *percentiles,
                                  int n_percentiles, double *d_out, void *stream)
{
    if (!d_cols || !percentiles || !d_out) return fail(BISIP_EINVAL, "null argument");
    if (n_columns < 1 || n_columns > 0x7fffffffLL || n < 1 || n_percentiles < 1 || n_percentiles > 1024)
        return fail(BISIP_EINVAL, "bad shape");
    std::vector<long long> lo(n_percentiles);
    std::vector<double> t(n_percentiles);
    int rc = percentile_ranks(n, percentiles, n_percentiles, lo, t);
    if (rc != BISIP_OK) return rc;
    return select_columns(d_cols, n, n_columns, n_percentiles, lo, t, d_out, (hipStream_t)stream);
}

int bisip_chain_percentiles_dev(const double *d_chain, int64_t n_samples, int64_t sample_stride,
                                int64_t n_ensembles, int64_t walkers_per_ensemble, int ndim,
                                const double *percentiles, int n_percentiles, double *d_out,
                                void *d_work, int64_t work_bytes, void *stream)
{
    if (ndim < 1 || ndim > BISIP_MAX_NDIM) return fail(BISIP_EINVAL, "ndim=%d out of range", ndim);
    return percentiles_impl(d_chain, n_samples, sample_stride, n_ensembles, walkers_per_ensemble, ndim, percentiles,
                            n_percentiles, d_out, d_work, work_bytes, stream);
}

}  // extern "C"

namespace {

int percentiles_impl(const double *d_chain, int64_t n_samples, int64_t sample_stride, int64_t n_ensembles,
                     int64_t walkers_per_ensemble, int ndim, const double *percentiles, int n_percentiles,
                     double *d_out, void *d_work, int64_t work_bytes, void *stream)
{
    if (!d_chain || !percentiles || !d_out || !d_work) return fail(BISIP_EINVAL, "null argument");
    if (n_samples < 1 || n_ensembles < 1 || walkers_per_ensemble < 1 || n_percentiles < 1 || n_percentiles > 1024)
        return fail(BISIP_EINVAL, "bad shape");
    if (sample_stride < n_ensembles * walkers_per_ensemble * ndim)
        return fail(BISIP_EINVAL, "sample_stride smaller than one sample");
    const long long n = n_samples * walkers_per_ensemble, columns = n_ensembles * ndim, items = n * columns;
    if (items > 0x7fffffffLL) return fail(BISIP_EUNSUPPORTED, "chain of %lld values exceeds the 2^31 items of one sort", items);
    for (int k = 0; k < n_percentiles; ++k)
        if (!(percentiles[k] >= 0.0 && percentiles[k] <= 100.0)) return fail(BISIP_EINVAL, "percentiles must be in [0, 100]");
    size_t temp = 0;
    int rc = sort_temp_bytes(items, columns, n, &temp);
    if (rc != BISIP_OK) return rc;
    const size_t col_bytes = align256((size_t)items * 8);
    const size_t need = 2 * col_bytes + align256(temp) + align256((size_t)n_percentiles * 16);
    if (work_bytes < (int64_t)need) return fail(BISIP_EINVAL, "workspace of %lld bytes, need %zu", (long long)work_bytes, need);
    char *base = (char *)d_work;
    double *cols = (double *)base, *sorted = (double *)(base + col_bytes);
    void *d_temp = base + 2 * col_bytes;
    long long *d_lo = (long long *)(base + 2 * col_bytes + align256(temp));
    double *d_t = (double *)(d_lo + n_percentiles);
    hipStream_t st = (hipStream_t)stream;

    std::vector<long long> lo(n_percentiles);
    std::vector<double> t(n_percentiles);
    rc = percentile_ranks(n, percentiles, n_percentiles, lo, t);
    if (rc != BISIP_OK) return rc;

    GatherArgs g{d_chain, n_samples, sample_stride, n_ensembles, walkers_per_ensemble, ndim, cols};
    const long long rows = n_samples * n_ensembles * walkers_per_ensemble;
    const long long tiles = n_samples * n_ensembles * ((walkers_per_ensemble + 63) / 64) * ((ndim + 63) / 64);
    if (ndim >= 16 && tiles <= 0x7fffffffLL)
        hipLaunchKernelGGL(k_gather_columns_tiled, dim3((unsigned)tiles), dim3(256), 0, st, g);
    else
        hipLaunchKernelGGL(k_gather_columns, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, g);
    HIP_TRY(hipGetLastError());
    // Select, do not sort: up to SEL_MAX_P percentiles per launch, one workgroup per column.  More percentiles
    // than that go through in groups when the columns are few (a handful of very long columns is where the
    // segmented sort is at its worst: benchmarks/micro/select_long_columns.py); with many columns AND many
    // percentiles one sort serves them all.
    const char *force_sort = std::getenv("BISIP_PERCENTILE_SORT");
    if ((n_percentiles <= SEL_MAX_P || columns < 64) && columns <= 0x7fffffffLL && !(force_sort && force_sort[0] == '1')) {
        return select_columns(cols, n, columns, n_percentiles, lo, t, d_out, st);
    }
    HIP_TRY(hipMemcpyAsync(d_lo, lo.data(), sizeof(long long) * n_percentiles, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_t, t.data(), sizeof(double) * n_percentiles, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // lo / t are stack-lifetime host buffers

    using Counting = hipcub::CountingInputIterator<int>;
    using Offsets = hipcub::TransformInputIterator<int, SegmentOffset, Counting>;
    Offsets begin(Counting(0), SegmentOffset{n}), end(Counting(1), SegmentOffset{n});
    HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(d_temp, temp, (const double *)cols, sorted, (int)items,
                                                       (int)columns, begin, end, 0, 64, st));
    LerpArgs l{sorted, n, columns, n_percentiles, d_lo, d_t, d_out};
    hipLaunchKernelGGL(k_percentile_lerp, dim3((unsigned)((columns * n_percentiles + 255) / 256)), dim3(256), 0, st, l);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

}  // namespace

// chain_stats.hip -- percentiles of a device-resident chain (bisip_chain_percentiles_dev).
//
// The reference summarises a fit with np.percentile(chain, p, axis=0) over the flattened
// chain (src/bisip/utils.py:37-53, get_param_percentile; default p = [2.5, 50, 97.5]).  For
// a batch of spectra the chain lives in HBM, so the percentiles are taken there:
//   1. k_gather_columns: (sample, walker, parameter) -> one contiguous column per
//      (ensemble, parameter), coalesced on both sides;
//   2. rocPRIM segmented radix sort of the E*ndim columns (library sort: hipCUB header);
//   3. k_percentile_lerp: NumPy's 'linear' rule between the two neighbouring order statistics
//      (indices and weights are computed on the host exactly as numpy does).
#include "host.h"

#include <hipcub/hipcub.hpp>

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
    // numpy.lib._function_base_impl._lerp
    a.out[idx] = t >= 0.5 ? y - d * (1.0 - t) : x + d * t;
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

    // numpy's virtual index for method='linear' (alpha = beta = 1), evaluated as numpy does
    std::vector<long long> lo(n_percentiles);
    std::vector<double> t(n_percentiles);
    for (int k = 0; k < n_percentiles; ++k) {
        const double q = percentiles[k] / 100.0;
        double v = ((double)n * q + (1.0 + q * (1.0 - 1.0 - 1.0))) - 1.0;
        if (v < 0) v = 0;
        if (v > (double)(n - 1)) v = (double)(n - 1);
        const double f = std::floor(v);
        lo[k] = (long long)f;
        t[k] = v - f;
    }
    HIP_TRY(hipMemcpyAsync(d_lo, lo.data(), sizeof(long long) * n_percentiles, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_t, t.data(), sizeof(double) * n_percentiles, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // lo / t are stack-lifetime host buffers

    GatherArgs g{d_chain, n_samples, sample_stride, n_ensembles, walkers_per_ensemble, ndim, cols};
    const long long rows = n_samples * n_ensembles * walkers_per_ensemble;
    hipLaunchKernelGGL(k_gather_columns, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, g);
    HIP_TRY(hipGetLastError());
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

// ensemble_gram.hip -- sums and second moments of an ensemble's positions on the device
// (bisip_ensemble_gram_dev): what emcee's initial-state test needs of a (W, ndim) ensemble.
//
// emcee refuses an initial ensemble whose walkers do not span the parameter space: the condition number of the
// centred, column-scaled positions must stay below 1e8 (the reference's fit() runs into this test through
// emcee.EnsembleSampler.run_mcmc, src/bisip/models.py:111-118; bisip_amd/sampler.py:walkers_independent restates
// it).  On the host that test reads the whole ensemble -- 26 ms at a million walkers, more than the 200
// iterations that follow -- while the ensemble is on its way to the device anyway.  Here the device, which has
// the rows, forms  S_j = sum_i (x_ij - x_0j)  and  P_jk = sum_i (x_ij - x_0j)(x_ik - x_0k)  (shifted by walker 0:
// the shift lies inside the cloud, so nothing cancels when the host forms the centred Gram matrix
// P - S S^T / W from them), ndim (ndim + 3) / 2 numbers come back and the 7 x 7 decision stays on the host.  A NaN
// or an inf anywhere makes the sums non-finite: the host then takes its slow, exact route (and refuses).
//
// Mapping: workgroups of 256 lanes stride over tiles of 256 rows; a tile is staged through LDS with coalesced
// 16-byte loads (stage_theta) and read back one row per lane; a lane keeps its ndim + ndim (ndim + 1) / 2 running
// sums in registers (ndim <= 8); lanes, then workgroups, are combined in a fixed order (the result does not
// depend on scheduling).  HBM-bound: 8 ndim bytes per walker, read once.
#include "host.h"

using namespace bisip;
using namespace bisip::host;

namespace {

constexpr int GRAM_BLK = 256;
constexpr int GRAM_MAX_GROUPS = 1024;

template <int NDIM>
__global__ __launch_bounds__(GRAM_BLK) void k_gram_partial(const double *__restrict__ x, long long W, double *__restrict__ partial)
{
    constexpr int NV = NDIM + NDIM * (NDIM + 1) / 2;
    __shared__ __attribute__((aligned(16))) double lds[GRAM_BLK * NDIM];
    __shared__ double red[GRAM_BLK / 64][NV];
    double shift[NDIM], acc[NV];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) shift[q] = x[q];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    const long long tiles = (W + GRAM_BLK - 1) / GRAM_BLK;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long row0 = tile * GRAM_BLK;
        __syncthreads();                                  // the previous tile has been read
        stage_theta<NDIM, GRAM_BLK, true>(x, W, row0, lds);
        __syncthreads();
        if (row0 + threadIdx.x < W) {
            double d[NDIM];
#pragma unroll
            for (int q = 0; q < NDIM; ++q) { d[q] = lds[threadIdx.x * NDIM + q] - shift[q]; acc[q] += d[q]; }
            int v = NDIM;
#pragma unroll
            for (int j = 0; j < NDIM; ++j)
#pragma unroll
                for (int k = j; k < NDIM; ++k, ++v) acc[v] = fma(d[j], d[k], acc[v]);
        }
    }
    // lanes of a wave (butterfly: every lane ends with the same sum), then the four waves in order
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double s = acc[v];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        acc[v] = s;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) red[wave][v] = acc[v];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
        for (int w = 0; w < GRAM_BLK / 64; ++w) s += red[w][threadIdx.x];
        partial[(long long)blockIdx.x * NV + threadIdx.x] = s;
    }
}

// out[v] = sum over the groups' partials, in group order
__global__ __launch_bounds__(64) void k_gram_finish(const double *__restrict__ partial, int groups, int nv, double *__restrict__ out)
{
    const int v = threadIdx.x;
    if (v >= nv) return;
    double s = 0.0;
    for (int g = 0; g < groups; ++g) s += partial[(long long)g * nv + v];
    out[v] = s;
}

int groups_for(long long W)
{
    const long long tiles = (W + GRAM_BLK - 1) / GRAM_BLK;
    return (int)(tiles < GRAM_MAX_GROUPS ? tiles : GRAM_MAX_GROUPS);
}

}  // namespace

extern "C" {

int64_t bisip_ensemble_gram_workspace(int64_t W, int ndim)
{
    if (W < 1 || ndim < 1 || ndim > 8) return 0;
    return (int64_t)groups_for(W) * (ndim + ndim * (ndim + 1) / 2);
}

int bisip_ensemble_gram_dev(const double *d_coords, int64_t W, int ndim, double *d_out, double *d_work, void *stream)
{
    if (!d_coords || !d_out || !d_work) return fail(BISIP_EINVAL, "null argument");
    if (W < 1) return fail(BISIP_EINVAL, "W=%lld < 1", (long long)W);
    if (ndim < 1 || ndim > 8) return fail(BISIP_EUNSUPPORTED, "bisip_ensemble_gram_dev holds its sums in registers: ndim=%d not in [1,8]", ndim);
    if ((uintptr_t)d_coords % 8) return fail(BISIP_EINVAL, "coords must be 8-byte aligned");
    const int groups = groups_for(W), nv = ndim + ndim * (ndim + 1) / 2;
    hipStream_t st = (hipStream_t)stream;
    switch (ndim) {
#define X(n) case n: hipLaunchKernelGGL((k_gram_partial<n>), dim3((unsigned)groups), dim3(GRAM_BLK), 0, st, d_coords, (long long)W, d_work); break;
        X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
    }
    hipLaunchKernelGGL(k_gram_finish, dim3(1), dim3(64), 0, st, (const double *)d_work, groups, nv, d_out);
    HIP_TRY(hipGetLastError());
    return BISIP_OK;
}

}  // extern "C"

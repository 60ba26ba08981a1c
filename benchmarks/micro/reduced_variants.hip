// Variants of the headline kernel's memory path (theta (W,7) f64 in, logp (W,) out) to find
// the achievable ceiling for this 7:1 read:write stream.  Timing only (outputs are sums).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../../bisip_amd/csrc/kernels.h"
using namespace bisip;

// ceiling probe: pure stream, 16-B loads, each thread reduces 2 doubles, 8 B stores every 7 loads
__global__ __launch_bounds__(256) void k_stream_sum(const dbl2 *__restrict__ in, double *__restrict__ out, long long n2)
{
    // out[i] for i < n2/ (7/2)...: each block handles 256*7/2=896 dbl2 -> 256 outputs
    __shared__ double lds[1792];
    const long long b = blockIdx.x;
    const dbl2 *src = in + b * 896;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int i = r * 256 + threadIdx.x;
        if (i < 896) { dbl2 v = __builtin_nontemporal_load(src + i); lds[2 * i] = v.x; lds[2 * i + 1] = v.y; }
    }
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) s += lds[threadIdx.x * 7 + q];
    out[b * 256 + threadIdx.x] = s;
}

// direct strided loads, no LDS
template <int P>
__global__ __launch_bounds__(256) void k_direct(const LaunchArgs a, const ReducedArgs<P> r)
{
    constexpr int NDIM = P + 2;
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.W) return;
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = __builtin_nontemporal_load(a.theta + row * NDIM + q);
    a.out[row] = logprob_row_reduced<P>(th, r, a.lconst, a.b);
}

// persistent: grid = G blocks, grid-stride over 256-row tiles, next tile's loads issued
// before computing the current one (register double buffer)
template <int P>
__global__ __launch_bounds__(256) void k_persist(const LaunchArgs a, const ReducedArgs<P> r)
{
    constexpr int NDIM = P + 2;
    constexpr int N2 = 256 * NDIM / 2;  // 896
    __shared__ __attribute__((aligned(16))) double lds[2][256 * NDIM];
    const long long ntiles = a.W / 256;
    const dbl2 *base = reinterpret_cast<const dbl2 *>(a.theta);
    long long tile = blockIdx.x;
    dbl2 buf[4];
    auto issue = [&](long long t) {
        const dbl2 *src = base + t * N2;
#pragma unroll
        for (int q = 0; q < 4; ++q) { int i = q * 256 + threadIdx.x; if (i < N2) buf[q] = __builtin_nontemporal_load(src + i); }
    };
    if (tile < ntiles) issue(tile);
    int ph = 0;
    for (; tile < ntiles; tile += gridDim.x, ph ^= 1) {
        dbl2 *dst = reinterpret_cast<dbl2 *>(lds[ph]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { int i = q * 256 + threadIdx.x; if (i < N2) dst[i] = buf[q]; }
        const long long nxt = tile + gridDim.x;
        if (nxt < ntiles) issue(nxt);
        __syncthreads();
        double th[NDIM];
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[q] = lds[ph][threadIdx.x * NDIM + q];
        a.out[tile * 256 + threadIdx.x] = logprob_row_reduced<P>(th, r, a.lconst, a.b);
    }
}

template <int BLK>
__global__ __launch_bounds__(BLK) void k_lds_blk(const LaunchArgs a, const ReducedArgs<5> r)
{
    constexpr int NDIM = 7;
    __shared__ __attribute__((aligned(16))) double lds[BLK * NDIM];
    const long long row0 = (long long)blockIdx.x * BLK;
    stage_theta<NDIM, BLK, true>(a.theta, a.W, row0, lds);
    __syncthreads();
    const long long row = row0 + threadIdx.x;
    if (row >= a.W) return;
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = lds[threadIdx.x * NDIM + q];
    a.out[row] = logprob_row_reduced<5>(th, r, a.lconst, a.b);
}

// 128 threads, 256 rows per block: two rows per lane
__global__ __launch_bounds__(128) void k_two_rows(const LaunchArgs a, const ReducedArgs<5> r)
{
    constexpr int NDIM = 7;
    __shared__ __attribute__((aligned(16))) double lds[256 * NDIM];
    const long long row0 = (long long)blockIdx.x * 256;
    const dbl2 *src = reinterpret_cast<const dbl2 *>(a.theta + row0 * NDIM);
    dbl2 *dst = reinterpret_cast<dbl2 *>(lds);
#pragma unroll
    for (int q = 0; q < 7; ++q) dst[q * 128 + threadIdx.x] = __builtin_nontemporal_load(src + q * 128 + threadIdx.x);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double th[NDIM];
        const int lr = h * 128 + threadIdx.x;
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[q] = lds[lr * NDIM + q];
        a.out[row0 + lr] = logprob_row_reduced<5>(th, r, a.lconst, a.b);
    }
}

// 128 threads, 256 rows per block, lane l owns rows 2l and 2l+1 -> one 16-byte store per lane
__global__ __launch_bounds__(128) void k_pair_rows(const LaunchArgs a, const ReducedArgs<5> r)
{
    constexpr int NDIM = 7;
    __shared__ __attribute__((aligned(16))) double lds[256 * NDIM];
    const long long row0 = (long long)blockIdx.x * 256;
    const dbl2 *src = reinterpret_cast<const dbl2 *>(a.theta + row0 * NDIM);
    dbl2 *dst = reinterpret_cast<dbl2 *>(lds);
#pragma unroll
    for (int q = 0; q < 7; ++q) dst[q * 128 + threadIdx.x] = __builtin_nontemporal_load(src + q * 128 + threadIdx.x);
    __syncthreads();
    dbl2 o;
    {
        double th[NDIM];
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[q] = lds[(2 * threadIdx.x) * NDIM + q];
        o.x = logprob_row_reduced<5>(th, r, a.lconst, a.b);
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[q] = lds[(2 * threadIdx.x + 1) * NDIM + q];
        o.y = logprob_row_reduced<5>(th, r, a.lconst, a.b);
    }
    __builtin_nontemporal_store(o, reinterpret_cast<dbl2 *>(a.out + row0) + threadIdx.x);
}

// product structure, BLK=128, non-temporal output store
__global__ __launch_bounds__(128) void k_nt_store(const LaunchArgs a, const ReducedArgs<5> r)
{
    constexpr int NDIM = 7;
    __shared__ __attribute__((aligned(16))) double lds[128 * NDIM];
    const long long row0 = (long long)blockIdx.x * 128;
    stage_theta<NDIM, 128, true>(a.theta, a.W, row0, lds);
    __syncthreads();
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = lds[threadIdx.x * NDIM + q];
    __builtin_nontemporal_store(logprob_row_reduced<5>(th, r, a.lconst, a.b), a.out + row0 + threadIdx.x);
}

// product structure, BLK=128, plain (temporal) loads
__global__ __launch_bounds__(128) void k_plain_loads(const LaunchArgs a, const ReducedArgs<5> r)
{
    constexpr int NDIM = 7;
    __shared__ __attribute__((aligned(16))) double lds[128 * NDIM];
    const long long row0 = (long long)blockIdx.x * 128;
    const dbl2 *src = reinterpret_cast<const dbl2 *>(a.theta + row0 * NDIM);
    dbl2 *dst = reinterpret_cast<dbl2 *>(lds);
#pragma unroll
    for (int q = 0; q < 4; ++q) { int i = q * 128 + threadIdx.x; if (i < 448) dst[i] = src[i]; }
    __syncthreads();
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = lds[threadIdx.x * NDIM + q];
    a.out[row0 + threadIdx.x] = logprob_row_reduced<5>(th, r, a.lconst, a.b);
}

template <class F>
float timeit(F f, int reps = 20)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    const long long W = 1LL << 24;
    double *theta, *out;
    hipMalloc(&theta, W * 7 * 8); hipMalloc(&out, W * 8);
    std::vector<double> h(W * 7);
    unsigned long long s = 88172645463325252ULL;
    for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
    for (long long i = 0; i < W; ++i) h[i * 7] = 0.9 + 0.1 * (h[i * 7] + 1.0);
    hipMemcpy(theta, h.data(), W * 7 * 8, hipMemcpyHostToDevice);
    LaunchArgs a; a.theta = theta; a.out = out; a.W = W; a.cb = nullptr; a.N = 32; a.lconst = 1.0;
    for (int q = 0; q < MAXD; ++q) { a.b.lo[q] = -1; a.b.hi[q] = 1; } a.b.lo[0] = 0.9; a.b.hi[0] = 1.1;
    ReducedArgs<5> r;
    for (auto &x : r.R) x = 0.5; for (auto &x : r.bhat) x = 1.0; for (auto &x : r.e) x = 0.1; r.rest = 1.0;
    const double bytes = (double)W * 64;
    auto rep = [&](const char *name, float ms) { printf("%-34s %8.1f us  %7.1f GB/s  frac %.3f\n", name, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 8000); };
    // interleaved A/B rounds (same process), median of 7
    {
        const char *names[8] = {"stream", "blk64", "blk128", "product(blk128)", "blk128x2rows", "pair_rows_16B_store", "blk128_nt_store", "blk128_plain_loads"};
        std::vector<float> t[8];
        for (int round = 0; round < 7; ++round) {
            t[0].push_back(timeit([&] { hipLaunchKernelGGL(k_stream_sum, dim3(W / 256), dim3(256), 0, 0, (const dbl2 *)theta, out, W * 7 / 2); }, 10));
            t[1].push_back(timeit([&] { hipLaunchKernelGGL((k_lds_blk<64>), dim3(W / 64), dim3(64), 0, 0, a, r); }, 10));
            t[2].push_back(timeit([&] { hipLaunchKernelGGL((k_lds_blk<128>), dim3(W / 128), dim3(128), 0, 0, a, r); }, 10));
            t[3].push_back(timeit([&] { hipLaunchKernelGGL((k_logprob_pd_reduced<5, 128, true>), dim3(W / 128), dim3(128), 0, 0, a, r); }, 10));
            t[4].push_back(timeit([&] { hipLaunchKernelGGL((k_two_rows), dim3(W / 256), dim3(128), 0, 0, a, r); }, 10));
            t[5].push_back(timeit([&] { hipLaunchKernelGGL((k_pair_rows), dim3(W / 256), dim3(128), 0, 0, a, r); }, 10));
            t[6].push_back(timeit([&] { hipLaunchKernelGGL((k_nt_store), dim3(W / 128), dim3(128), 0, 0, a, r); }, 10));
            t[7].push_back(timeit([&] { hipLaunchKernelGGL((k_plain_loads), dim3(W / 128), dim3(128), 0, 0, a, r); }, 10));
        }
        for (int v = 0; v < 8; ++v) { std::sort(t[v].begin(), t[v].end()); printf("AB %-18s median %7.1f us  min %7.1f  frac(median) %.3f\n", names[v], t[v][3] * 1e3, t[v][0] * 1e3, bytes / t[v][3] / 1e6 / 8000); }
    }
    rep("stream_sum ceiling probe", timeit([&] { hipLaunchKernelGGL(k_stream_sum, dim3(W / 256), dim3(256), 0, 0, (const dbl2 *)theta, out, W * 7 / 2); }));
    rep("product k_logprob_pd_reduced", timeit([&] { hipLaunchKernelGGL((k_logprob_pd_reduced<5, 128, true>), dim3(W / 128), dim3(128), 0, 0, a, r); }));
    rep("lds BLK=128", timeit([&] { hipLaunchKernelGGL((k_lds_blk<128>), dim3(W / 128), dim3(128), 0, 0, a, r); }));
    rep("lds BLK=512", timeit([&] { hipLaunchKernelGGL((k_lds_blk<512>), dim3(W / 512), dim3(512), 0, 0, a, r); }));
    rep("lds BLK=1024", timeit([&] { hipLaunchKernelGGL((k_lds_blk<1024>), dim3(W / 1024), dim3(1024), 0, 0, a, r); }));
    rep("direct strided loads (no LDS)", timeit([&] { hipLaunchKernelGGL((k_direct<5>), dim3(W / 256), dim3(256), 0, 0, a, r); }));
    for (int g : {256, 512, 1024, 2048, 4096})
    {
        char nm[64]; snprintf(nm, 64, "persistent prefetch grid=%d", g);
        rep(nm, timeit([&] { hipLaunchKernelGGL((k_persist<5>), dim3(g), dim3(256), 0, 0, a, r); }));
    }
    return 0;
}

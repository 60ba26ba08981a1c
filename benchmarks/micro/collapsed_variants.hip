// Rows per lane for the PDCollapsed log-probability kernel (scalar-cache pressure vs registers).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o collapsed_variants collapsed_variants.hip
// k_logprob_xr<R>: R walkers per lane evaluated in lockstep, each record loaded into SGPRs once
// per R rows.  Prints time per launch and evals/s for R = 1..4 and two workgroup sizes, and
// checks every variant against R = 1 bit for bit.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../bisip_amd/csrc/kernels.h"

using namespace bisip;

template <class M, int BLK, int R, bool VEC>
__global__ __launch_bounds__(BLK) void k_logprob_xr(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    __shared__ __attribute__((aligned(16))) double lds[R * BLK * NDIM];
    const long long row0 = (long long)blockIdx.x * (R * BLK);
    stage_theta<NDIM, R * BLK, VEC, BLK>(a.theta, a.W, row0, lds);
    __syncthreads();
    double th[R][NDIM];
    bool ok[R];
    typename M::Setup s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[r][q] = lds[(threadIdx.x + r * BLK) * NDIM + q];
        ok[r] = in_prior<NDIM>(th[r], a.b) && (row0 + threadIdx.x + r * BLK < a.W);
        s[r] = M::setup(th[r]);
    }
    double acc0[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }
    const double *__restrict__ rec = a.cb;
#pragma unroll 2
    for (int j = 0; j < a.N; ++j, rec += M::REC) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double rr, ri;
            M::residual(s[r], rec, rr, ri);
            acc0[r] = fma(rr, rr, acc0[r]);
            acc1[r] = fma(ri, ri, acc1[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long row = row0 + threadIdx.x + r * BLK;
        if (row < a.W) a.out[row] = ok[r] ? fma(-0.5, acc0[r] + acc1[r], a.lconst) : -__builtin_inf();
    }
}

// x2 with the records software-pipelined: the scalar loads of frequency j+1 are issued
// before the FMAs of frequency j (the shipped loop waits for its own loads at the top of
// every iteration and relies on other waves to cover the scalar-cache latency).
template <class M, int BLK, int R, bool VEC>
__global__ __launch_bounds__(BLK) void k_logprob_xr_pf(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    constexpr int REC = M::REC;
    __shared__ __attribute__((aligned(16))) double lds[R * BLK * NDIM];
    const long long row0 = (long long)blockIdx.x * (R * BLK);
    stage_theta<NDIM, R * BLK, VEC, BLK>(a.theta, a.W, row0, lds);
    __syncthreads();
    double th[R][NDIM];
    bool ok[R];
    typename M::Setup s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[r][q] = lds[(threadIdx.x + r * BLK) * NDIM + q];
        ok[r] = in_prior<NDIM>(th[r], a.b) && (row0 + threadIdx.x + r * BLK < a.W);
        s[r] = M::setup(th[r]);
    }
    double acc0[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }
    const double *__restrict__ rec = a.cb;
    const double *last = a.cb + (long long)(a.N - 1) * REC;
    double A[REC], B[REC];
#pragma unroll
    for (int q = 0; q < REC; ++q) A[q] = rec[q];
    auto use = [&](const double (&cur)[REC]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double rr, ri;
            M::residual(s[r], cur, rr, ri);
            acc0[r] = fma(rr, rr, acc0[r]);
            acc1[r] = fma(ri, ri, acc1[r]);
        }
    };
    int j = 0;
    for (; j + 2 <= a.N; j += 2) {
        const double *r1 = rec + REC;                       // j+1 exists
        const double *r2 = r1 < last ? r1 + REC : last;     // j+2, clamped
#pragma unroll
        for (int q = 0; q < REC; ++q) B[q] = r1[q];
        use(A);
#pragma unroll
        for (int q = 0; q < REC; ++q) A[q] = r2[q];
        use(B);
        rec = r2;
    }
    if (j < a.N) use(A);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long row = row0 + threadIdx.x + r * BLK;
        if (row < a.W) a.out[row] = ok[r] ? fma(-0.5, acc0[r] + acc1[r], a.lconst) : -__builtin_inf();
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class F>
static float time_kernel(F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 20;
}

template <int P>
static void run(int N, long long W)
{
    using M = PDCollapsed<P>;
    constexpr int NDIM = M::NDIM;
    std::vector<double> theta((size_t)W * NDIM), cb((size_t)N * M::REC);
    srand(11);
    for (long long i = 0; i < W; ++i) {
        theta[i * NDIM] = 0.9 + 0.2 * (rand() / (RAND_MAX + 1.0));
        for (int q = 1; q < NDIM; ++q) theta[i * NDIM + q] = -1.0 + 2.0 * (rand() / (RAND_MAX + 1.0));
    }
    for (auto &v : cb) v = -0.5 + (rand() / (RAND_MAX + 1.0));
    double *d_theta, *d_cb, *d_ref, *d_out;
    CK(hipMalloc(&d_theta, theta.size() * 8)); CK(hipMalloc(&d_cb, cb.size() * 8));
    CK(hipMalloc(&d_ref, W * 8)); CK(hipMalloc(&d_out, W * 8));
    CK(hipMemcpy(d_theta, theta.data(), theta.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cb, cb.data(), cb.size() * 8, hipMemcpyHostToDevice));
    LaunchArgs a{};
    a.theta = d_theta; a.W = W; a.cb = d_cb; a.N = N; a.lconst = 1.5;
    for (int q = 0; q < NDIM; ++q) { a.b.lo[q] = q ? -1.0 : 0.9; a.b.hi[q] = q ? 1.0 : 1.1; }
    std::vector<double> ref(W), out(W);
    auto bench = [&](const char *name, auto kern, int blk, int rows, bool is_ref) {
        LaunchArgs b = a; b.out = is_ref ? d_ref : d_out;
        const unsigned grid = (unsigned)((W + (long long)blk * rows - 1) / ((long long)blk * rows));
        auto go = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(blk), 0, 0, b); };
        go(); CK(hipDeviceSynchronize());
        if (is_ref) CK(hipMemcpy(ref.data(), d_ref, W * 8, hipMemcpyDeviceToHost));
        else {
            CK(hipMemcpy(out.data(), d_out, W * 8, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (long long i = 0; i < W; ++i) bad += out[i] != ref[i];
            if (bad) printf("  !! %s differs from the one-row kernel in %zu rows\n", name, bad);
        }
        const float t = time_kernel(go);
        printf("P=%d N=%3d W=%lld  %-22s %8.1f us  %.3e evals/s\n", P, N, W, name, t * 1e3, W / (t * 1e-3));
    };
    bench("k_logprob BLK256", k_logprob<M, 256, true, 1>, 256, 1, true);
    bench("x2 BLK128 (shipped)", k_logprob_x2<M, 128, true>, 128, 2, false);
    bench("xr<2> BLK256", k_logprob_xr<M, 256, 2, true>, 256, 2, false);
    bench("pf<1> BLK256", k_logprob_xr_pf<M, 256, 1, true>, 256, 1, false);
    bench("pf<2> BLK128", k_logprob_xr_pf<M, 128, 2, true>, 128, 2, false);
    bench("pf<2> BLK256", k_logprob_xr_pf<M, 256, 2, true>, 256, 2, false);
    bench("pf<3> BLK128", k_logprob_xr_pf<M, 128, 3, true>, 128, 3, false);
    bench("xr<3> BLK128", k_logprob_xr<M, 128, 3, true>, 128, 3, false);
    bench("xr<4> BLK128", k_logprob_xr<M, 128, 4, true>, 128, 4, false);
    CK(hipFree(d_theta)); CK(hipFree(d_cb)); CK(hipFree(d_ref)); CK(hipFree(d_out));
}

int main()
{
    run<5>(32, 1LL << 24);
    run<5>(64, 1LL << 22);
    run<5>(20, (1LL << 22) + 13);
    return 0;
}

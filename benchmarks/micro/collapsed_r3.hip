// Round 3: where does the PDCollapsed log-probability kernel lose its issue slots?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o collapsed_r3 collapsed_r3.hip
// Every variant evaluates the same rows with the same per-row arithmetic and order as
// logprob_row<PDCollapsed<P>> and is compared with it bit for bit.  Besides time per launch the
// program prints the shader clock each kernel ran at (s_memtime against the constant 100 MHz
// s_memrealtime inside one workgroup of the launch) and, from the static instruction count of the
// loop, cycles per VALU instruction per SIMD -- the number that says whether a kernel is at the
// issue limit of the clock it was given or below it.
//
//   x2<BLK>        shipped: two rows per lane, records through the scalar cache
//   xr<BLK,R>      R rows per lane
//   nold<BLK,R>    the same FMAs on ONE record held in SGPRs for the whole loop: no loads at all
//                  (wrong values; the ceiling of this instruction mix)
//   lds<BLK,R>     records staged in LDS once per workgroup, read by all lanes at one address
//   pers<R>        persistent single-wave workgroups: the next block's theta rows are in flight
//                  while this block is evaluated
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../bisip_amd/csrc/kernels.h"

using namespace bisip;

struct Clk { long long t0, t1, r0, r1; };

// The clocks are read by NON-volatile inline assembly tied to a value by an in/out operand:
// clock64() / wall_clock64() and any `asm volatile` count as memory writes, after which the compiler
// no longer proves the record loads unclobbered and fetches them lane by lane through the vector
// path instead of s_load (first version of this file: every scalar-path variant 3x slower than the
// shipped kernel for that reason alone).  `tie` is something everything after (begin) or before
// (end) the measured region depends on, so the read cannot move across it.
__device__ __forceinline__ void clk_read(long long &t, long long &r, unsigned &tie)
{
    unsigned long long a, b;
    asm("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a), "=s"(b), "+s"(tie));
    t = (long long)a; r = (long long)b;
}
#define CLK_BEGIN() unsigned bid_ = blockIdx.x; long long t = 0, rt = 0; clk_read(t, rt, bid_); const bool probe = bid_ == gridDim.x / 2;
__device__ __forceinline__ void clk_end(Clk *c, bool on, long long t, long long r, double last)
{
    long long t1, r1;
    unsigned tie = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(last));
    clk_read(t1, r1, tie);
    if (on && threadIdx.x == 0) { c->t0 = t; c->r0 = r; c->t1 = t1 + (tie & 0); c->r1 = r1; }
}

template <class M, int BLK, int R, bool NOLOAD>
__global__ __launch_bounds__(BLK) void k_xr(const LaunchArgs a, Clk *clk)
{
    constexpr int NDIM = M::NDIM;
    __shared__ __attribute__((aligned(16))) double lds[R * BLK * NDIM];
    CLK_BEGIN()
    const long long row0 = (long long)bid_ * (R * BLK);
    stage_theta<NDIM, R * BLK, true, BLK>(a.theta, a.W, row0, lds);
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
    double fixed[M::REC];
    if constexpr (NOLOAD) {
#pragma unroll
        for (int q = 0; q < M::REC; ++q) fixed[q] = rec[q];
    }
#pragma unroll 2
    for (int j = 0; j < a.N; ++j) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double rr, ri;
            if constexpr (NOLOAD) M::residual(s[r], fixed, rr, ri);
            else M::residual(s[r], rec, rr, ri);
            acc0[r] = fma(rr, rr, acc0[r]);
            acc1[r] = fma(ri, ri, acc1[r]);
        }
        if constexpr (!NOLOAD) rec += M::REC;
        else asm("" : "+v"(acc0[0]));      // keeps the loop a loop
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long row = row0 + threadIdx.x + r * BLK;
        if (row < a.W) a.out[row] = ok[r] ? fma(-0.5, acc0[r] + acc1[r], a.lconst) : -__builtin_inf();
    }
    clk_end(clk, probe, t, rt, acc0[0] + acc1[R - 1]);
}

// records in LDS: every lane reads the same address (a broadcast: no bank conflict)
template <class M, int BLK, int R, int MAXN>
__global__ __launch_bounds__(BLK) void k_lds(const LaunchArgs a, Clk *clk)
{
    constexpr int NDIM = M::NDIM;
    constexpr int REC = M::REC;
    __shared__ __attribute__((aligned(16))) double lds[R * BLK * NDIM];
    __shared__ __attribute__((aligned(16))) double lrec[MAXN * REC];
    CLK_BEGIN()
    const long long row0 = (long long)bid_ * (R * BLK);
    stage_theta<NDIM, R * BLK, true, BLK>(a.theta, a.W, row0, lds);
    for (int i = threadIdx.x; i < a.N * REC / 2; i += BLK)
        reinterpret_cast<dbl2 *>(lrec)[i] = reinterpret_cast<const dbl2 *>(a.cb)[i];
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
    const double *rec = lrec;
#pragma unroll 2
    for (int j = 0; j < a.N; ++j, rec += REC) {
        double cur[REC];
#pragma unroll
        for (int q = 0; q < REC; ++q) cur[q] = rec[q];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double rr, ri;
            M::residual(s[r], cur, rr, ri);
            acc0[r] = fma(rr, rr, acc0[r]);
            acc1[r] = fma(ri, ri, acc1[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long row = row0 + threadIdx.x + r * BLK;
        if (row < a.W) a.out[row] = ok[r] ? fma(-0.5, acc0[r] + acc1[r], a.lconst) : -__builtin_inf();
    }
    clk_end(clk, probe, t, rt, acc0[0] + acc1[R - 1]);
}

// persistent single-wave workgroups: block b, b + G, b + 2G, ...; the theta rows of the next block
// are loaded into registers (16 bytes per lane per load) before this block's frequency loop and
// written to LDS after it.  Rows per block = 64 R.
template <class M, int R>
__global__ __launch_bounds__(64) void k_pers(const LaunchArgs a, Clk *clk)
{
    constexpr int NDIM = M::NDIM;
    constexpr int ROWS = 64 * R;
    constexpr int N2 = ROWS * NDIM / 2;          // dbl2 per block
    constexpr int LOADS = (N2 + 63) / 64;
    static_assert((ROWS * NDIM) % 2 == 0, "even doubles per block");
    __shared__ __attribute__((aligned(16))) double lds[ROWS * NDIM];
    CLK_BEGIN()
    const long long nblk = (a.W + ROWS - 1) / ROWS;
    dbl2 nxt[LOADS];
    auto fetch = [&](long long blk) {
        const long long base = blk * ROWS * NDIM;
        const long long avail = a.W * NDIM - base;      // doubles left
        const dbl2 *src = reinterpret_cast<const dbl2 *>(a.theta + base);
#pragma unroll
        for (int q = 0; q < LOADS; ++q) {
            const int i = q * 64 + threadIdx.x;
            if (i < N2 && 2LL * i + 1 < avail) nxt[q] = __builtin_nontemporal_load(src + i);
            else { nxt[q].x = (i < N2 && 2LL * i < avail) ? a.theta[base + 2 * i] : 0.0; nxt[q].y = 0.0; }
        }
    };
    long long blk = bid_;
    double last = 0.0;
    if (blk < nblk) fetch(blk);
    for (; blk < nblk; blk += gridDim.x) {
#pragma unroll
        for (int q = 0; q < LOADS; ++q) {
            const int i = q * 64 + threadIdx.x;
            if (i < N2) reinterpret_cast<dbl2 *>(lds)[i] = nxt[q];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the wave's own LDS writes (single-wave workgroup)
        __builtin_amdgcn_wave_barrier();
        const long long row0 = blk * ROWS;
        double th[R][NDIM];
        bool ok[R];
        typename M::Setup s[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int q = 0; q < NDIM; ++q) th[r][q] = lds[(threadIdx.x + r * 64) * NDIM + q];
            ok[r] = in_prior<NDIM>(th[r], a.b) && (row0 + threadIdx.x + r * 64 < a.W);
            s[r] = M::setup(th[r]);
        }
        if (blk + gridDim.x < nblk) fetch(blk + gridDim.x);
        double acc0[R], acc1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }
        // constant address space: the stores of earlier blocks would otherwise make the compiler
        // fetch the records through the vector path
        typedef const __attribute__((address_space(4))) double cdouble;
        cdouble *rec = (cdouble *)a.cb;
#pragma unroll 2
        for (int j = 0; j < a.N; ++j, rec += M::REC) {
            double cur[M::REC];
#pragma unroll
            for (int q = 0; q < M::REC; ++q) cur[q] = rec[q];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double rr, ri;
                M::residual(s[r], cur, rr, ri);
                acc0[r] = fma(rr, rr, acc0[r]);
                acc1[r] = fma(ri, ri, acc1[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long row = row0 + threadIdx.x + r * 64;
            if (row < a.W) a.out[row] = ok[r] ? fma(-0.5, acc0[r] + acc1[r], a.lconst) : -__builtin_inf();
        }
        last = acc0[0] + acc1[R - 1];
    }
    clk_end(clk, probe, t, rt, last);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class F>
static float time_kernel(F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 300; ++i) launch();          // ~0.1 s: clocks settled
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 50;
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
    Clk *d_clk;
    CK(hipMalloc(&d_theta, theta.size() * 8)); CK(hipMalloc(&d_cb, cb.size() * 8));
    CK(hipMalloc(&d_ref, W * 8)); CK(hipMalloc(&d_out, W * 8)); CK(hipMalloc(&d_clk, sizeof(Clk)));
    CK(hipMemcpy(d_theta, theta.data(), theta.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cb, cb.data(), cb.size() * 8, hipMemcpyHostToDevice));
    LaunchArgs a{};
    a.theta = d_theta; a.W = W; a.cb = d_cb; a.N = N; a.lconst = 1.5;
    for (int q = 0; q < NDIM; ++q) { a.b.lo[q] = q ? -1.0 : 0.9; a.b.hi[q] = q ? 1.0 : 1.1; }
    std::vector<double> ref(W), out(W);
    // VALU instructions per row, static: 2(P+1)+1+2 FMAs per frequency + ~14 compares/selects for the prior,
    // P+1 products, the tail; the PMC count of the shipped kernel is 558 at P=5, N=32
    const double instr_per_row = N * (2.0 * (P + 1) + 3.0) + 78.0;
    auto bench = [&](const char *name, auto kern, int blk, int rows, int fixed_grid, int mode) {
        LaunchArgs b = a; b.out = mode == 0 ? d_ref : d_out;
        unsigned grid = (unsigned)((W + (long long)blk * rows - 1) / ((long long)blk * rows));
        if (fixed_grid) grid = (unsigned)fixed_grid;
        auto go = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(blk), 0, 0, b, d_clk); };
        go(); CK(hipDeviceSynchronize());
        if (mode == 0) CK(hipMemcpy(ref.data(), d_ref, W * 8, hipMemcpyDeviceToHost));
        else if (mode == 1) {
            CK(hipMemcpy(out.data(), d_out, W * 8, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (long long i = 0; i < W; ++i) bad += out[i] != ref[i];
            if (bad) printf("  !! %s differs from the one-row kernel in %zu rows\n", name, bad);
        }
        const float t = time_kernel(go);
        Clk c;
        CK(hipMemcpy(&c, d_clk, sizeof(c), hipMemcpyDeviceToHost));
        const double ghz = (double)(c.t1 - c.t0) / (double)(c.r1 - c.r0) * 0.1;
        const double cyc = ghz * 1e9 * (t * 1e-3) * 1024.0 / (instr_per_row * W / 64.0);
        printf("P=%d N=%3d W=%lld  %-26s %8.1f us  %.3e evals/s  %.2f GHz  %.2f cyc/VALU/SIMD  block %lld cyc\n", P, N, W,
               name, t * 1e3, W / (t * 1e-3), ghz, cyc, c.t1 - c.t0);
    };
    bench("k_logprob 1 row BLK256", k_xr<M, 256, 1, false>, 256, 1, 0, 0);
    {   // the shipped kernel itself (no clock probe): the probe's effect on the others
        LaunchArgs b = a; b.out = d_out;
        const unsigned grid = (unsigned)((W + 511) / 512);
        const float t = time_kernel([&] { hipLaunchKernelGGL((k_logprob_x2<M, 256, true>), dim3(grid), dim3(256), 0, 0, b); });
        printf("P=%d N=%3d W=%lld  %-26s %8.1f us  %.3e evals/s\n", P, N, W, "k_logprob_x2 (shipped)", t * 1e3, W / (t * 1e-3));
    }
    bench("x2 BLK256 (shipped form)", k_xr<M, 256, 2, false>, 256, 2, 0, 1);
    bench("x2 BLK128", k_xr<M, 128, 2, false>, 128, 2, 0, 1);
    bench("x2 BLK64", k_xr<M, 64, 2, false>, 64, 2, 0, 1);
    bench("x4 BLK64", k_xr<M, 64, 4, false>, 64, 4, 0, 1);
    bench("x4 BLK128", k_xr<M, 128, 4, false>, 128, 4, 0, 1);
    bench("no loads x2 BLK256", k_xr<M, 256, 2, true>, 256, 2, 0, 2);
    bench("no loads x4 BLK128", k_xr<M, 128, 4, true>, 128, 4, 0, 2);
    bench("lds x2 BLK256", k_lds<M, 256, 2, 64>, 256, 2, 0, 1);
    bench("lds x4 BLK128", k_lds<M, 128, 4, 64>, 128, 4, 0, 1);
    bench("lds x4 BLK256", k_lds<M, 256, 4, 64>, 256, 4, 0, 1);
    bench("lds x2 BLK64", k_lds<M, 64, 2, 64>, 64, 2, 0, 1);
    bench("pers x2 grid 256x16", k_pers<M, 2>, 64, 2, 256 * 16, 1);
    bench("pers x2 grid 256x20", k_pers<M, 2>, 64, 2, 256 * 20, 1);
    bench("pers x2 grid 256x32", k_pers<M, 2>, 64, 2, 256 * 32, 1);
    bench("pers x4 grid 256x16", k_pers<M, 4>, 64, 4, 256 * 16, 1);
    CK(hipFree(d_theta)); CK(hipFree(d_cb)); CK(hipFree(d_ref)); CK(hipFree(d_out)); CK(hipFree(d_clk));
}

int main()
{
    run<5>(32, 1LL << 24);
    run<5>(64, 1LL << 22);
    run<5>(20, (1LL << 22) + 13);
    return 0;
}

// Output path of the whole-row batched forward kernel at N = 20 (the bundled spectra): variants of
// k_forward_rows (kernels.h), all bit-identical, timed with primed clocks.
//   SUB    walkers per LDS pass (32 = shipped in round 1, 64 = the whole wave at once)
//   NT     non-temporal stores or plain
//   DIRECT no LDS: every lane stores its own 2N-double row with 16-byte stores (stride 16 N B)
//   STAGE  theta rows fetched with coalesced 16-B loads through LDS instead of strided per-lane loads
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -o forward_rows_variants forward_rows_variants.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../bisip_amd/csrc/kernels.h"
using namespace bisip;

template <class M, int JC, int SUB, bool NT, bool DIRECT, bool STAGE>
__global__ __launch_bounds__(64) void k_rows(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    constexpr int ROWMAX = 2 * JC + 1;
    constexpr int LDS_DOUBLES = DIRECT ? (STAGE ? 64 * NDIM : 1) : (SUB * ROWMAX > 64 * NDIM ? SUB * ROWMAX : 64 * NDIM);
    __shared__ __attribute__((aligned(16))) double lds[LDS_DOUBLES];
    const int lane = threadIdx.x;
    const int N = a.N;
    const int rowlen = (2 * N) | 1;
    const long long blk = blockIdx.x;
    const long long row0 = blk * 64;
    const int rows_here = (int)((a.W - row0) < 64 ? (a.W - row0) : 64);
    const long long row = row0 + lane < a.W ? row0 + lane : a.W - 1;
    double th[NDIM];
    if constexpr (STAGE) {
        stage_theta<NDIM, 64, true>(a.theta, a.W, row0, lds);
        wave_lds_fence();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[q] = lds[(lane < rows_here ? lane : 0) * NDIM + q];
        wave_lds_fence();
    } else {
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[q] = a.theta[row * NDIM + q];
    }
    const typename M::Setup s = M::setup(th);
    double zr[JC], zi[JC];
#pragma unroll
    for (int jj = 0; jj < JC; ++jj) {
        zr[jj] = 0.0; zi[jj] = 0.0;
        if (jj < N) M::eval(s, a.cb + (long long)jj * M::REC + 4, zr[jj], zi[jj]);
    }
    if constexpr (DIRECT) {
        // N even: the row is 16 N bytes, 16-byte aligned; piece q = doubles 2q, 2q+1 of [re | im]
        if (lane < rows_here) {
            dbl2 *dst = reinterpret_cast<dbl2 *>(a.out + (row0 + lane) * 2 * N);
#pragma unroll
            for (int q = 0; q < JC; ++q) {
                if (q < N) {
                    // doubles 2q, 2q+1 of the concatenated row [zr[0..N) zi[0..N)]
                    dbl2 v;
                    const int i0 = 2 * q, i1 = 2 * q + 1;
                    v.x = i0 < N ? zr[i0 < JC ? i0 : 0] : zi[(i0 - N) < JC ? (i0 - N) : 0];
                    v.y = i1 < N ? zr[i1 < JC ? i1 : 0] : zi[(i1 - N) < JC ? (i1 - N) : 0];
                    if (NT) __builtin_nontemporal_store(v, dst + q); else dst[q] = v;
                }
            }
        }
    } else {
#pragma unroll 1
        for (int sub = 0; sub * SUB < rows_here; ++sub) {
            if ((lane / SUB) == sub) {
                double *r = lds + (lane % SUB) * rowlen;
#pragma unroll
                for (int jj = 0; jj < JC; ++jj)
                    if (jj < N) { r[jj] = zr[jj]; r[N + jj] = zi[jj]; }
            }
            wave_lds_fence();
            const int wn = (rows_here - sub * SUB) < SUB ? (rows_here - sub * SUB) : SUB;
            double *dst0 = a.out + (row0 + sub * SUB) * 2 * N;
            const int total = wn * N, dw = 64 / N, de = 64 - dw * N;
            int w = lane / N, e = lane - w * N;
            for (int q = lane; q < total; q += 64) {
                const double *src = lds + w * rowlen + 2 * e;
                dbl2 v;
                v.x = src[0];
                v.y = src[1];
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(dst0 + 2 * (long long)q));
                else *reinterpret_cast<dbl2 *>(dst0 + 2 * (long long)q) = v;
                w += dw; e += de;
                if (e >= N) { e -= N; ++w; }
            }
            wave_lds_fence();
        }
    }
}

// k_forward_tiled (kernels.h) without its grid-stride loop / next-block prefetch and with the
// store width fixed at compile time (N a multiple of 16, Z 16-byte aligned).
template <class M>
__global__ __launch_bounds__(64) void k_tiled_straight(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    constexpr int JC = 16;
    constexpr int ROW = JC + 1;
    __shared__ __attribute__((aligned(16))) double lds[64 * ROW];
    const int lane = threadIdx.x;
    const int N = a.N;
    const long long row0 = (long long)blockIdx.x * 64;
    const int rows_here = (int)((a.W - row0) < 64 ? (a.W - row0) : 64);
    const long long row = row0 + lane < a.W ? row0 + lane : a.W - 1;
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = a.theta[row * NDIM + q];
    const typename M::Setup s = M::setup(th);
    auto stream_out = [&](int part, int j0) {
        const int c = (lane & 7) << 1;
        const double *src = lds + (lane >> 3) * ROW + c;
        double *dst = a.out + (row0 + (lane >> 3)) * 2 * N + (long long)part * N + j0 + c;
#pragma unroll 4
        for (int w = lane >> 3; w < rows_here; w += 8, src += 8 * ROW, dst += 16 * (long long)N) {
            dbl2 v;
            v.x = src[0];
            v.y = src[1];
            __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(dst));
        }
    };
    for (int j0 = 0; j0 < N; j0 += JC) {
        const double *__restrict__ rec = a.cb + (long long)j0 * M::REC;
        double zim[JC];
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) {
            double zr, zi;
            M::eval(s, rec + (long long)jj * M::REC + 4, zr, zi);
            lds[lane * ROW + jj] = zr;
            zim[jj] = zi;
        }
        wave_lds_fence();
        stream_out(0, j0);
        wave_lds_fence();
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) lds[lane * ROW + jj] = zim[jj];
        wave_lds_fence();
        stream_out(1, j0);
        wave_lds_fence();
    }
}

template <class M>
void run_tiled(const char *name, int N, long long W, const double *lo, const double *hi);

template <class F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 300; ++i) f();          // primed clocks
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) f();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 50;
}

template <class M>
void run(const char *name, int N, long long W, const double *lo, const double *hi)
{
    std::vector<double> th(W * M::NDIM), cb((size_t)N * M::REC, 0.0);
    unsigned long long s = 777;
    for (long long r = 0; r < W; ++r)
        for (int q = 0; q < M::NDIM; ++q) { s = s * 6364136223846793005ull + 1442695040888963407ull; th[r * M::NDIM + q] = lo[q] + (hi[q] - lo[q]) * ((double)(s >> 11) / 9007199254740992.0); }
    for (int j = 0; j < N; ++j) { double *r = &cb[(size_t)j * M::REC]; for (int k = 0; k < M::REC; ++k) r[k] = 0.01 * (k + 1) + 0.001 * j; if (M::REC == 8) { r[4] = 4e4 / (j + 1.0); r[5] = std::log(r[4]); r[6] = std::sqrt(r[4]); } }
    double *d_th, *d_cb, *d_z, *d_ref;
    hipMalloc(&d_th, th.size() * 8); hipMalloc(&d_cb, cb.size() * 8); hipMalloc(&d_z, W * 2 * N * 8); hipMalloc(&d_ref, W * 2 * N * 8);
    hipMemcpy(d_th, th.data(), th.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_cb, cb.data(), cb.size() * 8, hipMemcpyHostToDevice);
    LaunchArgs a{}; a.theta = d_th; a.W = W; a.cb = d_cb; a.N = N; a.lconst = 0;
    const unsigned grid = (unsigned)((W + 63) / 64);
    const double bytes = (double)W * (2 * N + M::NDIM) * 8;
    std::vector<double> ref(W * 2 * N), got(W * 2 * N);
    a.out = d_ref;
    hipLaunchKernelGGL((k_forward_rows<M, 24, true>), dim3(grid), dim3(64), 0, 0, a);
    hipMemcpy(ref.data(), d_ref, ref.size() * 8, hipMemcpyDeviceToHost);
    a.out = d_z;
    auto report = [&](const char *variant, float ms) {
        hipMemcpy(got.data(), d_z, got.size() * 8, hipMemcpyDeviceToHost);
        const bool same = std::memcmp(got.data(), ref.data(), got.size() * 8) == 0;
        printf("%-14s N=%2d W=%lld  %-34s %8.1f us  %7.1f GB/s  %s\n", name, N, W, variant, ms * 1e3, bytes / (ms * 1e-3) / 1e9, same ? "bits ok" : "MISMATCH");
        hipMemset(d_z, 0, got.size() * 8);
    };
    report("shipped rows<24> (SUB 32, nt)", timeit([&] { hipLaunchKernelGGL((k_forward_rows<M, 24, true>), dim3(grid), dim3(64), 0, 0, a); }));
#define V(SUB, NT, DIRECT, STAGE, label) report(label, timeit([&] { hipLaunchKernelGGL((k_rows<M, 24, SUB, NT, DIRECT, STAGE>), dim3(grid), dim3(64), 0, 0, a); }));
    V(32, true, false, false, "SUB 32 nt")
    V(32, false, false, false, "SUB 32 plain")
    V(64, true, false, false, "SUB 64 nt")
    V(64, false, false, false, "SUB 64 plain")
    V(64, false, false, true, "SUB 64 plain, staged theta")
    V(32, false, false, true, "SUB 32 plain, staged theta")
    V(64, true, true, false, "direct rows nt")
    V(64, false, true, false, "direct rows plain")
    V(64, false, true, true, "direct rows plain, staged theta")
#undef V
    hipFree(d_th); hipFree(d_cb); hipFree(d_z); hipFree(d_ref);
}

template <class M>
void run_tiled(const char *name, int N, long long W, const double *lo, const double *hi)
{
    std::vector<double> th(W * M::NDIM), cb((size_t)N * M::REC, 0.0);
    unsigned long long s = 777;
    for (long long r = 0; r < W; ++r)
        for (int q = 0; q < M::NDIM; ++q) { s = s * 6364136223846793005ull + 1442695040888963407ull; th[r * M::NDIM + q] = lo[q] + (hi[q] - lo[q]) * ((double)(s >> 11) / 9007199254740992.0); }
    for (int j = 0; j < N; ++j) { double *r = &cb[(size_t)j * M::REC]; for (int k = 0; k < M::REC; ++k) r[k] = 0.01 * (k + 1) + 0.001 * j; if (M::REC == 8) { r[4] = 4e4 / (j + 1.0); r[5] = std::log(r[4]); r[6] = std::sqrt(r[4]); } }
    double *d_th, *d_cb, *d_z, *d_ref;
    hipMalloc(&d_th, th.size() * 8); hipMalloc(&d_cb, cb.size() * 8); hipMalloc(&d_z, W * 2 * N * 8); hipMalloc(&d_ref, W * 2 * N * 8);
    hipMemcpy(d_th, th.data(), th.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_cb, cb.data(), cb.size() * 8, hipMemcpyHostToDevice);
    LaunchArgs a{}; a.theta = d_th; a.W = W; a.cb = d_cb; a.N = N; a.lconst = 0;
    const unsigned grid = (unsigned)((W + 63) / 64);
    const double bytes = (double)W * (2 * N + M::NDIM) * 8;
    std::vector<double> ref(W * 2 * N), got(W * 2 * N);
    a.out = d_ref;
    hipLaunchKernelGGL((k_forward_tiled<M, true>), dim3(grid), dim3(64), 0, 0, a);
    hipMemcpy(ref.data(), d_ref, ref.size() * 8, hipMemcpyDeviceToHost);
    a.out = d_z;
    const float t0 = timeit([&] { hipLaunchKernelGGL((k_forward_tiled<M, true>), dim3(grid), dim3(64), 0, 0, a); });
    printf("%-14s N=%2d W=%lld  %-34s %8.1f us  %7.1f GB/s\n", name, N, W, "shipped tiled", t0 * 1e3, bytes / (t0 * 1e-3) / 1e9);
    hipMemset(d_z, 0, got.size() * 8);
    const float t1 = timeit([&] { hipLaunchKernelGGL((k_tiled_straight<M>), dim3(grid), dim3(64), 0, 0, a); });
    hipMemcpy(got.data(), d_z, got.size() * 8, hipMemcpyDeviceToHost);
    printf("%-14s N=%2d W=%lld  %-34s %8.1f us  %7.1f GB/s  %s\n", name, N, W, "straight-line tiled", t1 * 1e3, bytes / (t1 * 1e-3) / 1e9,
           std::memcmp(got.data(), ref.data(), got.size() * 8) == 0 ? "bits ok" : "MISMATCH");
    hipFree(d_th); hipFree(d_cb); hipFree(d_z); hipFree(d_ref);
}

int main()
{
    const double lo7[7] = {0.9, 0, 0, -15, -15, 0, 0}, hi7[7] = {1.1, 1, 1, 5, 5, 1, 1};
    const double lop[7] = {0.9, -1, -1, -1, -1, -1, -1}, hip_[7] = {1.1, 1, 1, 1, 1, 1, 1};
    run<PDCollapsed<5>>("PDCollapsed<5>", 20, 1 << 21, lop, hip_);
    run<ColeCole<2>>("ColeCole<2>", 20, 1 << 21, lo7, hi7);
    run<PDCollapsed<5>>("PDCollapsed<5>", 20, 100000, lop, hip_);
    run_tiled<PDCollapsed<5>>("PDCollapsed<5>", 32, 1 << 21, lop, hip_);
    run_tiled<PDCollapsed<5>>("PDCollapsed<5>", 64, 1 << 20, lop, hip_);
    run_tiled<ColeCole<2>>("ColeCole<2>", 32, 1 << 21, lo7, hi7);
    return 0;
}

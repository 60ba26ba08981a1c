// A/B of the batched-forward kernel's output path on one MI355X (build: hipcc -O3
// --offload-arch=gfx950 -ffp-contract=off -o forward_variants forward_variants.hip).
//   tiled : the shipped k_forward_tiled (16-frequency tile of one part through LDS, 128-B runs)
//   rows  : whole walker rows -- 2*JC doubles per lane in registers, 16 walkers at a time
//           through LDS, each store instruction writes 1 KB of consecutive Z
// Checks rows == tiled bit for bit, then times both (HIP events, 20 launches).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../bisip_amd/csrc/kernels.h"

using namespace bisip;

template <class M, int JC, bool VEC>
__global__ __launch_bounds__(64) void k_forward_rows(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    constexpr int SUB = 16;
    constexpr int ROW = 2 * JC + 1;
    __shared__ __attribute__((aligned(16))) double lds[SUB * ROW];
    const int lane = threadIdx.x;
    const int N = a.N;
    const bool wide = ((N & 1) == 0) && ((reinterpret_cast<unsigned long long>(a.out) & 15) == 0);
    const long long nblocks = (a.W + 63) / 64;
    double th_next[NDIM];
    auto request = [&](long long blk) {
        const long long row = blk * 64 + lane;
        const long long r = row < a.W ? row : a.W - 1;
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th_next[q] = a.theta[r * NDIM + q];
    };
    long long blk = blockIdx.x;
    if (blk < nblocks) request(blk);
    for (; blk < nblocks; blk += gridDim.x) {
        const long long row0 = blk * 64;
        const int rows_here = (int)((a.W - row0) < 64 ? (a.W - row0) : 64);
        double th[NDIM];
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[q] = th_next[q];
        if (blk + gridDim.x < nblocks) request(blk + gridDim.x);
        const typename M::Setup s = M::setup(th);
        for (int j0 = 0; j0 < N; j0 += JC) {
            const int jn = (N - j0) < JC ? (N - j0) : JC;
            const double *__restrict__ rec = a.cb + (long long)j0 * M::REC;
            double zr[JC], zi[JC];
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
                zr[jj] = 0.0; zi[jj] = 0.0;
                if (jj < jn) M::eval(s, rec + (long long)jj * M::REC + 4, zr[jj], zi[jj]);
            }
#pragma unroll 1
            for (int sub = 0; sub < 4; ++sub) {
                if (sub * SUB >= rows_here) break;   // wave-uniform
                if ((lane >> 4) == sub) {
                    double *row = lds + (lane & 15) * ROW;
#pragma unroll
                    for (int jj = 0; jj < JC; ++jj) { row[jj] = zr[jj]; row[JC + jj] = zi[jj]; }
                }
                wave_lds_fence();
                const int wn = (rows_here - sub * SUB) < SUB ? (rows_here - sub * SUB) : SUB;
                if (jn == JC && wide) {
                    // JC pairs per walker (2 parts x JC/2): lane -> (walker lane/JC.., pair lane%JC)
                    constexpr int WPI = 64 / JC;            // walkers per store instruction
                    const int r = lane % JC, w0 = lane / JC;
                    const int p = r / (JC / 2), jj = (r % (JC / 2)) * 2;
                    const double *src = lds + w0 * ROW + p * JC + jj;
                    double *dst = a.out + (row0 + sub * SUB + w0) * 2 * N + (long long)p * N + j0 + jj;
#pragma unroll 4
                    for (int w = w0; w < wn; w += WPI, src += WPI * ROW, dst += WPI * 2 * (long long)N) {
                        dbl2 v;
                        v.x = src[0];
                        v.y = src[1];
                        __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(dst));
                    }
                } else {
                    const int total = wn * 2 * jn;
                    for (int flat = lane; flat < total; flat += 64) {
                        const int w = flat / (2 * jn), r = flat - w * 2 * jn;
                        const int p = r / jn, jj = r - p * jn;
                        a.out[(row0 + sub * SUB + w) * 2 * N + (long long)p * N + j0 + jj] = lds[w * ROW + p * JC + jj];
                    }
                }
                wave_lds_fence();
            }
        }
    }
}

// tiled2: the shipped structure with (a) tile width chosen per launch so that tiles are equal
// (N=20 -> 10+10 instead of 16+4), (b) 16-byte stores for every even tile width, (c) JC as a
// template parameter (JC=24 takes N<=24 in one tile).
template <class M, int JC, bool VEC>
__global__ __launch_bounds__(64) void k_forward_tiled2(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    constexpr int ROW = JC + 1;
    __shared__ __attribute__((aligned(16))) double lds[64 * ROW];
    const int lane = threadIdx.x;
    const int N = a.N;
    const bool wide = ((N & 1) == 0) && ((reinterpret_cast<unsigned long long>(a.out) & 15) == 0);
    const long long nblocks = (a.W + 63) / 64;
    const int ntiles = (N + JC - 1) / JC;
    int tw = (N + ntiles - 1) / ntiles;
    tw += tw & 1;
    if (tw > JC) tw = JC;
    double th_next[NDIM];
    auto request = [&](long long blk) {
        const long long row = blk * 64 + lane;
        const long long r = row < a.W ? row : a.W - 1;
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th_next[q] = a.theta[r * NDIM + q];
    };
    auto stream_out = [&](long long row0, int rows_here, int part, int j0, int jn) {
        if (wide && (jn & 1) == 0) {
            const int hp = jn >> 1;            // 16-byte pieces per walker row of this tile
            const int wpi = 64 / hp;           // walkers per store instruction
            const int w0 = lane / hp, c = (lane - w0 * hp) << 1;
            if (w0 < wpi) {
                const double *src = lds + w0 * ROW + c;
                double *dst = a.out + (row0 + w0) * 2 * N + (long long)part * N + j0 + c;
#pragma unroll 4
                for (int w = w0; w < rows_here; w += wpi, src += wpi * ROW, dst += wpi * 2 * (long long)N) {
                    dbl2 v;
                    v.x = src[0];
                    v.y = src[1];
                    __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(dst));
                }
            }
        } else {
            const int total = rows_here * jn;
            for (int flat = lane; flat < total; flat += 64) {
                const int w = flat / jn, jj = flat - w * jn;
                a.out[(row0 + w) * 2 * N + (long long)part * N + j0 + jj] = lds[w * ROW + jj];
            }
        }
    };
    long long blk = blockIdx.x;
    if (blk < nblocks) request(blk);
    for (; blk < nblocks; blk += gridDim.x) {
        const long long row0 = blk * 64;
        const int rows_here = (int)((a.W - row0) < 64 ? (a.W - row0) : 64);
        double th[NDIM];
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[q] = th_next[q];
        if (blk + gridDim.x < nblocks) request(blk + gridDim.x);
        const typename M::Setup s = M::setup(th);
        for (int j0 = 0; j0 < N; j0 += tw) {
            const int jn = (N - j0) < tw ? (N - j0) : tw;
            const double *__restrict__ rec = a.cb + (long long)j0 * M::REC;
            double zim[JC];
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
                double zr = 0.0, zi = 0.0;
                if (jj < jn) M::eval(s, rec + (long long)jj * M::REC + 4, zr, zi);
                lds[lane * ROW + jj] = zr;
                zim[jj] = zi;
            }
            wave_lds_fence();
            stream_out(row0, rows_here, 0, j0, jn);
            wave_lds_fence();
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) lds[lane * ROW + jj] = zim[jj];
            wave_lds_fence();
            stream_out(row0, rows_here, 1, j0, jn);
            wave_lds_fence();
        }
    }
}

// rows2: N <= JC, whole rows.  Lane = walker computes all N frequencies (2N doubles in
// registers); SUB walkers at a time go through LDS laid out exactly like Z ([re 0..N) [im 0..N)
// per walker), so the SUB*2N doubles of a pass are ONE contiguous span of Z and every store
// instruction writes 1 KB (16-byte pieces) or 512 B (odd N) of consecutive addresses:
// no cache line is ever written in parts.
template <class M, int JC, bool VEC>
__global__ __launch_bounds__(64) void k_forward_rows2(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    constexpr int SUB = 32;
    constexpr int ROWMAX = 2 * JC + 1;
    __shared__ __attribute__((aligned(16))) double lds[SUB * ROWMAX];
    const int lane = threadIdx.x;
    const int N = a.N;
    const int rowlen = (2 * N) | 1;   // odd stride: conflict-free column writes
    const bool wide = ((N & 1) == 0) && ((reinterpret_cast<unsigned long long>(a.out) & 15) == 0);
    const long long nblocks = (a.W + 63) / 64;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long row0 = blk * 64;
        const int rows_here = (int)((a.W - row0) < 64 ? (a.W - row0) : 64);
        const long long row = row0 + lane < a.W ? row0 + lane : a.W - 1;
        double th[NDIM];
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[q] = a.theta[row * NDIM + q];
        const typename M::Setup s = M::setup(th);
        double zr[JC], zi[JC];
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) {
            zr[jj] = 0.0; zi[jj] = 0.0;
            if (jj < N) M::eval(s, a.cb + (long long)jj * M::REC + 4, zr[jj], zi[jj]);
        }
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
            if (wide) {
                // 16-byte piece q of the pass = walker q / N, doubles 2*(q % N)..+1
                const int total = wn * N, dw = 64 / N, de = 64 - dw * N;
                int w = lane / N, e = lane - w * N;
                for (int q = lane; q < total; q += 64) {
                    const double *src = lds + w * rowlen + 2 * e;
                    dbl2 v;
                    v.x = src[0];
                    v.y = src[1];
                    __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(dst0 + 2 * (long long)q));
                    w += dw; e += de;
                    if (e >= N) { e -= N; ++w; }
                }
            } else {
                const int M2 = 2 * N, total = wn * M2, dw = 64 / M2, de = 64 - dw * M2;
                int w = lane / M2, e = lane - w * M2;
                for (int f = lane; f < total; f += 64) {
                    dst0[f] = lds[w * rowlen + e];
                    w += dw; e += de;
                    if (e >= M2) { e -= M2; ++w; }
                }
            }
            wave_lds_fence();
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class M, class F>
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

template <class M>
static void run(const char *name, int N, long long W, const std::vector<double> &lo, const std::vector<double> &hi)
{
    constexpr int NDIM = M::NDIM;
    std::vector<double> theta((size_t)W * NDIM), cb((size_t)N * M::REC);
    srand(7);
    for (long long i = 0; i < W; ++i)
        for (int q = 0; q < NDIM; ++q) theta[i * NDIM + q] = lo[q] + (hi[q] - lo[q]) * (rand() / (RAND_MAX + 1.0));
    for (int j = 0; j < N; ++j) {
        double *r = &cb[(size_t)j * M::REC];
        for (int q = 0; q < M::REC; ++q) r[q] = 0.3 + 0.4 * (rand() / (RAND_MAX + 1.0));
        const double w = 2 * M_PI * pow(10.0, 3.78 - 5.7 * j / (N > 1 ? N - 1 : 1));
        if (M::REC == 8) { r[4] = w; r[5] = log(w); r[6] = sqrt(w); }
    }
    double *d_theta, *d_cb, *d_a, *d_b;
    const size_t zbytes = (size_t)W * 2 * N * 8;
    CK(hipMalloc(&d_theta, theta.size() * 8)); CK(hipMalloc(&d_cb, cb.size() * 8));
    CK(hipMalloc(&d_a, zbytes)); CK(hipMalloc(&d_b, zbytes));
    CK(hipMemcpy(d_theta, theta.data(), theta.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cb, cb.data(), cb.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_a, 0, zbytes)); CK(hipMemset(d_b, 0xff, zbytes));
    LaunchArgs a{};
    a.theta = d_theta; a.W = W; a.cb = d_cb; a.N = N; a.lconst = 0.0;
    long long blocks = (W + 63) / 64;
    auto go = [&](auto kern, double *out, long long per_cu) {
        LaunchArgs b = a; b.out = out;
        long long g = blocks < 256 * per_cu ? blocks : 256 * per_cu;
        hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(64), 0, 0, b);
    };
    go(k_forward_tiled<M, true>, d_a, 16);
    std::vector<double> za((size_t)W * 2 * N), zb(za.size());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(za.data(), d_a, zbytes, hipMemcpyDeviceToHost));
    auto check = [&](const char *what) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(zb.data(), d_b, zbytes, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < za.size(); ++i) bad += (za[i] != zb[i]) && !(za[i] != za[i] && zb[i] != zb[i]);
        if (bad) printf("  !! %s differs from tiled in %zu values\n", what, bad);
        CK(hipMemset(d_b, 0xff, zbytes));
    };
    const double gb = zbytes / 1e9;
    float t;
    for (long long per_cu : {16LL, 4096LL}) {
        t = time_kernel<M>([&] { go(k_forward_tiled<M, true>, d_a, per_cu); });
        printf("%-14s N=%3d W=%lld  tiled      x%-4lld %8.1f us  %7.1f GB/s\n", name, N, W, per_cu, t * 1e3, gb / (t * 1e-3));
    }
    for (long long per_cu : {32LL, 4096LL}) {
        go(k_forward_tiled2<M, 16, true>, d_b, per_cu); check("tiled2<16>");
        t = time_kernel<M>([&] { go(k_forward_tiled2<M, 16, true>, d_b, per_cu); });
        printf("%-14s N=%3d W=%lld  tiled2<16> x%-4lld %8.1f us  %7.1f GB/s\n", name, N, W, per_cu, t * 1e3, gb / (t * 1e-3));
    }
    if (N <= 24) {
        go(k_forward_rows2<M, 24, true>, d_b, 4096); check("rows2<24>");
        t = time_kernel<M>([&] { go(k_forward_rows2<M, 24, true>, d_b, 4096); });
        printf("%-14s N=%3d W=%lld  rows2<24>  x4096  %8.1f us  %7.1f GB/s\n", name, N, W, t * 1e3, gb / (t * 1e-3));
    }
    if (N <= 32) {
        go(k_forward_rows2<M, 32, true>, d_b, 4096); check("rows2<32>");
        t = time_kernel<M>([&] { go(k_forward_rows2<M, 32, true>, d_b, 4096); });
        printf("%-14s N=%3d W=%lld  rows2<32>  x4096  %8.1f us  %7.1f GB/s\n", name, N, W, t * 1e3, gb / (t * 1e-3));
    }
    for (long long per_cu : {4096LL}) {
        go(k_forward_tiled2<M, 24, true>, d_b, per_cu); check("tiled2<24>");
        t = time_kernel<M>([&] { go(k_forward_tiled2<M, 24, true>, d_b, per_cu); });
        printf("%-14s N=%3d W=%lld  tiled2<24> x%-4lld %8.1f us  %7.1f GB/s\n", name, N, W, per_cu, t * 1e3, gb / (t * 1e-3));
    }
    CK(hipFree(d_theta)); CK(hipFree(d_cb)); CK(hipFree(d_a)); CK(hipFree(d_b));
}

int main()
{
    const long long W = 1 << 21;
    run<PDCollapsed<5>>("PDCollapsed<5>", 32, W, {0.9, -1, -1, -1, -1, -1, -1}, {1.1, 1, 1, 1, 1, 1, 1});
    run<ColeCole<2>>("ColeCole<2>", 32, W, {0.9, 0, 0, -15, -15, 0, 0}, {1.1, 1, 1, 5, 5, 1, 1});
    run<PDCollapsed<5>>("PDCollapsed<5>", 64, W / 2, {0.9, -1, -1, -1, -1, -1, -1}, {1.1, 1, 1, 1, 1, 1, 1});
    run<PDCollapsed<5>>("PDCollapsed<5>", 20, W + 37, {0.9, -1, -1, -1, -1, -1, -1}, {1.1, 1, 1, 1, 1, 1, 1});
    run<Dias>("Dias", 21, W / 2 + 5, {0.9, 0, -20, 0, 0}, {1.1, 1, 0, 150, 1});
    run<ColeCole<2>>("ColeCole<2>", 20, W, {0.9, 0, 0, -15, -15, 0, 0}, {1.1, 1, 1, 5, 5, 1, 1});
    run<PDCollapsed<5>>("PDCollapsed<5>", 40, W / 2, {0.9, -1, -1, -1, -1, -1, -1}, {1.1, 1, 1, 1, 1, 1, 1});
    run<PDCollapsed<5>>("PDCollapsed<5>", 30, W, {0.9, -1, -1, -1, -1, -1, -1}, {1.1, 1, 1, 1, 1, 1, 1});
    run<PDCollapsed<5>>("PDCollapsed<5>", 7, W, {0.9, -1, -1, -1, -1, -1, -1}, {1.1, 1, 1, 1, 1, 1, 1});
    run<PDCollapsed<5>>("PDCollapsed<5>", 20, 1000, {0.9, -1, -1, -1, -1, -1, -1}, {1.1, 1, 1, 1, 1, 1, 1});
    run<PDCollapsed<5>>("PDCollapsed<5>", 32, 1000, {0.9, -1, -1, -1, -1, -1, -1}, {1.1, 1, 1, 1, 1, 1, 1});
    return 0;
}

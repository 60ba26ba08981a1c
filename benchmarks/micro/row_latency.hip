// One wave per SIMD running logprob_row<ColeCole<2>> (the arithmetic of a cfg5 half-step, no
// proposal / gather / commit): cycles per wave by s_memtime, for 1 / 2 / 4 waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -o row_latency row_latency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../../bisip_amd/csrc/kernels.h"
using namespace bisip;

template <class M, bool STAGE>
__global__ __launch_bounds__(64) void k_row(const double *theta, double *out, long long *cyc, const double *cb, int N, Bounds b)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const long long row = (long long)blockIdx.x * 64 + threadIdx.x;
    double th[M::NDIM];
#pragma unroll
    for (int q = 0; q < M::NDIM; ++q) th[q] = theta[row * M::NDIM + q];
    const double *rec = cb;
    if (STAGE) {
        for (int i = threadIdx.x; i < N * M::REC; i += 64) lds[i] = cb[i];
        __syncthreads();
        rec = lds;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t0 = __builtin_amdgcn_s_memtime();
    const ModelOperands o{rec, N, 1.0};
    const double lp = logprob_row<M, 1, STAGE>(th, o, b, 0);
    asm volatile("" :: "v"(lp));
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[row] = lp;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class M, bool STAGE>
void run(const char *name, int waves_per_simd, int N)
{
    const long long W = 65536LL * waves_per_simd;
    std::vector<double> th(W * M::NDIM), cb((size_t)N * M::REC);
    Bounds b;
    const double lo[7] = {0.9, 0, 0, -15, -15, 0, 0}, hi[7] = {1.1, 1, 1, 5, 5, 1, 1};
    for (int q = 0; q < 16; ++q) { b.lo[q] = q < 7 ? lo[q] : 0; b.hi[q] = q < 7 ? hi[q] : 0; }
    unsigned long long s = 12345;
    for (auto &x : th) { s = s * 6364136223846793005ull + 1; x = (double)(s >> 11) / 9007199254740992.0; }
    for (long long r = 0; r < W; ++r) for (int q = 0; q < M::NDIM; ++q) th[r * M::NDIM + q] = lo[q] + (hi[q] - lo[q]) * th[r * M::NDIM + q];
    for (int j = 0; j < N; ++j) { double *r = &cb[(size_t)j * M::REC]; r[0] = 0.9; r[1] = -0.05; r[2] = 1e3; r[3] = 1e4; r[4] = 1e3 / (j + 1); r[5] = std::log(r[4]); r[6] = std::sqrt(r[4]); }
    double *d_th, *d_out, *d_cb; long long *d_cyc;
    hipMalloc(&d_th, th.size() * 8); hipMalloc(&d_out, W * 8); hipMalloc(&d_cb, cb.size() * 8); hipMalloc(&d_cyc, W / 64 * 8);
    hipMemcpy(d_th, th.data(), th.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_cb, cb.data(), cb.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = STAGE ? (size_t)N * M::REC * 8 : 0;
    for (int rep = 0; rep < 200; ++rep) k_row<M, STAGE><<<W / 64, 64, lds>>>(d_th, d_out, d_cyc, d_cb, N, b);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int rep = 0; rep < 50; ++rep) k_row<M, STAGE><<<W / 64, 64, lds>>>(d_th, d_out, d_cyc, d_cb, N, b);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(W / 64);
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    printf("%-22s stage %d waves/SIMD %d: logprob_row median %6lld cycles (p10 %lld p90 %lld) = %.1f per frequency; launch %.2f us\n", name, (int)STAGE,
           waves_per_simd, c[c.size() / 2], c[c.size() / 10], c[c.size() * 9 / 10], (double)c[c.size() / 2] / N, ms * 1e3 / 50);
    hipFree(d_th); hipFree(d_out); hipFree(d_cb); hipFree(d_cyc);
}

int main()
{
    for (int wps : {1, 2, 4}) {
        run<ColeCole<2>, false>("ColeCole<2> N=32", wps, 32);
        run<ColeCole<2>, true>("ColeCole<2> N=32", wps, 32);
    }
    run<ColeCole<1>, false>("ColeCole<1> N=32", 1, 32);
    run<Dias, false>("Dias N=32", 1, 32);
    run<PDCollapsed<5>, false>("PDCollapsed<5> N=32", 1, 32);
    return 0;
}

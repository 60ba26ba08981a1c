// How fast does ONE wave per SIMD issue fp64 VALU instructions?  The stretch half-step of a
// 65,536-proposal ensemble is 1024 waves = one per SIMD; this measures cycles per v_fma_f64 for
// chains with ILP independent accumulators, at 1, 2 and 4 waves per SIMD, by s_memtime.
//   hipcc -O3 --offload-arch=gfx950 -o issue_latency issue_latency.hip && ./issue_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void k_chain(double *out, long long *cyc, long long *real, double seed, int iters)
{
    double a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = seed + threadIdx.x + i;
    const double b = seed * 0.5 + 0.25;
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / ILP; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i];
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        cyc[w] = t1 - t0;
        real[w] = r1 - r0;
    }
}

template <int ILP>
void run(int waves_per_simd, int block)
{
    const int iters = 200;
    const long long lanes = 1024LL * 64 * waves_per_simd;
    double *out; long long *cyc, *real;
    hipMalloc(&out, lanes * 8); hipMalloc(&cyc, lanes / 64 * 8); hipMalloc(&real, lanes / 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 40; ++rep) k_chain<ILP><<<lanes / block, block>>>(out, cyc, real, 1.0, iters);   // clocks up
    hipEventRecord(e0);
    k_chain<ILP><<<lanes / block, block>>>(out, cyc, real, 1.0, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(lanes / 64), r(lanes / 64);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), real, r.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
    const double n = 64.0 * iters;
    const double med = c[c.size() / 2], rmed = r[r.size() / 2];
    printf("ILP %d  waves/SIMD %d  block %3d : median %.2f cycles per fma per wave (p10 %.2f p90 %.2f), clock %.2f GHz, kernel %.1f us = %.2f cycles/fma/SIMD\n",
           ILP, waves_per_simd, block, med / n, c[c.size() / 10] / n, c[c.size() * 9 / 10] / n, med / rmed * 0.1, ms * 1e3,
           ms * 1e-3 * (med / rmed * 1e8) / (n * waves_per_simd));
    hipFree(out); hipFree(cyc); hipFree(real);
}

int main()
{
    for (int wps : {1, 2, 4}) {
        run<1>(wps, 64); run<2>(wps, 64); run<4>(wps, 64); run<8>(wps, 64);
    }
    run<1>(1, 256); run<4>(1, 256); run<8>(1, 256);
    // two-wave workgroups, two per CU (the persistent sampler's shape at 256 walkers per ensemble):
    // do the four waves of a CU land on four SIMDs?
    run<8>(1, 128); run<1>(1, 128);
    run<8>(1, 512); run<8>(1, 1024);
    return 0;
}

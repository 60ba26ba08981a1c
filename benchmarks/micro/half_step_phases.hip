// Phases of the cfg5 stretch half-step kernel (512 spectra x 128 slots, ColeCole<2>, N = 32) by
// s_memtime: stream + row gather | record staging | proposal + log-probability | commit.
// One wave per SIMD (1024 single-wave workgroups), like the product kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -o half_step_phases half_step_phases.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../bisip_amd/csrc/sampler_kernels.h"
using namespace bisip;

template <class LP, bool STAGED>
__global__ __launch_bounds__(64) void k_half_timed(const StretchArgs a, const LP lp, long long *stamps)
{
    constexpr int NDIM = LP::NDIM;
    const long long t = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long T0 = __builtin_amdgcn_s_memtime();
    const int i = a.active[t], p = a.partner[t];
    const double z = a.zz[t], fac = a.factor[t], lu = a.logu[t];
    double s[NDIM], c[NDIM];
#pragma unroll
    for (int k = 0; k < NDIM; ++k) { s[k] = a.coords[(long long)i * NDIM + k]; c[k] = a.coords[(long long)p * NDIM + k]; }
    const double old_lp = a.logp[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long T1 = __builtin_amdgcn_s_memtime();
    const double *staged = nullptr;
    const long long e = (long long)__builtin_amdgcn_readfirstlane(i / (int)lp.Wp);   // one spectrum per wave
    if constexpr (STAGED) {
        extern __shared__ __attribute__((aligned(16))) double lds_records[];
        const double *__restrict__ src = lp.records(e);
        for (int k = threadIdx.x; k < lp.n_freq() * LP::REC_DOUBLES; k += 64) lds_records[k] = src[k];
        __syncthreads();
        staged = lds_records;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long T2 = __builtin_amdgcn_s_memtime();
    double q[NDIM];
#pragma unroll
    for (int k = 0; k < NDIM; ++k) { const double d = c[k] - s[k]; q[k] = c[k] - d * z; }
    const auto loc = lp.local(e);
    const double new_lp = lp.template eval_ens<STAGED>(q, loc, 0, staged);
    asm volatile("" :: "v"(new_lp));
    const long long T3 = __builtin_amdgcn_s_memtime();
    const bool acc = (fac + new_lp) - old_lp > lu;
    double row[NDIM];
#pragma unroll
    for (int k = 0; k < NDIM; ++k) row[k] = acc ? q[k] : s[k];
    commit_row<NDIM>(a, i, row, acc ? new_lp : old_lp, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long T4 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { long long *o = stamps + 4 * blockIdx.x; o[0] = T1 - T0; o[1] = T2 - T1; o[2] = T3 - T2; o[3] = T4 - T3; }
}

int main()
{
    const int E = 512, Wp = 256, N = 32, ND = 7, REC = 8;
    const long long W = (long long)E * Wp, slots = W / 2;
    std::vector<double> coords(W * ND), logp(W, -10.0), cb((size_t)E * N * REC), lconst(E, 100.0), zz(slots), fac(slots), lu(slots);
    std::vector<int> act(slots), par(slots);
    unsigned long long s = 99;
    auto uni = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
    const double centre[7] = {1.0, 0.15, 0.5, -1.5, -12.0, 0.45, 0.6};
    for (long long w = 0; w < W; ++w) for (int q = 0; q < ND; ++q) coords[w * ND + q] = centre[q] + 1e-3 * (uni() - 0.5);
    for (int e = 0; e < E; ++e) for (int j = 0; j < N; ++j) { double *r = &cb[((size_t)e * N + j) * REC]; r[0] = 0.9; r[1] = -0.05; r[2] = 1e3; r[3] = 1e4; r[4] = 4e4 / (j + 1.0); r[5] = std::log(r[4]); r[6] = std::sqrt(r[4]); }
    for (int e = 0; e < E; ++e) for (int t = 0; t < Wp / 2; ++t) { const long long k = (long long)e * (Wp / 2) + t; act[k] = e * Wp + 2 * t; par[k] = e * Wp + 2 * (int)(uni() * (Wp / 2)) + 1; zz[k] = 0.5 + 1.5 * uni(); fac[k] = 6 * std::log(zz[k]); lu[k] = std::log(uni()); }
    StretchArgs a{};
    int *d_status, *d_nacc;
#define UP(dst, vec) hipMalloc((void **)&dst, vec.size() * sizeof(vec[0])); hipMemcpy((void *)dst, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice)
    double *d_cb, *d_lconst;
    UP(a.coords, coords); UP(a.logp, logp); UP(a.active, act); UP(a.partner, par); UP(a.zz, zz); UP(a.factor, fac); UP(a.logu, lu); UP(d_cb, cb); UP(d_lconst, lconst);
    hipMalloc(&d_status, 4); hipMemset(d_status, 0, 4); hipMalloc(&d_nacc, W * 4); hipMemset(d_nacc, 0, W * 4);
    hipMalloc(&a.chain_row, W * ND * 8); hipMalloc(&a.logp_row, W * 8);
    a.n_slots = slots; a.naccept = d_nacc; a.status = d_status;
    long long *d_st; hipMalloc(&d_st, slots / 64 * 4 * 8);
    Bounds b; const double lo[7] = {0.9, 0, 0, -15, -15, 0, 0}, hi[7] = {1.1, 1, 1, 5, 5, 1, 1};
    for (int q = 0; q < 16; ++q) { b.lo[q] = q < 7 ? lo[q] : 0; b.hi[q] = q < 7 ? hi[q] : 0; }
    auto report = [&](const char *name, float us) {
        std::vector<long long> st(slots / 64 * 4); hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
        const char *ph[4] = {"stream+gather", "stage records", "proposal+logprob", "accept+commit"};
        printf("%s: launch %.2f us;", name, us);
        for (int k = 0; k < 4; ++k) { std::vector<long long> v; for (size_t w = 0; w < st.size() / 4; ++w) v.push_back(st[4 * w + k]); std::sort(v.begin(), v.end()); printf("  %s %lld", ph[k], v[v.size() / 2]); }
        printf(" cycles (median over waves)\n");
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    {
        BatchGenericLP<ColeCole<2>, true, 1> lp; lp.cb = d_cb; lp.cb_stride = (long long)N * REC; lp.Wp = Wp; lp.lconst = d_lconst; lp.N = N; lp.b = b;
        for (int r = 0; r < 300; ++r) k_half_timed<decltype(lp), false><<<slots / 64, 64>>>(a, lp, d_st);
        hipEventRecord(e0); for (int r = 0; r < 100; ++r) k_half_timed<decltype(lp), false><<<slots / 64, 64>>>(a, lp, d_st); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1); report("scalar-cache records", ms * 10);
    }
    {
        BatchGenericLP<ColeCole<2>, true, 1> lp; lp.cb = d_cb; lp.cb_stride = (long long)N * REC; lp.Wp = Wp; lp.lconst = d_lconst; lp.N = N; lp.b = b;
        for (int r = 0; r < 300; ++r) k_half_timed<decltype(lp), true><<<slots / 64, 64, N * REC * 8>>>(a, lp, d_st);
        hipEventRecord(e0); for (int r = 0; r < 100; ++r) k_half_timed<decltype(lp), true><<<slots / 64, 64, N * REC * 8>>>(a, lp, d_st); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1); report("LDS-staged records  ", ms * 10);
    }
    return 0;
}

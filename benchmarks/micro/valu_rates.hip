// Issue-rate microbenchmark for the fp64 VALU instructions the transcendental kernels use.
// Each wave runs a long chain of independent instructions (8 accumulators); reports
// cycles per wave-instruction per SIMD with the chip full (4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITERS 200

#define KERNEL(name, body)                                                            \
    __global__ void name(double *out, double seed)                                    \
    {                                                                                 \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3,         \
               a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                     \
        double b = seed * 0.5 + 0.25;                                                 \
        int iv = (int)seed;                                                           \
        for (int it = 0; it < ITERS; ++it) {                                          \
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { body }              \
        }                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + iv; \
    }

#define OP8(INS) INS(a0) INS(a1) INS(a2) INS(a3) INS(a4) INS(a5) INS(a6) INS(a7)

#define FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define RCP(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define RSQ(x) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
#define SQRT(x) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));
#define RNDNE(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));
#define LDEXP(x) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(iv));
#define CVT(x) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(iv) : "v"(x));
#define DIVSCALE(x) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(b) : "vcc");
#define DIVFMAS(x) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(x) : "v"(b) : "vcc");
#define DIVFIXUP(x) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define MAXF(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define CMP(x) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define FRACT(x) asm volatile("v_fract_f64 %0, %0" : "+v"(x));
#define FREXPM(x) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x));
#define TRIGPRE(x) asm volatile("v_trig_preop_f64 %0, %0, %1" : "+v"(x) : "v"(iv));
#define ADDU32(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv) : "v"(iv));
#define FMA32(x) { float f = (float)x; asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); x = f; }

#define FMAS(x) asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(x) : "s"(sb));
#define FMACS(x) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x) : "s"(sb), "v"(b));
#define MOVS(x) { int lo_; asm volatile("v_mov_b32 %0, %1" : "=v"(lo_) : "s"(si)); iv += lo_; }
KERNEL(k_fma, OP8(FMA))
__global__ void k_fma_sgpr(double *out, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double sb = __builtin_amdgcn_readfirstlane((int)seed) * 0.5 + 0.25;
    sb = __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(sb) >> 32)) << 32) |
                              (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(sb)));
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) { OP8(FMAS) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void k_fmac_sgpr(double *out, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = seed * 0.5 + 0.25;
    double sb = b;
    sb = __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(sb) >> 32)) << 32) |
                              (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(sb)));
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) { OP8(FMACS) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
KERNEL(k_mul, OP8(MUL))
KERNEL(k_add, OP8(ADD))
KERNEL(k_rcp, OP8(RCP))
KERNEL(k_rsq, OP8(RSQ))
KERNEL(k_sqrt, OP8(SQRT))
KERNEL(k_rndne, OP8(RNDNE))
KERNEL(k_ldexp, OP8(LDEXP))
KERNEL(k_cvt, OP8(CVT))
KERNEL(k_divscale, OP8(DIVSCALE))
KERNEL(k_divfmas, OP8(DIVFMAS))
KERNEL(k_divfixup, OP8(DIVFIXUP))
KERNEL(k_max, OP8(MAXF))
KERNEL(k_cmp, OP8(CMP))
KERNEL(k_fract, OP8(FRACT))
KERNEL(k_frexpm, OP8(FREXPM))
KERNEL(k_addu32, OP8(ADDU32))

template <class K>
void run(const char *name, K kern, double *d_out)
{
    const int blocks = 256 * 4, threads = 256;  // 4 waves per SIMD on every CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    // wave-instructions per SIMD: blocks*threads/64 waves / (256 CUs * 4 SIMD) * REP*ITERS
    const double waves_per_simd = (double)blocks * threads / 64 / (256.0 * 4);
    const double instr = waves_per_simd * REP * ITERS;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-12s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at 2.4 GHz nominal)\n", name, ms, cycles / instr);
}

int main()
{
    double *d_out; hipMalloc(&d_out, sizeof(double) * 256 * 4 * 256);
    run("v_fma_f64", k_fma, d_out);
    run("v_fma_f64 sgpr", k_fma_sgpr, d_out);
    run("v_fmac_f64 sgpr", k_fmac_sgpr, d_out);
    run("v_mul_f64", k_mul, d_out);
    run("v_add_f64", k_add, d_out);
    run("v_max_f64", k_max, d_out);
    run("v_cmp_f64", k_cmp, d_out);
    run("v_rcp_f64", k_rcp, d_out);
    run("v_rsq_f64", k_rsq, d_out);
    run("v_sqrt_f64", k_sqrt, d_out);
    run("v_rndne_f64", k_rndne, d_out);
    run("v_fract_f64", k_fract, d_out);
    run("v_frexp_mant", k_frexpm, d_out);
    run("v_ldexp_f64", k_ldexp, d_out);
    run("v_cvt_i32_f64", k_cvt, d_out);
    run("v_div_scale", k_divscale, d_out);
    run("v_div_fmas", k_divfmas, d_out);
    run("v_div_fixup", k_divfixup, d_out);
    run("v_add_u32", k_addu32, d_out);
    return 0;
}

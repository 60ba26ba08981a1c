// What rate does a stream of INDEPENDENT fp64 FMAs -- nothing to wait for, no memory -- actually reach on this chip,
// cold and sustained, and at what shader clock?  The compute-bound log-probability kernels all end near
// 4.7-5.0e11 VALU wave-instructions per second (0.77-0.81 of 1024 SIMDs x 2.4 GHz / 4), whatever their dependency
// structure, occupancy or scalar traffic; this measures the ceiling they should be held to.
// Every wave runs `iters` rounds of 32 FMAs on 8 independent accumulators (fixed registers, operands with full mantissas: a stream of small integers draws less power); wave 0 of every
// workgroup brackets its loop with s_memtime (shader clock) and s_memrealtime (constant 100 MHz): the clock the
// wave itself saw.  Launched back to back for `seconds`; the first and the last launches are reported.
//   hipcc -O3 --offload-arch=gfx950 -w -o fp64_stream_ceiling fp64_stream_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define BODY8(FMT)                                  \
    FMT("v[20:21]") FMT("v[22:23]") FMT("v[24:25]") FMT("v[26:27]") \
    FMT("v[28:29]") FMT("v[30:31]") FMT("v[32:33]") FMT("v[34:35]")
#define CLOBBER "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"

#define KERNEL(name, INS)                                                                                  \
    __global__ void name(double *out, long long *clk, int iters, double seed)                              \
    {                                                                                                      \
        /* operands with full mantissas (seed, a, m arrive as data): d = d * m + a keeps every bit toggling */ \
        asm volatile("v_mov_b64 v[40:41], %0\n v_mov_b64 v[42:43], %1\n v_mov_b64 v[44:45], %2\n s_mov_b64 s[20:21], %3\n" \
                     "v_mov_b64 v[20:21], v[40:41]\n v_mov_b64 v[22:23], v[42:43]\n v_mov_b64 v[24:25], v[40:41]\n" \
                     "v_mov_b64 v[26:27], v[42:43]\n v_mov_b64 v[28:29], v[40:41]\n v_mov_b64 v[30:31], v[42:43]\n" \
                     "v_mov_b64 v[32:33], v[40:41]\n v_mov_b64 v[34:35], v[42:43]\n"                    \
                     :: "v"(seed * (1.0 + 1e-9 * threadIdx.x)), "v"(0.7853981633974483 + 1e-7 * threadIdx.x), "v"(1.0000002718281828), "s"(1.0000003141592653) \
                     : CLOBBER, "v40","v41","v42","v43","v44","v45","s20","s21");        \
        const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();                           \
        for (int it = 0; it < iters; ++it)                                                                 \
            asm volatile(BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) ::: CLOBBER);                         \
        const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();                           \
        double r;                                                                                          \
        asm volatile("v_add_f64 %0, v[20:21], v[34:35]" : "=v"(r));                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r + seed;                                             \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }        \
    }

#define F_VSV(d) "v_fma_f64 " d ", " d ", s[20:21], v[42:43]\n"
#define F_VVV(d) "v_fma_f64 " d ", " d ", v[44:45], v[42:43]\n"
#define M_VS(d) "v_mul_f64 " d ", " d ", s[20:21]\n"
KERNEL(k_fma_vsv, F_VSV)
KERNEL(k_fma_vvv, F_VVV)
KERNEL(k_mul_vs, M_VS)

template <class K>
void run(const char *name, K kern, int waves_per_simd, int iters, double seconds, double *d_out, long long *d_clk)
{
    const int threads = 256, blocks = 256 * waves_per_simd;
    hipEvent_t e[3];
    for (auto &x : e) hipEventCreate(&x);
    std::vector<long long> clk(2 * blocks);
    auto one = [&](double &ms, double &ghz) {
        hipEventRecord(e[0]);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, d_clk, iters, 1.2345678901234567);
        hipEventRecord(e[1]);
        hipEventSynchronize(e[1]);
        float f; hipEventElapsedTime(&f, e[0], e[1]); ms = f;
        hipMemcpy(clk.data(), d_clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
        double lo = 1e9, hi = 0, sum = 0;
        for (int b = 0; b < blocks; ++b) { const double g = clk[2 * b] / (clk[2 * b + 1] * 10.0); lo = std::min(lo, g); hi = std::max(hi, g); sum += g; }
        ghz = sum / blocks;
    };
    const double instr_per_simd = (double)waves_per_simd * 32.0 * iters;
    double ms, ghz;
    one(ms, ghz);                                       // after an idle moment
    printf("%-22s %d waves/SIMD, %7d rounds: first launch %8.3f ms  %5.2f cycles/instr at 2.4 nominal, wave clock %.3f GHz -> %4.2f cycles/instr at that clock\n",
           name, waves_per_simd, iters, ms, ms * 1e-3 * 2.4e9 / instr_per_simd, ghz, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
    // back to back for `seconds`, then one more measured launch straight after
    hipEventRecord(e[0]);
    int n = 0;
    double total = 0;
    while (total < seconds * 1e3) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, d_clk, iters, 1.2345678901234567);
        n += 20;
        hipEventRecord(e[2]); hipEventSynchronize(e[2]);
        float f; hipEventElapsedTime(&f, e[0], e[2]); total = f;
    }
    one(ms, ghz);
    const double mean_ms = total / n;
    printf("%-22s                       %5d launches back to back: mean %8.3f ms  %5.2f cycles/instr at 2.4 nominal = %5.3f of the nominal issue peak (%.3g wave-instr/s); the launch after them: %5.2f, wave clock %.3f GHz -> %4.2f at that clock\n",
           "", n, mean_ms, mean_ms * 1e-3 * 2.4e9 / instr_per_simd, instr_per_simd / (mean_ms * 1e-3 * 2.4e9) * 4.0,
           instr_per_simd * 1024.0 / (mean_ms * 1e-3), ms * 1e-3 * 2.4e9 / instr_per_simd, ghz, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 0.5;
    double *d_out; hipMalloc(&d_out, sizeof(double) * 256 * 8 * 256);
    long long *d_clk; hipMalloc(&d_clk, sizeof(long long) * 2 * 256 * 8);
    for (int waves : {4, 8})
        for (int iters : {400, 4000}) {
            run("fma d,d,s,v", k_fma_vsv, waves, iters, seconds, d_out, d_clk);
            run("fma d,d,v,v", k_fma_vvv, waves, iters, seconds, d_out, d_clk);
            run("mul d,d,s", k_mul_vs, waves, iters, seconds, d_out, d_clk);
        }
    return 0;
}

// Does a fp64 FMA cost more issue time when all THREE of its sources are vector registers, and does it matter in
// which register-file banks they lie?  (Dias2000's frequency loop is made of such FMAs and runs at 5.4-5.7 cycles
// per VALU instruction per SIMD where its siblings, whose FMAs take one source from scalar registers, run at
// 4.3-4.8: DESIGN.md section 3.3.)  Every kernel runs a long unrolled sequence of 8 independent accumulators
// (v[20:35]) with FIXED registers, 4 waves per SIMD on every CU; cycles per wave-instruction per SIMD at 2.4 GHz nominal.
//   hipcc -O3 --offload-arch=gfx950 -w -o fma_operands fma_operands.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 400
// 8 accumulators d = v[20:21] ... v[34:35]; sources named by the caller
#define BODY8(FMT)                                  \
    FMT("v[20:21]") FMT("v[22:23]") FMT("v[24:25]") FMT("v[26:27]") \
    FMT("v[28:29]") FMT("v[30:31]") FMT("v[32:33]") FMT("v[34:35]")

#define KERNEL(name, INS)                                                                                  \
    __global__ void name(double *out, double seed)                                                         \
    {                                                                                                      \
        asm volatile("v_cvt_f64_i32 v[40:41], %0\n v_mov_b64 v[42:43], v[40:41]\n v_mov_b64 v[44:45], v[40:41]\n" \
                     "v_mov_b64 v[46:47], v[40:41]\n v_mov_b64 v[48:49], v[40:41]\n"                    \
                     "v_mov_b64 v[20:21], v[40:41]\n v_mov_b64 v[22:23], v[40:41]\n v_mov_b64 v[24:25], v[40:41]\n" \
                     "v_mov_b64 v[26:27], v[40:41]\n v_mov_b64 v[28:29], v[40:41]\n v_mov_b64 v[30:31], v[40:41]\n" \
                     "v_mov_b64 v[32:33], v[40:41]\n v_mov_b64 v[34:35], v[40:41]\n"                    \
                     :: "v"((int)threadIdx.x & 3)                                                         \
                     : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35", \
                       "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49");                     \
        for (int it = 0; it < ITERS; ++it) {                                                               \
            asm volatile(BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS)                                       \
                         ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"); \
        }                                                                                                  \
        double r;                                                                                          \
        asm volatile("v_add_f64 %0, v[20:21], v[34:35]" : "=v"(r));                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r + seed;                                             \
    }

// d = d * b + c with b, c vector registers (64-bit tuples are even-aligned on gfx950: low dwords in bank 0 or 2 of 4)
#define F_VVV_SPREAD(d) "v_fma_f64 " d ", " d ", v[42:43], v[44:45]\n"
// ... all three low dwords in the SAME bank (d = 20, 24, ... and 40, 44, 48 are bank 0 for half of the accumulators)
#define F_VVV_SAME(d) "v_fma_f64 " d ", " d ", v[44:45], v[48:49]\n"
// d = d * b + b (two distinct vector registers)
#define F_VV(d) "v_fma_f64 " d ", " d ", v[42:43], v[42:43]\n"
// d = d * s + c (one source scalar)
#define F_VSV(d) "v_fma_f64 " d ", " d ", s[20:21], v[42:43]\n"
// d = d * s + s' is not encodable (one scalar source per VOP3 on gfx9); d = d * d + s
#define F_VVS(d) "v_fma_f64 " d ", " d ", " d ", s[20:21]\n"
#define M_VV(d) "v_mul_f64 " d ", " d ", v[42:43]\n"
#define M_VS(d) "v_mul_f64 " d ", " d ", s[20:21]\n"

KERNEL(k_fma_vvv_spread, F_VVV_SPREAD)
KERNEL(k_fma_vvv_same, F_VVV_SAME)
KERNEL(k_fma_vv, F_VV)
KERNEL(k_fma_vsv, F_VSV)
KERNEL(k_fma_vvs, F_VVS)
KERNEL(k_mul_vv, M_VV)
KERNEL(k_mul_vs, M_VS)

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
    const double waves_per_simd = (double)blocks * threads / 64 / (256.0 * 4);
    const double instr = waves_per_simd * 32.0 * ITERS;
    printf("%-34s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at 2.4 GHz nominal)\n", name, ms, ms * 1e-3 * 2.4e9 / instr);
}

int main()
{
    double *d_out; hipMalloc(&d_out, sizeof(double) * 256 * 4 * 256);
    run("fma d,d,v,v  (3 vector, sources in banks 2, 0)", k_fma_vvv_spread, d_out);
    run("fma d,d,v,v  (3 vector, sources both bank 0)", k_fma_vvv_same, d_out);
    run("fma d,d,v,v' (v = v': 2 vector)", k_fma_vv, d_out);
    run("fma d,d,s,v  (1 scalar source)", k_fma_vsv, d_out);
    run("fma d,d,d,s  (1 vector register)", k_fma_vvs, d_out);
    run("mul d,d,v", k_mul_vv, d_out);
    run("mul d,d,s", k_mul_vs, d_out);
    return 0;
}

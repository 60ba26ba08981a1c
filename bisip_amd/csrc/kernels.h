// kernels.h -- CDNA4 (gfx950) device code of the log-probability hot path.
//
// Mapping: ONE LANE PER WALKER.  A workgroup takes a contiguous block
// of BLK walkers; their theta rows (row-major (W,ndim), the layout emcee hands over)
// are fetched with fully coalesced 16-byte loads into LDS and read back one row per
// lane (a transposition, not a reuse buffer).  Everything that does not depend on
// the walker -- the frequency array, the measured spectrum, 1/sigma^2, the
// pre-reduced model operands -- is wave-uniform, so it is read through the scalar
// cache into SGPRs (s_load) and costs no vector registers, no LDS bandwidth and no
// cross-lane traffic.  With one walker per lane there is no cross-lane reduction
// at all: the sum over frequencies is a sequential in-register accumulation, so a
// walker's result does not depend on where it sits in the batch.
//
// All arithmetic is IEEE binary64; no fast-math, no MFMA (elementwise complex
// arithmetic).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "../../include/bisip_hip.h"

namespace bisip {

constexpr int MAXD = BISIP_MAX_NDIM;

typedef double dbl2 __attribute__((ext_vector_type(2)));

struct Bounds {
    double lo[MAXD];
    double hi[MAXD];
    // BOUNDS_FAST: set by the host (bound_flags in bisip_hip.hip) when, everywhere inside this box, the
    // denominators of one frequency of ColeCole<D> / Shin stay within [1, 2^225] -- their product over a
    // group of <= 4 neither overflows nor underflows, so ONE reciprocal serves the group (rcp_batch_n) --
    // and no exponent needs clamping.  Wave-uniform (kernel argument): logprob_row picks, once per row,
    // between two instantiations of the frequency loop (FAST / safe); the loop itself has no branch.
    // Dias: the bit says that the product of two frequencies' denominators is a normal number everywhere in
    // the box, so frequencies 2k and 2k+1 share one reciprocal.
    int flags = 0;
};
constexpr int BOUNDS_FAST = 1;
// BOUNDS_GRID (only together with BOUNDS_FAST): the frequencies of at least one spectrum of the context lie on a
// geometric grid, ln w_{16k+q} = ln w_{16k} + q * dlnw (q < 16) to 4e-15 (grid_step in host_precompute.cpp; rec[7] of
// every record of a spectrum holds ITS dlnw, 0 for a spectrum on no grid: logprob_row looks there).  The exponentials of the per-frequency models are then taken once per block of
// GRID_BLOCK = 16 frequencies and stepped by multiplication (GridSteps below).
constexpr int BOUNDS_GRID = 2;
constexpr int GRID_MAX_TERMS = 3;     // ColeCole with up to three modes, Shin (two elements)
// The safe loop clamps the exponent y of 2^y = |(i w tau)^c| (ColeCole) or Q w^n (Shin) at 500: beyond,
// 2^y squared would overflow, the denominator become inf and rcp_nr(inf) NaN where the reference's term
// quietly vanishes (a round-2 defect that boxes with c up to 22 and more exposed).  It takes one
// reciprocal per term.  Only boxes a user widened far beyond the reference's run it.
constexpr double EXP2_CLAMP = 500.0;

struct LaunchArgs {
    const double *__restrict__ theta;  // (W, NDIM) row-major
    double *__restrict__ out;          // (W,) log-probabilities   | (W,2,N) forward
    long long W;
    const double *__restrict__ cb;     // per-frequency records (wave-uniform reads)
    int N;
    double lconst;                     // -sum ln sigma^2
    Bounds b;
    // forward kernels on a batch of spectra: rows [e*Wp, (e+1)*Wp) read the records of
    // spectrum e at cb + e*cb_stride (Wp a multiple of 64, so a 64-walker block never
    // straddles two spectra and the record pointer stays wave-uniform).  Wp = 0: one spectrum.
    long long Wp = 0, cb_stride = 0;
};

// ---------------------------------------------------------------------------------
// theta staging: BLK rows, coalesced global -> LDS, then one row per lane.
// ---------------------------------------------------------------------------------
// ROWS rows are staged by THREADS lanes (THREADS == ROWS unless a lane owns several rows)
template <int NDIM, int ROWS, bool VEC, int THREADS = ROWS>
__device__ __forceinline__ void stage_theta(const double *__restrict__ theta, long long W,
                                            long long row0, double *lds)
{
    constexpr int BLK = THREADS;
    constexpr int CHUNK = ROWS * NDIM;  // doubles per block
    const int t = threadIdx.x;
    const long long base = row0 * NDIM;
    const long long avail = (W - row0) * (long long)NDIM;  // doubles left from row0
    if constexpr (VEC && (CHUNK % 2 == 0)) {
        const dbl2 *__restrict__ src = reinterpret_cast<const dbl2 *>(theta + base);
        dbl2 *dst = reinterpret_cast<dbl2 *>(lds);
        constexpr int N2 = CHUNK / 2;
        constexpr int ROUNDS = (N2 + BLK - 1) / BLK;
        if (avail >= CHUNK) {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int i = r * BLK + t;
                if ((r + 1) * BLK <= N2 || i < N2) dst[i] = __builtin_nontemporal_load(src + i);
            }
        } else {
            for (int i = t; i < N2; i += BLK) {
                if (2LL * i + 1 < avail) dst[i] = src[i];
                else if (2LL * i < avail) lds[2 * i] = theta[base + 2 * i];
            }
        }
    } else {
        if (avail >= CHUNK) {
#pragma unroll
            for (int r = 0; r < (CHUNK + BLK - 1) / BLK; ++r) {
                const int i = r * BLK + t;
                if ((r + 1) * BLK <= CHUNK || i < CHUNK) lds[i] = __builtin_nontemporal_load(theta + base + i);
            }
        } else {
            for (int i = t; i < CHUNK; i += BLK)
                if (i < avail) lds[i] = theta[base + i];
        }
    }
}

// strict open box, NaN -> false   (reference src/bisip/models.py:64-69)
template <int NDIM>
__device__ __forceinline__ bool in_prior(const double (&th)[NDIM], const Bounds &b)
{
    bool ok = true;
#pragma unroll
    for (int q = 0; q < NDIM; ++q) ok = ok && (b.lo[q] < th[q]) && (th[q] < b.hi[q]);
    return ok;
}

// ---------------------------------------------------------------------------------
// fp64 building blocks for the transcendental-bound models.  On gfx950 every fp64 VALU
// op issues in ~4 cycles per wave except v_rcp/v_rsq/v_sqrt (~16; measured with
// benchmarks/micro/valu_rates.hip), so the cost of a kernel is its instruction count.
// ---------------------------------------------------------------------------------

// exp(x) for finite x: Cody-Waite reduction by ln2 (hi/lo split), degree-11 polynomial
// (coefficients: Chebyshev fit of (e^r-1-r)/r^2 on |r| <= ln2/2, max error 0.83 ulp
// against a 50-digit reference), scaling by v_ldexp (saturates to +inf / 0 by itself).
// Same structure as the device library's exp minus its overflow/underflow selects,
// which the callers' argument ranges make dead code.
__device__ __forceinline__ double exp_finite(double x)
{
    const double t = rint(x * 0x1.71547652b82fep+0);
    double r = fma(t, -0x1.62e42fefa39efp-1, x);
    r = fma(t, -0x1.abc9e3b39803fp-56, r);
    double p = 0x1.af38a9b0ec855p-26;
    p = fma(p, r, 0x1.289185613a3d6p-22);
    p = fma(p, r, 0x1.71de0dae63bb3p-19);
    p = fma(p, r, 0x1.a019b90d2ae7ap-16);
    p = fma(p, r, 0x1.a01a01a7c41d5p-13);
    p = fma(p, r, 0x1.6c16c1788bd9p-10);
    p = fma(p, r, 0x1.11111111109b3p-7);
    p = fma(p, r, 0x1.5555555553d63p-5);
    p = fma(p, r, 0x1.5555555555556p-3);
    p = fma(p, r, 0x1.0000000000001p-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)t);
}

// 2^y for finite y: t = rint(y), f = y - t is exact, degree-11 polynomial in f (Chebyshev
// fit of (2^f - 1 - f ln2)/f^2 on |f| <= 1/2, max error 0.89 ulp).  The per-frequency
// kernels fold log2(e) into per-walker constants so that exp(c*(ln w + lt)) is
// exp2_finite(fma(c*log2e, ln w, c*log2e*lt)): no range-reduction multiplies at all.
__device__ __forceinline__ double exp2_finite(double y)
{
    const double t = rint(y);
    const double f = y - t;
    double p = 0x1.e9bbe9e45e5a6p-32;
    p = fma(p, f, 0x1.e5ea03c5ae3ccp-28);
    p = fma(p, f, 0x1.b525087314718p-24);
    p = fma(p, f, 0x1.62bfe45ac08ccp-20);
    p = fma(p, f, 0x1.ffcbfc61f8673p-17);
    p = fma(p, f, 0x1.4309130379f8bp-13);
    p = fma(p, f, 0x1.5d87fe78a5dc3p-10);
    p = fma(p, f, 0x1.3b2ab6fba385bp-7);
    p = fma(p, f, 0x1.c6b08d704a0c0p-5);
    p = fma(p, f, 0x1.ebfbdff82c590p-3);
    p = fma(p, f, 0x1.62e42fefa39efp-1);
    p = fma(p, f, 1.0);
    return ldexp(p, (int)t);
}

constexpr double LOG2E = 0x1.71547652b82fep+0;

// 1/x for normal x with a normal reciprocal: hardware estimate r (2^-24.4 on gfx950) and ONE cubic step,
// 1/x = r (1 + e + e^2 + ...) with e = 1 - x r: three FMAs, the neglected e^3 is 1e-22.  <= 1.00 ulp over 4M
// doubles, the same as two Newton steps (four FMAs, a chain one longer): benchmarks/micro/rcp_accuracy.hip.
// 4 instructions instead of the 10 of an IEEE division.  Every caller's argument is |1 + z|^2-like and
// bounded away from 0 (see the model comments).
__device__ __forceinline__ double rcp_nr(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

// An FMA / product / sum whose LAST operand is wave-uniform and stays in scalar registers.  Left to itself
// the compiler turns p = fma(p, t, CONSTANT) into two v_mov_b32 (the constant into vector registers) and a
// v_fmac: three vector instructions per Horner step in straight-line code (inside loops it hoists the
// constants into scalar registers by itself).  Same operation, same bits.
__device__ __forceinline__ double fma_s(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ double mul_s(double a, double c)
{
    double d;
    asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(c));
    return d;
}
__device__ __forceinline__ double add_s(double a, double c)
{
    double d;
    asm("v_add_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(c));
    return d;
}

// sin(c pi/2), cos(c pi/2) for c in [0, 1] -- the Cole-Cole exponent / the CPE exponent inside a prior box that
// BOUNDS_FAST vouches for.  The general sincospi spends 85 instructions per call on range reduction, special
// values and constants moved into vector registers; here x = c/2 lies in [0, 1/2], y = min(x, 1/2 - x) (exact)
// in [0, 1/4], and
//   sin(pi y) = pi y + y^3 S(y^2),   cos(pi y) = 1 + y^2 C(y^2),
// S of degree 5 and C of degree 6 (Chebyshev fits of (sin(pi y)/y - pi)/y^2 and (cos(pi y) - 1)/y^2 on
// y^2 in [0, 1/16], 50-digit arithmetic: truncation 4e-17 and 1e-19), 27 instructions.  Against long double
// sin / cos on 2e7 arguments (dense near c = 0 and c = 1, where cos(c pi/2) -> 0 keeps its RELATIVE accuracy
// because 1/2 - x is exact): 1.5 ulp at worst for both.
__device__ __forceinline__ void sincos_unit(double c, double *sn, double *cs)
{
    const double x = 0.5 * c;
    const bool upper = x > 0.25;
    const double y = upper ? 0.5 - x : x;
    const double t = y * y;
    double ps = add_s(mul_s(t, 0x1.e4a9d9166f052p-12), -0x1.e3027dea82bd7p-8);
    ps = fma_s(ps, t, 0x1.50783208843ebp-4);
    ps = fma_s(ps, t, -0x1.32d2cce500387p-1);
    ps = fma_s(ps, t, 0x1.466bc6775a476p+1);
    ps = fma_s(ps, t, -0x1.4abbce625be52p+2);
    double pc = add_s(mul_s(t, -0x1.b2f3eb054afcdp-14), 0x1.f9ce245cada0bp-10);
    pc = fma_s(pc, t, -0x1.a6d1eef479be1p-6);
    pc = fma_s(pc, t, 0x1.e1f5068688d5bp-3);
    pc = fma_s(pc, t, -0x1.55d3c7e3cb241p+0);
    pc = fma_s(pc, t, 0x1.03c1f081b5ac0p+2);
    pc = fma_s(pc, t, -0x1.3bd3cc9be45dep+2);
    const double s = fma(0x1.921fb54442d18p+1, y, (y * t) * ps);
    const double k = fma(t, pc, 1.0);
    *sn = upper ? k : s;
    *cs = upper ? s : k;
}

// K independent values in lockstep: every Horner / Newton step is issued for all K before the
// next step, so a wave has K independent dependency chains in flight.  One fp64 VALU op takes
// ~8.3 cycles to its dependent successor but a SIMD can start one every ~4.5 cycles: with ONE
// wave per SIMD -- every sampler half-step of <= 65,536 proposals, every emcee-sized call -- a
// single chain runs at half the issue rate (benchmarks/micro/issue_latency.hip: 8.3 / 6.5 / 5.4
// cycles per instruction at 1 / 2 / 4 chains).  Same operations per value, so the same bits.
template <int K>
__device__ __forceinline__ void exp2_finite_n(const double (&y)[K], double (&out)[K])
{
    double t[K], f[K], p[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { t[k] = rint(y[k]); f[k] = y[k] - t[k]; p[k] = 0x1.e9bbe9e45e5a6p-32; }
#define BISIP_HORNER(c) _Pragma("unroll") for (int k = 0; k < K; ++k) p[k] = fma(p[k], f[k], c);
    BISIP_HORNER(0x1.e5ea03c5ae3ccp-28) BISIP_HORNER(0x1.b525087314718p-24) BISIP_HORNER(0x1.62bfe45ac08ccp-20)
    BISIP_HORNER(0x1.ffcbfc61f8673p-17) BISIP_HORNER(0x1.4309130379f8bp-13) BISIP_HORNER(0x1.5d87fe78a5dc3p-10)
    BISIP_HORNER(0x1.3b2ab6fba385bp-7) BISIP_HORNER(0x1.c6b08d704a0c0p-5) BISIP_HORNER(0x1.ebfbdff82c590p-3)
    BISIP_HORNER(0x1.62e42fefa39efp-1) BISIP_HORNER(1.0)
#undef BISIP_HORNER
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = ldexp(p[k], (int)t[k]);
}

template <int K>
__device__ __forceinline__ void rcp_nr_n(const double (&x)[K], double (&r)[K])
{
    double e[K];
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = __builtin_amdgcn_rcp(x[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) e[k] = fma(-x[k], r[k], 1.0);
#pragma unroll
    for (int k = 0; k < K; ++k) e[k] = fma(e[k], e[k], e[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = fma(r[k], e[k], r[k]);
}

// 1/x_k for K values from ONE reciprocal of their product (Montgomery's trick): 3(K-1)
// multiplications + rcp_nr instead of K rcp_nr.  v_rcp_f64 issues at a quarter of the FMA rate, so a
// reciprocal costs 7 issue slots and a pair by this route 10 instead of 14.  F independent groups
// (one per frequency) run in lockstep.  The caller guarantees that products of K arguments stay
// normal (BOUNDS_FAST).  These groups are formed WITHIN one frequency (three modes and more, and a last
// unpaired frequency of the paired models; rcp_joint below takes the pairs), the same in every kernel, so a
// walker's value does not depend on which kernel evaluated it.
template <int F, int K>
__device__ __forceinline__ void rcp_batch_n(const double (&x)[F][K], double (&r)[F][K])
{
    if constexpr (K == 1) {
        double xx[F], rr[F];
#pragma unroll
        for (int f = 0; f < F; ++f) xx[f] = x[f][0];
        rcp_nr_n<F>(xx, rr);
#pragma unroll
        for (int f = 0; f < F; ++f) r[f][0] = rr[f];
    } else {
        double pre[F][K], top[F], inv[F];
#pragma unroll
        for (int f = 0; f < F; ++f) pre[f][0] = x[f][0];
#pragma unroll
        for (int k = 1; k < K; ++k)
#pragma unroll
            for (int f = 0; f < F; ++f) pre[f][k] = pre[f][k - 1] * x[f][k];
#pragma unroll
        for (int f = 0; f < F; ++f) top[f] = pre[f][K - 1];
        rcp_nr_n<F>(top, inv);
#pragma unroll
        for (int k = K - 1; k >= 1; --k)
#pragma unroll
            for (int f = 0; f < F; ++f) { r[f][k] = inv[f] * pre[f][k - 1]; inv[f] = inv[f] * x[f][k]; }
#pragma unroll
        for (int f = 0; f < F; ++f) r[f][0] = inv[f];
    }
}

// D reciprocals per frequency in groups of at most 4 (the product of 4 denominators <= 2^225 is normal)
template <int F, int D>
__device__ __forceinline__ void rcp_groups(const double (&x)[F][D], double (&r)[F][D])
{
    if constexpr (D <= 4) rcp_batch_n<F, D>(x, r);
    else {
        constexpr int H = D - 4;
        double xa[F][4], ra[F][4], xb[F][H], rb[F][H];
#pragma unroll
        for (int f = 0; f < F; ++f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) xa[f][k] = x[f][k];
#pragma unroll
            for (int k = 0; k < H; ++k) xb[f][k] = x[f][4 + k];
        }
        rcp_batch_n<F, 4>(xa, ra);
        rcp_groups<F, H>(xb, rb);
#pragma unroll
        for (int f = 0; f < F; ++f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[f][k] = ra[f][k];
#pragma unroll
            for (int k = 0; k < H; ++k) r[f][4 + k] = rb[f][k];
        }
    }
}

// The denominators of TWO frequencies (2k, 2k+1) from ONE reciprocal (models with PAIRED, FAST loops only):
// K = 2 (one denominator per frequency) or 4 (two per frequency, values k = (frequency k / 2, term k % 2)).
// Four go as a tree -- (x0 x1)(x2 x3), one reciprocal, two half products back, four factors -- so that the
// dependent chain is 2 + 4 + 2 instructions instead of the 3 + 4 + 3 of the linear form (same nine
// multiplications): 13 instructions and 16 issue slots for what two per-frequency groups do in 14 and 20;
// K = 2: 7 and 10 instead of 8 and 14.  BOUNDS_FAST bounds every denominator by [1, 2^225]: a product of four
// is a normal number.  The pairs are (2k, 2k+1) in every kernel and a last unpaired frequency keeps its
// per-frequency group: a walker's value does not depend on which kernel evaluated it.
template <int K>
__device__ __forceinline__ void rcp_joint(const double (&x)[K], double (&r)[K])
{
    static_assert(K == 2 || K == 4, "one or two denominators per frequency");
    if constexpr (K == 2) {
        const double t = rcp_nr(x[0] * x[1]);
        r[0] = t * x[1];
        r[1] = t * x[0];
    } else {
        const double p01 = x[0] * x[1], p23 = x[2] * x[3];
        const double t = rcp_nr(p01 * p23);
        const double i01 = t * p23, i23 = t * p01;
        r[0] = i01 * x[1];
        r[1] = i01 * x[0];
        r[2] = i23 * x[3];
        r[3] = i23 * x[2];
    }
}

// ---------------------------------------------------------------------------------
// Forward models.  Each exposes
//   NDIM, REC (doubles per frequency record; rec[0..3] = y_re, y_im, 1/s2_re, 1/s2_im)
//   Setup / setup(theta row)           per-walker quantities, computed once per lane
//   eval(setup, rec+4, zr, zi)         complex impedance at one frequency (forward())
//   residual(setup, rec, rr, ri)       y - Z at one frequency for the log-likelihood, with
//                                      the walker-constant part of Z folded into per-walker
//                                      constants (fewer instructions than y - eval())
//   residual2(setup, recA, recB, ...)  (8-double-record models) the same for TWO frequencies with
//                                      their dependency chains interleaved (exp2_finite_n);
//                                      identical values; used where one wave per SIMD runs alone
// ---------------------------------------------------------------------------------

// PolynomialDecomposition, collapsed: Z_j = R0*(1 - sum_p a_p G[j,p]).
// G[j,p] = sum_k log_taus[p,k]*K[j,k] is walker-independent because c_exp and the tau
// grid are fixed per model (reference src/bisip/models.py:195-209, cython_funcs.pyx:84-93).
template <int P>
struct PDCollapsed {
    static constexpr int NDIM = P + 2;
    static constexpr int REC = 4 + 2 * (P + 1);
    struct Setup {
        double r0;
        double a[P + 1];
        double b[P + 1];  // r0 * a_p
    };
    __device__ static __forceinline__ Setup setup(const double (&th)[NDIM], const bool = false)
    {
        Setup s;
        s.r0 = th[0];
#pragma unroll
        for (int p = 0; p <= P; ++p) {
            s.a[p] = th[1 + p];  // ascending a0..aP
            s.b[p] = th[0] * th[1 + p];
        }
        return s;
    }
    static constexpr bool HAS_FAST = false;
    static constexpr bool HAS_GRID = false;
    static constexpr bool PAIRED = false;
    // Log-prob records are pre-weighted by 1/sigma (rows of the weighted design matrix):
    //   rec = ys_re, ys_im, -s_re, pad | s_re*G_re[0..P] | s_im*G_im[0..P]
    // so (y - Z)/sigma = ys + r0*(-s) + sum_p b_p (s G_p): 2(P+1)+1 FMAs, and the caller
    // squares without a further multiply.
    static constexpr bool WEIGHTED = true;
    template <bool FAST = false>
    __device__ static __forceinline__ void residual(const Setup &s, const double *__restrict__ rec,
                                                    double &rr, double &ri)
    {
        rr = fma(s.r0, rec[2], rec[0]);
        ri = rec[1];
#pragma unroll
        for (int p = 0; p <= P; ++p) {
            rr = fma(s.b[p], rec[4 + p], rr);
            ri = fma(s.b[p], rec[4 + P + 1 + p], ri);
        }
    }
    // Z = r0 - sum_p b_p G_p  (m = unweighted G_re[0..P], G_im[0..P])
    __device__ static __forceinline__ void eval(const Setup &s, const double *__restrict__ m,
                                                double &zr, double &zi)
    {
        zr = s.r0;
        zi = 0.0;
#pragma unroll
        for (int p = 0; p <= P; ++p) {
            zr = fma(-s.b[p], m[p], zr);
            zi = fma(-s.b[p], m[P + 1 + p], zi);
        }
    }
};

// PeltonColeCole with D modes (reference src/bisip/cython_funcs.pyx:33-34, 49-62):
// (i w e^lt)^c = exp(c (ln w + lt)) * (cos(c pi/2) + i sin(c pi/2)).
template <int D>
struct ColeCole {
    static constexpr int NDIM = 1 + 3 * D;
    static constexpr int REC = 8;  // y_re y_im iv_re iv_im | w lnw sqrt(w) pad
    static constexpr bool WEIGHTED = false;
    struct Setup {
        double r0;
        double m[D], lt[D], c[D], cs[D], sn[D];
        double A[D], c2[D], clt2[D], C;  // residual(): m r0, c log2e, c log2e lt, r0 - sum A
        double cs2[D], Acs[D], Asn[D];   // the fast residual: 2 cos, A cos, A sin
    };
    static constexpr bool HAS_FAST = true;
    static constexpr bool HAS_GRID = true;
    // FAST, one or two modes: frequencies 2k and 2k+1 take their D denominators each from ONE reciprocal
    // (rcp_joint); three modes and more keep a group per frequency (six denominators: no normal product)
    static constexpr bool PAIRED = D <= 2;
    // unit (wave-uniform): every c of the row lies in [0, 1] (BOUNDS_FAST and the row inside the prior)
    __device__ static __forceinline__ Setup setup(const double (&th)[NDIM], const bool unit = false)
    {
        Setup s;
        s.r0 = th[0];
        s.C = th[0];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            s.m[i] = th[1 + i];
            s.lt[i] = th[1 + D + i];
            s.c[i] = th[1 + 2 * D + i];
            if (unit) sincos_unit(s.c[i], &s.sn[i], &s.cs[i]);
            else sincospi(0.5 * s.c[i], &s.sn[i], &s.cs[i]);
            s.A[i] = s.m[i] * th[0];
            s.c2[i] = s.c[i] * LOG2E;
            s.clt2[i] = s.c2[i] * s.lt[i];
            s.C -= s.A[i];
            s.cs2[i] = 2.0 * s.cs[i];
            s.Acs[i] = s.A[i] * s.cs[i];
            s.Asn[i] = s.A[i] * s.sn[i];
        }
        return s;
    }
    // Z = (r0 - sum A_i) + sum A_i (1+x_i)^-1  =>  y - Z accumulates -A_i conj(1+x_i)/|1+x_i|^2
    // F frequencies in lockstep (F = 2: the pair (2k, 2k+1), residual2; F = 1: a last unpaired frequency, the
    // safe loop, three modes and more): per frequency the D exponentials, then the D reciprocals -- FAST: one
    // per PAIR of frequencies (PAIRED) or per frequency's group of modes; safe: one per mode, exponents
    // clamped -- then the modes accumulated in ascending order.  Every FAST loop takes the pairs (2k, 2k+1)
    // with F = 2 and only a last unpaired frequency with F = 1; elsewhere the operations per frequency are
    // the same for every F: the same bits whichever kernel evaluates a walker.
    // the D exponentials of one frequency are 2^(exp_a(i) ln w + exp_b(i))   (see GridSteps below)
    static constexpr int NEXP = D;
    __device__ static __forceinline__ double exp_a(const Setup &s, int i) { return s.c2[i]; }
    __device__ static __forceinline__ double exp_b(const Setup &s, int i) { return s.clt2[i]; }
    template <int F, bool FAST>
    __device__ static __forceinline__ void residual_n(const Setup &s, const double *const (&rec)[F],
                                                      double (&rr)[F], double (&ri)[F])
    {
        constexpr int K = F * D;     // value k = (frequency k / D, mode k % D)
        double y[K], e[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            y[k] = fma(s.c2[k % D], rec[k / D][5], s.clt2[k % D]);
            if constexpr (!FAST) y[k] = fmin(y[k], EXP2_CLAMP);
        }
        exp2_finite_n<K>(y, e);
        residual_from_exp<F, FAST>(s, rec, e, rr, ri);
    }
    // everything after the exponentials e[f * D + i] = (w_f tau_i)^c_i
    template <int F, bool FAST>
    __device__ static __forceinline__ void residual_from_exp(const Setup &s, const double *const (&rec)[F],
                                                             const double (&e)[F * D], double (&rr)[F],
                                                             double (&ri)[F])
    {
        constexpr int K = F * D;
        double zr[F], zi[F];
        if constexpr (FAST) {
            // Inside a box BOUNDS_FAST vouches for (0 <= c <= 1: cos >= 0, nothing cancels) the term is taken as
            //   A conj(1+x)/|1+x|^2 = (A + A cos e - i A sin e) / (1 + e (e + 2 cos)),   e = |x|,
            // with A cos, A sin, 2 cos per walker: six instructions per (frequency, mode) besides the reciprocal
            // instead of seven (1 + x itself is never formed; cos^2 + sin^2 = 1 to the rounding of the pair).
            double den[F][D], nr[F][D], ni[F][D], inv[F][D];
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    const double ee = e[f * D + i];
                    den[f][i] = fma(ee, ee + s.cs2[i], 1.0);
                    nr[f][i] = fma(ee, s.Acs[i], s.A[i]);
                    ni[f][i] = ee * s.Asn[i];
                }
            if constexpr (F == 2 && PAIRED) {
                double flat[K], r[K];
#pragma unroll
                for (int k = 0; k < K; ++k) flat[k] = den[k / D][k % D];
                rcp_joint<K>(flat, r);
#pragma unroll
                for (int k = 0; k < K; ++k) inv[k / D][k % D] = r[k];
            } else if constexpr (D >= 2) rcp_groups<F, D>(den, inv);
            else {
                double flat[K], r[K];
#pragma unroll
                for (int k = 0; k < K; ++k) flat[k] = den[k][0];
                rcp_nr_n<K>(flat, r);
#pragma unroll
                for (int k = 0; k < K; ++k) inv[k][0] = r[k];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    zr[f] = fma(nr[f][i], inv[f][i], i == 0 ? s.C : zr[f]);
                    zi[f] = i == 0 ? ni[f][i] * inv[f][i] : fma(ni[f][i], inv[f][i], zi[f]);
                }
            }
        } else {
            double dr[F][D], di[F][D], den[F][D], inv[F][D];
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    dr[f][i] = fma(e[f * D + i], s.cs[i], 1.0);
                    di[f][i] = e[f * D + i] * s.sn[i];
                }
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int i = 0; i < D; ++i) den[f][i] = fma(dr[f][i], dr[f][i], di[f][i] * di[f][i]);
            double flat[K], r[K];
#pragma unroll
            for (int k = 0; k < K; ++k) flat[k] = den[k / D][k % D];
            rcp_nr_n<K>(flat, r);
#pragma unroll
            for (int k = 0; k < K; ++k) inv[k / D][k % D] = r[k];
#pragma unroll
            for (int i = 0; i < D; ++i) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const double t = s.A[i] * inv[f][i];
                    zr[f] = fma(t, dr[f][i], i == 0 ? s.C : zr[f]);
                    zi[f] = i == 0 ? t * di[f][i] : fma(t, di[f][i], zi[f]);
                }
            }
        }
        // Z = C + sum_i A_i conj(1+x_i)/|1+x_i|^2 is accumulated first (modes in ascending order, the real part from
        // C) and taken from the measured value last: y - Z.  With the measured value as the START of the chain
        // the compiler moves it from its scalar registers into vector registers first (two v_mov_b32 per
        // frequency and part for a v_fmac); as the operand of the final add it stays scalar.
#pragma unroll
        for (int f = 0; f < F; ++f) { rr[f] = rec[f][0] - zr[f]; ri[f] = rec[f][1] + zi[f]; }
    }
    template <bool FAST = false>
    __device__ static __forceinline__ void residual(const Setup &s, const double *__restrict__ rec,
                                                    double &rr, double &ri)
    {
        const double *const r1[1] = {rec};
        double a[1], b[1];
        residual_n<1, FAST>(s, r1, a, b);
        rr = a[0];
        ri = b[0];
    }
    template <bool FAST = false>
    __device__ static __forceinline__ void residual2(const Setup &s, const double *__restrict__ ra,
                                                     const double *__restrict__ rb, double (&rr)[2],
                                                     double (&ri)[2])
    {
        const double *const r2[2] = {ra, rb};
        residual_n<2, FAST>(s, r2, rr, ri);
    }
    // Z = (r0 - sum A_i) + sum A_i conj(1+x_i)/|1+x_i|^2   (m = w, ln w, sqrt w)
    __device__ static __forceinline__ void eval(const Setup &s, const double *__restrict__ m,
                                                double &zr, double &zi)
    {
        const double lnw = m[1];
        zr = s.C;
        zi = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double y = fma(s.c2[i], lnw, s.clt2[i]);
            y = y > EXP2_CLAMP ? EXP2_CLAMP : y;     // any theta may be asked of forward(); NaN stays NaN
            const double e = exp2_finite(y);
            const double dr = fma(e, s.cs[i], 1.0);  // 1 + x, >= 1 because cos(c pi/2) >= 0
            const double di = e * s.sn[i];
            const double t = s.A[i] * rcp_nr(fma(dr, dr, di * di));
            zr = fma(t, dr, zr);
            zi = fma(-t, di, zi);
        }
    }
};

// Dias2000 (reference src/bisip/cython_funcs.pyx:36-40, 64-73)
struct Dias {
    static constexpr int NDIM = 5;
    static constexpr int REC = 8;
    static constexpr bool WEIGHTED = false;
    struct Setup {
        double r0, m, tau, taup;
        double A, C, teh, teh2;  // r0 m, r0 - A, tau |eta| / sqrt(2) and its square
    };
    // tau' is clamped at 1e50: delta = 0 or m = 1 (prior bounds; forward() may be asked for them) make it
    // infinite and the reference's complex arithmetic then gives Z = r0 (1 - m).  A huge finite tau' reaches
    // the same limit without inf * 0 (NaN, from 0/0, is kept): beyond 1e50 the term r0 m / den is below
    // 1e-45 in absolute terms.  1e50 (not 1e100) so that the PRODUCT of two frequencies' X^2 + Y^2 stays
    // finite inside the reference's box (the shared reciprocal below; bound_flags in bisip_hip.hip checks it).
    static constexpr double TAUP_MAX = 1e50;
    __device__ static __forceinline__ Setup setup(const double (&th)[NDIM], const bool = false)
    {
        Setup s;
        s.r0 = th[0];
        s.m = th[1];
        s.tau = exp_finite(th[2]);
        double taup = s.tau * (1.0 / th[4] - 1.0) / (1.0 - s.m);
        if (fabs(taup) > TAUP_MAX) taup = copysign(TAUP_MAX, taup);
        s.taup = taup;
        s.A = th[0] * th[1];
        s.C = th[0] - s.A;
        s.teh = s.tau * fabs(th[3]) * 0.70710678118654752440;
        s.teh2 = s.teh * s.teh;
        return s;
    }
    // Z = r0 (1-m) + r0 m / den,  den = 1 + i a (1 + 1/mu),  a = w tau',  mu = i w tau + (i w tau'')^0.5,
    // tau'' = tau^2 eta^2.  With u = sqrt(w_j) (precomputed per frequency: no square root per (walker,
    // frequency)), (i w tau'')^0.5 = u teh (1 + i), so mu = u (teh + i g) with g = u tau + teh, and
    // |mu|^2 = w n2, n2 = teh^2 + g^2.  The inner division goes (1 + 1/mu = (|mu|^2 + conj mu)/|mu|^2) and the
    // common factor w cancels between numerator and denominator:
    //   1/den = n2 (X - iY)/(X^2 + Y^2),   X = n2 + b g,   Y = b (u n2 + teh),   b = tau' u
    // -- 8 instructions up to D = X^2 + Y^2, ONE reciprocal per frequency.  Inside the prior b > 0 and every
    // term is positive: nothing cancels.
    static constexpr bool HAS_FAST = true;
    static constexpr bool HAS_GRID = false;
    // FAST (bound_flags: everywhere in the box D lies in [1e-140, 1e140]): the reciprocals of frequencies 2k and
    // 2k+1 come from ONE reciprocal of D_2k D_2k+1 (v_rcp_f64 issues at a quarter of the FMA rate).  The pairs
    // are (2k, 2k+1) in every kernel, a last unpaired frequency takes its own reciprocal: a walker's value does
    // not depend on which kernel evaluated it.
    static constexpr bool PAIRED = true;
    struct Den { double X, Y, n2, D; };
    __device__ static __forceinline__ Den den(const Setup &s, double u)
    {
        Den d;
        const double b = s.taup * u;
        const double g = fma(u, s.tau, s.teh);
        d.n2 = fma(g, g, s.teh2);
        d.X = fma(b, g, d.n2);
        d.Y = b * fma(u, d.n2, s.teh);
        d.D = fma(d.X, d.X, d.Y * d.Y);
        return d;
    }
    template <bool FAST = false>
    __device__ static __forceinline__ void residual(const Setup &s, const double *__restrict__ rec,
                                                    double &rr, double &ri)
    {
        const Den d = den(s, rec[6]);
        const double t = d.n2 * (s.A * rcp_nr(d.D));
        rr = rec[0] - fma(t, d.X, s.C);      // y - Z with Z = C + t (X - iY); the measured value last (see ColeCole)
        ri = rec[1] + t * d.Y;
    }
    // two frequencies with their dependency chains interleaved; FAST: one reciprocal for the pair
    template <bool FAST = false>
    __device__ static __forceinline__ void residual2(const Setup &s, const double *__restrict__ ra,
                                                     const double *__restrict__ rb, double (&rr)[2],
                                                     double (&ri)[2])
    {
        const double *__restrict__ rec[2] = {ra, rb};
        Den d[2];
        double Ainv[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) d[f] = den(s, rec[f][6]);
        if constexpr (FAST) {
            const double Ar = s.A * rcp_nr(d[0].D * d[1].D);
            Ainv[0] = Ar * d[1].D;
            Ainv[1] = Ar * d[0].D;
        } else {
            const double D[2] = {d[0].D, d[1].D};
            double inv[2];
            rcp_nr_n<2>(D, inv);
            Ainv[0] = s.A * inv[0];
            Ainv[1] = s.A * inv[1];
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const double t = d[f].n2 * Ainv[f];
            rr[f] = rec[f][0] - fma(t, d[f].X, s.C);
            ri[f] = rec[f][1] + t * d[f].Y;
        }
    }
    // Z = r0 (1-m) + r0 m n2 (X - iY)/(X^2 + Y^2)   (m = w, ln w, sqrt w); its own reciprocal: any theta may
    // be asked of forward()
    __device__ static __forceinline__ void eval(const Setup &s, const double *__restrict__ m,
                                                double &zr, double &zi)
    {
        const Den d = den(s, m[2]);
        const double t = d.n2 * (s.A * rcp_nr(d.D));
        zr = fma(t, d.X, s.C);
        zi = -(t * d.Y);
    }
};

// Shin2015 (reference src/bisip/cython_funcs.pyx:42-44, 96-108): two CPE||R elements.
struct Shin {
    static constexpr int NDIM = 6;
    static constexpr int REC = 8;
    static constexpr bool WEIGHTED = false;
    struct Setup {
        double invR[2], Q[2], n[2], cs[2], sn[2];
        double n2[2], lq2[2];  // residual(): n log2e, log_Q log2e
    };
    static constexpr bool HAS_FAST = true;
    static constexpr bool HAS_GRID = true;
    // FAST: frequencies 2k and 2k+1 take their four |y_i|^2 from ONE reciprocal (rcp_joint)
    static constexpr bool PAIRED = true;
    // unit (wave-uniform): both n of the row lie in [0, 1] (BOUNDS_FAST and the row inside the prior)
    __device__ static __forceinline__ Setup setup(const double (&th)[NDIM], const bool unit = false)
    {
        Setup s;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // R = 0 (a prior bound; forward() may be asked for it) makes the term vanish in the
            // reference: 1/0 = inf, inf^-1 = 0.  Clamping 1/R keeps |y|^2 finite here and the
            // term at <= 2^-110 = 8e-34 instead of inf * 0; 2^110 (1e70 until round 4) so that the PRODUCT
            // of the FOUR |y|^2 of a pair of frequencies is a normal number (rcp_joint; with the exponent
            // bound of BOUNDS_FAST every |y|^2 <= 2^223).
            double ir = 1.0 / th[i];
            if (fabs(ir) > 0x1p110) ir = copysign(0x1p110, ir);
            s.invR[i] = ir;
            s.Q[i] = exp_finite(th[2 + i] > 700.0 ? 700.0 : th[2 + i]);     // finite (forward() only; NaN stays NaN)
            s.n[i] = th[4 + i];
            if (unit) sincos_unit(s.n[i], &s.sn[i], &s.cs[i]);
            else sincospi(0.5 * s.n[i], &s.sn[i], &s.cs[i]);
            s.n2[i] = s.n[i] * LOG2E;
            s.lq2[i] = th[2 + i] * LOG2E;
        }
        return s;
    }
    // Q (iw)^n = 2^(n log2e ln w + log_Q log2e) (cs + i sn);  Z = sum_i conj(y_i)/|y_i|^2.
    // F frequencies in lockstep; FAST: the pair (2k, 2k+1) (F = 2) takes its four reciprocals from one
    // reciprocal of the product of the four |y_i|^2, a last unpaired frequency (F = 1) its two from the
    // product of two; safe: one each, exponents clamped.
    static constexpr int NEXP = 2;
    __device__ static __forceinline__ double exp_a(const Setup &s, int i) { return s.n2[i]; }
    __device__ static __forceinline__ double exp_b(const Setup &s, int i) { return s.lq2[i]; }
    template <int F, bool FAST>
    __device__ static __forceinline__ void residual_n(const Setup &s, const double *const (&rec)[F],
                                                      double (&rr)[F], double (&ri)[F])
    {
        constexpr int K = 2 * F;     // value k = (frequency k / 2, element k % 2)
        double y[K], p[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            y[k] = fma(s.n2[k % 2], rec[k / 2][5], s.lq2[k % 2]);
            if constexpr (!FAST) y[k] = fmin(y[k], EXP2_CLAMP);
        }
        exp2_finite_n<K>(y, p);
        residual_from_exp<F, FAST>(s, rec, p, rr, ri);
    }
    // everything after the exponentials p[f * 2 + i] = Q_i w_f^n_i
    template <int F, bool FAST>
    __device__ static __forceinline__ void residual_from_exp(const Setup &s, const double *const (&rec)[F],
                                                             const double (&p)[F * 2], double (&rr)[F],
                                                             double (&ri)[F])
    {
        constexpr int K = 2 * F;
        double yr[F][2], yi[F][2], den[F][2], inv[F][2];
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                yr[f][i] = fma(p[f * 2 + i], s.cs[i], s.invR[i]);    // yr >= 1/R > 1
                yi[f][i] = p[f * 2 + i] * s.sn[i];
                den[f][i] = fma(yr[f][i], yr[f][i], yi[f][i] * yi[f][i]);
            }
        if constexpr (FAST && F == 2) {
            const double flat[4] = {den[0][0], den[0][1], den[1][0], den[1][1]};
            double r[4];
            rcp_joint<4>(flat, r);
#pragma unroll
            for (int k = 0; k < 4; ++k) inv[k / 2][k % 2] = r[k];
        } else if constexpr (FAST) rcp_batch_n<F, 2>(den, inv);
        else {
            double flat[K], r[K];
#pragma unroll
            for (int k = 0; k < K; ++k) flat[k] = den[k / 2][k % 2];
            rcp_nr_n<K>(flat, r);
#pragma unroll
            for (int k = 0; k < K; ++k) inv[k / 2][k % 2] = r[k];
        }
        // real part: the measured value starts the chain (a scalar addend of the first fma); imaginary part: the
        // model's sum first and the measured value in the final add -- as the start of ITS chain the compiler
        // moves it into vector registers first (two v_mov_b32 per frequency; see ColeCole)
#pragma unroll
        for (int f = 0; f < F; ++f) {
            rr[f] = fma(-yr[f][1], inv[f][1], fma(-yr[f][0], inv[f][0], rec[f][0]));
            ri[f] = rec[f][1] + fma(yi[f][1], inv[f][1], yi[f][0] * inv[f][0]);
        }
    }
    template <bool FAST = false>
    __device__ static __forceinline__ void residual(const Setup &s, const double *__restrict__ rec,
                                                    double &rr, double &ri)
    {
        const double *const r1[1] = {rec};
        double a[1], b[1];
        residual_n<1, FAST>(s, r1, a, b);
        rr = a[0];
        ri = b[0];
    }
    template <bool FAST = false>
    __device__ static __forceinline__ void residual2(const Setup &s, const double *__restrict__ ra,
                                                     const double *__restrict__ rb, double (&rr)[2],
                                                     double (&ri)[2])
    {
        const double *const r2[2] = {ra, rb};
        residual_n<2, FAST>(s, r2, rr, ri);
    }
    // Z = sum_i conj(y_i)/|y_i|^2,  y_i = Q (iw)^n + 1/R   (m = w, ln w, sqrt w)
    // forward() may be asked for any theta.  With R < 0 and n > 1 (boxes widened beyond the reference's) the
    // two terms of Re y can cancel -- Z then amplifies the rounding of Q w^n a thousandfold -- so the power is
    // taken with the REFERENCE'S roundings: e^(n ln w), the product n ln w rounded as its cpow rounds it, times
    // Q = e^logQ (a walker constant), instead of one exponential of the fused exponent.
    __device__ static __forceinline__ void eval(const Setup &s, const double *__restrict__ m,
                                                double &zr, double &zi)
    {
        const double lnw = m[1];
        constexpr double PMAX = 0x1p500;          // beyond, |y|^2 would overflow (EXP2_CLAMP)
        zr = 0.0;
        zi = 0.0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            double t = s.n[i] * lnw;
            t = t > 350.0 ? 350.0 : t;               // NaN stays NaN
            double p = s.Q[i] * exp_finite(t);
            p = p > PMAX ? PMAX : p;
            const double yr = fma(p, s.cs[i], s.invR[i]), yi = p * s.sn[i];  // yr >= 1/R > 1
            const double inv = rcp_nr(fma(yr, yr, yi * yi));
            zr = fma(yr, inv, zr);
            zi = fma(-yi, inv, zi);
        }
    }
};

// Geometric frequency grids (BOUNDS_GRID).  The exponentials of ColeCole / Shin at frequency f are
// 2^(a_i ln w_f + b_i) with per-walker a_i, b_i: sixteen of their ~33 instructions per (frequency, term).
// On a grid ln w_{16k+q} = ln w_{16k} + q dlnw they are, for the block of GRID_BLOCK = 16 frequencies that
// starts at 16k,
//   quarter_0 = 2^(a_i ln w_{16k} + b_i)          one exponential per block and term,
//   quarter_m = quarter_{m-1} S4_i                  m = 1, 2, 3: the four frequencies from 16k + 4m,
//   quarter_m,  quarter_m S1_i,  quarter_m S2_i,  quarter_m S3_i        for q mod 4 = 0..3,
// with the per-walker steps S1 = 2^(a_i dlnw), S2 = S1 S1, S3 = S2 S1, S4 = S2 S2 -- a sixteenth of the
// exponentials plus fifteen multiplications (blocks of eight until round 4: one instruction more per frequency
// and term).  The value at frequency f is defined by THIS formula (block start 16*(f/16), quarter (f/4)%4 reached
// by multiplying quarter by quarter, step f%4, in that order) in every kernel, whichever lane evaluates it and in
// whatever order: the same bits everywhere, as for the other paths.  Rounding: the block's exponential as before;
// the quarters and the step add <= 22 ulp (2.5e-15 of a term whose parity tolerance is 1e-12 of max |Z|); the
// grid's own deviation <= 4e-15 in the exponent's ln w, the size of the rounding of ln w.
constexpr int GRID_BLOCK = 16;
template <class M>
struct GridSteps {
    double S[4][M::NEXP];     // 2^(a dlnw) to the powers 1, 2, 3 and 4
};

template <class M>
__device__ __forceinline__ GridSteps<M> grid_steps(const typename M::Setup &s, double dlnw)
{
    GridSteps<M> g;
    double y[M::NEXP], r[M::NEXP];
#pragma unroll
    for (int i = 0; i < M::NEXP; ++i) y[i] = M::exp_a(s, i) * dlnw;
    exp2_finite_n<M::NEXP>(y, r);
#pragma unroll
    for (int i = 0; i < M::NEXP; ++i) {
        g.S[0][i] = r[i];
        g.S[1][i] = r[i] * r[i];
        g.S[2][i] = g.S[1][i] * r[i];
        g.S[3][i] = g.S[1][i] * g.S[1][i];
    }
    return g;
}

template <class M>
__device__ __forceinline__ void grid_base(const typename M::Setup &s, double lnw0, double (&base)[M::NEXP])
{
    double y[M::NEXP];
#pragma unroll
    for (int i = 0; i < M::NEXP; ++i) y[i] = fma(M::exp_a(s, i), lnw0, M::exp_b(s, i));
    exp2_finite_n<M::NEXP>(y, base);
}

// the next quarter of a block starts from the previous quarter's exponential times S4
template <class M>
__device__ __forceinline__ void grid_half(const GridSteps<M> &g, double (&base)[M::NEXP])
{
#pragma unroll
    for (int i = 0; i < M::NEXP; ++i) base[i] = base[i] * g.S[3][i];
}

// exponentials of step Q (compile time) of a half-block
template <class M, int Q>
__device__ __forceinline__ void grid_at(const GridSteps<M> &g, const double (&base)[M::NEXP], double *e)
{
#pragma unroll
    for (int i = 0; i < M::NEXP; ++i) {
        if constexpr (Q == 0) e[i] = base[i];
        else e[i] = base[i] * g.S[Q - 1][i];
    }
}

// ---------------------------------------------------------------------------------
// log-probability of ONE row held in registers -- shared by the batch kernels below and
// by the stretch-move kernels in sampler_kernels.h, so a walker's value is the same
// bits whichever kernel evaluates it.
// ---------------------------------------------------------------------------------
struct ModelOperands {
    const double *__restrict__ cb;  // per-frequency LOG-PROB records (weighted for PDCollapsed)
    int N;
    double lconst;
};

// Summation order (the same in every kernel, whatever L): ONE pair of running sums -- real
// and imaginary squared residuals -- accumulated in ascending frequency order.
//
// L = 1: a plain loop in one lane.
// L in {2, 4, 8}: L adjacent lanes (an aligned pair / quad / half-row; 8 in the multi-workgroup sampler) cooperate on one walker for launches
// with too few walkers to fill the chip.  In each round lane g evaluates the residual of
// frequency j0+g -- the expensive part, in parallel -- and then the running sums travel
// through the group: for step = 0..L-1 every lane forms "its term added to the sums" and all
// lanes adopt the result of lane `step` (a DPP quad broadcast, no LDS).  The additions are
// therefore performed in exactly the order and with exactly the operands of the L = 1 loop,
// so the result is BIT-IDENTICAL for every L: a walker's log-probability does not depend on
// how many lanes evaluated it, and launches of any size and the sampler kernels agree exactly.
template <int STEP, int L>
__device__ __forceinline__ double group_broadcast(double x)
{
    static_assert(L == 2 || L == 4 || L == 8, "an aligned pair, quad or half-row of lanes");
    // quad_perm selecting lane STEP % 4 of each aligned group of min(L, 4) inside a quad
    constexpr int S4 = STEP % 4;
    constexpr int P0 = S4, P1 = S4, P2 = (L >= 4) ? S4 : 2 + S4, P3 = (L >= 4) ? S4 : 2 + S4;
    constexpr int CTRL = P0 | (P1 << 2) | (P2 << 4) | (P3 << 6);
    const long long bits = __double_as_longlong(x);
    int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xf, 0xf, false);
    if constexpr (L == 8) {
        // Eight lanes = half a DPP row (two quads).  Each quad now holds ITS lane STEP % 4; the quad that does not
        // contain lane STEP takes the other one's: a row shift by four lanes, written only into that quad of every
        // half-row (bank mask: quads 1, 3 from quads 0, 2 -- or the reverse).
        if constexpr (STEP < 4) {
            lo = __builtin_amdgcn_update_dpp(lo, lo, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
            hi = __builtin_amdgcn_update_dpp(hi, hi, 0x114, 0xf, 0xa, false);
        } else {
            lo = __builtin_amdgcn_update_dpp(lo, lo, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
            hi = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xf, 0x5, false);
        }
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <class M, int L, int STEP>
__device__ __forceinline__ void rotate_sums(double rr, double ri, const double *__restrict__ rec,
                                            int j0, int N, double &acc0, double &acc1)
{
    if constexpr (STEP < L) {
        double c0, c1;
        if constexpr (M::WEIGHTED) {
            c0 = fma(rr, rr, acc0);
            c1 = fma(ri, ri, acc1);
        } else {
            c0 = fma(rr * rr, rec[2], acc0);
            c1 = fma(ri * ri, rec[3], acc1);
        }
        const double n0 = group_broadcast<STEP, L>(c0), n1 = group_broadcast<STEP, L>(c1);
        if (j0 + STEP < N) { acc0 = n0; acc1 = n1; }   // uniform inside the group
        rotate_sums<M, L, STEP + 1>(rr, ri, rec, j0, N, acc0, acc1);
    }
}

// The travelling sums for lanes that hold TWO residuals each (a pair with its shared reciprocal).  The sums visit
// the lanes in order; lane STEP adds its two terms, in ascending frequency, to what arrives and hands the result
// on -- ONE exchange per lane and sum, not one per term: every lane forms "the arriving sums plus my terms" (its
// own, speculatively; a term past the last frequency is skipped) and all adopt lane STEP's.  The additions are
// those of the one-lane loop, in its order, on its operands: the same bits.  (Until round 5 every TERM was handed
// round: four times the DPP traffic on the chain that bounds a small launch.)
template <int L, int STEP>
__device__ __forceinline__ void rotate_pair_sums(const double (&rr)[2], const double (&ri)[2],
                                                 const double (&iv)[2][2], int jb, int N, int g, double &acc0,
                                                 double &acc1)
{
    if constexpr (STEP < L) {
        double c0 = acc0, c1 = acc1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const double t0 = fma(rr[q] * rr[q], iv[q][0], c0), t1 = fma(ri[q] * ri[q], iv[q][1], c1);
            if (jb + 2 * g + q < N) { c0 = t0; c1 = t1; }
        }
        acc0 = group_broadcast<STEP, L>(c0);
        acc1 = group_broadcast<STEP, L>(c1);
        rotate_pair_sums<L, STEP + 1>(rr, ri, iv, jb, N, g, acc0, acc1);
    }
}

// Two 8-double frequency records (128 B) from LDS into registers, asynchronously: the ds_reads
// are issued here, and lds_pair_wait() -- an s_waitcnt that names the same registers, so every
// use is ordered after it -- is placed by the CALLER after the arithmetic of the current pair.
// Written as inline assembly because the compiler, left to itself, sinks a load whose first use
// is in the next loop iteration down to that use and waits there (and a volatile pointer loses
// the LDS address space: flat loads, one wait each).
// `tie` is an operand of the arithmetic that must come AFTER the reads are in flight (the
// compiler may move ordinary arithmetic across a volatile asm it does not depend on).
__device__ __forceinline__ void lds_pair_issue(const double *lds_ptr, dbl2 (&v)[8], double &tie)
{
    const unsigned a = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) double *)lds_ptr;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v[0]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(v[1]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(v[2]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(v[3]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(v[4]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:80" : "=v"(v[5]) : "v"(a));
    asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(v[6]) : "v"(a));
    asm volatile("ds_read_b128 %0, %2 offset:112" : "=v"(v[7]), "+v"(tie) : "v"(a));
}

__device__ __forceinline__ void lds_pair_wait(dbl2 (&v)[8])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}

// LDSREC: o.cb points at records staged in LDS (16-byte aligned); the pipelined loop then
// fetches the next pair with lds_pair_issue / lds_pair_wait around the current pair's arithmetic.
// FAST: the model's fast residual (ColeCole / Shin: shared reciprocals, no exponent clamp) -- a compile-time
// choice inside the loop, made once per row by logprob_row from the wave-uniform BOUNDS_FAST.
template <class M, int L, bool LDSREC, bool FAST>
__device__ __forceinline__ void logprob_sums(const typename M::Setup &s, const ModelOperands &o, const int g,
                                             double &acc0, double &acc1)
{
    if constexpr (L == 1 && LDSREC && M::REC == 8 && !M::WEIGHTED) {
        // records staged in LDS (persistent sampler): two frequencies at a time, their dependency
        // chains interleaved (residual2), and the NEXT pair's records in flight while this pair is
        // evaluated.  With one wave per SIMD nothing else hides a record read, and a dependent
        // fp64 instruction issues only every ~8 cycles (benchmarks/micro/issue_latency.hip,
        // row_latency.hip: 15.8-16.9k -> 12.8k cycles per 32-frequency double-Cole-Cole row).  The sums
        // still take their terms one by one in ascending frequency order: same bits as the loop below.
        const double *__restrict__ rec = o.cb;
        constexpr int R2 = 2 * M::REC;
        double cur[R2];
        dbl2 buf[8];
        int j = 0;
        if (o.N >= 2) {
            lds_pair_issue(rec, buf, acc0);
            lds_pair_wait(buf);
#pragma unroll
            for (int q = 0; q < 8; ++q) { cur[2 * q] = buf[q].x; cur[2 * q + 1] = buf[q].y; }
        }
        for (; j + 1 < o.N; j += 2, rec += R2) {
            lds_pair_issue((j + 3 < o.N) ? rec + R2 : rec, buf, cur[5]);   // last pair: a harmless re-read
            double rr[2], ri[2];
            M::template residual2<FAST>(s, cur, cur + M::REC, rr, ri);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                acc0 = fma(rr[f] * rr[f], cur[f * M::REC + 2], acc0);
                acc1 = fma(ri[f] * ri[f], cur[f * M::REC + 3], acc1);
            }
            // the compiler must not float the wait above the arithmetic: tie it to its results
            asm volatile("" : "+v"(acc0), "+v"(acc1));
            lds_pair_wait(buf);
#pragma unroll
            for (int q = 0; q < 8; ++q) { cur[2 * q] = buf[q].x; cur[2 * q + 1] = buf[q].y; }
        }
        if (j < o.N) {
            double rr, ri;
            M::template residual<FAST>(s, rec, rr, ri);
            acc0 = fma(rr * rr, rec[2], acc0);
            acc1 = fma(ri * ri, rec[3], acc1);
        }
    } else if constexpr (L == 1 && FAST && M::PAIRED) {
        // pairs (2k, 2k+1) share a reciprocal (Dias, ColeCole<1>, <2>, Shin); the sums still take their terms one by one
        const double *__restrict__ rec = o.cb;
        int j = 0;
        for (; j + 1 < o.N; j += 2, rec += 2 * M::REC) {
            double rr[2], ri[2];
            M::template residual2<true>(s, rec, rec + M::REC, rr, ri);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                acc0 = fma(rr[f] * rr[f], rec[f * M::REC + 2], acc0);
                acc1 = fma(ri[f] * ri[f], rec[f * M::REC + 3], acc1);
            }
        }
        if (j < o.N) {
            double rr, ri;
            M::template residual<true>(s, rec, rr, ri);
            acc0 = fma(rr * rr, rec[2], acc0);
            acc1 = fma(ri * ri, rec[3], acc1);
        }
    } else if constexpr (L == 1) {
        const double *__restrict__ rec = o.cb;
#pragma unroll 2
        for (int j = 0; j < o.N; ++j, rec += M::REC) {
            double rr, ri;
            M::template residual<FAST>(s, rec, rr, ri);
            if constexpr (M::WEIGHTED) {
                acc0 = fma(rr, rr, acc0);
                acc1 = fma(ri, ri, acc1);
            } else {
                acc0 = fma(rr * rr, rec[2], acc0);
                acc1 = fma(ri * ri, rec[3], acc1);
            }
        }
    } else if constexpr (FAST && M::PAIRED) {
        // L lanes per walker: lane g takes PAIR j0/2 + g of every group of L pairs (the pairs of the L = 1 loop:
        // 2k, 2k+1 with one reciprocal; a last unpaired frequency its own), then the sums travel through the
        // group pair by pair, term by term
        const int last = o.N - 1;
        for (int j0 = 0; j0 < o.N; j0 += 2 * L) {
            const int ja = (j0 + 2 * g < o.N) ? j0 + 2 * g : (last & ~1);    // clamp: its results are never adopted
            const double *__restrict__ ra = o.cb + (long long)ja * M::REC;
            double rr[2], ri[2], iv[2][2];
            if (ja + 1 < o.N) {
                M::template residual2<true>(s, ra, ra + M::REC, rr, ri);
                iv[1][0] = ra[M::REC + 2]; iv[1][1] = ra[M::REC + 3];
            } else {
                M::template residual<true>(s, ra, rr[0], ri[0]);
                rr[1] = 0.0; ri[1] = 0.0; iv[1][0] = 0.0; iv[1][1] = 0.0;
            }
            iv[0][0] = ra[2]; iv[0][1] = ra[3];
            rotate_pair_sums<L, 0>(rr, ri, iv, j0, o.N, g, acc0, acc1);
        }
    } else {
        for (int j0 = 0; j0 < o.N; j0 += L) {
            const int j = (j0 + g < o.N) ? j0 + g : o.N - 1;   // clamp: its result is never adopted
            const double *__restrict__ rec = o.cb + (long long)j * M::REC;
            double rr, ri;
            M::template residual<FAST>(s, rec, rr, ri);
            rotate_sums<M, L, 0>(rr, ri, rec, j0, o.N, acc0, acc1);
        }
    }
}

// The travelling sums for lanes that hold FOUR residuals each (a block): as rotate_pair_sums -- lane STEP adds its
// four terms in ascending frequency to the arriving sums, one exchange per lane and sum.
template <int L, int STEP>
__device__ __forceinline__ void rotate_block_sums(const double (&rr)[4], const double (&ri)[4],
                                                  const double (&iv)[4][2], int jb, int N, int g, double &acc0,
                                                  double &acc1)
{
    if constexpr (STEP < L) {
        double c0 = acc0, c1 = acc1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double t0 = fma(rr[q] * rr[q], iv[q][0], c0), t1 = fma(ri[q] * ri[q], iv[q][1], c1);
            if (jb + 4 * g + q < N) { c0 = t0; c1 = t1; }
        }
        acc0 = group_broadcast<STEP, L>(c0);
        acc1 = group_broadcast<STEP, L>(c1);
        rotate_block_sums<L, STEP + 1>(rr, ri, iv, jb, N, g, acc0, acc1);
    }
}

// The same sums on a geometric frequency grid (BOUNDS_GRID; the host guarantees N >= 8 and BOUNDS_FAST):
// one loop per lane layout as above, the exponentials from grid_base / grid_at instead of one exp2 each.
template <class M, int L, bool LDSREC>
__device__ __forceinline__ void logprob_sums_grid(const typename M::Setup &s, const ModelOperands &o, const int g,
                                                  double &acc0, double &acc1)
{
    static_assert(M::REC == 8 && !M::WEIGHTED, "per-frequency models with 8-double records");
    constexpr int NE = M::NEXP, REC = M::REC;
    if constexpr (L == 1 && LDSREC) {
        // pairs of frequencies from LDS, the next pair in flight (see logprob_sums): a block of four
        // frequencies is two pairs, the first takes the block's exponential
        const double *__restrict__ rec = o.cb;
        constexpr int R2 = 2 * REC;
        double cur[R2], base[NE];
        dbl2 buf[8];
        lds_pair_issue(rec, buf, acc0);
        lds_pair_wait(buf);
#pragma unroll
        for (int q = 0; q < 8; ++q) { cur[2 * q] = buf[q].x; cur[2 * q + 1] = buf[q].y; }
        const GridSteps<M> gs = grid_steps<M>(s, cur[7]);
        int j = 0;
        // KIND 0: first pair of a block (takes the block's exponential); 2: first pair of a later quarter (the
        // previous quarter's exponential times S4); 1: the second pair of a quarter
        auto pair = [&](auto kind) {
            constexpr int KIND = decltype(kind)::value;
            lds_pair_issue((j + 3 < o.N) ? rec + R2 : rec, buf, cur[5]);   // last pair: a harmless re-read
            double e[2 * NE], rr[2], ri[2];
            if constexpr (KIND != 1) {
                if constexpr (KIND == 0) grid_base<M>(s, cur[5], base);
                else grid_half<M>(gs, base);
                grid_at<M, 0>(gs, base, e);
                grid_at<M, 1>(gs, base, e + NE);
            } else {
                grid_at<M, 2>(gs, base, e);
                grid_at<M, 3>(gs, base, e + NE);
            }
            const double *const r2[2] = {cur, cur + REC};
            M::template residual_from_exp<2, true>(s, r2, e, rr, ri);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                acc0 = fma(rr[f] * rr[f], cur[f * REC + 2], acc0);
                acc1 = fma(ri[f] * ri[f], cur[f * REC + 3], acc1);
            }
            asm volatile("" : "+v"(acc0), "+v"(acc1));
            lds_pair_wait(buf);
#pragma unroll
            for (int q = 0; q < 8; ++q) { cur[2 * q] = buf[q].x; cur[2 * q + 1] = buf[q].y; }
            j += 2;
            rec += R2;
        };
        using K0 = std::integral_constant<int, 0>;
        using K1 = std::integral_constant<int, 1>;
        using K2 = std::integral_constant<int, 2>;
        for (; j + GRID_BLOCK - 1 < o.N;) {
            pair(K0{});
            pair(K1{});
#pragma unroll 1
            for (int quarter = 1; quarter < GRID_BLOCK / 4; ++quarter) {
                pair(K2{});
                pair(K1{});
            }
        }
        // fewer than a block left: up to seven pairs and a single frequency, in the block's order
        int done = 0;                           // pairs of the last, partial block
        if (j + 1 < o.N) { pair(K0{}); done = 1; }
#pragma unroll 1
        while (j + 1 < o.N) {
            if (done & 1) pair(K1{});
            else pair(K2{});
            ++done;
        }
        if (j < o.N) {                          // one frequency left: step 0 or 2 of the quarter it falls in
            double e[NE], rr[1], ri[1];
            if (done == 0) { grid_base<M>(s, rec[5], base); grid_at<M, 0>(gs, base, e); }
            else if ((done & 1) == 0) { grid_half<M>(gs, base); grid_at<M, 0>(gs, base, e); }
            else grid_at<M, 2>(gs, base, e);
            const double *const r1[1] = {rec};
            M::template residual_from_exp<1, true>(s, r1, e, rr, ri);
            acc0 = fma(rr[0] * rr[0], rec[2], acc0);
            acc1 = fma(ri[0] * ri[0], rec[3], acc1);
        }
    } else if constexpr (L == 1) {
        const double *__restrict__ rec = o.cb;
        const GridSteps<M> gs = grid_steps<M>(s, rec[7]);
        double base[NE];
        auto one = [&](auto Q, const double *__restrict__ r) {
            double e[NE], rr[1], ri[1];
            grid_at<M, decltype(Q)::value>(gs, base, e);
            const double *const r1[1] = {r};
            M::template residual_from_exp<1, true>(s, r1, e, rr, ri);
            acc0 = fma(rr[0] * rr[0], r[2], acc0);
            acc1 = fma(ri[0] * ri[0], r[3], acc1);
        };
        // steps Q and Q+1 of a quarter: a pair (2k, 2k+1) -- quarters start at multiples of four -- with ONE
        // reciprocal where the model pairs its frequencies
        auto two = [&](auto Q, const double *__restrict__ r) {
            constexpr int q = decltype(Q)::value;
            if constexpr (M::PAIRED) {
                double e[2 * NE], rr[2], ri[2];
                grid_at<M, q>(gs, base, e);
                grid_at<M, q + 1>(gs, base, e + NE);
                const double *const r2[2] = {r, r + REC};
                M::template residual_from_exp<2, true>(s, r2, e, rr, ri);
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    acc0 = fma(rr[f] * rr[f], r[f * REC + 2], acc0);
                    acc1 = fma(ri[f] * ri[f], r[f * REC + 3], acc1);
                }
            } else {
                one(std::integral_constant<int, q>{}, r);
                one(std::integral_constant<int, q + 1>{}, r + REC);
            }
        };
        auto four = [&](const double *__restrict__ r) {
            two(std::integral_constant<int, 0>{}, r);
            two(std::integral_constant<int, 2>{}, r + 2 * REC);
        };
        auto upto3 = [&](const double *__restrict__ r, int left) {
            if (left > 1) two(std::integral_constant<int, 0>{}, r);
            else one(std::integral_constant<int, 0>{}, r);
            if (left > 2) one(std::integral_constant<int, 2>{}, r + 2 * REC);
        };
        int j = 0;
        for (; j + GRID_BLOCK - 1 < o.N; j += GRID_BLOCK, rec += GRID_BLOCK * REC) {
            grid_base<M>(s, rec[5], base);
            four(rec);
#pragma unroll
            for (int quarter = 1; quarter < GRID_BLOCK / 4; ++quarter) {
                grid_half<M>(gs, base);
                four(rec + 4 * quarter * REC);
            }
        }
        if (j < o.N) {                           // a partial block: whole quarters, then up to three frequencies
            grid_base<M>(s, rec[5], base);
            for (;;) {
                if (j + 3 < o.N) {
                    four(rec);
                    j += 4;
                    rec += 4 * REC;
                    if (j >= o.N) break;
                    grid_half<M>(gs, base);
                } else {
                    upto3(rec, o.N - j);
                    break;
                }
            }
        }
    } else {
        // L lanes per walker: lane g takes BLOCK jb/4 + g of every group of L blocks -- its own exponential,
        // three steps, four residuals with independent dependency chains (these launches run one wave per
        // SIMD or less: chains, not instruction counts, are what they wait for) -- and the sums then travel
        // through the group block by block, term by term: the additions of the L = 1 loop in its order.
        // (One frequency per lane and round, as logprob_sums does it, would have every lane take the block's
        // exponential AND the step: slower than no grid at all -- measured, micro/grid_small_ensembles.py.)
        const GridSteps<M> gs = grid_steps<M>(s, o.cb[7]);
        const int last = o.N - 1;
        for (int jb = 0; jb < o.N; jb += 4 * L) {
            const int j4 = (jb + 4 * g < o.N) ? jb + 4 * g : (last & ~3);   // clamp: its results are never adopted
            double base[NE], rr[4], ri[4], iv[4][2];
            grid_base<M>(s, o.cb[(long long)(j4 & ~(GRID_BLOCK - 1)) * REC + 5], base);
            const int quarter = (j4 & (GRID_BLOCK - 1)) >> 2;           // its quarter: times S4 once per quarter before it
#pragma unroll
            for (int m = 1; m < GRID_BLOCK / 4; ++m)
#pragma unroll
                for (int i = 0; i < NE; ++i) base[i] = base[i] * (m <= quarter ? gs.S[3][i] : 1.0);
            auto one = [&](auto Q) {
                constexpr int q = decltype(Q)::value;
                const int j = (j4 + q <= last) ? j4 + q : last;
                const double *__restrict__ rec = o.cb + (long long)j * REC;
                double e[NE], r1r[1], r1i[1];
                grid_at<M, q>(gs, base, e);
                const double *const r1[1] = {rec};
                M::template residual_from_exp<1, true>(s, r1, e, r1r, r1i);
                rr[q] = r1r[0]; ri[q] = r1i[0]; iv[q][0] = rec[2]; iv[q][1] = rec[3];
            };
            // steps Q, Q+1 of the lane's quarter: the pair (2k, 2k+1) of the L = 1 loop with its one reciprocal
            // when both frequencies exist; a last unpaired frequency (and the clamped re-reads past the end,
            // whose results are never adopted) one by one
            auto two = [&](auto Q) {
                constexpr int q = decltype(Q)::value;
                if constexpr (M::PAIRED) {
                    if (j4 + q + 1 <= last) {
                        const double *__restrict__ rec = o.cb + (long long)(j4 + q) * REC;
                        double e[2 * NE], r2r[2], r2i[2];
                        grid_at<M, q>(gs, base, e);
                        grid_at<M, q + 1>(gs, base, e + NE);
                        const double *const r2[2] = {rec, rec + REC};
                        M::template residual_from_exp<2, true>(s, r2, e, r2r, r2i);
#pragma unroll
                        for (int f = 0; f < 2; ++f) {
                            rr[q + f] = r2r[f]; ri[q + f] = r2i[f];
                            iv[q + f][0] = rec[f * REC + 2]; iv[q + f][1] = rec[f * REC + 3];
                        }
                        return;
                    }
                }
                one(std::integral_constant<int, q>{});
                one(std::integral_constant<int, q + 1>{});
            };
            two(std::integral_constant<int, 0>{});
            two(std::integral_constant<int, 2>{});
            rotate_block_sums<L, 0>(rr, ri, iv, jb, o.N, g, acc0, acc1);
        }
    }
}

template <class M, int L = 1, bool LDSREC = false>
__device__ __forceinline__ double logprob_row(const double (&th)[M::NDIM], const ModelOperands &o,
                                              const Bounds &b, const int g = 0)
{
    static_assert(L == 1 || L == 2 || L == 4 || L == 8, "lanes per walker");
    if (!in_prior<M::NDIM>(th, b)) return -__builtin_inf();  // never touches the forward model
    // (inside a box that BOUNDS_FAST vouches for, the per-walker constants take their short route as well: sincos_unit)
    const typename M::Setup s = M::setup(th, M::HAS_FAST && (b.flags & BOUNDS_FAST) != 0);
    double acc0 = 0.0, acc1 = 0.0;
    if constexpr (M::HAS_FAST) {
        // (up to three exponentials per frequency: with four or five the steps' registers push the
        // persistent kernels into scratch -- for every loop of the kernel, not only this one; bound_flags
        // never sets the bit for those)
        if constexpr (M::HAS_GRID) {
            if constexpr (M::NEXP <= GRID_MAX_TERMS) {
                // per SPECTRUM: rec[7] holds the spectrum's own step, 0 when its frequencies are on no grid -- in a
                // batch every spectrum runs the loop a context of its own would run
                if ((b.flags & BOUNDS_GRID) && o.cb[7] != 0.0) {
                    logprob_sums_grid<M, L, LDSREC>(s, o, g, acc0, acc1);
                    return fma(-0.5, acc0 + acc1, o.lconst);
                }
            }
        }
        if (b.flags & BOUNDS_FAST) logprob_sums<M, L, LDSREC, true>(s, o, g, acc0, acc1);
        else logprob_sums<M, L, LDSREC, false>(s, o, g, acc0, acc1);
    } else {
        logprob_sums<M, L, LDSREC, false>(s, o, g, acc0, acc1);
    }
    return fma(-0.5, acc0 + acc1, o.lconst);
}

// ---------------------------------------------------------------------------------
// log-probability, one lane per walker, any model above.
// ---------------------------------------------------------------------------------
template <class M, int BLK, bool VEC, int L = 1>
__global__ __launch_bounds__(BLK) void k_logprob(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    constexpr int ROWS = BLK / L;  // walkers per workgroup
    __shared__ __attribute__((aligned(16))) double lds[ROWS * NDIM];
    const long long row0 = (long long)blockIdx.x * ROWS;
    stage_theta<NDIM, ROWS, VEC, BLK>(a.theta, a.W, row0, lds);
    __syncthreads();
    const int r = threadIdx.x / L, g = threadIdx.x % L;
    const long long row = row0 + r;
    const bool live = row < a.W;   // every lane stays for the wavefront exchanges of L > 1
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = lds[(live ? r : 0) * NDIM + q];
    const ModelOperands o{a.cb, a.N, a.lconst};
    const double lp = logprob_row<M, L>(th, o, a.b, g);
    if (live && g == 0) a.out[row] = lp;
}

// ---------------------------------------------------------------------------------
// Two walkers per lane.  For PDCollapsed the scalar cache, not the VALU, is the limiter with
// one walker per lane: every frequency needs 128 B of operands for only 15 FMAs.  Evaluating
// two rows in lockstep reuses each operand loaded into SGPRs for two FMAs, halving the
// scalar traffic per FMA.  Same per-row arithmetic and order as logprob_row.
// ---------------------------------------------------------------------------------
template <class M, int BLK, bool VEC>
__global__ __launch_bounds__(BLK) void k_logprob_x2(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    __shared__ __attribute__((aligned(16))) double lds[2 * BLK * NDIM];
    const long long row0 = (long long)blockIdx.x * (2 * BLK);
    stage_theta<NDIM, 2 * BLK, VEC, BLK>(a.theta, a.W, row0, lds);
    __syncthreads();
    double th[2][NDIM];
    bool ok[2];
    typename M::Setup s[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[r][q] = lds[(threadIdx.x + r * BLK) * NDIM + q];
        ok[r] = in_prior<NDIM>(th[r], a.b) && (row0 + threadIdx.x + r * BLK < a.W);
        s[r] = M::setup(th[r]);
    }
    // same summation order as logprob_row (ascending frequency), two rows in lockstep
    double acc0[2] = {0.0, 0.0}, acc1[2] = {0.0, 0.0};
    const double *__restrict__ rec = a.cb;
#pragma unroll 2
    for (int j = 0; j < a.N; ++j, rec += M::REC) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double rr, ri;
            M::residual(s[r], rec, rr, ri);
            if constexpr (M::WEIGHTED) {
                acc0[r] = fma(rr, rr, acc0[r]);
                acc1[r] = fma(ri, ri, acc1[r]);
            } else {
                acc0[r] = fma(rr * rr, rec[2], acc0[r]);
                acc1[r] = fma(ri * ri, rec[3], acc1[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const long long row = row0 + threadIdx.x + r * BLK;
        if (row < a.W) a.out[row] = ok[r] ? fma(-0.5, acc0[r] + acc1[r], a.lconst) : -__builtin_inf();
    }
}

// ---------------------------------------------------------------------------------
// PolynomialDecomposition, QR-reduced (BISIP_VARIANT_REDUCED).
// The forward model is linear in b = R0*(1, a_0..a_P), so with the weighted design
// matrix A (2N x n, n = P+2) factored once on the host, A = Q [R;0]:
//   chi^2(b) = |y/s - A b|^2 = rest + | e + R (bhat - b) |^2     (exact for ANY bhat,
//   e = Q^T(y/s)[:n] - R bhat carried explicitly).
// Per walker: n products, n subtractions, n(n+1)/2 FMAs, n squares -- the kernel is
// then bound by streaming theta in and logp out (8*(ndim+1) B/eval).
// ---------------------------------------------------------------------------------
// The compensated form also carries the LOW word of the triangle, R = R + Rlo to twice the working
// precision.  Rounding R itself to double is a formulation error no arithmetic can undo: on a degree-9
// design it leaves 1e-9 ... 3e-8 in a log-probability on the shell logp = 0 (60-digit evaluation of the
// reference's formula, benchmarks/exact_shell_rows.py) -- as much as the reference's own rounding there.
template <int P, bool COMP>
struct ReducedLow {};
// Rlo = Rl - (double)Rl of a long double Rl has at most 11 significant bits (64 - 53): a float holds it
// exactly, at half the registers (the persistent sampler keeps a spectrum's operands in VGPRs).
template <int P>
struct ReducedLow<P, true> {
    static constexpr int TRI = (P + 2) * (P + 3) / 2;
    float Rlo[(TRI + 1) & ~1];         // even count: the doubles that follow stay 8-byte aligned
};

template <int P, bool COMP = false>
struct ReducedArgs : ReducedLow<P, COMP> {
    static constexpr int n = P + 2;
    double R[n * (n + 1) / 2];  // packed upper triangle, row-major
    double bhat[n];
    double e[n];
    double elo[n];              // e = e + elo to twice the precision (used by the compensated form)
    double rest;
};

// a - b = s + err exactly (Knuth's TwoSum on a and -b; no ordering assumption, 6 flops)
__device__ __forceinline__ void two_diff(double a, double b, double &s, double &err)
{
    s = a - b;
    const double bb = s - a;
    err = (a - (s - bb)) + ((-b) - bb);
}

// COMP = false: plain double arithmetic (n products, n subtractions, n(n+1)/2 FMAs).
// COMP = true : the same number in COMPENSATED arithmetic, for designs whose triangle R has
// entries up to 1e10 (polynomial degree >= 6: nearly collinear columns) and walkers that sit in
// the flat valley of the posterior, where every row of R (bhat - b) cancels by many orders of
// magnitude:
//   * d = bhat - R0*a as an unevaluated sum d + dl: the product's rounding error comes from one
//     fma (TwoProduct), the subtraction's from TwoSum -- d + dl is exact;
//   * each row  e_i + sum_j R_ij d_j  is accumulated as a double-double (Ogita-Rump-Oishi Dot2:
//     TwoProduct per term, TwoSum per addition, the error terms summed separately), plus the
//     first-order terms R_ij dl_j, Rlo_ij d_j and the low word of e_i.
// Result: as if the row sums were formed in twice the working precision and rounded once --
// measured on posterior-valley probes 1e-13 .. 1e-16 relative where the plain form AND the
// per-frequency form (the reference's own kind of sum) give 1e-12 .. 7e-11.  ~11 flops per
// matrix entry instead of 1, still independent of the number of frequencies.
// RA: the operand struct -- ReducedArgs<P, COMP>, or, for COMP = false, any struct with R, bhat, e and rest
// (a batch whose spectra run different tiers keeps one register copy of the larger layout: BatchReducedLP).
// PRIOR_FIRST: decide the prior BEFORE the sums, so that its 4(P+2) scalar registers are free again when the
// operands of the sums arrive (degree 5, compensated, single spectrum: 68 spilled scalars -> 0, 659 -> 503 vector
// instructions).  1: behind a volatile asm -- the single-spectrum kernels, whose operands are kernel arguments.
// 2: behind a plain asm -- the batch kernels, whose operands come from memory through the scalar path: a volatile
// asm counts as a store, everything loaded after it leaves the scalar path for the vector one (a first attempt:
// 9.3e10 -> 5.7e10 evals/s), while the plain one, which the scheduler may move, still removes most of their spills
// (degree 5, compensated: 144 -> 40; plain tier 30 -> 0).
template <int P, bool COMP = false, class RA = ReducedArgs<P, COMP>, int PRIOR_FIRST = 0>
__device__ __forceinline__ double logprob_row_reduced(const double (&th)[P + 2], const RA &r, double lconst,
                                                      const Bounds &b)
{
    constexpr int n = P + 2;
    int inside = 1;
    if constexpr (PRIOR_FIRST != 0) {
        inside = in_prior<n>(th, b) ? 1 : 0;
        if constexpr (PRIOR_FIRST == 1) asm volatile("" : "+v"(inside));
        else asm("" : "+v"(inside));
    }
    double chi2 = r.rest;
    if constexpr (!COMP) {
        double d[n];
        d[0] = r.bhat[0] - th[0];
#pragma unroll
        for (int q = 1; q < n; ++q) d[q] = r.bhat[q] - th[0] * th[q];
        int k = 0;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double u = r.e[i];
#pragma unroll
            for (int j = i; j < n; ++j, ++k) u = fma(r.R[k], d[j], u);
            chi2 = fma(u, u, chi2);
        }
    } else {
        double d[n], dl[n];
        two_diff(r.bhat[0], th[0], d[0], dl[0]);
#pragma unroll
        for (int q = 1; q < n; ++q) {
            const double p = th[0] * th[q];
            const double pe = fma(th[0], th[q], -p);   // th0*thq = p + pe exactly
            double err;
            two_diff(r.bhat[q], p, d[q], err);
            dl[q] = err - pe;
        }
        int k = 0;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double s = r.e[i], c = r.elo[i];
#pragma unroll
            for (int j = i; j < n; ++j, ++k) {
                const double Rk = r.R[k];
                const double h = Rk * d[j];
                const double l = fma(Rk, d[j], -h);    // Rk*d[j] = h + l exactly
                const double t = s + h;
                const double bb = t - s;
                const double er = (s - (t - bb)) + (h - bb);   // s + h = t + er exactly
                s = t;
                c += er + l;
                c = fma(Rk, dl[j], c);
                c = fma((double)r.Rlo[k], d[j], c);
            }
            const double u = s + c;
            chi2 = fma(u, u, chi2);
        }
    }
    const double lp = fma(-0.5, chi2, lconst);
    if constexpr (PRIOR_FIRST != 0) return inside ? lp : -__builtin_inf();
    else return in_prior<n>(th, b) ? lp : -__builtin_inf();
}

template <int P, int BLK, bool VEC, bool COMP = false>
__global__ __launch_bounds__(BLK) void k_logprob_pd_reduced(const LaunchArgs a,
                                                            const ReducedArgs<P, COMP> r)
{
    constexpr int NDIM = P + 2;
    __shared__ __attribute__((aligned(16))) double lds[BLK * NDIM];
    const long long row0 = (long long)blockIdx.x * BLK;
    stage_theta<NDIM, BLK, VEC>(a.theta, a.W, row0, lds);
    __syncthreads();
    const long long row = row0 + threadIdx.x;
    if (row >= a.W) return;
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = lds[threadIdx.x * NDIM + q];
    a.out[row] = logprob_row_reduced<P, COMP, ReducedArgs<P, COMP>, 1>(th, r, a.lconst, a.b);
}

// ---------------------------------------------------------------------------------
// PolynomialDecomposition, loop-faithful (BISIP_VARIANT_FAITHFUL): the reference's own
// structure  M_k = sum_p a_p L[p,k];  z_j = sum_k M_k K[j,k]  (cython_funcs.pyx:84-90),
// each z_j summed over k in the reference's order, with only the walker-independent
// K[j,k] hoisted.  Frequencies are processed in blocks of JB = 16 whose 32 partial sums
// live in registers; k runs in the outer loop and M_k is rebuilt per block (6 of every 38
// FMAs), so no per-lane M[S] array exists and S, N are plain runtime sizes.
// cb layout (doubles):  LT (S, 8): L[0..P][k] padded to 8
//                       then per frequency block b (N padded to a multiple of 16, padding has
//                       weight 0):  y_re[16] y_im[16] iv_re[16] iv_im[16] | per k: K_re[16] K_im[16]
// ---------------------------------------------------------------------------------
template <int P, int BLK, bool VEC>
__global__ __launch_bounds__(BLK) void k_logprob_pd_faithful(const LaunchArgs a, const int S)
{
    constexpr int NDIM = P + 2;
    constexpr int JB = 16;
    __shared__ __attribute__((aligned(16))) double lds[BLK * NDIM];
    const long long row0 = (long long)blockIdx.x * BLK;
    stage_theta<NDIM, BLK, VEC>(a.theta, a.W, row0, lds);
    __syncthreads();
    const long long row = row0 + threadIdx.x;
    if (row >= a.W) return;
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = lds[threadIdx.x * NDIM + q];
    double lp = -__builtin_inf();
    if (in_prior<NDIM>(th, a.b)) {
        const double *__restrict__ LT = a.cb;
        const double *__restrict__ blk = a.cb + (long long)S * 8;
        const int nblocks = (a.N + JB - 1) / JB;
        const long long blk_stride = 4 * JB + (long long)S * 2 * JB;
        double acc0 = 0.0, acc1 = 0.0;
        for (int b = 0; b < nblocks; ++b, blk += blk_stride) {
            double sr[JB], si[JB];
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) { sr[jj] = 0.0; si[jj] = 0.0; }
            const double *__restrict__ Kk = blk + 4 * JB;
            for (int k = 0; k < S; ++k, Kk += 2 * JB) {
                double mk = 0.0;
#pragma unroll
                for (int p = 0; p <= P; ++p) mk = fma(th[1 + p], LT[k * 8 + p], mk);
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) {
                    sr[jj] = fma(mk, Kk[jj], sr[jj]);
                    si[jj] = fma(mk, Kk[JB + jj], si[jj]);
                }
            }
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) {
                const double zr = th[0] * (1.0 - sr[jj]), zi = th[0] * (0.0 - si[jj]);
                const double rr = blk[jj] - zr, ri = blk[JB + jj] - zi;
                acc0 = fma(rr * rr, blk[2 * JB + jj], acc0);
                acc1 = fma(ri * ri, blk[3 * JB + jj], acc1);
            }
        }
        lp = fma(-0.5, acc0 + acc1, a.lconst);
    }
    a.out[row] = lp;
}

// ---------------------------------------------------------------------------------
// PolynomialDecomposition, ONE WAVE PER WALKER (BISIP_VARIANT_WAVE) -- the mapping the
// north star sketches, built so the design decision in DESIGN.md §3.1 rests on a measurement:
// the 2N (frequency, part) points of the spectrum are spread over the 64 lanes, the
// spectrum operands (weighted design rows: ys, -s, s*G[0..P]) are staged through LDS once
// per workgroup and then held in registers, walkers stream through the wave (theta row in
// SGPRs), every lane evaluates its own weighted residual and a wavefront butterfly
// (6 x lane-exchange + add) reduces the 2N squares to one log-prob, written by lane 0.
// Same operands and arithmetic per point as PDCollapsed; only the summation order over
// points differs (tree instead of sequential).  T = ceil(2N/64) points per lane, N <= 64.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int P, int T>
__global__ __launch_bounds__(256) void k_logprob_pd_wave(const LaunchArgs a)
{
    constexpr int NDIM = P + 2;
    constexpr int REC = 4 + 2 * (P + 1);
    extern __shared__ __attribute__((aligned(16))) double lds_rec[];  // N * REC staged records
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < a.N * REC; i += 256) lds_rec[i] = a.cb[i];
    __syncthreads();
    // this lane's points: p = lane + 64*t  ->  (part = p / N, j = p % N)
    double ys[T], ms[T], g[T][P + 1];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int pt = lane + 64 * t;
        const bool on = pt < 2 * a.N;
        const int part = on ? pt / a.N : 0, j = on ? pt - part * a.N : 0;
        const double *r = lds_rec + j * REC;
        ys[t] = on ? r[part] : 0.0;
        ms[t] = (on && part == 0) ? r[2] : 0.0;   // -s_re for real points, 0 for imaginary
#pragma unroll
        for (int p = 0; p <= P; ++p) g[t][p] = on ? r[4 + part * (P + 1) + p] : 0.0;
    }
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * 256) >> 6;
    for (long long w = wave; w < a.W; w += nwaves) {
        const long long wu = (long long)__builtin_amdgcn_readfirstlane((int)(w >> 31)) << 31 |
                             (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w & 0x7fffffff));
        double th[NDIM];
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th[q] = a.theta[wu * NDIM + q];   // wave-uniform -> scalar loads
        double sq = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            double r = fma(th[0], ms[t], ys[t]);
#pragma unroll
            for (int p = 0; p <= P; ++p) r = fma(th[0] * th[1 + p], g[t][p], r);
            sq = fma(r, r, sq);
        }
        const double chi2 = wave_sum(sq);
        if (lane == 0)
            a.out[wu] = in_prior<NDIM>(th, a.b) ? fma(-0.5, chi2, a.lconst) : -__builtin_inf();
    }
}

// ---------------------------------------------------------------------------------
// Gaussian log-likelihood of a model response the CALLER computed: Z (W,2,N) -> (W,).
// The reference's _log_likelihood takes any callable f (src/bisip/models.py:59-62); when f
// is not one of the four built-in forward models the host evaluates it and only the
// reduction runs here.  One wave per row: the 2N terms of a row are contiguous, lane l
// takes elements l, l+64, ... (coalesced), then a fixed butterfly sums the 64 partials.
// rec[0..3] of every frequency record = y_re, y_im, 1/s2_re, 1/s2_im.
// ---------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_loglike_z(const double *__restrict__ Z,
                                                          double *__restrict__ out, long long W,
                                                          const double *__restrict__ cb, int rec,
                                                          int N, double lconst)
{
    const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= W) return;   // whole waves leave together
    const int lane = threadIdx.x & 63;
    const double *__restrict__ z = Z + row * 2 * N;
    double acc = 0.0;
    for (int i = lane; i < 2 * N; i += 64) {
        const int part = i >= N, j = i - part * N;
        const double r = cb[(long long)j * rec + part] - z[i];
        acc = fma(r * r, cb[(long long)j * rec + 2 + part], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = fma(-0.5, acc, lconst);
}

// ---------------------------------------------------------------------------------
// Batch of independent spectra (BASELINE config 5): E spectra share the model shape and
// the prior box; rows [e*Wp, (e+1)*Wp) of theta belong to spectrum e, whose operands sit
// at cb + e*cb_stride.  When a wave never straddles two spectra (UNIFORM: Wp % 64 == 0)
// the spectrum index is made scalar with readfirstlane, so operands still arrive through
// the scalar cache; otherwise they are ordinary per-lane loads.
// ---------------------------------------------------------------------------------
struct BatchArgs {
    const double *__restrict__ theta;
    double *__restrict__ out;
    long long W;              // total rows = E * Wp
    long long Wp;             // rows per spectrum
    const double *__restrict__ cb;
    long long cb_stride;      // doubles between two spectra's records
    const double *__restrict__ lconst;  // (E,)
    const void *__restrict__ red;       // (E,) ReducedArgs<P, COMP> (reduced variants only)
    // compensated launch of a batch whose spectra need different tiers (BISIP_VARIANT_AUTO): tier[e] = 0
    // sends spectrum e through the plain arithmetic on its plain operands -- every spectrum runs what a
    // context of its own would run.  tier == nullptr: every spectrum compensated.
    const void *__restrict__ red_plain; // (E,) ReducedArgs<P, false>
    const unsigned char *__restrict__ tier;
    int N;
    Bounds b;
};

template <bool UNIFORM>
__device__ __forceinline__ long long spectrum_of(long long row, long long Wp)
{
    // rows and Wp are below 2^31 (checked on the host): a 32-bit division, a third of the
    // instructions of the 64-bit one
    const int e = (int)((unsigned)row / (unsigned)Wp);
    return UNIFORM ? (long long)__builtin_amdgcn_readfirstlane(e) : (long long)e;
}

template <class M, bool UNIFORM>
__global__ __launch_bounds__(64) void k_logprob_batch(const BatchArgs a)
{
    constexpr int NDIM = M::NDIM;
    const long long row = (long long)blockIdx.x * 64 + threadIdx.x;
    if (row >= a.W) return;
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = a.theta[row * NDIM + q];
    const long long e = spectrum_of<UNIFORM>(row, a.Wp);
    const ModelOperands o{a.cb + e * a.cb_stride, a.N, a.lconst[e]};
    a.out[row] = logprob_row<M, 1>(th, o, a.b);
}

template <int P, bool UNIFORM, bool COMP = false>
__global__ __launch_bounds__(64) void k_logprob_batch_reduced(const BatchArgs a)
{
    constexpr int NDIM = P + 2;
    const long long row = (long long)blockIdx.x * 64 + threadIdx.x;
    if (row >= a.W) return;
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = a.theta[row * NDIM + q];
    const long long e = spectrum_of<UNIFORM>(row, a.Wp);
    if constexpr (COMP) {
        if (a.tier && !a.tier[e]) {
            const ReducedArgs<P, false> *__restrict__ rp = reinterpret_cast<const ReducedArgs<P, false> *>(a.red_plain) + e;
            a.out[row] = logprob_row_reduced<P, false, ReducedArgs<P, false>, 2>(th, *rp, a.lconst[e], a.b);
            return;
        }
    }
    const ReducedArgs<P, COMP> *__restrict__ r = reinterpret_cast<const ReducedArgs<P, COMP> *>(a.red) + e;
    a.out[row] = logprob_row_reduced<P, COMP, ReducedArgs<P, COMP>, 2>(th, *r, a.lconst[e], a.b);
}

// The same for big batches whose spectra hold a multiple of BLK walkers: the headline kernel's streaming
// structure (theta rows through the LDS transposition, non-temporal loads, 128-lane workgroups), the
// workgroup's spectrum known from blockIdx alone so that its operands come through the scalar path.
template <int P, int BLK, bool VEC, bool COMP = false>
__global__ __launch_bounds__(BLK) void k_logprob_batch_reduced_stream(const BatchArgs a)
{
    constexpr int NDIM = P + 2;
    __shared__ __attribute__((aligned(16))) double lds[BLK * NDIM];
    const long long row0 = (long long)blockIdx.x * BLK;
    stage_theta<NDIM, BLK, VEC>(a.theta, a.W, row0, lds);
    __syncthreads();
    const long long row = row0 + threadIdx.x;
    if (row >= a.W) return;
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = lds[threadIdx.x * NDIM + q];
    const unsigned e = (unsigned)row0 / (unsigned)a.Wp;          // one spectrum per workgroup (Wp % BLK == 0)
    const ReducedArgs<P, false> *__restrict__ rp = reinterpret_cast<const ReducedArgs<P, false> *>(a.red_plain) + e;
    const ReducedArgs<P, COMP> *__restrict__ r = reinterpret_cast<const ReducedArgs<P, COMP> *>(a.red) + e;
    if constexpr (COMP) {
        if (a.tier && !a.tier[e]) {                              // uniform: the whole workgroup takes the plain tier
            a.out[row] = logprob_row_reduced<P, false, ReducedArgs<P, false>, 2>(th, *rp, a.lconst[e], a.b);
            return;
        }
    }
    a.out[row] = logprob_row_reduced<P, COMP, ReducedArgs<P, COMP>, 2>(th, *r, a.lconst[e], a.b);
}

template <class M>
__global__ __launch_bounds__(256) void k_forward_batch(const BatchArgs a)
{
    constexpr int NDIM = M::NDIM;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = a.W * (long long)a.N;
    if (idx >= total) return;
    const long long wk = idx / a.N;
    const int j = (int)(idx - wk * a.N);
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = a.theta[wk * NDIM + q];
    const typename M::Setup s = M::setup(th);
    double zr, zi;
    M::eval(s, a.cb + (wk / a.Wp) * a.cb_stride + (long long)j * M::REC + 4, zr, zi);
    a.out[wk * 2 * a.N + j] = zr;
    a.out[wk * 2 * a.N + a.N + j] = zi;
}

// ---------------------------------------------------------------------------------
// Batched forward(): Z is (W,2,N).  One lane per walker (per-walker setup -- sincos, exp --
// is computed once, as in the log-prob kernels); a wave evaluates 16 frequencies for its
// 64 walkers into an LDS tile and streams the tile out so that every store instruction
// writes whole 128-byte runs of Z.  Bound by writing Z (16*N B per walker).
// Used when N is a multiple of 16 (runs = whole cache lines) or N > 32; other N <= 32 --
// the bundled spectra have N = 20 -- go through k_forward_rows below, because a 16+4 split
// writes most cache lines in parts (measured 1.5 vs 3.3 TB/s at N = 20).
// ---------------------------------------------------------------------------------
// The workgroup of k_forward_tiled is ONE wave, whose LDS operations execute in order; all
// it needs between writing a tile and reading it back is "LDS results have landed" plus a
// compiler barrier.  __syncthreads() would also wait vmcnt(0), i.e. for the previous tile's
// global stores to be acknowledged, serialising store latency with the next tile's
// arithmetic (measured: 2.9 -> see DESIGN.md TB/s).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// M::eval on a frequency record whose model part (doubles 4..REC) is read through the CONSTANT address space.
// The records are never written while a kernel runs, but a kernel that stores Z between two tiles cannot prove
// that its own stores leave them alone, and every record load after the first store leaves the scalar path for
// the vector one: a tile's 16 x 12 doubles of PolynomialDecomposition operands sat in vector registers (246 of
// them, two waves per SIMD; one from degree 6 on).  Loads from address space 4 are scalar whatever precedes them.
template <class M>
__device__ __forceinline__ void eval_const(const typename M::Setup &s, const double *rec, double &zr, double &zi)
{
    typedef const double __attribute__((address_space(4))) *const_ptr;
    const const_ptr c = (const_ptr)(rec + 4);
    double m[M::REC - 4];
#pragma unroll
    for (int q = 0; q < M::REC - 4; ++q) m[q] = c[q];
    M::eval(s, m, zr, zi);
}

template <class M, bool VEC>
__global__ __launch_bounds__(64) void k_forward_tiled(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    constexpr int JC = 16;          // frequencies per tile
    constexpr int ROW = JC + 1;     // padded tile row (doubles): conflict-free ds_write_b64
    // one (64 walkers x 16 frequencies) tile of ONE part at a time: the real parts go through
    // LDS first while the imaginary parts wait in registers -- half the LDS, twice the waves
    __shared__ __attribute__((aligned(16))) double lds[64 * ROW];
    const int lane = threadIdx.x;
    const int N = a.N;
    const bool wide = ((N & 1) == 0) && ((reinterpret_cast<unsigned long long>(a.out) & 15) == 0);
    const long long nblocks = (a.W + 63) / 64;
    // persistent: each wave walks blocks of 64 walkers; the next block's theta rows are
    // requested (strided per-lane loads, 8*NDIM B each) before this block is evaluated
    double th_next[NDIM];
    auto request = [&](long long blk) {
        const long long row = blk * 64 + lane;
        const long long r = row < a.W ? row : a.W - 1;
#pragma unroll
        for (int q = 0; q < NDIM; ++q) th_next[q] = a.theta[r * NDIM + q];
    };
    // stream the LDS tile (jn columns of `part`) to Z: runs of jn doubles per walker
    auto stream_out = [&](long long row0, int rows_here, int part, int j0, int jn) {
        if (jn == JC && wide) {
            // lane l always writes columns c, c+1 of walkers (l>>3) + 8*it: loop-invariant
            // addressing, the loop is two LDS reads and one 16-byte store
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
        for (int j0 = 0; j0 < N; j0 += JC) {
            const int jn = (N - j0) < JC ? (N - j0) : JC;
            const double *__restrict__ rec = a.cb + (a.Wp ? (row0 / a.Wp) * a.cb_stride : 0) + (long long)j0 * M::REC;
            double zim[JC];
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
                double zr = 0.0, zi = 0.0;
                if (jj < jn) eval_const<M>(s, rec + (long long)jj * M::REC, zr, zi);
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

// k_forward_tiled for its common case, straight-line: N a multiple of 16 and Z 16-byte aligned --
// one block of 64 walkers per workgroup, no grid-stride loop, store width fixed at compile time
// (+3 % at N = 32 / 64, benchmarks/micro/forward_rows_variants.hip).  Same values.
template <class M>
__global__ __launch_bounds__(64) void k_forward_tiled16(const LaunchArgs a)
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
    const double *__restrict__ cb = a.cb + (a.Wp ? (row0 / a.Wp) * a.cb_stride : 0);
    // lane l always writes columns c, c+1 of walkers (l>>3) + 8*it
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
        const double *__restrict__ rec = cb + (long long)j0 * M::REC;
        double zim[JC];
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) {
            double zr, zi;
            eval_const<M>(s, rec + (long long)jj * M::REC, zr, zi);
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

// forward() written COLUMN-major: out (spectra, 2N, Wp) -- one contiguous column per (spectrum, part, frequency).
// What the percentile kernels read (chain_stats.hip), so the model-space bands of a survey need no
// row -> column transposition (a 27 GB round trip for 4096 spectra): lane = walker, every store instruction
// writes 64 consecutive doubles of one column, no LDS staging needed.  Same M::eval, same values as Z.
template <class M>
__global__ __launch_bounds__(64) void k_forward_columns(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    const int N = a.N;
    const long long row = (long long)blockIdx.x * 64 + threadIdx.x;
    if (row >= a.W) return;
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = a.theta[row * NDIM + q];
    const typename M::Setup s = M::setup(th);
    const long long Wp = a.Wp ? a.Wp : a.W;
    const long long e = row / Wp, w = row - e * Wp;     // a block of 64 rows may straddle two spectra here: per lane
    double *__restrict__ col = a.out + e * 2 * N * Wp + w;
    if (Wp % 64 == 0) {
        // ... unless the spectra hold whole waves: one spectrum per wave, its records through the scalar path
        const long long e0 = __builtin_amdgcn_readfirstlane((int)e);
        const double *__restrict__ cb = a.cb + e0 * a.cb_stride;
        for (int j = 0; j < N; ++j) {
            double zr, zi;
            eval_const<M>(s, cb + (long long)j * M::REC, zr, zi);
            __builtin_nontemporal_store(zr, col + (long long)j * Wp);
            __builtin_nontemporal_store(zi, col + (long long)(N + j) * Wp);
        }
        return;
    }
    const double *__restrict__ cb = a.cb + e * a.cb_stride;
    for (int j = 0; j < N; ++j) {
        double zr, zi;
        M::eval(s, cb + (long long)j * M::REC + 4, zr, zi);
        __builtin_nontemporal_store(zr, col + (long long)j * Wp);
        __builtin_nontemporal_store(zi, col + (long long)(N + j) * Wp);
    }
}

// Batched forward(), N <= JC: whole rows.  Lane = walker computes all N frequencies (2N doubles in
// registers); SUB walkers at a time go through LDS laid out exactly like Z ([re 0..N) [im 0..N)
// per walker), so the SUB*2N doubles of a pass are ONE contiguous span of Z and every store
// instruction writes 1 KB (16-byte pieces, WIDE: N even and Z 16-byte aligned) or 512 B of
// consecutive addresses: no cache line is ever written in parts.
// One block of 64 walkers per single-wave workgroup, straight-line (a grid-stride loop around
// this body and a run-time choice between the two store widths cost 30 % at N = 20:
// 4.4 -> 6.2 TB/s, benchmarks/micro/forward_rows_variants.hip).
template <class M, int JC, bool WIDE>
__global__ __launch_bounds__(64) void k_forward_rows(const LaunchArgs a)
{
    constexpr int NDIM = M::NDIM;
    constexpr int SUB = 32;
    constexpr int ROWMAX = 2 * JC + 1;
    __shared__ __attribute__((aligned(16))) double lds[SUB * ROWMAX];
    const int lane = threadIdx.x;
    const int N = a.N;
    const int rowlen = (2 * N) | 1;   // odd stride: conflict-free column writes
    const long long row0 = (long long)blockIdx.x * 64;
    const int rows_here = (int)((a.W - row0) < 64 ? (a.W - row0) : 64);
    const long long row = row0 + lane < a.W ? row0 + lane : a.W - 1;
    double th[NDIM];
#pragma unroll
    for (int q = 0; q < NDIM; ++q) th[q] = a.theta[row * NDIM + q];
    const typename M::Setup s = M::setup(th);
    const double *__restrict__ cb = a.cb + (a.Wp ? (row0 / a.Wp) * a.cb_stride : 0);
    double zr[JC], zi[JC];
#pragma unroll
    for (int jj = 0; jj < JC; ++jj) {
        zr[jj] = 0.0; zi[jj] = 0.0;
        if (jj < N) M::eval(s, cb + (long long)jj * M::REC + 4, zr[jj], zi[jj]);
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
        if constexpr (WIDE) {
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

}  // namespace bisip

// What does a barrier among the workgroups of ONE XCD cost on MI355X, and what does a stretch half-step's
// exchange -- every lane writes a row of (ndim + 2) doubles, the barrier, every lane reads a row some lane of
// another workgroup wrote -- cost on top of it?  (A chip-wide barrier with agent-scope fences was measured in
// grid_barrier.hip: 3.2-14 us, more than the kernel boundary it would replace.)
//
// Groups: the grid is 8 * G workgroups; workgroups with equal blockIdx.x % 8 form a group (the dispatcher deals
// blocks round-robin over the 8 XCDs, so a group shares one XCD and one L2 -- OBSERVED, not promised: every
// workgroup reports HW_REG_XCC_ID and the host counts the groups that really sat on one XCD).  `spread` groups
// take consecutive blocks instead (blockIdx.x / G): their members sit on all eight XCDs.
// `active` = how many of the 8 groups run (the others leave at once): 1 = one ensemble alone, 8 = a group per XCD.
//
// Protocols (one lane per workgroup arrives and polls; bounded spin, so every wave ends):
//   fence  plain payload; __threadfence() + agent atomic add + sc1 poll + __threadfence()  (placement-independent)
//   sc1    payload stored and loaded write-through / L1-bypassing (sc1), every storing wave drains (vmcnt(0)),
//          workgroup barrier, agent atomic add, sc1 poll, workgroup barrier, sc1 loads      (placement-independent:
//          MI355X_MICROARCH.md, inter-workgroup visibility, "Valid forms")
//   l2     plain payload stores (the line stays in the XCD's L2), drain, workgroup-scope atomic add (executes in
//          that L2), sc1 poll and sc1 payload loads (bypass L1, served by the L2)            (ONE XCD only)
//
//   hipcc -O3 --offload-arch=gfx950 -w -o xcd_barrier xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ROW = 9;                      // ndim + 2 doubles at ndim = 7 (cfg4)
enum { P_FENCE = 0, P_SC1 = 1, P_L2 = 2 };

struct Args {
    unsigned *counters;     // one per group, 256 B apart
    double *rows;           // [2][groups][lanes per group][ROW]
    long long *cyc;         // per group: 100 MHz ticks of the loop (member 0, lane 0)
    int *fail, *bad, *xcc;  // timeouts, stale rows, XCC id per workgroup
    int G, iters, active, spread, exchange;
};

__device__ __forceinline__ int xcc_id()
{
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}

template <int PROTO>
__device__ __forceinline__ void store_row(double *p, const double (&v)[ROW])
{
#pragma unroll
    for (int q = 0; q < ROW; ++q) {
        if constexpr (PROTO == P_SC1) __hip_atomic_store(p + q, v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p[q] = v[q];
    }
}

template <int PROTO>
__device__ __forceinline__ void load_row(const double *p, double (&v)[ROW])
{
#pragma unroll
    for (int q = 0; q < ROW; ++q) {
        if constexpr (PROTO == P_FENCE) v[q] = p[q];
        else v[q] = __hip_atomic_load(p + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_load ... sc1
    }
}

template <int PROTO>
__device__ __forceinline__ bool group_barrier(unsigned *counter, unsigned target, unsigned limit)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's payload stores have left
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if constexpr (PROTO == P_FENCE) __threadfence();
        if constexpr (PROTO == P_L2) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > limit) { ok = false; break; }
        }
        if constexpr (PROTO == P_FENCE) __threadfence();
    }
    __syncthreads();
    return ok;
}

template <int PROTO>
__global__ __launch_bounds__(512) void k_rounds(const Args a)
{
    const int grp = a.spread ? blockIdx.x / a.G : blockIdx.x % 8;
    const int member = a.spread ? blockIdx.x % a.G : blockIdx.x / 8;
    if (threadIdx.x == 0) a.xcc[blockIdx.x] = xcc_id();
    if (grp >= a.active) return;
    unsigned *counter = a.counters + grp * 64;
    const int n = a.G * blockDim.x, me = member * blockDim.x + threadIdx.x;
    double *base = a.rows + (size_t)grp * 2 * n * ROW;
    const long long t0 = wall_clock64();
    unsigned seed = 2654435761u * (unsigned)(me + 1);
    for (int it = 0; it < a.iters; ++it) {
        double *buf = base + (size_t)(it & 1) * n * ROW;
        if (a.exchange) {
            double v[ROW];
#pragma unroll
            for (int q = 0; q < ROW; ++q) v[q] = me * 16.0 + q + it;
            store_row<PROTO>(buf + (size_t)me * ROW, v);
        }
        if (!group_barrier<PROTO>(counter, (unsigned)(it + 1) * a.G, 1u << 20)) {
            if (threadIdx.x == 0) atomicAdd(a.fail, 1);
            break;
        }
        if (a.exchange) {
            seed = seed * 1664525u + 1013904223u;
            const int other = (int)((seed >> 8) % (unsigned)n);      // any lane of the group, mostly another workgroup
            double v[ROW];
            load_row<PROTO>(buf + (size_t)other * ROW, v);
            double s = 0;
#pragma unroll
            for (int q = 0; q < ROW; ++q) s += v[q];
            const double want = ROW * (other * 16.0 + it) + ROW * (ROW - 1) / 2;
            if (s != want) atomicAdd(a.bad, 1);
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0 && member == 0) a.cyc[grp] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------
// The exchange a stretch half-step really needs, laid out for it: ONE LANE PER WALKER that keeps its own row in
// registers, rows padded to 8 doubles (64 B, aligned) in memory.  In round `it` the lanes of parity it & 1 are
// active: each gathers the row of a random walker of the other parity (4 x 16-B sc1 loads; written in round
// it - 1), checks it, and stores its own row (4 x 16-B stores, a wave's rows contiguous); then the barrier.
// One buffer, as in the sampler: a row is rewritten only after a barrier that follows every read of it.
// ---------------------------------------------------------------------------------------------------------
typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int PROTO>
__device__ __forceinline__ void store_row64(double *p, const dbl2 (&v)[4])
{
    if constexpr (PROTO == P_SC1) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
                     "global_store_dwordx4 %0, %3, off offset:32 sc1\n\tglobal_store_dwordx4 %0, %4, off offset:48 sc1"
                     :: "v"(p), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory");
    } else {
        dbl2 *d = reinterpret_cast<dbl2 *>(p);
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
}

__device__ __forceinline__ void gather_row64(const double *p, dbl2 (&v)[4])
{
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p) : "memory");
}

template <int PROTO>
__global__ __launch_bounds__(1024) void k_owner(const Args a)
{
    const int grp = a.spread ? blockIdx.x / a.G : blockIdx.x % 8;
    const int member = a.spread ? blockIdx.x % a.G : blockIdx.x / 8;
    if (threadIdx.x == 0) a.xcc[blockIdx.x] = xcc_id();
    if (grp >= a.active) return;
    unsigned *counter = a.counters + grp * 64;
    const int n = a.G * blockDim.x, me = member * blockDim.x + threadIdx.x;
    double *rows = a.rows + (size_t)grp * n * 8;
    dbl2 mine[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { mine[q].x = me * 16.0 + 2 * q - 1; mine[q].y = me * 16.0 + 2 * q + 1 - 1; }
    store_row64<PROTO>(rows + (size_t)me * 8, mine);
    unsigned round = 0;
    bool ok = group_barrier<PROTO>(counter, ++round * a.G, 1u << 20);
    const long long t0 = wall_clock64();
    unsigned seed = 2654435761u * (unsigned)(me + 1);
    for (int it = 0; ok && it < a.iters; ++it) {
        if ((me & 1) == (it & 1)) {
            seed = seed * 1664525u + 1013904223u;
            const int other = (int)(((seed >> 8) % (unsigned)(n / 2)) * 2 + (1 - (me & 1)));
            dbl2 v[4];
            gather_row64(rows + (size_t)other * 8, v);
            double s = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) s += v[q].x + v[q].y;
            if (s != 8 * (other * 16.0 + (it - 1)) + 28.0) atomicAdd(a.bad, 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) { mine[q].x = me * 16.0 + 2 * q + it; mine[q].y = me * 16.0 + 2 * q + 1 + it; }
            store_row64<PROTO>(rows + (size_t)me * 8, mine);
        }
        ok = group_barrier<PROTO>(counter, ++round * a.G, 1u << 20);
        if (!ok && threadIdx.x == 0) atomicAdd(a.fail, 1);
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0 && member == 0) a.cyc[grp] = t1 - t0;
}

template <int PROTO, bool OWNER = false>
static void run(const char *name, int G, int lanes, int active, int spread, int exchange, Args a)
{
    a.G = G; a.active = active; a.spread = spread; a.exchange = exchange;
    const int grid = 8 * G;
    std::vector<long long> c(8);
    std::vector<int> xcc(grid);
    int f = 0, b = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(a.counters, 0, 8 * 64 * 4); hipMemset(a.fail, 0, 4); hipMemset(a.bad, 0, 4); hipMemset(a.cyc, 0, 64);
        if (OWNER) hipLaunchKernelGGL(k_owner<PROTO>, dim3(spread ? G * active : grid), dim3(lanes), 0, 0, a);
        else hipLaunchKernelGGL(k_rounds<PROTO>, dim3(grid), dim3(lanes), 0, 0, a);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    }
    hipMemcpy(c.data(), a.cyc, 64, hipMemcpyDeviceToHost);
    hipMemcpy(&f, a.fail, 4, hipMemcpyDeviceToHost); hipMemcpy(&b, a.bad, 4, hipMemcpyDeviceToHost);
    hipMemcpy(xcc.data(), a.xcc, grid * 4, hipMemcpyDeviceToHost);
    int one_xcd = 0;
    for (int g = 0; g < active; ++g) {
        bool same = true;
        for (int m = 0; m < G; ++m) {
            const int blk = spread ? g * G + m : m * 8 + g, blk0 = spread ? g * G : g;
            same = same && xcc[blk] == xcc[blk0];
        }
        one_xcd += same;
    }
    double worst = 0, best = 1e30;
    for (int g = 0; g < active; ++g) { const double us = c[g] * 0.01 / a.iters; worst = us > worst ? us : worst; best = us < best ? us : best; }
    printf("%-5s %-8s G=%2d x %3d lanes, %d group(s) %-6s: %5.2f us per round (slowest group; fastest %5.2f), %d/%d groups on one XCD, "
           "%d stale%s\n", name, OWNER ? "owner" : (exchange ? "exchange" : "barrier"), G, lanes, active, spread ? "spread" : "by%8", worst, best, one_xcd, active,
           b, f ? "  (TIMED OUT)" : "");
    fflush(stdout);
}

int main()
{
    Args a{};
    hipMalloc(&a.counters, 8 * 64 * 4); hipMalloc(&a.cyc, 64); hipMalloc(&a.fail, 4); hipMalloc(&a.bad, 4); hipMalloc(&a.xcc, 2048 * 4);
    hipMalloc(&a.rows, (size_t)8 * 2 * 32 * 1024 * ROW * 8);
    a.iters = 2000;
    // one lane per walker, 64-B rows: W = G x lanes walkers in ONE group; by%8 = on one XCD, spread = over the chip
    for (int W : {2048, 8192, 32768}) {
        for (int lanes : {256, 512, 1024}) {
            const int G = W / lanes;
            if (G <= 32 && G >= 2) {
                run<P_L2, true>("l2", G, lanes, 1, 0, 1, a);
                run<P_SC1, true>("sc1", G, lanes, 1, 0, 1, a);
            }
        }
        for (int lanes : {64, 128, 256, 512}) {
            const int G = W / lanes;
            if (G <= 256 && G >= 8) run<P_SC1, true>("sc1", G, lanes, 1, 1, 1, a);
        }
    }
    for (int exchange : {0, 1})
        for (int lanes : {256, 512})
            for (int G : {8, 16, 32}) {
                for (int active : {1, 8}) {
                    run<P_L2>("l2", G, lanes, active, 0, exchange, a);
                    run<P_SC1>("sc1", G, lanes, active, 0, exchange, a);
                }
                run<P_SC1>("sc1", G, lanes, 1, 1, exchange, a);
                run<P_FENCE>("fence", G, lanes, 1, 0, exchange, a);
            }
    return 0;
}

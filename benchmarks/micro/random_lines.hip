// What does the chip deliver when every lane reads (and writes) whole 64-byte lines at RANDOM places of a buffer --
// the access pattern of the stretch move on a big ensemble, where emcee's random split makes the walker's own row
// and its partner's row random lines of the packed state?  The half-step kernel moves
// 160 B per proposal (its stream drawn in place) in 29-31 us per 524,288 proposals; this measures what the pattern
// itself allows.
// Patterns, per "proposal" p (524,288 of them over a state of 1,048,576 rows of 64 B = 64 MB):
//   gather1   read row a[p]                                        (one lane per proposal, 4 x 16 B)
//   gather2   read rows a[p] and b[p]
//   move      read rows a[p], b[p], write row a[p]                  (the half-step's three lines)
//   move+seq  the same plus 24 B of sequential stream per proposal and a 64-B sequential chain row
//   tail46    two rows read, 24 B of stream, the row written by 46 % of the proposals; +atomic: and a 4-byte counter bumped (atomicAdd) there
// each with one lane per proposal and with four lanes per proposal (16 B each: one line per quad).
// a = a random half of a random permutation (every row at most once), b = random rows of the other half.
//   hipcc -O3 --offload-arch=gfx950 -w -o random_lines random_lines.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int MODE, int L>
__global__ __launch_bounds__(256) void k(const dbl2 *__restrict__ state_r, dbl2 *state_w, const int *__restrict__ a,
                                         const int *__restrict__ b, const double *__restrict__ seq, dbl2 *chain, long long n, int *cnt)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long p = t / L;
    const int g = t % L;
    if (p >= n) return;
    const int ia = a[p];
    dbl2 acc = {0.0, 0.0};
    constexpr int PER = 4 / L;                       // 16-byte pieces per lane
#pragma unroll
    for (int q = 0; q < PER; ++q) acc += state_r[(long long)ia * 4 + g * PER + q];
    if (MODE >= 1) {
        const int ib = b[p];
#pragma unroll
        for (int q = 0; q < PER; ++q) acc += state_r[(long long)ib * 4 + g * PER + q];
    }
    if (MODE >= 3) acc.x += seq[p * 3] + seq[p * 3 + 1] + seq[p * 3 + 2];
    if (MODE == 4 || MODE == 5) {                   // the half-step's tail: 46 % accepted -> row written, counter bumped
        // (every lane needs its rows: without this the compiler moves the loads under the branch and 54 % of them go)
        asm volatile("" : "+v"(acc.x), "+v"(acc.y));
        if ((ia * 2654435761u >> 16) % 100 < 46) {
#pragma unroll
            for (int q = 0; q < PER; ++q) state_w[(long long)ia * 4 + g * PER + q] = acc;
            if (MODE == 4 && g == 0) atomicAdd(cnt + ia, 1);
        }
    } else if (MODE >= 2) {
#pragma unroll
        for (int q = 0; q < PER; ++q) state_w[(long long)ia * 4 + g * PER + q] = acc;
    } else if (acc.x == 12345.678) {
        state_w[0] = acc;
    }
    if (MODE == 3) {
#pragma unroll
        for (int q = 0; q < PER; ++q) __builtin_nontemporal_store(acc, &chain[p * 4 + g * PER + q]);
    }
}

static int *g_cnt;
template <int MODE, int L>
void run(const char *name, double bytes_per, dbl2 *st, int *a, int *b, double *seq, dbl2 *chain, long long n)
{
    const unsigned grid = (unsigned)((n * L + 255) / 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, L>), dim3(grid), dim3(256), 0, 0, st, st, a, b, seq, chain, n, g_cnt);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 200;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, L>), dim3(grid), dim3(256), 0, 0, st, st, a, b, seq, chain, n, g_cnt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-10s %d lane(s) per proposal: %6.1f us per launch, %5.0f B per proposal -> %5.2f TB/s\n", name, L, us, bytes_per,
           bytes_per * n / (us * 1e-6) / 1e12);
}

int main(int argc, char **argv)
{
    const long long W = argc > 1 ? atoll(argv[1]) : 1048576, n = W / 2;
    std::vector<int> perm(W);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937_64 rng(5);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<int> ha(perm.begin(), perm.begin() + n), hb(n);
    for (long long i = 0; i < n; ++i) hb[i] = perm[n + rng() % (W - n)];
    dbl2 *st, *chain; int *a, *b; double *seq;
    hipMalloc(&st, W * 64); hipMemset(st, 0, W * 64);
    hipMalloc(&chain, n * 64);
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&seq, n * 24); hipMemset(seq, 0, n * 24);
    hipMemcpy(a, ha.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), n * 4, hipMemcpyHostToDevice);
    hipMalloc(&g_cnt, W * 4); hipMemset(g_cnt, 0, W * 4);
    printf("state %lld rows x 64 B = %.0f MB, %lld proposals per launch\n", W, W * 64 / 1e6, n);
    run<0, 1>("gather1", 64 + 4, st, a, b, seq, chain, n);   run<0, 4>("gather1", 64 + 4, st, a, b, seq, chain, n);
    run<1, 1>("gather2", 128 + 8, st, a, b, seq, chain, n);  run<1, 4>("gather2", 128 + 8, st, a, b, seq, chain, n);
    run<2, 1>("move", 192 + 8, st, a, b, seq, chain, n);     run<2, 4>("move", 192 + 8, st, a, b, seq, chain, n);
    run<3, 1>("move+seq", 192 + 8 + 24 + 64, st, a, b, seq, chain, n); run<3, 4>("move+seq", 192 + 8 + 24 + 64, st, a, b, seq, chain, n);
    run<5, 1>("tail46", 128 + 8 + 24 + 0.46 * 64, st, a, b, seq, chain, n);
    run<4, 1>("tail46+atomic", 128 + 8 + 24 + 0.46 * 68, st, a, b, seq, chain, n);
    // the same bytes from SEQUENTIAL rows, for scale
    std::iota(ha.begin(), ha.end(), 0);
    for (long long i = 0; i < n; ++i) hb[i] = (int)(n + i);
    hipMemcpy(a, ha.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), n * 4, hipMemcpyHostToDevice);
    printf("sequential rows (a[p] = p, b[p] = n + p):\n");
    run<2, 1>("move", 192 + 8, st, a, b, seq, chain, n);     run<2, 4>("move", 192 + 8, st, a, b, seq, chain, n);
    run<3, 4>("move+seq", 192 + 8 + 24 + 64, st, a, b, seq, chain, n);
    return 0;
}

// Store-pattern ceilings for the forward kernel's output Z (W,2,N) f64, N = 32:
// how fast can 1 GiB be written as (a) a linear stream, (b) 128-byte runs hopping between
// 512-byte rows (what a 16-frequency tile produces), (c) 256-byte runs, (d) whole 512-byte rows.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl2 __attribute__((ext_vector_type(2)));

// each wave-instruction writes 64 lanes x 16 B; RUN = bytes per contiguous run; rows of 512 B;
// a block of 64 lanes covers 64 walkers (32 KiB of Z) tile by tile like k_forward_tiled
template <int RUN, bool NT>
__global__ __launch_bounds__(64) void k_runs(dbl2 *out, long long nwalkers)
{
    const long long w0 = (long long)blockIdx.x * 64;
    if (w0 >= nwalkers) return;
    constexpr int LANES_PER_RUN = RUN / 16;
    constexpr int RUNS_PER_ROW = 512 / RUN;                 // tiles per row half... (re,im each 256 B)
    const int lane = threadIdx.x;
    dbl2 v = {(double)lane, 1.0};
    // a tile = one RUN-byte column block of every walker row's re and im halves
    for (int tile = 0; tile < 256 / RUN * 1; ++tile) {      // tiles over j (256 B per part)
        // per tile: 64 walkers x 2 parts x RUN bytes
        const int total16 = 64 * 2 * LANES_PER_RUN;         // 16-byte pieces in the tile
        for (int flat = lane; flat < total16; flat += 64) {
            const int w = flat / (2 * LANES_PER_RUN), c = flat % (2 * LANES_PER_RUN);
            const int part = c / LANES_PER_RUN, jj = c % LANES_PER_RUN;
            dbl2 *dst = out + ((w0 + w) * 512 + part * 256 + tile * RUN) / 16 + jj;
            if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
    }
    (void)RUNS_PER_ROW;
}

template <bool NT>
__global__ __launch_bounds__(256) void k_linear(dbl2 *out, long long n16)
{
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    dbl2 v = {1.0, 2.0};
    for (; i < n16; i += stride) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
}

template <class F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}

int main()
{
    const long long W = 1LL << 21, bytes = W * 512;
    dbl2 *out; hipMalloc(&out, bytes);
    auto rep = [&](const char *n, float ms) { printf("%-40s %8.1f us %7.1f GB/s\n", n, ms * 1e3, bytes / ms / 1e6); };
    rep("linear stream 16B plain, grid 4096", timeit([&] { hipLaunchKernelGGL(k_linear<false>, dim3(4096), dim3(256), 0, 0, out, bytes / 16); }));
    rep("linear stream 16B nt, grid 4096", timeit([&] { hipLaunchKernelGGL(k_linear<true>, dim3(4096), dim3(256), 0, 0, out, bytes / 16); }));
    rep("linear stream 16B plain, grid 65536", timeit([&] { hipLaunchKernelGGL(k_linear<false>, dim3(65536), dim3(256), 0, 0, out, bytes / 16); }));
    rep("runs 128 B plain", timeit([&] { hipLaunchKernelGGL((k_runs<128, false>), dim3(W / 64), dim3(64), 0, 0, out, W); }));
    rep("runs 128 B nt", timeit([&] { hipLaunchKernelGGL((k_runs<128, true>), dim3(W / 64), dim3(64), 0, 0, out, W); }));
    rep("runs 256 B plain", timeit([&] { hipLaunchKernelGGL((k_runs<256, false>), dim3(W / 64), dim3(64), 0, 0, out, W); }));
    rep("runs 256 B nt", timeit([&] { hipLaunchKernelGGL((k_runs<256, true>), dim3(W / 64), dim3(64), 0, 0, out, W); }));
    return 0;
}

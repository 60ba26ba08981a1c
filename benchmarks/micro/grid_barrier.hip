// What does a device-wide barrier cost on MI355X (256 CUs in 8 XCDs, one L2 per XCD)?
// A grid-resident stretch-move kernel would need one per half-step (the other half reads the rows
// just written, on any CU), in place of a kernel boundary (~4.2 us launch-to-launch for a tiny kernel).
//   barrier = __threadfence() (agent-scope release/acquire: L2 write-back + invalidate across XCDs)
//             + one atomic per workgroup on a monotonic counter + spin (bounded: every wave exits)
// Also times barrier + "exchange": every lane writes a row and, after the barrier, reads a row
// written by a workgroup on another XCD, checking the value (coherence check).
//   hipcc -O3 --offload-arch=gfx950 -w -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, unsigned limit)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > limit) { ok = false; break; }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ void k_barriers(unsigned *counter, int iters, long long *cyc, int *fail)
{
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
        if (!grid_barrier(counter, (unsigned)(it + 1) * gridDim.x, 1u << 22)) { if (threadIdx.x == 0) atomicAdd(fail, 1); break; }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_exchange(unsigned *counter, int iters, double *rows, long long *cyc, int *fail, int *bad)
{
    const int n = gridDim.x * blockDim.x, me = blockIdx.x * blockDim.x + threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        double *buf = rows + (size_t)(it & 1) * n * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) buf[(size_t)me * 8 + q] = me * 8.0 + q + it;
        if (!grid_barrier(counter, (unsigned)(it + 1) * gridDim.x, 1u << 22)) { if (threadIdx.x == 0) atomicAdd(fail, 1); break; }
        const int other = (me + blockDim.x * 3 + 17) % n;      // a lane of another workgroup (another XCD)
        double s = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += buf[(size_t)other * 8 + q];
        const double want = 8.0 * (other * 8.0 + it) + 28.0;
        if (s != want) atomicAdd(bad, 1);
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    unsigned *counter; long long *cyc; int *fail, *bad; double *rows;
    hipMalloc(&counter, 4); hipMalloc(&cyc, 8); hipMalloc(&fail, 4); hipMalloc(&bad, 4); hipMalloc(&rows, 2ull * 1024 * 256 * 8 * 8);
    const int iters = 2000;
    for (int block : {64, 256}) {
        for (int grid : {64, 128, 256}) {
            long long c; int f, b;
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(counter, 0, 4); hipMemset(fail, 0, 4);
                k_barriers<<<grid, block>>>(counter, iters, cyc, fail);
                hipDeviceSynchronize();
            }
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
            printf("barrier only      grid %3d x %3d lanes: %.2f us per barrier%s\n", grid, block, c * 0.01 / iters, f ? "  (TIMED OUT)" : "");
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(counter, 0, 4); hipMemset(fail, 0, 4); hipMemset(bad, 0, 4);
                k_exchange<<<grid, block>>>(counter, iters, rows, cyc, fail, bad);
                hipDeviceSynchronize();
            }
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost); hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
            printf("write+barrier+read grid %3d x %3d lanes: %.2f us per round, %d stale reads%s\n", grid, block, c * 0.01 / iters, b, f ? "  (TIMED OUT)" : "");
        }
    }
    return 0;
}

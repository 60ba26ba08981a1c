// After a stream has drained and the host has synchronised, how long does the NEXT kernel take to come
// back?  (A 0.3 s sampler run followed by a 0.6 ms summary kernel sometimes keeps the host waiting 13-36 ms
// while the device-side markers around the kernel stay 0.6 ms apart: benchmarks/micro/summary_after_run.py.)
// Pure HIP, no torch: a long kernel, hipDeviceSynchronize, a pause on the host, a short kernel timed from
// launch to the end of hipStreamSynchronize -- with 1 stream, and with extra streams that have been used.
//   hipcc -O2 --offload-arch=gfx950 -o idle_queue_wakeup idle_queue_wakeup.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

__global__ void k_spin(long long cycles, int *sink)
{
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}
__global__ void k_short(double *x, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] += 1.0; }

static double ms_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }

int main(int argc, char **argv)
{
    const int extra = argc > 1 ? atoi(argv[1]) : 0;        // extra streams, each used once
    const int pause_us = argc > 2 ? atoi(argv[2]) : 1000;   // host pause between the sync and the next launch
    const double long_ms = argc > 3 ? atof(argv[3]) : 100;  // duration of the long kernel
    const char *what = argc > 4 ? argv[4] : "";             // c: pageable device->host copies after the sync (as a sampler's
                                                            // run ends); m: hipMalloc + hipFree of 1 GB; e: hipEvent markers
    const bool copies = strchr(what, 'c'), mallocs = strchr(what, 'm'), events = strchr(what, 'e');
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int *sink; double *x; const int n = 1 << 20;
    std::vector<double> host(n);
    hipMalloc(&sink, 4); hipMalloc(&x, n * 8); hipMemset(x, 0, n * 8);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    std::vector<hipStream_t> others(extra);
    for (auto &s : others) { hipStreamCreateWithFlags(&s, hipStreamNonBlocking); k_short<<<n / 256, 256, 0, s>>>(x, n); }
    hipDeviceSynchronize();
    int slow = 0; double worst = 0, sum = 0; const int reps = 40;
    for (int r = 0; r < reps; ++r) {
        k_spin<<<256, 64, 0, st>>>((long long)(long_ms * 1e5), sink);   // s_memtime ticks at 100 MHz
        hipStreamSynchronize(st);
        if (copies) { int h; hipMemcpy(&h, sink, 4, hipMemcpyDeviceToHost); hipMemcpy(host.data(), x, n * 8, hipMemcpyDeviceToHost); hipMemcpy(host.data(), x, n, hipMemcpyDeviceToHost); }
        if (mallocs) { void *p; hipMalloc(&p, 1ull << 30); hipFree(p); }
        std::this_thread::sleep_for(std::chrono::microseconds(pause_us));
        const auto t = std::chrono::steady_clock::now();
        if (events) hipEventRecord(e0, st);
        k_short<<<n / 256, 256, 0, st>>>(x, n);
        if (events) hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        const double ms = ms_since(t);
        sum += ms; if (ms > worst) worst = ms; if (ms > 2.0) ++slow;
    }
    printf("[%s] extra streams %d, pause %d us, long kernel %.0f ms: short kernel launch->sync mean %.3f ms, worst %.3f ms, %d of %d above 2 ms\n",
           what, extra, pause_us, long_ms, sum / reps, worst, slow, reps);
    return 0;
}

// stream_call_probe.hip -- which HIP calls wait when the stream already holds queued work?  (round 5, diagnosis of the c5 replay:
// he_poly_alloc's zero fill took ~0.5 ms per call with 16 callers and ~5 us alone.)  One stream, 40 queued kernels of ~250 us each;
// the host-side latency of each call issued behind them.
//   hipcc --offload-arch=gfx950 -O2 tools/stream_call_probe.hip -o /tmp/stream_call_probe && /tmp/stream_call_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(uint64_t *p, long iters) {
    uint64_t x = p[threadIdx.x];
    for (long i = 0; i < iters; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
    p[threadIdx.x] = x;
}
__global__ void tiny(uint64_t *p) { p[threadIdx.x] += 1; }
static double us(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint64_t *a, *b, *c; const size_t bytes = 15u << 20;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, 4096);
    hipMemset(c, 0, 4096);
    hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
    // calibrate the spin kernel to ~250 us
    long iters = 20000;
    for (int t = 0; t < 6; t++) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, c, iters); hipStreamSynchronize(s);
        double d = us(t0); if (t > 0) iters = (long)(iters * 250.0 / d);
    }
    const char *names[] = {"hipMemsetAsync 15 MiB", "hipMemsetAsync 256 KiB", "hipMemcpyAsync D2D 15 MiB", "hipMemcpy2DAsync D2D 15 MiB", "kernel launch", "hipEventRecord", "hipEventQuery"};
    for (int which = 0; which < 7; which++) {
        for (int busy = 0; busy < 2; busy++) {
            hipStreamSynchronize(s);
            if (busy) for (int k = 0; k < 40; k++) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, c, iters);
            double tot = 0, mx = 0;
            const int reps = 8;
            for (int r = 0; r < reps; r++) {
                auto t0 = std::chrono::steady_clock::now();
                switch (which) {
                    case 0: hipMemsetAsync(a, 0, bytes, s); break;
                    case 1: hipMemsetAsync(a, 0, 256 << 10, s); break;
                    case 2: hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, s); break;
                    case 3: hipMemcpy2DAsync(b, bytes, a, bytes, bytes, 1, hipMemcpyDeviceToDevice, s); break;
                    case 4: hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, c); break;
                    case 5: hipEventRecord(e, s); break;
                    case 6: hipEventQuery(e); break;
                }
                double d = us(t0); tot += d; if (d > mx) mx = d;
            }
            printf("%-30s stream %s: mean %8.1f us  max %8.1f us per call\n", names[which], busy ? "busy (40 x 250 us queued)" : "idle", tot / reps, mx);
        }
    }
    hipStreamSynchronize(s);
    (void)hipGetLastError();
    return 0;
}

// forkjoin_probe.hip -- what a fork / join between two HIP streams costs on this machine, against the same kernels in one stream.
// A "step" is three dependent phases of short kernels (~10 us each, one workgroup per CU); the middle phase has two independent
// kernels.  (a) all four in one stream; (b) the second middle kernel on a side stream, forked and joined with events.
//   hipcc --offload-arch=gfx950 -O3 tools/forkjoin_probe.hip -o /tmp/forkjoin_probe && /tmp/forkjoin_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin(uint64_t *out, int cycles) {
    const uint64_t t0 = wall_clock64();  // 100 MHz
    while ((int64_t)(wall_clock64() - t0) < cycles) {}
    if (threadIdx.x == 0) out[blockIdx.x] = t0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e1, e2;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    uint64_t *buf;
    CK(hipMalloc((void **)&buf, 4 * 256 * 8));
    const int steps = 2000;
    for (int us : {5, 10, 20}) {
        const int cyc = us * 100;  // wall_clock64 ticks at 100 MHz
        auto one = [&](bool fork) {
            for (int i = 0; i < steps; i++) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, buf, cyc);
                if (fork) {
                    hipEventRecord(e1, s0);
                    hipStreamWaitEvent(s1, e1, 0);
                    hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, buf + 256, cyc);
                    hipEventRecord(e2, s1);
                } else {
                    hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, buf + 256, cyc);
                }
                hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s0, buf + 512, 3 * cyc);
                if (fork) hipStreamWaitEvent(s0, e2, 0);
                hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, buf + 768, cyc);
            }
            hipStreamSynchronize(s0);
            hipStreamSynchronize(s1);
        };
        for (int fork = 0; fork < 2; fork++) {
            one(fork != 0);
            const auto t0 = std::chrono::steady_clock::now();
            one(fork != 0);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("kernels of %2d us (x1, x1 || x3, x1): %s  %.1f us per step  (sum of kernel times: %d us serial, %d us with overlap)\n", us,
                        fork ? "fork/join" : "one stream", dt / steps * 1e6, 6 * us, 5 * us);
        }
    }
    return 0;
}

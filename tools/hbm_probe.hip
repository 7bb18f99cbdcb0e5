// hbm_probe.hip -- what HBM gives a streaming kernel on MI355X by read : write mix (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe && /tmp/hbm_probe
// Each workgroup of 256 threads handles contiguous 2 KiB rows (one 8-byte word per thread), R input streams and W output streams
// of 1 GiB each, plain or non-temporal accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int R, int W, bool NT>
__global__ void __launch_bounds__(256) mix(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, size_t words) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) {
        uint64_t acc = i;
#pragma unroll
        for (int r = 0; r < R; r++) acc += NT ? __builtin_nontemporal_load(&in[(size_t)r * words + i]) : in[(size_t)r * words + i];
#pragma unroll
        for (int w = 0; w < W; w++) {
            if (NT) __builtin_nontemporal_store(acc + w, &out[(size_t)w * words + i]);
            else out[(size_t)w * words + i] = acc + w;
        }
        if (W == 0 && acc == 0x123456789abcdefull) out[0] = acc;
    }
}

template <int R, int W, bool NT>
void run(const uint64_t *in, uint64_t *out, size_t words) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 16;
    mix<R, W, NT><<<grid, 256>>>(in, out, words);
    hipEventRecord(e0);
    for (int k = 0; k < 5; k++) mix<R, W, NT><<<grid, 256>>>(in, out, words);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("read %d : write %d  %s  %7.3f ms  %6.0f GB/s\n", R, W, NT ? "nt   " : "plain", ms, (double)(R + W) * words * 8 / ms * 1e-6);
}

int main() {
    const size_t words = (size_t)1 << 26;  // 512 MiB per stream
    uint64_t *in, *out;
    hipMalloc(&in, 4 * words * 8); hipMalloc(&out, 12 * words * 8);
    hipMemset(in, 1, 4 * words * 8);
    run<1, 0, false>(in, out, words); run<4, 0, false>(in, out, words); run<4, 0, true>(in, out, words);
    run<0, 1, false>(in, out, words); run<0, 4, false>(in, out, words); run<0, 4, true>(in, out, words);
    run<1, 1, false>(in, out, words); run<1, 1, true>(in, out, words);
    run<2, 1, true>(in, out, words); run<1, 4, false>(in, out, words); run<1, 4, true>(in, out, words);
    run<3, 12, true>(in, out, words); run<3, 12, false>(in, out, words); run<4, 2, true>(in, out, words);
    return 0;
}

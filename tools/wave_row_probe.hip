// wave_row_probe.hip -- feasibility probe (round 5): a 4096-coefficient forward row transform in exact doubles done by ONE
// wavefront -- 64 coefficients per lane, two radix-64 rounds, ONE wave-local exchange (no workgroup barrier) -- against the
// production shape (256 threads x 16 coefficients, three radix-16 rounds, three exchanges, one barrier).  The question it answers:
// does a lone wave per SIMD with 64 independent butterfly chains keep the double-precision pipe busier than two to four waves that
// wait on exchanges?  Not part of the product; run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/wave_row_probe.hip -o /tmp/wave_row_probe && /tmp/wave_row_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef const double __attribute__((address_space(4))) *cptrd;
__device__ __forceinline__ double ldcd(const double *p, size_t i) { return ((cptrd)(uintptr_t)p)[i]; }
__device__ __forceinline__ double modmul_f64(double a, double w, double q, double qi) {
    const double h = a * w;
    const double l = __fma_rn(a, w, -h);
    const double c = rint(h * qi);
    return __fma_rn(-c, q, h) + l;
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// offsets of the per-stage blocks of the lane-major twiddle copy: stage s in 6..11 holds 2^(s-6) twiddles per lane
__host__ __device__ constexpr int twl_off(int s) { return 64 * ((1 << (s - 6)) - 1); }

// rows: [nrows][4096] doubles (integers, |x| < q), tw: [4096] doubles (the row's twiddles, index (1 << s) + group as ring/ntt.go)
template <bool STORE_T>
__global__ void __launch_bounds__(256, 1) wave_row_kernel(const double *in, double *out, const double *tw, int nrows, double q, double qi) {
    __shared__ double tile[4][4096];
    __shared__ double twl[64 * 63];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // round-B twiddles, lane-major: twl[off(s) + j * 64 + lane] = tw[(1 << s) + (lane << (s - 6)) + j]
    for (int i = threadIdx.x; i < 64 * 63; i += 256) {
        int s = 6, rem = i;
        while (rem >= 64 * (1 << (s - 6))) { rem -= 64 * (1 << (s - 6)); s++; }
        const int j = rem >> 6, lane = rem & 63;
        twl[i] = tw[(1 << s) + (lane << (s - 6)) + j];
    }
    __syncthreads();
    double *t = tile[wv];
    for (int row = blockIdx.x * 4 + wv; row < nrows; row += gridDim.x * 4) {
        const double *src = in + (size_t)row * 4096;
        double x[64];
#pragma unroll
        for (int k = 0; k < 64; k++) x[k] = __builtin_nontemporal_load(&src[k * 64 + l]);
        // round A: stages 0..5 across the register index (distances 2048 .. 64 coefficients), twiddles wave-uniform
#pragma unroll
        for (int s = 0; s < 6; s++) {
            const int d = 32 >> s;
#pragma unroll
            for (int k = 0; k < 64; k++) {
                if (k & d) continue;
                const double w = ldcd(tw, (size_t)((1 << s) + (k >> (6 - s))));
                const double r = modmul_f64(x[k + d], w, q, qi);
                const double U = x[k];
                x[k] = U + r;
                x[k + d] = U - r;
            }
        }
        // the one exchange: lane l, register k (element 64 k + l) -> lane k, register l; XOR swizzle, no padding
#pragma unroll
        for (int k = 0; k < 64; k++) t[k * 64 + (l ^ k)] = x[k];
        wave_sync();
#pragma unroll
        for (int k = 0; k < 64; k++) x[k] = t[l * 64 + (k ^ l)];
        // round B: stages 6..11 inside the lane's 64 contiguous coefficients
#pragma unroll
        for (int s = 6; s < 12; s++) {
            const int d = 32 >> (s - 6);
#pragma unroll
            for (int k = 0; k < 64; k++) {
                if (k & d) continue;
                const double w = twl[twl_off(s) + (k >> (12 - s)) * 64 + l];
                const double r = modmul_f64(x[k + d], w, q, qi);
                const double U = x[k];
                x[k] = U + r;
                x[k + d] = U - r;
            }
        }
        double *dst = out + (size_t)row * 4096;
        if constexpr (STORE_T) {  // back to the coalesced order through the tile
            wave_sync();
#pragma unroll
            for (int k = 0; k < 64; k++) t[l * 64 + (k ^ l)] = x[k];
            wave_sync();
#pragma unroll
            for (int k = 0; k < 64; k++) __builtin_nontemporal_store(t[k * 64 + (l ^ k)], &dst[k * 64 + l]);
            wave_sync();
        } else {
#pragma unroll
            for (int k = 0; k < 64; k++) dst[l * 64 + k] = x[k];
        }
    }
}

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((unsigned __int128)a * b % q); }

int main() {
    const uint64_t q = 35184372744193ull;
    const int nrows = 256 * 4 * 24;  // 24 rows per wave at one workgroup per CU
    std::vector<double> h_in((size_t)nrows * 4096), h_tw(4096), h_out((size_t)nrows * 4096);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (auto &v : h_in) v = (double)(rnd() % q);
    for (auto &v : h_tw) v = (double)(rnd() % q);
    double *d_in, *d_out, *d_tw;
    hipMalloc(&d_in, h_in.size() * 8); hipMalloc(&d_out, h_in.size() * 8); hipMalloc(&d_tw, 4096 * 8);
    hipMemcpy(d_in, h_in.data(), h_in.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_tw, h_tw.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; variant++) {
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL((wave_row_kernel<true>), dim3(256), dim3(256), 0, 0, d_in, d_out, d_tw, nrows, (double)q, 1.0 / (double)q);
            else hipLaunchKernelGGL((wave_row_kernel<false>), dim3(256), dim3(256), 0, 0, d_in, d_out, d_tw, nrows, (double)q, 1.0 / (double)q);
        };
        launch(); hipDeviceSynchronize();
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 1; }
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("variant %s: %.4f ms per %d rows -> %.2f TB/s of row traffic (read + write), %.1f ns per row-CU slot\n",
               variant == 0 ? "coalesced stores (second exchange)" : "lane-contiguous stores", ms, nrows, nrows * 65536.0 / (ms * 1e-3) / 1e12,
               ms * 1e6 / (nrows / 1024.0));
        // check row 0 and the last row against the same network in integers
        hipMemcpy(h_out.data(), d_out, h_out.size() * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int row : {0, nrows - 1}) {
            std::vector<uint64_t> x(4096);
            for (int e = 0; e < 4096; e++) x[e] = (uint64_t)h_in[(size_t)row * 4096 + e];
            for (int s = 0; s < 12; s++) {
                const int dist = 2048 >> s;
                for (int e = 0; e < 4096; e++) {
                    if (e & dist) continue;
                    const uint64_t w = (uint64_t)h_tw[(1 << s) + (e >> (12 - s))];
                    const uint64_t r = mulmod(x[e + dist], w, q), U = x[e];
                    x[e] = (U + r) % q; x[e + dist] = (U + q - r) % q;
                }
            }
            for (int e = 0; e < 4096; e++) {
                double g = h_out[(size_t)row * 4096 + e];
                long long gi = (long long)g % (long long)q; if (gi < 0) gi += q;
                if (g != (double)(long long)g || (uint64_t)gi != x[e]) bad++;
            }
        }
        printf("  check: %d mismatches\n", bad);
    }
    return 0;
}

// f64_modmul_probe.hip -- feasibility probe: exact modular multiplication of integers below 2^48 carried in
// double-precision registers (error-free product via FMA + rounded quotient) vs the 64-bit integer Montgomery
// product, on gfx950.  Also validates exactness against __int128 on random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned __int128 u128;

__device__ __forceinline__ double modmul_f64(double a, double w, double q, double qi) {
    const double h = a * w;
    const double l = __fma_rn(a, w, -h);
    const double c = rint(h * qi);
    const double d = __fma_rn(-c, q, h);
    return d + l;  // in (-1.5q, 1.5q), exact integer
}
__device__ __forceinline__ uint64_t mred_lazy(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {
    u128 m = (u128)x * y;
    uint64_t H = (uint64_t)(((u128)((uint64_t)m * qinv) * q) >> 64);
    return (uint64_t)(m >> 64) - H + q;
}

__global__ void __launch_bounds__(256) k_f64(double *buf, int iters, double q, double qi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a0 = buf[i * 4], a1 = buf[i * 4 + 1], a2 = buf[i * 4 + 2], a3 = buf[i * 4 + 3];
    const double w = a0;
    for (int k = 0; k < iters; k++) {
        a0 = modmul_f64(a0, w, q, qi); a1 = modmul_f64(a1, w, q, qi);
        a2 = modmul_f64(a2, w, q, qi); a3 = modmul_f64(a3, w, q, qi);
    }
    buf[i * 4] = a0; buf[i * 4 + 1] = a1; buf[i * 4 + 2] = a2; buf[i * 4 + 3] = a3;
}
__global__ void __launch_bounds__(256) k_int(uint64_t *buf, int iters, uint64_t q, uint64_t qinv) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a0 = buf[i * 4], a1 = buf[i * 4 + 1], a2 = buf[i * 4 + 2], a3 = buf[i * 4 + 3];
    const uint64_t w = a0 | 1;
    for (int k = 0; k < iters; k++) {
        a0 = mred_lazy(a0, w, q, qinv); a1 = mred_lazy(a1, w, q, qinv);
        a2 = mred_lazy(a2, w, q, qinv); a3 = mred_lazy(a3, w, q, qinv);
    }
    buf[i * 4] = a0; buf[i * 4 + 1] = a1; buf[i * 4 + 2] = a2; buf[i * 4 + 3] = a3;
}
// exactness: r = modmul_f64(a, w) must satisfy r == a*w mod q (as a signed representative, |r| < 1.5q)
__global__ void k_check(const int64_t *a, const uint64_t *w, int n, uint64_t q, unsigned long long *bad, double *maxabs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double r = modmul_f64((double)a[i], (double)w[i], (double)q, 1.0 / (double)q);
    const __int128 prod = (__int128)a[i] * (__int128)w[i];
    __int128 want = prod % (__int128)q;
    if (want < 0) want += q;
    __int128 got = (__int128)(long long)r % (__int128)q;
    if (got < 0) got += q;
    if (got != want || r != rint(r) || fabs(r) >= 1.5 * (double)q) atomicAdd(bad, 1ull);
}

int main() {
    const uint64_t q = 35184372744193ull;  // 45-bit NTT prime of the bench chain
    const size_t n = (size_t)1 << 24;
    void *buf; hipMalloc(&buf, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 256;
    float ms;
    // f64
    {
        double *h = (double *)malloc(n * 8);
        for (size_t i = 0; i < n; i++) h[i] = (double)((i * 2654435761ull + 12345) % q);
        hipMemcpy(buf, h, n * 8, hipMemcpyHostToDevice); free(h);
        k_f64<<<n / 4 / 256, 256>>>((double *)buf, 8, (double)q, 1.0 / (double)q);
        hipEventRecord(e0); k_f64<<<n / 4 / 256, 256>>>((double *)buf, iters, (double)q, 1.0 / (double)q); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("f64 exact modmul : %.3e /s\n", (double)n * iters / (ms * 1e-3));
    }
    {
        hipMemset(buf, 0x5a, n * 8);
        uint64_t inv = q; for (int i = 0; i < 6; i++) inv *= 2 - q * inv;
        k_int<<<n / 4 / 256, 256>>>((uint64_t *)buf, 8, q, inv);
        hipEventRecord(e0); k_int<<<n / 4 / 256, 256>>>((uint64_t *)buf, iters, q, inv); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("int MRedLazy     : %.3e /s\n", (double)n * iters / (ms * 1e-3));
    }
    // exactness over random signed operands with |a| < 24q, 0 <= w < q, several primes
    const uint64_t primes[3] = {35184372744193ull, 1099511480321ull /*40-bit*/, 281474976546817ull /*48-bit*/};
    for (uint64_t p : primes) {
        const int m = 1 << 22;
        int64_t *ha = (int64_t *)malloc(m * 8); uint64_t *hw = (uint64_t *)malloc(m * 8);
        uint64_t s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
        for (int i = 0; i < m; i++) {
            ha[i] = (int64_t)(rnd() % (48 * p)) - (int64_t)(24 * p);
            hw[i] = rnd() % p;
            if (i < 8) { ha[i] = (i & 1) ? (int64_t)(24 * p - 1) : -(int64_t)(24 * p - 1); hw[i] = p - 1 - (i >> 1); }
        }
        int64_t *da; uint64_t *dw; unsigned long long *dbad; double *dmax;
        hipMalloc(&da, m * 8); hipMalloc(&dw, m * 8); hipMalloc(&dbad, 8); hipMalloc(&dmax, 8);
        hipMemcpy(da, ha, m * 8, hipMemcpyHostToDevice); hipMemcpy(dw, hw, m * 8, hipMemcpyHostToDevice); hipMemset(dbad, 0, 8);
        k_check<<<m / 256, 256>>>(da, dw, m, p, dbad, dmax);
        unsigned long long bad; hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
        printf("exactness q=%llu (%d bits, |a| < 24q): %llu mismatches of %d\n", (unsigned long long)p, 64 - __builtin_clzll(p), bad, m);
        free(ha); free(hw);
    }
    return 0;
}

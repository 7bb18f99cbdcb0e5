// solinas_probe.hip -- VERDICT r5 item 5: a product that uses the STRUCTURE of the reference's primes instead of Montgomery's.
// Every modulus GenModuli draws is q = 2^k -+ c with c = j 2N -+ 1 small (ring/primes.go:24-110); with D = c 2^(64-k) < 2^32,
//     2^64 = D (mod q')     for q' = q 2^(64-k) = 2^64 - D   (a multiple of q: any residue mod q' is a lazy residue mod q)
// so a 128-bit product folds at the word boundary:  a w = H 2^64 + L = L + D H,  D H = E 2^64 + E', ... -- 4 + 2 + 1
// v_mad_u64_u32, no v_mul_lo_u32, plain (non-Montgomery) twiddles and no second twiddle word (what sank the Shoup rows).
// What the count leaves out, and this probe measures: every fold produces a carry that is itself worth D, the representatives
// fill the whole 64-bit word (no room for a lazy butterfly: U + V and U - V need their own wrap corrections), and on gfx950 a
// v_add_co / v_addc costs what a v_mad_u64_u32 costs (tools/instr_probe.hip: 4.7 vs 5.2 cycles per wave instruction).
//   product chains: hipcc's rendering of the Solinas product (full-range representatives, exact carries), the same with the
//   butterfly's modular add / subtract, against the production 16-instruction Montgomery sequence and hipcc's own Montgomery
//   hipcc --offload-arch=gfx950 -O3 tools/solinas_probe.hip -o /tmp/solinas_probe && /tmp/solinas_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
typedef unsigned __int128 u128;

// a w mod q' (q' = 2^64 - D), any 64-bit a, w < 2^61: result any 64-bit representative
__device__ __forceinline__ uint64_t solinas_mul(uint64_t a, uint64_t w, uint32_t D) {
    const u128 p = (u128)a * w;
    const uint64_t H = (uint64_t)(p >> 64), L = (uint64_t)p;
    const u128 e = (u128)H * D;                       // < 2^96
    const uint64_t Elo = (uint64_t)e;
    uint64_t S = L + Elo;
    const uint64_t F = (uint64_t)(e >> 64) + (S < L);  // <= 2^32
    const uint64_t G = F * D;                          // < 2^64 (F D <= 2^64 - 2^33 + ... fits: F < 2^32 + 1, checked on the host)
    uint64_t R = S + G;
    if (R < S) R += D;                                 // the wrapped sum is below G < 2^64 - D: no second wrap
    return R;
}
__device__ __forceinline__ uint64_t addmod(uint64_t u, uint64_t v, uint32_t D) {  // u + v mod q'
    uint64_t s = u + v;
    if (s < u) { s += D; if (s < D) s += D; }
    return s;
}
__device__ __forceinline__ uint64_t submod(uint64_t u, uint64_t v, uint32_t D) {  // u - v mod q'
    uint64_t s = u - v;
    if (u < v) { const uint64_t t = s - D; s = t > s ? t - D : t; }
    return s;
}
// hipcc's Montgomery (two 32-bit rounds, csrc/modarith.h mred_lazy_w32 restated)
__device__ __forceinline__ uint64_t mont_c(uint64_t x, uint64_t w, uint64_t q, uint32_t nqinv) {
    u128 t = (u128)x * w;
    uint32_t m = (uint32_t)t * nqinv;
    t = (t + (u128)m * q) >> 32;
    m = (uint32_t)t * nqinv;
    t = (t + (u128)m * q) >> 32;
    return (uint64_t)t;
}
__device__ __forceinline__ uint64_t mont16(uint64_t x, uint64_t w, uint64_t q, uint64_t qinv) {
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32), nq = (uint32_t)(0 - qinv);
    uint64_t r;
    asm("v_mad_u64_u32 v[116:117], vcc, %[x0], %[w0], 0\n\t"
        "v_mad_u64_u32 v[118:119], vcc, %[x0], %[w1], 0\n\t"
        "v_mad_u64_u32 v[120:121], vcc, %[x1], %[w1], 0\n\t"
        "v_mad_u64_u32 v[118:119], vcc, %[x1], %[w0], v[118:119]\n\t"
        "v_mul_lo_u32 v122, v116, %[nq]\n\t"
        "v_mad_u64_u32 v[116:117], vcc, v122, %[q0], v[116:117]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 v119, vcc, 0, v119, vcc\n\t"
        "v_mad_u64_u32 v[118:119], vcc, v122, %[q1], v[118:119]\n\t"
        "v_add_co_u32_e32 v118, vcc, v118, v117\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 v119, vcc, 0, v119, vcc\n\t"
        "v_mul_lo_u32 v122, v118, %[nq]\n\t"
        "v_mad_u64_u32 v[118:119], vcc, v122, %[q0], v[118:119]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 v121, vcc, 0, v121, vcc\n\t"
        "v_add_co_u32_e32 v120, vcc, v120, v119\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 v121, vcc, 0, v121, vcc\n\t"
        "v_mad_u64_u32 %[r], vcc, v122, %[q1], v[120:121]"
        : [r] "=v"(r)
        : [x0] "v"(x0), [x1] "v"(x1), [w0] "v"(w0), [w1] "v"(w1), [q0] "s"(q0), [q1] "s"(q1), [nq] "s"(nq)
        : "v116", "v117", "v118", "v119", "v120", "v121", "v122", "vcc");
    return r;
}

__global__ void check_kernel(const uint64_t *a, const uint64_t *b, const uint64_t *w, uint32_t D, uint64_t *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[3 * i] = solinas_mul(a[i], w[i], D);
    out[3 * i + 1] = addmod(a[i], b[i], D);
    out[3 * i + 2] = submod(a[i], b[i], D);
}
// WHICH: 0 Solinas product chain, 1 Solinas butterfly chain (product + add + sub), 2 hipcc Montgomery product, 3 hand-written
// Montgomery product, 4 Montgomery lazy butterfly (hand-written product + the production range handling)
template <int WHICH>
__global__ void __launch_bounds__(256) rate_kernel(uint64_t *buf, int iters, uint64_t q, uint64_t qinv, uint32_t D, uint64_t w, uint64_t wm) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a0 = buf[i * 4], a1 = buf[i * 4 + 1], a2 = buf[i * 4 + 2], a3 = buf[i * 4 + 3];
    const uint64_t twoq = 2 * q;
    for (int k = 0; k < iters; k++) {
        if constexpr (WHICH == 0) {
            a0 = solinas_mul(a0, w, D); a1 = solinas_mul(a1, w, D); a2 = solinas_mul(a2, w, D); a3 = solinas_mul(a3, w, D);
        } else if constexpr (WHICH == 1) {  // two butterflies per iteration: (a0, a1), (a2, a3)
            uint64_t t = solinas_mul(a1, w, D), u = a0;
            a0 = addmod(u, t, D); a1 = submod(u, t, D);
            t = solinas_mul(a3, w, D); u = a2;
            a2 = addmod(u, t, D); a3 = submod(u, t, D);
        } else if constexpr (WHICH == 2) {
            a0 = mont_c(a0, wm, q, (uint32_t)(0 - qinv)); a1 = mont_c(a1, wm, q, (uint32_t)(0 - qinv));
            a2 = mont_c(a2, wm, q, (uint32_t)(0 - qinv)); a3 = mont_c(a3, wm, q, (uint32_t)(0 - qinv));
        } else if constexpr (WHICH == 3) {
            a0 = mont16(a0, wm, q, qinv); a1 = mont16(a1, wm, q, qinv); a2 = mont16(a2, wm, q, qinv); a3 = mont16(a3, wm, q, qinv);
        } else {  // the production forward butterfly on [0, 4q): U' = U - 2q if U >= 2q; X = U' + T; Y = U' + 2q - T, T = MRedLazy(V, w) in [0, 2q)
            uint64_t t = mont16(a1, wm, q, qinv), u = a0 >= twoq ? a0 - twoq : a0;
            a0 = u + t; a1 = u + twoq - t;
            t = mont16(a3, wm, q, qinv); u = a2 >= twoq ? a2 - twoq : a2;
            a2 = u + t; a3 = u + twoq - t;
        }
    }
    buf[i * 4] = a0; buf[i * 4 + 1] = a1; buf[i * 4 + 2] = a2; buf[i * 4 + 3] = a3;
}

int main() {
    // q = 2^k - c primes of the library's own chains (bench.py: the 55-bit special prime of c3 below 2^55; a 61-bit one of c4)
    struct { uint64_t q; int k; } ms[2] = {{36028797017456641ull, 55}, {2305843009211596801ull, 61}};
    std::mt19937_64 rng(7);
    for (auto m : ms) {
        const uint64_t q = m.q, c = ((uint64_t)1 << m.k) - q;
        const uint64_t D64 = c << (64 - m.k);
        if (c >> 31 || D64 >> 32) { std::printf("q = %llu: D does not fit 32 bits\n", (unsigned long long)q); continue; }
        const uint32_t D = (uint32_t)D64;
        const uint64_t qp = 0 - (uint64_t)D;  // q' = 2^64 - D
        const size_t n = 1 << 20;
        std::vector<uint64_t> a(n), b(n), w(n), out(3 * n);
        for (size_t i = 0; i < n; i++) {
            a[i] = i < 16 ? (i & 1 ? ~0ull - (i >> 1) : (i >> 1)) : rng();
            b[i] = i < 32 ? (i & 2 ? ~0ull - (i >> 2) : (i >> 2)) : rng();
            w[i] = i % 5 == 0 ? q - 1 : rng() % q;
        }
        uint64_t *da, *db, *dw, *dout;
        hipMalloc((void **)&da, n * 8); hipMalloc((void **)&db, n * 8); hipMalloc((void **)&dw, n * 8); hipMalloc((void **)&dout, 3 * n * 8);
        hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
        hipMemcpy(dw, w.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, da, db, dw, D, dout, n);
        hipMemcpy(out.data(), dout, 3 * n * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) {
            if (out[3 * i] % q != (uint64_t)(((u128)(a[i] % q) * w[i]) % q)) bad++;
            if (out[3 * i + 1] % q != (uint64_t)(((u128)(a[i] % q) + b[i] % q) % q)) bad++;
            if (out[3 * i + 2] % q != (uint64_t)(((u128)(a[i] % q) + q - b[i] % q) % q)) bad++;
        }
        std::printf("q = %llu = 2^%d - %llu, q' = 2^64 - %u (= %llu q): %zu of %zu product / sum / difference residues wrong\n", (unsigned long long)q, m.k,
                    (unsigned long long)c, D, (unsigned long long)(qp / q), bad, 3 * n);
        hipFree(da); hipFree(db); hipFree(dw); hipFree(dout);
    }
    const uint64_t q = ms[0].q;
    const uint32_t D = (uint32_t)((((uint64_t)1 << 55) - q) << 9);
    uint64_t qinv = 1;
    for (int i = 0; i < 6; i++) qinv *= 2 - q * qinv;
    const uint64_t w = 0x123456789abcdull % q, wm = (uint64_t)(((u128)w << 64) % q);
    const size_t threads = 256 * 256 * 16;
    uint64_t *buf;
    hipMalloc((void **)&buf, threads * 4 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[5] = {"Solinas product, full-range representatives (hipcc)   ", "Solinas butterfly: product + add + subtract (hipcc)    ",
                            "Montgomery product, two 32-bit rounds (hipcc)          ", "Montgomery product, 16 VALU + 4 wait states (by hand) ",
                            "Montgomery lazy butterfly (production: by hand + range) "};
    for (int which = 0; which < 5; which++) {
        const int iters = 2000;
        float ms_ = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(buf, 1, threads * 4 * 8);
            hipEventRecord(e0, 0);
            switch (which) {
                case 0: hipLaunchKernelGGL(rate_kernel<0>, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, buf, iters, q, qinv, D, w, wm); break;
                case 1: hipLaunchKernelGGL(rate_kernel<1>, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, buf, iters, q, qinv, D, w, wm); break;
                case 2: hipLaunchKernelGGL(rate_kernel<2>, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, buf, iters, q, qinv, D, w, wm); break;
                case 3: hipLaunchKernelGGL(rate_kernel<3>, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, buf, iters, q, qinv, D, w, wm); break;
                default: hipLaunchKernelGGL(rate_kernel<4>, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, buf, iters, q, qinv, D, w, wm); break;
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_, e0, e1);
        }
        const double per = which == 1 || which == 4 ? 2.0 : 4.0;  // butterflies (one product each) or products per thread and iteration
        std::printf("%s: %.3e %s/s\n", names[which], (double)threads * per * iters / (ms_ * 1e-3), which == 1 || which == 4 ? "butterflies" : "products");
    }
    return 0;
}

// shoup_probe.hip -- a 12-slot Shoup product for the integer-class moduli (q < 2^61), against the 16-instruction Montgomery
// column product of csrc/kernels.hip (mred_lazy_col_asm): is it right, and how fast does it issue?
//
//   r = a w - c q  (mod 2^64),  c ~ floor(a w' / 2^64),  w' = floor(w 2^64 / q) precomputed beside the twiddle w < q
//
// * the quotient estimate drops the a0 w'0 partial product: c is short by at most 1, so r lies in [0, 3q) for ANY 64-bit a
//   (a butterfly's multiplicand needs no range correction at all);
// * a w - c q is ONE chain of v_mad_u64_u32 on the negated modulus nq = 2^64 - q: the low cross words a0 w1 + a1 w0 + c0 nq1 +
//   c1 nq0 accumulate in the low half of a pair whose high half is never read, the two full products a0 w0 + c0 nq0 in another.
// 11 VALU instructions + one wait state (the carry of the quotient's middle column), against 16 + 4.
//   hipcc --offload-arch=gfx950 -O3 tools/shoup_probe.hip -o /tmp/shoup_probe && /tmp/shoup_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

__device__ __forceinline__ uint64_t shoup12(uint64_t a, uint64_t w, uint64_t wp, uint64_t nq) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t p0 = (uint32_t)wp, p1 = (uint32_t)(wp >> 32), n0 = (uint32_t)nq, n1 = (uint32_t)(nq >> 32);
    uint32_t rl, rh;
    asm("v_mad_u64_u32 v[100:101], vcc, %[a1], %[p0], 0\n\t"                 // M = a1 w'0
        "v_mad_u64_u32 v[100:101], vcc, %[a0], %[p1], v[100:101]\n\t"        // M += a0 w'1, carry -> vcc
        "v_mov_b32 v102, v101\n\t"                                           // T = {M.hi, carry}
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 v103, vcc, 0, 0, vcc\n\t"
        "v_mad_u64_u32 v[104:105], vcc, %[a1], %[p1], v[102:103]\n\t"        // C = a1 w'1 + T  (the quotient estimate)
        "v_mul_lo_u32 v106, %[a0], %[w1]\n\t"                                // X.lo = a0 w1 (low word; X.hi is never read)
        "v_mad_u64_u32 v[106:107], vcc, %[a1], %[w0], v[106:107]\n\t"        // X += a1 w0
        "v_mad_u64_u32 v[106:107], vcc, v104, %[n1], v[106:107]\n\t"         // X += c0 nq1
        "v_mad_u64_u32 v[106:107], vcc, v105, %[n0], v[106:107]\n\t"         // X += c1 nq0
        "v_mad_u64_u32 v[108:109], vcc, %[a0], %[w0], 0\n\t"                 // R = a0 w0
        "v_mad_u64_u32 v[108:109], vcc, v104, %[n0], v[108:109]\n\t"         // R += c0 nq0
        "v_add_u32 v109, v109, v106\n\t"                                     // R.hi += X.lo
        "v_mov_b32 %[rl], v108\n\tv_mov_b32 %[rh], v109"                     // (hand-off only: a kernel would use v[108:109] in place)
        : [rl] "=&v"(rl), [rh] "=&v"(rh)
        : [a0] "v"(a0), [a1] "v"(a1), [w0] "v"(w0), [w1] "v"(w1), [p0] "v"(p0), [p1] "v"(p1), [n0] "s"(n0), [n1] "s"(n1)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "vcc");
    return ((uint64_t)rh << 32) | rl;
}
// the production Montgomery product, as csrc/kernels.hip has it (result (x w + m q) / 2^64 in [0, 2q), x < 4q)
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

__global__ void check_kernel(const uint64_t *a, const uint64_t *w, const uint64_t *wp, uint64_t nq, uint64_t *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = shoup12(a[i], w[i], wp[i], nq);
}
template <int WHICH>
__global__ void __launch_bounds__(256) rate_kernel(uint64_t *buf, int iters, uint64_t q, uint64_t qinv, uint64_t nq, uint64_t w, uint64_t wp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a0 = buf[i * 4], a1 = buf[i * 4 + 1], a2 = buf[i * 4 + 2], a3 = buf[i * 4 + 3];
    for (int k = 0; k < iters; k++) {
        if constexpr (WHICH == 0) {
            a0 = shoup12(a0, w, wp, nq); a1 = shoup12(a1, w, wp, nq); a2 = shoup12(a2, w, wp, nq); a3 = shoup12(a3, w, wp, nq);
        } else {
            a0 = mont16(a0, w, q, qinv); a1 = mont16(a1, w, q, qinv); a2 = mont16(a2, w, q, qinv); a3 = mont16(a3, w, q, qinv);
        }
    }
    buf[i * 4] = a0; buf[i * 4 + 1] = a1; buf[i * 4 + 2] = a2; buf[i * 4 + 3] = a3;
}

int main() {
    typedef unsigned __int128 u128;
    const uint64_t qs[3] = {0x7fffffffe90001ull /* 55 bits */, 0xffffffffffc0001ull /* 60 bits */, 0x1fffffffffe00001ull /* 61 bits */};
    std::mt19937_64 rng(7);
    for (uint64_t q : qs) {
        const size_t n = 1 << 20;
        std::vector<uint64_t> a(n), w(n), wp(n), out(n);
        for (size_t i = 0; i < n; i++) {
            a[i] = i < 8 ? (i & 1 ? ~0ull : 0ull) + (i >> 1) : rng();  // any 64-bit multiplicand, the extremes included
            w[i] = i % 5 == 0 ? q - 1 : rng() % q;
            wp[i] = (uint64_t)(((u128)w[i] << 64) / q);
        }
        uint64_t *da, *dw, *dp, *dout;
        hipMalloc((void **)&da, n * 8); hipMalloc((void **)&dw, n * 8); hipMalloc((void **)&dp, n * 8); hipMalloc((void **)&dout, n * 8);
        hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), n * 8, hipMemcpyHostToDevice);
        hipMemcpy(dp, wp.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, da, dw, dp, 0 - q, dout, n);
        hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost);
        size_t bad = 0, over = 0;
        uint64_t mx = 0;
        for (size_t i = 0; i < n; i++) {
            const uint64_t want = (uint64_t)(((u128)a[i] * w[i]) % q);
            if (out[i] % q != want) bad++;
            if (out[i] >= 3 * q) over++;
            if (out[i] > mx) mx = out[i];
        }
        std::printf("q = %#llx: %zu of %zu residues wrong, %zu outside [0, 3q), largest result %.3f q\n", (unsigned long long)q, bad, n, over, (double)mx / (double)q);
        hipFree(da); hipFree(dw); hipFree(dp); hipFree(dout);
    }
    // issue rate: four independent chains per thread, as he_probe_modmul
    const uint64_t q = qs[0];
    uint64_t qinv = 1;
    for (int i = 0; i < 6; i++) qinv *= 2 - q * qinv;
    const uint64_t w = 0x123456789abcdull % q, wp = (uint64_t)(((u128)w << 64) / q);
    const size_t threads = 256 * 256 * 16;
    uint64_t *buf;
    hipMalloc((void **)&buf, threads * 4 * 8);
    hipMemset(buf, 1, threads * 4 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        const int iters = 2000;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, buf, iters, q, qinv, 0 - q, w, wp);
            else hipLaunchKernelGGL(rate_kernel<1>, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, buf, iters, q, qinv, 0 - q, w, wp);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::printf("%s: %.3e products/s\n", which == 0 ? "Shoup, 11 VALU + 1 wait state  " : "Montgomery, 16 VALU + 4 wait states", (double)threads * 4 * iters / (ms * 1e-3));
    }
    return 0;
}

// instr_probe.hip -- issue-rate probe for the integer instructions the 64-bit modular multiply
// is made of (gfx950).  Not part of the product; run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/instr_probe.hip -o /tmp/instr_probe && /tmp/instr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int WHICH>
__global__ void __launch_bounds__(256) probe(uint32_t *out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 0x9e3779b9u + threadIdx.x;
    uint64_t c0 = a0, c1 = a1, c2 = a2, c3 = a3;
    for (int i = 0; i < iters; i++) {
        if constexpr (WHICH == 0) {  // v_mad_u64_u32, 4 independent chains
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %3, vcc, %4, %2, %3\n"
                              "v_mad_u64_u32 %5, vcc, %6, %2, %5\n v_mad_u64_u32 %7, vcc, %8, %2, %7\n"
                              : "+v"(c0), "+v"(a0), "+v"(b), "+v"(c1), "+v"(a1), "+v"(c2), "+v"(a2), "+v"(c3), "+v"(a3)::"vcc");)
        } else if constexpr (WHICH == 1) {  // v_mul_lo_u32
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if constexpr (WHICH == 2) {  // v_mul_hi_u32
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if constexpr (WHICH == 3) {  // v_add_u32
            REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if constexpr (WHICH == 4) {  // v_add_co_u32 + v_addc_co_u32 (64-bit add)
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
        } else if constexpr (WHICH == 5) {  // v_lshl_add_u64 (64-bit add in one op)
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(c0));)
        } else if constexpr (WHICH == 6) {  // v_mad_u32_u24
            REP8(asm volatile("v_mad_u32_u24 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_mad_u32_u24 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if constexpr (WHICH == 7) {  // v_cmp_ge_u64 + 2x v_cndmask
            REP8(asm volatile("v_cmp_ge_u64 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc\n v_cmp_ge_u64 vcc, %2, %0\n v_cndmask_b32 %3, %3, %1, vcc\n"
                              : "+v"(c0), "+v"(a0), "+v"(c1), "+v"(a1)::"vcc");)
        } else if constexpr (WHICH == 8) {  // v_fma_f64
            double d0 = __longlong_as_double(c0 | 0x3ff0000000000000ull), d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));)
            c0 = __double_as_longlong(d0 + d1 + d2 + d3);
        } else if constexpr (WHICH >= 10 && WHICH <= 16) {  // double-precision neighbours of v_fma_f64 and the conversions
            double d0 = __longlong_as_double(c0 | 0x3ff0000000000000ull), d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
            if constexpr (WHICH == 10) {
                REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                                  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));)
            } else if constexpr (WHICH == 11) {
                REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                                  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));)
            } else if constexpr (WHICH == 12) {
                REP8(asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n"
                                  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
            } else if constexpr (WHICH == 13) {
                REP8(asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7\n"
                                  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
            } else if constexpr (WHICH == 14) {
                REP8(asm volatile("v_cvt_u32_f64 %0, %4\n v_cvt_u32_f64 %1, %5\n v_cvt_u32_f64 %2, %6\n v_cvt_u32_f64 %3, %7\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));)
            } else if constexpr (WHICH == 15) {
                REP8(asm volatile("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4\n"
                                  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0));)
            } else {
                REP8(asm volatile("v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3\n"
                                  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
            }
            c0 = __double_as_longlong(d0 + d1 + d2 + d3);
        } else if constexpr (WHICH >= 17 && WHICH <= 21) {  // cheap 32-bit / 64-bit shuffling ops
            if constexpr (WHICH == 17) {
                REP8(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
            } else if constexpr (WHICH == 18) {
                REP8(asm volatile("v_alignbit_b32 %0, %0, %4, 29\n v_alignbit_b32 %1, %1, %4, 29\n v_alignbit_b32 %2, %2, %4, 29\n v_alignbit_b32 %3, %3, %4, 29\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
            } else if constexpr (WHICH == 19) {
                REP8(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
            } else if constexpr (WHICH == 20) {
                REP8(asm volatile("v_lshrrev_b64 %0, 29, %0\n v_lshrrev_b64 %1, 29, %1\n v_lshrrev_b64 %2, 29, %2\n v_lshrrev_b64 %3, 29, %3\n"
                                  : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));)
            } else {
                REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
            }
        } else if constexpr (WHICH == 9) {  // v_mul_u32_u24 / v_mul_hi_u32_u24
            REP8(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)c0 ^ (uint32_t)c1 ^ (uint32_t)c2 ^ (uint32_t)c3 ^ b;
}

template <int W>
void run(const char *name, uint32_t *d) {
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<W><<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    probe<W><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks*4 waves spread on 1024 SIMDs
    double wave_instr_per_simd = (double)blocks * 4 / 1024.0 * iters * 32;
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction (at 2.4 GHz nominal)\n", name, ms, cycles / wave_instr_per_simd);
}

int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_mad_u64_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_u32", d); run<3>("v_add_u32", d);
    run<4>("v_add_co/addc pair (per op)", d); run<5>("v_lshl_add_u64", d); run<6>("v_mad_u32_u24", d);
    run<7>("v_cmp_ge_u64+cndmask (per op)", d); run<8>("v_fma_f64", d); run<9>("v_mul_u32_u24/hi_u24", d);
    run<10>("v_add_f64", d); run<11>("v_mul_f64", d); run<12>("v_rndne_f64", d); run<13>("v_cvt_f64_u32", d);
    run<14>("v_cvt_u32_f64", d); run<15>("v_ldexp_f64", d); run<16>("v_floor_f64", d); run<17>("v_and_b32", d);
    run<18>("v_alignbit_b32", d); run<19>("v_mov_b32", d); run<20>("v_lshrrev_b64", d); run<21>("v_cndmask_b32", d);
    return 0;
}

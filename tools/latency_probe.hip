// latency_probe.hip -- dependent-issue latency of the VALU instructions the modular products are made of (gfx950).
// For each instruction: cycles per instruction of ONE wave running NCHAIN independent dependency chains (s_memtime around an
// unrolled loop), for NCHAIN = 1, 2, 4, 8, alone on its SIMD and with a second wave on the same SIMD.  Not part of the product:
//   hipcc --offload-arch=gfx950 -O3 tools/latency_probe.hip -o /tmp/latency_probe && /tmp/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// OP: 0 v_fma_f64, 1 v_mul_f64, 2 v_add_f64, 3 v_rndne_f64, 4 v_mad_u64_u32, 5 v_mul_hi_u32, 6 v_mul_lo_u32, 7 v_add_co/addc pair,
//     8 v_lshl_add_u64
template <int OP, int NCHAIN>
__global__ void __launch_bounds__(512) probe(uint64_t *out, int iters) {
    double d[8];
    uint64_t c[8];
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { d[i] = 1.0 + threadIdx.x * 1e-9 + i; c[i] = threadIdx.x + i; a[i] = threadIdx.x * 2654435761u + i; }
    double w = 1.0000001;
    uint32_t b = 0x9e3779b9u + threadIdx.x;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        REP16(
            _Pragma("unroll") for (int i = 0; i < NCHAIN; i++) {
                if constexpr (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(w));
                else if constexpr (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(w));
                else if constexpr (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(w));
                else if constexpr (OP == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
                else if constexpr (OP == 4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b) : "vcc");
                else if constexpr (OP == 5) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if constexpr (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if constexpr (OP == 7) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(b) : "vcc");
                else if constexpr (OP == 8) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[i]) : "v"(c[7]));
            })
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += (uint64_t)__double_as_longlong(d[i]) ^ c[i] ^ a[i];
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t1 - t0;
    if (s == 0x1234567) out[1] = s;
}

template <int OP, int NCHAIN>
double run(uint64_t *dbuf, int threads) {
    const int blocks = 256, iters = 200;
    probe<OP, NCHAIN><<<blocks, threads>>>(dbuf, iters);
    hipDeviceSynchronize();
    const int waves = blocks * threads / 64;
    std::vector<uint64_t> h(2 * waves);
    hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int i = 0; i < waves; i++) v.push_back((double)h[2 * i]);
    std::sort(v.begin(), v.end());
    const double n_instr = (double)iters * 16 * NCHAIN * (OP == 7 ? 2 : 1);
    return v[v.size() / 2] / n_instr;
}

template <int OP>
void row(const char *name, uint64_t *dbuf) {
    // 256 threads = 4 waves = one per SIMD; 512 threads = two per SIMD
    printf("%-16s 1 wave/SIMD: chains 1/2/4/8 = %5.2f %5.2f %5.2f %5.2f   2 waves/SIMD: %5.2f %5.2f %5.2f %5.2f  cycles per instruction per wave\n", name,
           run<OP, 1>(dbuf, 256), run<OP, 2>(dbuf, 256), run<OP, 4>(dbuf, 256), run<OP, 8>(dbuf, 256),
           run<OP, 1>(dbuf, 512), run<OP, 2>(dbuf, 512), run<OP, 4>(dbuf, 512), run<OP, 8>(dbuf, 512));
}

int main() {
    uint64_t *dbuf;
    hipMalloc(&dbuf, 256 * 8 * 2 * 8 + 64);
    row<0>("v_fma_f64", dbuf); row<1>("v_mul_f64", dbuf); row<2>("v_add_f64", dbuf); row<3>("v_rndne_f64", dbuf);
    row<4>("v_mad_u64_u32", dbuf); row<5>("v_mul_hi_u32", dbuf); row<6>("v_mul_lo_u32", dbuf); row<7>("v_add_co+addc", dbuf);
    row<8>("v_lshl_add_u64", dbuf);
    return 0;
}

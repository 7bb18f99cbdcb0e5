// modarith.h -- 64-bit modular arithmetic primitives shared by the host-side table
// builders and the gfx950 kernels of libhering.
//
// Word-level semantics follow the reference's ring/modular_reduction.go (cited per
// function) because several ring ops return *lazy* representatives whose exact
// 64-bit word is observable through the API (SURVEY.md section 8a note); the
// kernels are otherwise free to use their own reduction schedule.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HE_HD __host__ __device__ __forceinline__
#else
#define HE_HD inline
#endif

namespace he {

typedef unsigned __int128 u128;

// 64x64 -> 128 multiply.  On gfx950 this lowers to four v_mad_u64_u32 (no native
// 64-bit multiplier); on the host to one MUL.
HE_HD void mul64wide(uint64_t a, uint64_t b, uint64_t &hi, uint64_t &lo) {
    u128 m = (u128)a * b;
    hi = (uint64_t)(m >> 64);
    lo = (uint64_t)m;
}
HE_HD uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) >> 64); }

// MRedLazy: x*y*2^-64 mod q in [0, 2q)      (ring/modular_reduction.go:90-95)
HE_HD uint64_t mred_lazy(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {
    uint64_t ahi, alo;
    mul64wide(x, y, ahi, alo);
    uint64_t H = mulhi64(alo * qinv, q);
    return ahi - H + q;
}
// MRed: canonical                            (ring/modular_reduction.go:78-86)
HE_HD uint64_t mred(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {
    uint64_t r = mred_lazy(x, y, q, qinv);
    return r >= q ? r - q : r;
}
// Montgomery reduction of a 128-bit value (hi,lo) -> [0, 2q) when hi < q.
HE_HD uint64_t mred128_lazy(uint64_t hi, uint64_t lo, uint64_t q, uint64_t qinv) {
    uint64_t H = mulhi64(lo * qinv, q);
    return hi - H + q;
}
// CRed                                        (ring/modular_reduction.go:200-205)
HE_HD uint64_t cred(uint64_t a, uint64_t q) { return a >= q ? a - q : a; }
// BRedAddLazy / BRedAdd: a mod q for any 64-bit a (ring/modular_reduction.go:110-124)
HE_HD uint64_t bred_add_lazy(uint64_t a, uint64_t q, uint64_t brc0) { return a - mulhi64(a, brc0) * q; }
HE_HD uint64_t bred_add(uint64_t a, uint64_t q, uint64_t brc0) { return cred(bred_add_lazy(a, q, brc0), q); }
// BRedLazy / BRed: x*y mod q with the 128-bit Barrett constant (ring/modular_reduction.go:127-196)
HE_HD uint64_t bred_lazy(uint64_t x, uint64_t y, uint64_t q, uint64_t brc0, uint64_t brc1) {
    uint64_t mhi, mlo, hhi, hlo;
    mul64wide(x, y, mhi, mlo);
    uint64_t r = mhi * brc0;
    mul64wide(mlo, brc0, hhi, hlo);
    r += hhi;
    uint64_t lhi = mulhi64(mlo, brc1);
    uint64_t s0 = hlo + lhi;
    r += (uint64_t)(s0 < hlo);
    mul64wide(mhi, brc1, hhi, hlo);
    r += hhi;
    uint64_t s1 = hlo + s0;
    r += (uint64_t)(s1 < hlo);
    return mlo - r * q;
}
HE_HD uint64_t bred(uint64_t x, uint64_t y, uint64_t q, uint64_t brc0, uint64_t brc1) {
    return cred(bred_lazy(x, y, q, brc0, brc1), q);
}
// MFormLazy / MForm: a*2^64 mod q            (ring/modular_reduction.go:11-45)
HE_HD uint64_t mform_lazy(uint64_t a, uint64_t q, uint64_t brc0, uint64_t brc1) {
    uint64_t mhi = mulhi64(a, brc1);
    return (uint64_t)(0 - (a * brc0 + mhi)) * q;
}
HE_HD uint64_t mform(uint64_t a, uint64_t q, uint64_t brc0, uint64_t brc1) { return cred(mform_lazy(a, q, brc0, brc1), q); }
// IMFormLazy / IMForm: a*2^-64 mod q         (ring/modular_reduction.go:49-65)
HE_HD uint64_t imform_lazy(uint64_t a, uint64_t q, uint64_t qinv) { return q - mulhi64(a * qinv, q); }
HE_HD uint64_t imform(uint64_t a, uint64_t q, uint64_t qinv) { return cred(imform_lazy(a, q, qinv), q); }

// Per-modulus constants as the kernels see them (one 80-byte record per RNS limb).
struct ModConst {
    uint64_t q;      // Modulus
    uint64_t qinv;   // MRedConstant = q^-1 mod 2^64
    uint64_t brc0;   // BRedConstant[0] = floor(2^128/q) >> 64
    uint64_t brc1;   // BRedConstant[1] = floor(2^128/q) mod 2^64
    uint64_t ninv;   // NInv = MForm(N^-1)
    uint64_t r2;     // 2^128 mod q (MForm(x) = MRed(x, r2)), kernels only
    uint64_t pad0, pad1;
    double rq;       // 1.0 / (double)q, IEEE-rounded (the f64 kernels would otherwise each spend ~28 VALU ops on the division)
    uint64_t pad2;   // keeps the record a multiple of 16 bytes
};

}  // namespace he

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
// The same residue by two 32-bit reduction rounds (word-serial Montgomery, "CIOS"): x*y*2^-64 mod q in [0, 2q) for
// q < 2^61 (the largest modulus the library accepts, as the reference), x + q < 2^64 (any lazy operand of the kernels: x < 4q)
// and y < q.  gfx950 has no 64-bit multiplier; the full-width form
// above costs 11 multiplies and ~15 carry / select instructions (a 64x64->128 product, a 64-bit low product and a 64x64 high
// product), this one 8 v_mad_u64_u32 + 2 v_mul_lo_u32 + ~6 adds: the running sum never exceeds 96 bits and every partial
// product is a 32x32+64 multiply-add.  m = -T q^-1 mod 2^64 is the same number either way (computed word by word here), so the
// result is the same element of [0, 2q) as mred_lazy's EXCEPT when T q^-1 = 0 mod 2^64, where the reference's formula returns
// r + q: use it where only the residue class matters (butterflies, canonical MRed), never for an API-visible *Lazy word.
HE_HD uint64_t mred_lazy_w32(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32);
    const uint32_t q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32), nq = (uint32_t)(0 - qinv);  // -q^-1 mod 2^32
    // round 0: T = x * y0 (96 bits) ; T' = (T + m q) / 2^32
    const uint64_t t = (uint64_t)x0 * y0;
    const uint64_t u = (uint64_t)x1 * y0 + (t >> 32);           // T = u 2^32 + lo32(t)
    uint32_t m = (uint32_t)t * nq;
    uint64_t c = (uint64_t)m * q0 + (uint32_t)t;                // low word cancels
    const uint64_t v = (uint64_t)m * q1 + u + (c >> 32);        // T' < 2^64 (see the bounds above)
    // round 1: T'' = T' + x * y1 2^32-aligned ; r = (T'' + m q) / 2^32
    const uint64_t a = (uint64_t)x0 * y1 + (uint32_t)v;         // T'' = b 2^32 + lo32(a)
    const uint64_t b = (uint64_t)x1 * y1 + (a >> 32) + (v >> 32);
    m = (uint32_t)a * nq;
    c = (uint64_t)m * q0 + (uint32_t)a;
    return (uint64_t)m * q1 + b + (c >> 32);
}
HE_HD uint64_t mred_w32(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {  // canonical: equals mred
    const uint64_t r = mred_lazy_w32(x, y, q, qinv);
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

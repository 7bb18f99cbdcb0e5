// host_math.h -- host-side (CPU, set-up only) construction of the per-ring constants the
// kernels consume.  These are the *mathematical definitions* of the reference's tables
//  computed with plain 128-bit modular arithmetic; the
// resulting words must equal the reference's (ring/subring.go:99-159, ring/ring.go:329-346,
// ring/basis_extension.go:25-49,101-172) because NTT outputs depend on the choice of psi
// and lazy ModUp outputs on the constants.  Independent of oracle/ by construction.
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

#include "modarith.h"

namespace he {

uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m);
uint64_t powmod(uint64_t a, uint64_t e, uint64_t m);
uint64_t invmod(uint64_t a, uint64_t p);  // p prime
bool is_prime_u64(uint64_t n);
std::vector<uint64_t> unique_prime_factors(uint64_t n);
uint64_t to_mont(uint64_t a, uint64_t q);  // a * 2^64 mod q

struct SubRingHost {
    ModConst mc;
    uint64_t primroot;
    std::vector<uint64_t> roots_fwd, roots_bwd;  // [N]
};
// returns false and fills err if (N, q) does not define an NTT-enabled SubRing
bool build_subring(int logN, uint64_t q, SubRingHost &out, std::string &err);
// Conjugate-invariant SubRing Z[X+X^-1]/(X^2N+1) (ring/ring.go:260, NthRoot = 4N).  The transform of
// ring/ntt.go:757-786 / :1104-1152 is a fold with roots[1] around a standard butterfly network whose stage
// with h blocks uses roots4N[2h+i]; roots_fwd/bwd are returned already remapped to that network
// (tw[h+i] = roots4N[2h+i], N entries), the fold twiddles in mc.pad0 (forward) / mc.pad1 (backward),
// and mc.ninv = MForm((2N)^-1).
bool build_subring_ci(int logN, uint64_t q, SubRingHost &out, std::string &err);

// RescaleConstants[j-1][i] = MForm(q_i - q_j^-1 mod q_i), i < j   (ring/ring.go:329-346)
std::vector<std::vector<uint64_t>> build_rescale_constants(const std::vector<uint64_t> &moduli);

// GenModUpConstants(S, D): a[i], T[j][i], vt[j][v]  (ring/basis_extension.go:101-172)
struct ModUpHost {
    int nsrc, ndst;
    std::vector<uint64_t> a, T, vt;
};
ModUpHost build_modup_constants(const std::vector<uint64_t> &S, const std::vector<uint64_t> &D);

// floor(prod(S)/2) mod m  (S all odd)
uint64_t half_product_mod(const std::vector<uint64_t> &S, uint64_t m);
// (prod(S))^-1 mod q, Montgomery form  (ring/basis_extension.go:25-49)
uint64_t inv_product_mont(const std::vector<uint64_t> &S, uint64_t q);
// multiword (little-endian) scalar mod q
uint64_t words_mod(const uint64_t *w, int n, uint64_t q);

}  // namespace he

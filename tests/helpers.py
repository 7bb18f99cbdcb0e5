"""Shared helpers for the parity tests (host-side, numpy + Python big integers)."""
from __future__ import annotations

import numpy as np

MASK64 = (1 << 64) - 1


def prod(xs):
    r = 1
    for x in xs:
        r *= int(x)
    return r


def rng_for(config_index: int):
    """PCG64 seeded with 0x1A77160 + config_index.  HERING_SEED_OFFSET shifts every seed (soak runs of
    the parity suites on fresh random inputs; golden-vector tests do not draw from here)."""
    import os
    return np.random.Generator(np.random.PCG64(0x1A77160 + config_index + 1000003 * int(os.environ.get("HERING_SEED_OFFSET", "0"))))


def uniform_poly(rng, moduli, N):
    """[limbs, N] with limb i uniform in [0, q_i)."""
    out = np.empty((len(moduli), N), dtype=np.uint64)
    for i, q in enumerate(moduli):
        out[i] = rng.integers(0, int(q), size=N, dtype=np.uint64)
    return out


def rand_bigints(rng, bound: int, n: int):
    """n uniform Python ints in [0, bound)."""
    nbytes = (bound.bit_length() + 7) // 8 + 8
    out = []
    for _ in range(n):
        out.append(int.from_bytes(rng.bytes(nbytes), "little") % bound)
    return out


def set_coefficients_bigint(coeffs, moduli):
    """ring.SetCoefficientsBigint: limb i = coeff mod q_i (non-negative residue)."""
    out = np.empty((len(moduli), len(coeffs)), dtype=np.uint64)
    for i, q in enumerate(moduli):
        q = int(q)
        out[i] = np.array([c % q for c in coeffs], dtype=np.uint64)
    return out


def div_round(a: int, b: int) -> int:
    """utils/bignum/int.go:52-64 (round half away from zero, truncated quotient)."""
    sign = (1 if a >= 0 else -1) * (1 if b >= 0 else -1)
    qt = abs(a) // abs(b) * sign
    r = a - qt * b
    if 2 * abs(r) >= abs(b):
        qt += 1 if (a >= 0) == (b >= 0) else -1
    return qt


def bitrev(x: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r

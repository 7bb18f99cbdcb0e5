"""Host-side RLWE key material for the semantic (decrypt-and-check) tests.

Restates, with OUR OWN seeded PRNG (the reference's blake2b-keyed sampler lives
in an un-vendored dependency), the key shapes of:
  * secret key: ternary, NTT + Montgomery         core/rlwe/keygenerator.go:58-70
  * zero-encryption in QP                         core/rlwe/encryptor.go:406-435
  * evaluation key skIn -> skOut                  core/rlwe/keygenerator.go:287-330
  * gadget P * skIn on the digit's Q-limbs        core/rlwe/gadgetciphertext.go:172-241
All arithmetic goes through the CPU oracle (test infrastructure).
"""
from __future__ import annotations

import numpy as np

from oracle import oracle as O
from tests.helpers import prod


def small_to_rns(vals, moduli):
    """signed small integers -> [limbs, N] residues."""
    out = np.empty((len(moduli), len(vals)), dtype=np.uint64)
    v = np.asarray(vals, dtype=np.int64)
    for i, q in enumerate(moduli):
        out[i] = np.where(v < 0, np.uint64(int(q)) - (-v).astype(np.uint64), v.astype(np.uint64))
    return out


class SecretKey:
    def __init__(self, rng, ringQ: O.Ring, ringP: O.Ring, vals=None):
        N = ringQ.N
        self.vals = rng.integers(-1, 2, size=N) if vals is None else np.asarray(vals)
        self.Q = ringQ.unop("MForm", ringQ.NTT(small_to_rns(self.vals, ringQ.moduli)))
        # ringP = None: parameters without special primes (core/rlwe/keygenerator.go:64 `if levelP > -1`)
        self.P = ringP.unop("MForm", ringP.NTT(small_to_rns(self.vals, ringP.moduli))) if ringP is not None else None


def automorphism_secret(rng, ringQ, ringP, sk: SecretKey, galel: int) -> SecretKey:
    """pi_galel(sk) in the coefficient domain (ring/automorphism.go:113); on a conjugate-invariant ring in the NTT domain, as
    the reference's key generator does for either type (core/rlwe/keygenerator.go:158-169)."""
    if getattr(ringQ, "conjugate_invariant", False):
        out = SecretKey.__new__(SecretKey)
        out.vals = None
        out.Q = ringQ.AutomorphismNTTWithIndex(sk.Q, ringQ.AutomorphismNTTIndex(galel))
        out.P = ringP.AutomorphismNTTWithIndex(sk.P, ringP.AutomorphismNTTIndex(galel)) if ringP is not None else None
        return out
    N = ringQ.N
    out = np.zeros(N, dtype=np.int64)
    for i in range(N):
        raw = i * galel
        idx = raw & (N - 1)
        sign = -1 if (raw >> (N.bit_length() - 1)) & 1 else 1
        out[idx] = sign * sk.vals[i]
    return SecretKey(rng, ringQ, ringP, vals=out)


def gen_evaluation_key(rng, ringQ: O.Ring, ringP: O.Ring, sk_in_Q: np.ndarray, sk_out: SecretKey,
                       sigma: float = 3.2) -> O.EvaluationKey:
    """evk[d][k] (k=0: b, k=1: a), NTT + Montgomery, at max levels."""
    N = ringQ.N
    LQ, LP = len(ringQ.moduli), len(ringP.moduli)
    levelQ, levelP = LQ - 1, LP - 1
    beta = O.BaseRNSDecompositionVectorSize(levelQ, levelP)
    P = prod(ringP.moduli)
    kq = np.zeros((beta, 2, LQ, N), dtype=np.uint64)
    kp = np.zeros((beta, 2, LP, N), dtype=np.uint64)
    p_times_skin = ringQ.MulScalarBigint(sk_in_Q, P)
    for d in range(beta):
        e = np.clip(np.rint(rng.normal(0.0, sigma, size=N)), -19, 19).astype(np.int64)
        aQ = np.stack([rng.integers(0, int(q), size=N, dtype=np.uint64) for q in ringQ.moduli])
        aP = np.stack([rng.integers(0, int(q), size=N, dtype=np.uint64) for q in ringP.moduli])
        bQ = ringQ.unop("MForm", ringQ.NTT(small_to_rns(e, ringQ.moduli)))
        bP = ringP.unop("MForm", ringP.NTT(small_to_rns(e, ringP.moduli)))
        bQ = ringQ.binop("MulCoeffsMontgomeryThenSub", aQ, sk_out.Q, bQ)
        bP = ringP.binop("MulCoeffsMontgomeryThenSub", aP, sk_out.P, bP)
        lo, hi = d * LP, min((d + 1) * LP, LQ)
        tmp = ringQ.binop("Add", bQ, p_times_skin)
        bQ[lo:hi] = tmp[lo:hi]
        kq[d, 0], kq[d, 1], kp[d, 0], kp[d, 1] = bQ, aQ, bP, aP
    return O.EvaluationKey(kq, kp)


def centered(ring: O.Ring, poly_coeff: np.ndarray, limb: int = 0) -> np.ndarray:
    q = int(ring.moduli[limb])
    v = poly_coeff[limb].astype(object)
    return np.array([int(x) - q if int(x) > q // 2 else int(x) for x in v], dtype=object)


def phase(ringQ: O.Ring, ct: np.ndarray, skQ: np.ndarray) -> np.ndarray:
    """Decrypt degree-d NTT-domain ct by Horner (core/rlwe/decryptor.go:49-93); returns NTT-domain phase."""
    level = ct.shape[1] - 1
    s = skQ[: level + 1]
    acc = ct[-1].copy()
    sub = O.Ring(ringQ.N, ringQ.moduli[: level + 1], getattr(ringQ, "conjugate_invariant", False))
    for i in range(ct.shape[0] - 2, -1, -1):
        acc = sub.binop("Add", sub.binop("MulCoeffsMontgomery", acc, s), ct[i])
    return acc


def noise_log2(ringQ: O.Ring, ntt_poly: np.ndarray) -> float:
    level = ntt_poly.shape[0] - 1
    sub = O.Ring(ringQ.N, ringQ.moduli[: level + 1], getattr(ringQ, "conjugate_invariant", False))
    c = centered(sub, sub.INTT(ntt_poly), 0)
    m = max(abs(int(x)) for x in c)
    return float(np.log2(m)) if m > 0 else 0.0


def gen_evaluation_key_base2(rng, ringQ: O.Ring, ringP: O.Ring, sk_in_Q: np.ndarray, sk_out: SecretKey, pw2: int,
                             sigma: float = 3.2) -> O.EvaluationKey:
    """Base-2 gadget key (core/rlwe/gadgetciphertext.go:172-241 with BaseTwoDecomposition = pw2, one P limb, or none --
    ringP = None, `levelP != -1` :183 --): block (i, j) encrypts P * 2^(j*pw2) * skIn on Q-limb i only;
    nj[i] = ceil(bits(q_i) / pw2) (core/rlwe/params.go:523-540)."""
    N = ringQ.N
    LQ, LP = len(ringQ.moduli), (len(ringP.moduli) if ringP is not None else 0)
    assert LP <= 1
    nj = [(int(q).bit_length() + pw2 - 1) // pw2 for q in ringQ.moduli]
    D = sum(nj)
    P = prod(ringP.moduli) if LP else 1
    kq = np.zeros((D, 2, LQ, N), dtype=np.uint64)
    kp = np.zeros((D, 2, LP, N), dtype=np.uint64)
    blk = 0
    for i in range(LQ):
        for j in range(nj[i]):
            e = np.clip(np.rint(rng.normal(0.0, sigma, size=N)), -19, 19).astype(np.int64)
            aQ = np.stack([rng.integers(0, int(q), size=N, dtype=np.uint64) for q in ringQ.moduli])
            bQ = ringQ.unop("MForm", ringQ.NTT(small_to_rns(e, ringQ.moduli)))
            bQ = ringQ.binop("MulCoeffsMontgomeryThenSub", aQ, sk_out.Q, bQ)
            g = ringQ.MulScalarBigint(sk_in_Q, P << (j * pw2))
            bQ[i] = ringQ.binop("Add", bQ, g)[i]
            kq[blk, 0], kq[blk, 1] = bQ, aQ
            if LP:
                aP = np.stack([rng.integers(0, int(q), size=N, dtype=np.uint64) for q in ringP.moduli])
                bP = ringP.unop("MForm", ringP.NTT(small_to_rns(e, ringP.moduli)))
                kp[blk, 0], kp[blk, 1] = ringP.binop("MulCoeffsMontgomeryThenSub", aP, sk_out.P, bP), aP
            blk += 1
    return O.EvaluationKey(kq, kp, pw2=pw2, nj=nj)


def gen_galois_keys(rng, ringQ: O.Ring, ringP: O.Ring, sk: SecretKey, galels):
    """rlwe.KeyGenerator.GenGaloisKeysNew (core/rlwe/keygenerator.go:188-259): key for galEl switches
    pi_{galEl^-1}(sk)... -> sk, i.e. skIn = sk, skOut = pi_{galEl^-1}(sk)."""
    nth = 2 * ringQ.N
    out = {}
    for g in galels:
        if g in out:
            continue
        ginv = pow(int(g), nth - 1, nth)  # core/rlwe/params.go:587
        sk_out = automorphism_secret(rng, ringQ, ringP, sk, ginv)
        out[int(g)] = gen_evaluation_key(rng, ringQ, ringP, sk.Q, sk_out)
    return out


def small_plaintext_qp(rng, ringQ: O.Ring, ringP: O.Ring, bound: int = 8):
    """An 'encoded diagonal': a small polynomial in NTT + Montgomery form over Q and P."""
    vals = rng.integers(-bound, bound + 1, size=ringQ.N)
    return (ringQ.unop("MForm", ringQ.NTT(small_to_rns(vals, ringQ.moduli))),
            ringP.unop("MForm", ringP.NTT(small_to_rns(vals, ringP.moduli))))


def downstream_primes(bits: int, nth_root: int, n: int, avoid=()):
    """ring.NTTFriendlyPrimesGenerator.NextDownstreamPrimes (ring/primes.go): primes 2^bits + 1 - k*nth_root, k >= 1."""
    out, x = [], (1 << bits) + 1
    while len(out) < n:
        x -= nth_root
        if O.IsPrime(x) and x not in avoid:
            out.append(x)
    return out


def bfv_encrypt(rng, ringQ: O.Ring, sk: SecretKey, m, t: int, sigma: float = 3.2) -> np.ndarray:
    """(c0, c1) NTT with phase floor(Q/t) * m + e (BFV / the bgv package's MSB encoding at scale 1)."""
    Q = prod(ringQ.moduli)
    delta = Q // t
    N = ringQ.N
    e = np.clip(np.rint(rng.normal(0.0, sigma, size=N)), -19, 19).astype(np.int64)
    pt = [(int(mi) * delta + int(ei)) % Q for mi, ei in zip(m, e)]
    ptr = np.array([[x % int(qi) for x in pt] for qi in ringQ.moduli], dtype=np.uint64)
    c1 = np.stack([rng.integers(0, int(q), size=N, dtype=np.uint64) for q in ringQ.moduli])
    c0 = ringQ.binop("Sub", ringQ.NTT(ptr), ringQ.binop("MulCoeffsMontgomery", c1, sk.Q))
    return np.stack([c0, c1])


def bfv_decrypt(ringQ: O.Ring, ct: np.ndarray, sk: SecretKey, t: int):
    """round(t/Q * centred phase) mod t, coefficient domain"""
    Q = prod(ringQ.moduli)
    ph = ringQ.INTT(phase(ringQ, ct, sk.Q))
    w = [(Q // int(qi)) * pow(Q // int(qi), -1, int(qi)) for qi in ringQ.moduli]
    out = []
    for j in range(ringQ.N):
        x = sum(int(ph[i, j]) * w[i] for i in range(len(w))) % Q
        if x > Q // 2:
            x -= Q
        out.append(((2 * x * t + Q) // (2 * Q)) % t)
    return np.array(out, dtype=np.int64)


def negacyclic_mul_mod(a, b, t: int):
    N = len(a)
    full = np.convolve(np.asarray(a, dtype=object), np.asarray(b, dtype=object))
    out = [int(full[k]) - (int(full[k + N]) if k + N < len(full) else 0) for k in range(N)]
    return np.array([x % t for x in out], dtype=np.int64)


def bgv_encrypt(rng, ringQ: O.Ring, sk: SecretKey, m, t: int, scale: int = 1, sigma: float = 3.2) -> np.ndarray:
    """BGV ciphertext as the reference's bgv package holds it (schemes/bgv/encoder.go:395-406): phase =
    centred(m * scale mod t) * (T^-1 mod Q) + e; m is a polynomial of R_t (coefficient domain)."""
    Q = prod(ringQ.moduli)
    tinv = pow(t, -1, Q)
    N = ringQ.N
    e = np.clip(np.rint(rng.normal(0.0, sigma, size=N)), -19, 19).astype(np.int64)
    pt = []
    for mi, ei in zip(m, e):
        v = int(mi) * scale % t
        v = v - t if v > t // 2 else v
        pt.append((v * tinv + int(ei)) % Q)
    ptr = np.array([[x % int(qi) for x in pt] for qi in ringQ.moduli], dtype=np.uint64)
    c1 = np.stack([rng.integers(0, int(q), size=N, dtype=np.uint64) for q in ringQ.moduli])
    c0 = ringQ.binop("Sub", ringQ.NTT(ptr), ringQ.binop("MulCoeffsMontgomery", c1, sk.Q[: len(ringQ.moduli)]))
    return np.stack([c0, c1])


def bgv_decrypt(ringQ: O.Ring, ct: np.ndarray, sk: SecretKey, t: int, scale: int = 1):
    """(T * phase mod Q, centred) mod t, divided by the scale"""
    Q = prod(ringQ.moduli)
    ph = ringQ.INTT(phase(ringQ, ct, sk.Q))
    w = [(Q // int(qi)) * pow(Q // int(qi), -1, int(qi)) for qi in ringQ.moduli]
    sinv = pow(scale, -1, t)
    out = []
    for j in range(ringQ.N):
        x = sum(int(ph[i, j]) * w[i] for i in range(len(w))) % Q
        y = x * t % Q
        if y > Q // 2:
            y -= Q
        out.append(y % t * sinv % t)
    return np.array(out, dtype=np.int64)


# ---- CKKS canonical embedding (test-side encoder: slots z_j = m(zeta_j), zeta_j = exp(i pi 5^j / N)) --------------------
def ckks_slot_roots(N: int) -> np.ndarray:
    g, out = 1, []
    for _ in range(N // 2):
        out.append(np.exp(1j * np.pi * g / N))
        g = g * 5 % (2 * N)
    return np.array(out)


def ckks_encode(z, N: int, scale, moduli) -> np.ndarray:
    """slots -> NTT-domain plaintext [limbs][N] at `scale` (real coefficients (2/N) Re(sum_j z_j conj(zeta_j)^k), rounded)"""
    zeta = ckks_slot_roots(N)
    k = np.arange(N)
    V = zeta[:, None] ** k[None, :]  # [N/2][N]
    coeffs = (2.0 / N) * np.real(np.conj(V).T @ np.asarray(z, dtype=complex))
    ints = [int(round(float(c) * float(scale))) for c in coeffs]
    res = np.array([[x % int(q) for x in ints] for q in moduli], dtype=np.uint64)
    return O.Ring(N, moduli).NTT(res)


def ckks_encrypt(rng, ringQ: O.Ring, sk: SecretKey, z, scale, sigma: float = 3.2) -> np.ndarray:
    N = ringQ.N
    pt = ckks_encode(z, N, scale, ringQ.moduli)
    e = np.clip(np.rint(rng.normal(0.0, sigma, size=N)), -19, 19).astype(np.int64)
    c1 = np.stack([rng.integers(0, int(q), size=N, dtype=np.uint64) for q in ringQ.moduli])
    c0 = ringQ.binop("Add", ringQ.binop("Sub", pt, ringQ.binop("MulCoeffsMontgomery", c1, sk.Q[: len(ringQ.moduli)])),
                     ringQ.NTT(small_to_rns(e, ringQ.moduli)))
    return np.stack([c0, c1])


def ckks_decrypt(ringQ: O.Ring, ct: np.ndarray, sk: SecretKey, scale) -> np.ndarray:
    Q = prod(ringQ.moduli)
    N = ringQ.N
    ph = ringQ.INTT(phase(ringQ, ct, sk.Q))
    w = [(Q // int(qi)) * pow(Q // int(qi), -1, int(qi)) for qi in ringQ.moduli]
    coeffs = np.empty(N)
    for j in range(N):
        x = sum(int(ph[i, j]) * w[i] for i in range(len(w))) % Q
        if x > Q // 2:
            x -= Q
        coeffs[j] = float(x / scale) if not isinstance(scale, float) else x / scale
    zeta = ckks_slot_roots(N)
    return (zeta[:, None] ** np.arange(N)[None, :]) @ coeffs

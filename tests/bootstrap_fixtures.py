"""Toy CKKS bootstrapping instance for the end-to-end tests (test infrastructure: numpy canonical embedding, dense DFT
matrices, our own seeded keys).  Full packing (N/2 slots): slots z = U w with U[j,k] = zeta_j^k, w = c_lo + i c_hi the two
halves of the plaintext coefficient vector, so CoeffsToSlots is multiplication by (a multiple of) U^-1 and SlotsToCoeffs by U."""
import numpy as np

from oracle import circuits as OC
from oracle import oracle as O
from tests.rlwe_fixtures import (SecretKey, ckks_slot_roots, gen_evaluation_key, gen_galois_keys, small_to_rns)


def encode_diag_qp(vec, N, scale, ringQ, ringP):
    """slot vector -> (Q, P) polynomials, NTT + Montgomery, at `scale`"""
    zeta = ckks_slot_roots(N)
    V = zeta[:, None] ** np.arange(N)[None, :]
    coeffs = (2.0 / N) * np.real(np.conj(V).T @ np.asarray(vec, dtype=complex))
    ints = np.array([int(round(float(c) * float(scale))) for c in coeffs], dtype=object)
    out = []
    for ring in (ringQ, ringP):
        res = np.array([[int(x) % int(q) for x in ints] for q in ring.moduli], dtype=np.uint64)
        out.append(ring.unop("MForm", ring.NTT(res)))
    return tuple(out)


def dense_matrix_lt(M, N, N1, scale, levelQ, ringQ, ringP):
    """BSGS linear transformation of the dense n x n complex matrix M acting on slots: (M z)_j = sum_k diag_k[j] z_{j+k};
    diagonal k = j0 + i is encoded pre-rotated by -j0 (lintrans.go:271-295)"""
    n = N // 2
    idx = np.arange(n)
    Vec = {}
    sub = O.Ring(N, ringQ.moduli[: levelQ + 1])
    for k in range(n):
        diag = M[idx, (idx + k) % n]
        if np.max(np.abs(diag)) < 1e-14:
            continue
        j0 = (k // N1) * N1
        Vec[k] = encode_diag_qp(np.roll(diag, j0), N, scale, sub, ringP)  # rotate right by j0 = left by -j0
    return OC.LinearTransformation(Vec, levelQ, len(ringP.moduli) - 1, n, N1)


def sparse_ternary(rng, N, h):
    v = np.zeros(N, dtype=np.int64)
    v[rng.choice(N, size=h, replace=False)] = rng.choice([-1, 1], size=h)
    return v


def special_fft_layers(N):
    """U = L_log(n) ... L_1 Bitrev with U[j,k] = zeta_j^k: the radix-2 layers of the CKKS 'special' FFT (the structure the
    reference factorises its homomorphic DFT on, circuits/ckks/dft/dft.go:368-470), as dense n x n matrices"""
    n, M = N // 2, 2 * N
    rot = [pow(5, j, M) for j in range(n)]
    ksi = np.exp(2j * np.pi * np.arange(M) / M)
    layers, ln = [], 2
    while ln <= n:
        L = np.zeros((n, n), dtype=complex)
        lenh, lenq = ln >> 1, ln << 2
        for i in range(0, n, ln):
            for j in range(lenh):
                w = ksi[(rot[j] % lenq) * (M // lenq)]
                L[i + j, i + j], L[i + j, i + j + lenh] = 1, w
                L[i + j + lenh, i + j], L[i + j + lenh, i + j + lenh] = 1, -w
        layers.append(L)
        ln <<= 1
    return layers


class ToyBootstrap:
    """N = 512, fourteen 55-bit moduli (CoeffsToSlots 2, EvalMod 8, SlotsToCoeffs 2, two left), sparse secret h = 16,
    Delta = 2^45.  The DFT is factorised into two sparse matrices per direction (groups of four special-FFT layers, 31 and
    16 diagonals); the bit-reversal is dropped on both sides, EvalMod being slot-wise (as the reference does)."""

    def __init__(self, rng, logN=9, K=12, deg=30, r=3, h=16):
        from fractions import Fraction
        from lattigo_amd import mod1 as M1
        self.N = N = 1 << logN
        self.q, self.p = O.GenModuli(logN + 1, [55] * 14, [55, 55])
        self.ringQ, self.ringP = O.Ring(N, self.q), O.Ring(N, self.p)
        self.oev = O.Evaluator(self.ringQ, self.ringP)
        self.sk = SecretKey(rng, self.ringQ, self.ringP, vals=sparse_ternary(rng, N, h))
        self.top = top = len(self.q) - 1
        self.Se = Fraction(1 << 55)
        self.Delta = float(1 << 45)
        self.K = K
        n, nth = N // 2, 2 * N
        layers = special_fft_layers(N)
        half = len(layers) // 2
        A = np.eye(n, dtype=complex)
        for L in layers[:half]:
            A = L @ A
        Bm = np.eye(n, dtype=complex)
        for L in layers[half:]:
            Bm = L @ Bm
        zeta = ckks_slot_roots(N)
        self.U = zeta[:, None] ** np.arange(n)[None, :]
        q0 = float(self.q[0])
        g = float(self.Se) / (q0 * K) / 2.0
        self.N1 = 16
        depth = deg.bit_length() + r
        self.stc_level = top - 2 - depth  # level at which SlotsToCoeffs starts
        sc = lambda level: Fraction(int(self.q[level]))
        self.cts_scale = [sc(top), sc(top - 1)]
        self.stc_scale = [sc(self.stc_level), sc(self.stc_level - 1)]
        # U^-1 z = P^-1 A^-1 B^-1 z: apply B^-1 (with the gain), then A^-1; the result is in bit-reversed order
        self.cts = [dense_matrix_lt(g * np.linalg.inv(Bm), N, self.N1, self.cts_scale[0], top, self.ringQ, self.ringP),
                    dense_matrix_lt(np.linalg.inv(A), N, self.N1, self.cts_scale[1], top - 1, self.ringQ, self.ringP)]
        self.stc = [dense_matrix_lt(A, N, self.N1, self.stc_scale[0], self.stc_level, self.ringQ, self.ringP),
                    dense_matrix_lt(Bm, N, self.N1, self.stc_scale[1], self.stc_level - 1, self.ringQ, self.ringP)]
        rots = set()
        for lt in self.cts + self.stc:
            _, r1, r2 = OC.BSGSIndex(list(lt.Vec.keys()), n, self.N1)
            rots |= set(r1) | set(r2)
        self.n_diagonals = [len(lt.Vec) for lt in self.cts + self.stc]
        gal = [OC.GaloisElement(nth, k) for k in sorted(rots) if k] + [nth - 1]
        self.gks = gen_galois_keys(rng, self.ringQ, self.ringP, self.sk, gal)
        self.rlk = gen_evaluation_key(rng, self.ringQ, self.ringP,
                                      self.ringQ.binop("MulCoeffsMontgomery", self.sk.Q, self.sk.Q), self.sk)
        self.mod1_params = M1.Mod1Parameters(int(self.q[0]), LevelQ=top - 2, LogScale=55, Mod1Type=M1.CosContinuous, K=K,
                                             Mod1Degree=deg, DoubleAngle=r)

    def encrypt_level0(self, rng, z):
        """level-0 ciphertext with phase Delta * c + e (c the plaintext coefficients of the slots z)"""
        from tests.rlwe_fixtures import ckks_encrypt
        r0 = O.Ring(self.N, self.q[:1])
        return ckks_encrypt(rng, r0, self.sk, z, self.Delta)

    def oracle_bootstrapper(self):
        from lattigo_amd import bootstrapping as BS
        from lattigo_amd import mod1 as M1
        ce = OC.CKKSCtEvaluator(self.oev, self.rlk)
        be = OC.OracleBootstrapBackend(ce, OC.LinTransEvaluator(self.oev, self.gks), OC.InnerSumEvaluator(self.oev, self.gks))
        return BS.Bootstrapper(be, M1.Mod1Evaluator(ce, self.mod1_params), self.cts, self.cts_scale, self.stc, self.stc_scale)

    def decode(self, res):
        """slots of the refreshed ciphertext, undoing the (Delta / 2^55) gain of the circuit"""
        from tests.rlwe_fixtures import ckks_decrypt
        sub = O.Ring(self.N, self.q[: res.level + 1])
        return ckks_decrypt(sub, np.stack(res.Value), self.sk, res.Scale) * (2.0 ** 55 / self.Delta)

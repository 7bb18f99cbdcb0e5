"""Toy CKKS bootstrapping instance for the end-to-end tests (test infrastructure: numpy canonical embedding, dense DFT
matrices, our own seeded keys).  Full packing (N/2 slots): slots z = U w with U[j,k] = zeta_j^k, w = c_lo + i c_hi the two
halves of the plaintext coefficient vector, so CoeffsToSlots is multiplication by (a multiple of) U^-1 and SlotsToCoeffs by U."""
import math

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
        from drivers import mod1 as M1
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
        from drivers import bootstrapping as BS
        from oracle import polyeval_ref as PR
        ce = OC.CKKSCtEvaluator(self.oev, self.rlk)
        be = OC.OracleBootstrapBackend(ce, OC.LinTransEvaluator(self.oev, self.gks), OC.InnerSumEvaluator(self.oev, self.gks))
        # EvalMod on the oracle side is the oracle's own restatement (polynomial evaluator + mod1), not the product driver
        return BS.Bootstrapper(be, PR.Mod1Ref(ce, self.mod1_params), self.cts, self.cts_scale, self.stc, self.stc_scale)

    def decode(self, res):
        """slots of the refreshed ciphertext, undoing the (Delta / 2^55) gain of the circuit"""
        from tests.rlwe_fixtures import ckks_decrypt
        sub = O.Ring(self.N, self.q[: res.level + 1])
        return ckks_decrypt(sub, np.stack(res.Value), self.sk, res.Scale) * (2.0 ** 55 / self.Delta)


from drivers.dft import (bitrev_indices, diag_matmul, fast_encode_rns, layer_diagonals, special_fft,  # noqa: E402,F401
                             special_ifft)


def run_functional_bootstrap(logN, logq_res, n_stc, evalmod_bits, n_cts, cts_bits, stc_bits, logp, cts_groups, stc_groups, h_dense, h_sparse,
                             K=16, deg=30, r=3, log_delta=52, log_se=60, ctx=None, seed=5500, min_bits=15, mod1_type=0):
    """One full bootstrap with real keys and factorised DFT matrices; ctx = None runs it on the oracle backend, a
    lattigo_amd.Context on the device.  Returns a dict with the precision and timings."""
    import time
    from fractions import Fraction
    from drivers import bootstrapping as BS
    from drivers import lintrans as LT
    from drivers import mod1 as M1
    from tests.helpers import prod, rng_for
    from tests.rlwe_fixtures import phase
    device = ctx is not None
    if device:
        import lattigo_amd as la
        from lattigo_amd import rlwe as R
        from drivers import schemes as S
    t_start = time.time()
    N, n, nth = 1 << logN, 1 << (logN - 1), 2 << logN
    depth = deg.bit_length() + r
    q, p = O.GenModuli(logN + 1, list(logq_res) + [stc_bits] * n_stc + [evalmod_bits] * depth + [cts_bits] * n_cts, list(logp))
    top, LP = len(q) - 1, len(p)
    rng = rng_for(seed)
    oQ, oP = O.Ring(N, q), O.Ring(N, p)
    oev = O.Evaluator(oQ, oP)
    sk = SecretKey(rng, oQ, oP, vals=sparse_ternary(rng, N, h_dense))
    sks = SecretKey(rng, oQ, oP, vals=sparse_ternary(rng, N, h_sparse))
    if device:
        gQ, gP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
        gev = la.Evaluator(gQ, gP)
        up_key = lambda k: gev.NewEvaluationKey(k.q, k.p)
    else:
        up_key = lambda k: k
    Se = Fraction(1 << log_se)
    q0 = float(q[0])
    gain = float(Se) / (q0 * K) / 2.0
    lns = [2 << i for i in range(logN - 1)]  # butterfly spans of L_1 .. L_(logN-1):  U = L_last ... L_1 Bitrev

    def group(layers, inverse):
        acc = None
        for ln in layers:  # forward: L_b ... L_a (left-multiply); inverse: (L_b ... L_a)^-1 = L_a^-1 ... L_b^-1 (right-multiply)
            d = layer_diagonals(N, ln, inverse)
            acc = d if acc is None else (diag_matmul(acc, d, n) if inverse else diag_matmul(d, acc, n))
        return acc

    stc_top = top - len(cts_groups) - depth

    def build_lt(diags, level, first_gain=1.0):
        ks = sorted(diags)
        N1 = LT.FindBestBSGSRatio(ks, n, 1)
        scale = Fraction(int(q[level]))
        sub = O.Ring(N, q[: level + 1])
        vec = {}
        for k in ks:
            j0 = ((k // N1) * N1) & (n - 1)
            v = np.roll(diags[k] * first_gain, j0)
            rq = sub.unop("MForm", sub.NTT(fast_encode_rns(v, N, scale, q[: level + 1])))
            rp = oP.unop("MForm", oP.NTT(fast_encode_rns(v, N, scale, p)))
            vec[k] = (rq, rp)
        _, r1, r2 = LT.BSGSIndex(ks, n, N1)
        lt = OC.LinearTransformation(vec, level, LP - 1, n, N1)
        return lt, scale, set(r1) | set(r2)

    if device:  # product path: drivers.dft encodes the factors with the device's NTT / MForm
        from drivers import dft as DFT
        enc = DFT.Encoder(gQ, gP)
        cts, cts_sc, r_a = DFT.NewMatrices(enc, DFT.HomomorphicEncode, cts_groups, top, gain)
        stc, stc_sc, r_b = DFT.NewMatrices(enc, DFT.HomomorphicDecode, stc_groups, stc_top)
        rots = r_a | r_b
    else:
        cts, cts_sc, stc, stc_sc, rots = [], [], [], [], set()
        for i, (a, b) in enumerate(cts_groups):
            lt, sc, rr = build_lt(group(lns[a:b], True), top - i, gain if i == 0 else 1.0)
            cts.append(lt); cts_sc.append(sc); rots |= rr
        for i, (a, b) in enumerate(stc_groups):
            lt, sc, rr = build_lt(group(lns[a:b], False), stc_top - i)
            stc.append(lt); stc_sc.append(sc); rots |= rr
    ndiag = sum(len(m.Vec) for m in cts + stc)
    t_mats = time.time()

    gks = {}
    for k in sorted(rots):
        if k:
            g = OC.GaloisElement(nth, k)
            gks[g] = up_key(gen_galois_keys(rng, oQ, oP, sk, [g])[g])
    gks[nth - 1] = up_key(gen_galois_keys(rng, oQ, oP, sk, [nth - 1])[nth - 1])
    rlk = up_key(gen_evaluation_key(rng, oQ, oP, oQ.binop("MulCoeffsMontgomery", sk.Q, sk.Q), sk))
    d2s = up_key(gen_evaluation_key(rng, oQ, oP, sk.Q, sks))
    s2d = up_key(gen_evaluation_key(rng, oQ, oP, sks.Q, sk))
    t_keys = time.time()

    Delta = float(1 << log_delta)
    z = rng.uniform(-1, 1, size=n) + 1j * rng.uniform(-1, 1, size=n)
    r0 = O.Ring(N, q[:1])
    pt = r0.NTT(fast_encode_rns(z, N, Delta, q[:1]))
    e = np.clip(np.rint(rng.normal(0.0, 3.2, size=N)), -19, 19).astype(np.int64)
    c1 = np.stack([rng.integers(0, int(q[0]), size=N, dtype=np.uint64)])
    c0 = r0.binop("Add", r0.binop("Sub", pt, r0.binop("MulCoeffsMontgomery", c1, sk.Q[:1])), r0.NTT(small_to_rns(e, q[:1])))

    pm = M1.Mod1Parameters(int(q[0]), LevelQ=top - len(cts), LogScale=log_se, Mod1Type=mod1_type, K=K, Mod1Degree=deg,
                           DoubleAngle=r, LogMessageRatio=int(round(math.log2(q0))) - log_delta)
    if device:
        ggks = R.GaloisKeySet(gks)
        ce = S.CKKSCiphertextEvaluator(gev, rlk)
        be = BS.DeviceBootstrapBackend(ce, LT.LinTransEvaluator(gev, ggks), R.InnerSumEvaluator(gev, ggks), d2s, s2d)
        ct_in = S.Ciphertext([la.Poly(gQ, 1).upload(c0), la.Poly(gQ, 1).upload(c1)], 0, 1)
    else:
        ce = OC.CKKSCtEvaluator(oev, rlk)
        be = OC.OracleBootstrapBackend(ce, OC.LinTransEvaluator(oev, gks), OC.InnerSumEvaluator(oev, gks), d2s, s2d)
        ct_in = OC.Ct([c0, c1], 1)
    boot = BS.Bootstrapper(be, M1.Mod1Evaluator(ce, pm), cts, cts_sc, stc, stc_sc)
    if device:
        boot.Bootstrap(ct_in, Se)  # warm-up (plans, buffer cache)
        ctx.sync()
    t0 = time.time()
    res = boot.Bootstrap(ct_in, Se)
    if device:
        ctx.sync()
    t_boot = time.time() - t0
    assert res.Degree() == 1 and res.Scale == Se and res.level == len(logq_res) - 1

    lv = res.level
    sub = O.Ring(N, q[: lv + 1])
    ct_out = np.stack([v.download()[0][: lv + 1] for v in res.Value]) if device else np.stack(res.Value)
    import hashlib
    ct_digest = hashlib.sha256(np.ascontiguousarray(ct_out, dtype=np.uint64).tobytes()).hexdigest()  # the refreshed ciphertext's words
    ph = sub.INTT(phase(oQ, ct_out, sk.Q))
    Ql = prod(q[: lv + 1])
    w = [(Ql // int(qi)) * pow(Ql // int(qi), -1, int(qi)) for qi in q[: lv + 1]]
    coeffs = np.empty(N)
    for j in range(N):
        x = sum(int(ph[i, j]) * w[i] for i in range(lv + 1)) % Ql
        if x > Ql // 2:
            x -= Ql
        coeffs[j] = float(Fraction(x) / Se)
    got = special_fft(coeffs[:n] + 1j * coeffs[n:], N) * (float(Se) / Delta)
    errs = np.abs(got - z)
    err = float(np.max(errs))
    mean_bits = float(np.mean(-np.log2(np.maximum(errs, 1e-300))))
    out = {"mean_precision_bits": mean_bits, "logN": logN, "limbs_Q": top + 1, "limbs_P": LP, "dft_diagonals": ndiag, "galois_keys": len(gks),
           "bootstrap_ms": t_boot * 1e3, "precision_bits": float(-np.log2(err)), "max_slot_error": err, "output_level": lv,
           "host_setup_s": {"matrices": t_mats - t_start, "keys": t_keys - t_mats}, "backend": "device" if device else "oracle",
           "ct_sha256": ct_digest, "seed": seed}
    assert out["precision_bits"] > min_bits, out
    return out

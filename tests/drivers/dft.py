"""Host-side set-up for the homomorphic DFT of CKKS bootstrapping (the role of circuits/ckks/dft/dft.go:368-600 and of the
CKKS encoder's special FFT, schemes/ckks/encoder.go): the "special" FFT over the slot roots zeta_j = exp(i pi 5^j / N),
its radix-2 layers in diagonal form, products of layers (the sparse factors CoeffsToSlots / SlotsToCoeffs are made of), and a
double-precision encoder of slot vectors into RNS plaintext polynomials.  numpy, O(n log n); nothing here touches the device
-- encoded polynomials are uploaded by the caller.

Conventions: n = N / 2 slots, z = U w with U[j, k] = zeta_j^k and w = c_lo + i c_hi the two halves of the real coefficient
vector; U = L_log(n) ... L_1 Bitrev.  A matrix in diagonal form is {offset: vector} with (A z)[r] = sum_a A[a][r] z[(r+a) mod n]."""
from __future__ import annotations

import numpy as np

_TW_CACHE = {}


def _layer_twiddles(N, ln):
    if (N, ln) not in _TW_CACHE:
        _TW_CACHE[(N, ln)] = _layer_twiddles_uncached(N, ln)
    return _TW_CACHE[(N, ln)]


def _layer_twiddles_uncached(N, ln):
    n, M = N // 2, 2 * N
    lenh, lenq = ln >> 1, ln << 2
    j = np.arange(lenh)
    rot = np.array([pow(5, int(x), M) for x in j]) if lenh <= 4096 else None
    if rot is None:  # 5^j mod M incrementally
        rot = np.empty(lenh, dtype=np.int64)
        g = 1
        for i in range(lenh):
            rot[i] = g
            g = g * 5 % M
    w = np.exp(2j * np.pi * ((rot % lenq) * (M // lenq)) / M)  # [lenh]
    return np.tile(w, n // ln)  # per block, indexed by (block, j)


def layer_diagonals(N, ln, inverse=False):
    """special-FFT layer of butterfly span `ln` as {offset: vector} with (A z)[r] = sum_a A[a][r] * z[(r + a) mod n]"""
    n = N // 2
    lenh = ln >> 1
    w = _layer_twiddles(N, ln)  # one twiddle per butterfly, in row order of the first halves
    first = (np.arange(n) % ln) < lenh
    d0, dp, dm = np.zeros(n, dtype=complex), np.zeros(n, dtype=complex), np.zeros(n, dtype=complex)
    if not inverse:  # (u, v) -> (u + w v, u - w v)
        d0[first], dp[first] = 1.0, w
        dm[~first], d0[~first] = 1.0, -w
    else:  # u = (u' + v') / 2, v = (u' - v') / (2 w)
        d0[first], dp[first] = 0.5, 0.5
        dm[~first], d0[~first] = 0.5 / w, -0.5 / w
    out = {0: d0, lenh % n: dp}
    key = (-lenh) % n
    out[key] = out.get(key, 0) + dm
    return out


def diag_matmul(A, B, n):
    """C = A B in diagonal form: C[a + b][r] += A[a][r] * B[b][(r + a) mod n]"""
    C = {}
    for a, va in A.items():
        for b, vb in B.items():
            k = (a + b) % n
            t = va * np.roll(vb, -a)
            C[k] = C[k] + t if k in C else t
    return {k: v for k, v in C.items() if np.max(np.abs(v)) > 1e-13}


def bitrev_indices(n):
    b = n.bit_length() - 1
    idx = np.arange(n)
    out = np.zeros(n, dtype=np.int64)
    for i in range(b):
        out |= ((idx >> i) & 1) << (b - 1 - i)
    return out


def special_fft(w_vec, N):
    """z = U w (slots from the complex half-coefficient vector), O(n log n)"""
    n = N // 2
    v = np.asarray(w_vec, dtype=complex)[bitrev_indices(n)].copy()
    ln = 2
    while ln <= n:
        lenh = ln >> 1
        tw = _layer_twiddles(N, ln).reshape(n // ln, lenh)
        blk = v.reshape(n // ln, ln)
        u, t = blk[:, :lenh].copy(), blk[:, lenh:] * tw
        blk[:, :lenh], blk[:, lenh:] = u + t, u - t
        ln <<= 1
    return v


def special_ifft(z, N):
    """w = U^-1 z"""
    n = N // 2
    v = np.asarray(z, dtype=complex).copy()
    ln = n
    while ln >= 2:
        lenh = ln >> 1
        tw = _layer_twiddles(N, ln).reshape(n // ln, lenh)
        blk = v.reshape(n // ln, ln)
        a, b = blk[:, :lenh].copy(), blk[:, lenh:].copy()
        blk[:, :lenh], blk[:, lenh:] = (a + b) / 2, (a - b) / (2 * tw)
        ln >>= 1
    out = np.empty(n, dtype=complex)
    out[bitrev_indices(n)] = v
    return out


def fast_encode_rns(vec, N, scale, moduli):
    """slot vector -> coefficient-domain residues [limbs][N] (uint64) of round(scale * coefficients), via the special iFFT"""
    w = special_ifft(vec, N)
    coeffs = np.concatenate([w.real, w.imag]) * float(scale)
    ints = np.rint(coeffs).astype(np.int64)
    out = np.empty((len(moduli), N), dtype=np.uint64)
    for i, q in enumerate(moduli):
        out[i] = np.mod(ints, np.int64(q)).astype(np.uint64)
    return out


# ---- factor lists and their encoding onto the device -----------------------------------------------------------------------------

HomomorphicEncode, HomomorphicDecode = 0, 1  # CoeffsToSlots (inverse DFT) / SlotsToCoeffs (DFT), circuits/ckks/dft/dft.go:24-31


def factor_diagonals(N, groups, inverse):
    """The sparse factors of the homomorphic (inverse) DFT, each the product of the radix-2 layers lns[a:b] for (a, b) in groups,
    in diagonal form -- the matrices circuits/ckks/dft/dft.go:368-470 (GenMatrices) merges layer by layer.  The bit-reversal
    permutation is not part of any factor: CoeffsToSlots followed by SlotsToCoeffs cancels it, and EvalMod is slot-wise."""
    n = N >> 1
    lns = [2 << i for i in range(N.bit_length() - 2)]
    out = []
    for a, b in groups:
        acc = None
        for ln in lns[a:b]:  # forward: L_b ... L_a (left-multiply); inverse: (L_b ... L_a)^-1 = L_a^-1 ... L_b^-1 (right-multiply)
            d = layer_diagonals(N, ln, inverse)
            acc = d if acc is None else (diag_matmul(acc, d, n) if inverse else diag_matmul(d, acc, n))
        out.append(acc)
    return out


class Encoder:
    """CKKS slot encoder in double precision over device rings (schemes/ckks/encoder.go:160-330 Encode / Embed and :520-640 Decode,
    with float64 where the reference switches to big.Float above 53 bits of precision): the special inverse FFT and the rounding
    run in numpy on the host, the NTT and the Montgomery form on the device."""

    def __init__(self, ringQ, ringP=None):
        self.ringQ, self.ringP = ringQ, ringP
        self.N = ringQ.N

    def _up(self, ring, nl, values, scale, montgomery):
        from lattigo_amd.ring import Poly
        p = Poly(ring, nl).upload(fast_encode_rns(values, self.N, scale, ring.ModuliChain()[:nl]))
        r = ring.AtLevel(nl - 1)
        r.NTT(p, p)
        if montgomery:
            r.MForm(p, p)
        return p

    def Encode(self, values, level, scale, montgomery=False):
        """values: N/2 complex slots -> an NTT-domain plaintext of level+1 limbs holding round(scale * iFFT(values))."""
        return self._up(self.ringQ, level + 1, np.asarray(values, dtype=np.complex128), scale, montgomery)

    def EncodeQP(self, values, levelQ, scale):
        """The (Q, P) pair in NTT + Montgomery form a linear-transformation diagonal is stored as
        (circuits/common/lintrans/lintrans.go:177-260 Encode)."""
        v = np.asarray(values, dtype=np.complex128)
        return self._up(self.ringQ, levelQ + 1, v, scale, True), self._up(self.ringP, self.ringP.Level() + 1, v, scale, True)

    def Decode(self, pt, scale, is_ntt=True):
        """pt: a device Poly (batch 1) -> N/2 complex slots.  CRT reconstruction in Python big integers (host)."""
        from fractions import Fraction
        from lattigo_amd.ring import Poly
        nl = pt.n_limbs
        r = self.ringQ.AtLevel(nl - 1)
        t = pt
        if is_ntt:
            t = Poly(self.ringQ, nl)
            r.INTT(pt, t)
        res = t.download()[0][:nl]
        q = [int(x) for x in self.ringQ.ModuliChain()[:nl]]
        Q = 1
        for x in q:
            Q *= x
        w = [(Q // qi) * pow(Q // qi, -1, qi) for qi in q]
        sc = Fraction(scale)
        coeffs = np.empty(self.N)
        for j in range(self.N):
            x = sum(int(res[i, j]) * w[i] for i in range(nl)) % Q
            coeffs[j] = float(Fraction(x - Q if x > Q // 2 else x) / sc)
        n = self.N >> 1
        return special_fft(coeffs[:n] + 1j * coeffs[n:], self.N)


def encode_linear_transformation(encoder: Encoder, diags, level, scale, gain=1.0, log_bsgs_ratio=1):
    """One factor in diagonal form -> a device LinearTransformation at `level` with every diagonal pre-rotated for the
    baby-step/giant-step evaluation (circuits/common/lintrans/lintrans.go:177-260), plus the rotations it needs."""
    from . import lintrans as LT
    n = encoder.N >> 1
    ks = sorted(diags)
    N1 = LT.FindBestBSGSRatio(ks, n, log_bsgs_ratio)
    vec = {}
    for k in ks:
        j0 = ((k // N1) * N1) & (n - 1)
        vec[k] = encoder.EncodeQP(np.roll(diags[k] * gain, j0), level, scale)
    _, r1, r2 = LT.BSGSIndex(ks, n, N1)
    return LT.LinearTransformation(vec, level, encoder.ringP.Level(), n, N1), set(r1) | set(r2)


def NewMatrices(encoder: Encoder, kind, groups, level_start, gain=1.0):
    """The encoded factor list of CoeffsToSlots (kind = HomomorphicEncode) or SlotsToCoeffs (HomomorphicDecode) -- the role of
    circuits/ckks/dft/dft.go:168-260 NewMatrixFromLiteral.  Factor i is encoded at level level_start - i with scale q[level]
    (so the rescale after it restores the ciphertext's scale); `gain` multiplies the first factor.  Returns (matrices, scales,
    rotations)."""
    from fractions import Fraction
    q = encoder.ringQ.ModuliChain()
    mats, scales, rots = [], [], set()
    for i, d in enumerate(factor_diagonals(encoder.N, groups, kind == HomomorphicEncode)):
        sc = Fraction(int(q[level_start - i]))
        lt, rr = encode_linear_transformation(encoder, d, level_start - i, sc, gain if i == 0 else 1.0)
        mats.append(lt); scales.append(sc); rots |= rr
    return mats, scales, rots

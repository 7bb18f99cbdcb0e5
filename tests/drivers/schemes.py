"""Host-side mirror of the ring-level call sites of ``schemes.Evaluator`` (schemes/schemes.go:14-28) for CKKS and BGV:
``Add/Sub/Mul/MulRelin/MulThenAdd/MulRelinThenAdd/Relinearize/Rescale`` on ciphertext x ciphertext and ciphertext x
plaintext operands, as drivers over the device-resident operators.  Everything that is floating-point scale management in
the reference (CKKS scale ratios, plaintext encoding) stays with the caller: these methods are the polynomial arithmetic a
call performs once levels and scales have been decided.  A ciphertext is a list of ``Poly`` (degree + 1 entries), a
plaintext a single ``Poly``, all in the NTT domain."""
from __future__ import annotations

import numpy as np

from lattigo_amd.ring import Poly
from lattigo_amd.rlwe import EvaluationKey, Evaluator


class _Base:
    def __init__(self, evaluator: Evaluator):
        self.eval, self.ringQ = evaluator, evaluator.ringQ

    def _tmp(self, level, B):
        """a temporary the next operation overwrites in full: contents unspecified (no zero-fill launch)"""
        return Poly(self.ringQ, level + 1, B, zero=False)

    # Evaluator.Add / Sub for operands of equal scale (schemes/ckks/evaluator.go:65,226; schemes/bgv/evaluator.go:168,350
    # -> rlwe evaluateInPlace): component-wise, the operand of higher degree is copied (negated for Sub) beyond the other's
    def Add(self, level, op0, op1, opOut):
        self._addsub(level, op0, op1, opOut, False)

    def Sub(self, level, op0, op1, opOut):
        self._addsub(level, op0, op1, opOut, True)

    def _addsub(self, level, op0, op1, opOut, sub):
        r = self.ringQ.AtLevel(level)
        d0, d1 = len(op0), len(op1)
        for i in range(min(d0, d1)):
            (r.Sub if sub else r.Add)(op0[i], op1[i], opOut[i])
        for i in range(min(d0, d1), max(d0, d1)):
            if d0 > d1:
                if opOut[i] is not op0[i]:
                    opOut[i].CopyLvl(level, op0[i])
            elif sub:
                r.Neg(op1[i], opOut[i])
            elif opOut[i] is not op1[i]:
                opOut[i].CopyLvl(level, op1[i])

    def Relinearize(self, level, op0, rlk: EvaluationKey, opOut):
        self.eval.Relinearize(level, op0, rlk, opOut)

    def Rescale(self, level, op0, opOut, nbRescales: int = 1):
        self.eval.Rescale(level, nbRescales, op0, opOut)

    def _ct_ct_then_add(self, level, c00, c01, op1, rlk, opOut):
        """the shared tail of mulRelinThenAdd (schemes/ckks/evaluator.go:1131-1155, schemes/bgv/evaluator.go:1288-1314)"""
        r = self.ringQ.AtLevel(level)
        B = op1[0].batch
        r.MulCoeffsMontgomeryThenAdd(c00, op1[0], opOut[0])  # c0 += c[0]*c[0]
        r.MulCoeffsMontgomeryThenAdd(c00, op1[1], opOut[1])  # c1 += c[0]*c[1]
        r.MulCoeffsMontgomeryThenAdd(c01, op1[0], opOut[1])  # c1 += c[1]*c[0]
        if rlk is not None:
            c2 = self._tmp(level, B)
            r.MulCoeffsMontgomery(c01, op1[1], c2)
            tmp = [self._tmp(level, B), self._tmp(level, B)]
            self.eval.GadgetProduct(level, c2, rlk, tmp)
            r.Add(opOut[0], tmp[0], opOut[0])
            r.Add(opOut[1], tmp[1], opOut[1])
        else:
            r.MulCoeffsMontgomeryThenAdd(c01, op1[1], opOut[2])  # c2 += c[1]*c[1]


class CKKSEvaluator(_Base):
    """schemes/ckks Evaluator, ring level."""

    # Evaluator.Mul / MulRelin, ct x ct (schemes/ckks/evaluator.go:764-840)
    def MulRelin(self, level, op0, op1, rlk: EvaluationKey | None, opOut):
        self.eval.CKKSMulRelin(level, op0, op1, rlk, opOut)

    # ... ct x pt (or pt x ct) branch (:842-870)
    def MulPlaintext(self, level, op0, pt: Poly, opOut):
        r = self.ringQ.AtLevel(level)
        c0 = self._tmp(level, pt.batch)
        r.MForm(pt, c0)
        for a, o in zip(op0, opOut):
            r.MulCoeffsMontgomery(c0, a, o)

    # Evaluator.MulThenAdd / MulRelinThenAdd, ct x ct (:1081-1155; the scale-ratio rescaling of opOut, :1087-1096, is the
    # caller's: it is a MulScalar by an integer decided from floating-point scales)
    def MulRelinThenAdd(self, level, op0, op1, rlk: EvaluationKey | None, opOut):
        r = self.ringQ.AtLevel(level)
        B = op0[0].batch
        c00, c01 = self._tmp(level, B), self._tmp(level, B)
        r.MForm(op0[0], c00)
        r.MForm(op0[1], c01)
        self._ct_ct_then_add(level, c00, c01, op1, rlk, opOut)

    # ... ct x pt branch (:1158-1170)
    def MulPlaintextThenAdd(self, level, op0, pt: Poly, opOut):
        r = self.ringQ.AtLevel(level)
        c00 = self._tmp(level, pt.batch)
        r.MForm(pt, c00)
        for a, o in zip(op0, opOut):
            r.MulCoeffsMontgomeryThenAdd(a, c00, o)


def bgv_match_scales_binary(scale0: int, scale1: int, t: int):
    """bgv.Evaluator.matchScalesBinary (schemes/bgv/evaluator.go:1569-1608): (r0, r1, e) with r0 * scale0 = r1 * scale1 mod t
    and minimal |r0| + |r1|"""
    from math import gcd
    if gcd(scale0, t) != 1:
        raise ValueError("cannot matchScalesBinary: invalid ciphertext scale: gcd(scale, t) != 1")
    thalf = t >> 1
    center = lambda x: t - x if x >= thalf else x
    a, b = t, 0
    A, Bv = pow(scale0, t - 2, t) * scale1 % t, 1
    r0, r1 = A, Bv
    e = center(A) + 1
    while A != 0:
        qq = a // A
        a, A = A, a % A
        b, Bv = Bv, (t + b - Bv * qq % t) % t
        if A != 0 and gcd(A, t) == 1:
            tmp = center(A) + center(Bv)
            if tmp < e:
                e = tmp
                r0, r1 = A, Bv
    return r0, r1, e


class BGVEvaluator(_Base):
    """schemes/bgv Evaluator (standard tensoring), ring level; t = plaintext modulus."""

    def __init__(self, evaluator: Evaluator, t: int):
        super().__init__(evaluator)
        self.t = int(t)
        # tMontgomery = MForm(T * 2^64 mod q_i) (schemes/bgv/evaluator.go:60-62)
        self.tMontgomery = np.array([((self.t << 64) % int(q)) * (1 << 64) % int(q) for q in self.ringQ.ModuliChain()],
                                    dtype=np.uint64)

    # Evaluator.Mul / MulRelin, ct x ct (schemes/bgv/evaluator.go:592-667)
    def MulRelin(self, level, op0, op1, rlk: EvaluationKey | None, opOut):
        self.eval.BGVMulRelin(level, self.t, op0, op1, rlk, opOut)

    # ... ct x pt branch (:669-683)
    def MulPlaintext(self, level, op0, pt: Poly, opOut):
        r = self.ringQ.AtLevel(level)
        c00 = self._tmp(level, pt.batch)
        r.MulRNSScalarMontgomery(pt, self.tMontgomery, c00)
        for a, o in zip(op0, opOut):
            r.MulCoeffsMontgomery(a, c00, o)

    # Evaluator.MulThenAdd / MulRelinThenAdd, ct x ct (:1230-1314).  scales = (op0.Scale, op1.Scale, opOut.Scale) mod t;
    # returns the new opOut scale
    def MulRelinThenAdd(self, level, op0, op1, rlk: EvaluationKey | None, opOut, scales=(1, 1, 1)) -> int:
        r = self.ringQ.AtLevel(level)
        B = op0[0].batch
        s0, s1, so = (int(x) % self.t for x in scales)
        r0, target = 1, s0 * s1 % self.t
        if so != target:  # :1267-1276
            r0, r1, _ = bgv_match_scales_binary(target, so, self.t)
            for o in opOut:
                r.MulScalar(o, r1, o)
            so = so * r1 % self.t
        c00, c01 = self._tmp(level, B), self._tmp(level, B)
        r.MulRNSScalarMontgomery(op0[0], self.tMontgomery, c00)
        r.MulRNSScalarMontgomery(op0[1], self.tMontgomery, c01)
        if r0 != 1:  # :1283-1286
            r.MulScalar(c00, r0, c00)
            r.MulScalar(c01, r0, c01)
        self._ct_ct_then_add(level, c00, c01, op1, rlk, opOut)
        return so


# ----------------------------------------------------------------------------------------------------------------
# bgv.Evaluator on rlwe.Ciphertext objects (level, degree and the scale in Z_t travel with the polynomials): the
# schemes.Evaluator surface that circuits/common/polynomial is written against.
# ----------------------------------------------------------------------------------------------------------------
class Ciphertext:
    """rlwe.Ciphertext: Value = degree+1 device polynomials (NTT), Level, Scale (an integer mod t for BGV)"""

    def __init__(self, value, level: int, scale: int = 1):
        self.Value, self.level, self.Scale = list(value), level, int(scale)

    def Degree(self):
        return len(self.Value) - 1

    def Level(self):
        return self.level


class BGVCiphertextEvaluator:
    """schemes/bgv Evaluator (standard tensoring) at the rlwe.Ciphertext level: Add (ciphertext or integer, with the
    scale-matching branch), Mul / MulRelin (ciphertext or integer), MulThenAdd (integer), Relinearize, Rescale, with the
    reference's level and scale bookkeeping (schemes/bgv/evaluator.go:122-260, 384-470, 500-560, 1056-1140, 1363-1393)."""

    def __init__(self, evaluator: Evaluator, t: int, rlk: EvaluationKey | None = None):
        self.eval, self.t, self.rlk = evaluator, int(t), rlk
        self.ringQ = evaluator.ringQ
        self.low = BGVEvaluator(evaluator, t)
        self.Q = [int(q) for q in self.ringQ.ModuliChain()]
        self.tInvModQ, Qi = [], 1  # schemes/bgv/encoder.go:67-71
        for q in self.Q:
            Qi *= q
            self.tInvModQ.append(pow(self.t, -1, Qi))

    # -- allocation helpers
    def NewCiphertext(self, degree: int, level: int, batch: int = 1) -> Ciphertext:
        return Ciphertext([Poly(self.ringQ, level + 1, batch) for _ in range(degree + 1)], level, 1)

    def _new_result(self, degree: int, level: int, batch: int) -> Ciphertext:
        """the result of an XNew method: every component is overwritten in full by the operation that follows"""
        return Ciphertext([Poly(self.ringQ, level + 1, batch, zero=False) for _ in range(degree + 1)], level, 1)

    def CopyNew(self, ct: Ciphertext) -> Ciphertext:
        out = self._new_result(ct.Degree(), ct.level, ct.Value[0].batch)
        for a, b in zip(ct.Value, out.Value):
            b.CopyLvl(ct.level, a)
        out.Scale = ct.Scale
        return out

    def _resize(self, ct: Ciphertext, degree: int, level: int):
        """rlwe.Element.Resize: grow (zero polynomials) or shrink the degree, set the level"""
        B = ct.Value[0].batch if ct.Value else 1
        while len(ct.Value) < degree + 1:
            ct.Value.append(Poly(self.ringQ, ct.Value[0].n_limbs if ct.Value else level + 1, B))
        del ct.Value[degree + 1:]
        ct.level = level

    def _centered(self, x: int) -> int:
        x %= self.t
        return x - self.t if x > (self.t >> 1) else x

    # -- Add (schemes/bgv/evaluator.go:122-205)
    def Add(self, op0: Ciphertext, op1, opOut: Ciphertext):
        if isinstance(op1, Ciphertext):
            level = min(op0.level, op1.level, opOut.level)
            self._resize(opOut, max(op0.Degree(), op1.Degree()), level)
            r = self.ringQ.AtLevel(level)
            if op0.Scale == op1.Scale:  # evaluateInPlace (:207-224)
                small, large = (op0, op1) if op0.Degree() <= op1.Degree() else (op1, op0)
                for i in range(small.Degree() + 1):
                    r.Add(op0.Value[i], op1.Value[i], opOut.Value[i])
                for i in range(small.Degree() + 1, large.Degree() + 1):
                    if opOut.Value[i] is not large.Value[i]:
                        opOut.Value[i].CopyLvl(level, large.Value[i])
                opOut.Scale = max(op0.Scale, op1.Scale)
            else:  # matchScaleThenEvaluateInPlace (:226-243)
                r0, r1, _ = bgv_match_scales_binary(op0.Scale, op1.Scale, self.t)
                for i in range(op0.Degree() + 1):
                    r.MulScalar(op0.Value[i], r0, opOut.Value[i])
                for i in range(op0.Degree() + 1, opOut.Degree() + 1):
                    opOut.Value[i].Zero()
                for i in range(op1.Degree() + 1):
                    r.MulScalarThenAdd(op1.Value[i], r1, opOut.Value[i])
                opOut.Scale = op0.Scale * r0 % self.t
            return
        # integer operand (:144-172): brought to the scale of op0, centred, times T^-1 mod Q, added to c0
        level = min(op0.level, opOut.level)
        self._resize(opOut, op0.Degree(), level)
        v = self._centered(int(op1) * op0.Scale) * self.tInvModQ[level]
        r = self.ringQ.AtLevel(level)
        r.AddScalarBigint(op0.Value[0], v, opOut.Value[0])
        if op0 is not opOut:
            for i in range(1, op0.Degree() + 1):
                opOut.Value[i].CopyLvl(level, op0.Value[i])
            opOut.Scale = op0.Scale

    # -- Mul / MulRelin (:384-470, 500-560, tensorStandard :592-685)
    def _tensor(self, op0: Ciphertext, op1: Ciphertext, relin: bool, opOut: Ciphertext):
        level = min(op0.level, op1.level, opOut.level)
        if op0.Degree() != 1 or op1.Degree() != 1:
            raise ValueError("cannot tensor: operands must be of degree 1")
        self._resize(opOut, 1 if relin else 2, level)
        if relin and self.rlk is None:
            raise KeyError("cannot Tensor: cannot Relinearize: RelinearizationKey is nil")
        self.eval.BGVMulRelin(level, self.t, op0.Value, op1.Value, self.rlk if relin else None, opOut.Value)
        opOut.Scale = op0.Scale * op1.Scale % self.t

    def Mul(self, op0: Ciphertext, op1, opOut: Ciphertext):
        if isinstance(op1, Ciphertext):
            self._tensor(op0, op1, False, opOut)
            return
        level = min(op0.level, opOut.level)  # integer operand (:414-437)
        self._resize(opOut, op0.Degree(), level)
        r = self.ringQ.AtLevel(level)
        v = self._centered(int(op1))
        for i in range(op0.Degree() + 1):
            r.MulScalarBigint(op0.Value[i], v, opOut.Value[i])
        opOut.Scale = op0.Scale

    def MulRelin(self, op0: Ciphertext, op1, opOut: Ciphertext):
        if isinstance(op1, Ciphertext):
            self._tensor(op0, op1, True, opOut)
        else:
            self.Mul(op0, op1, opOut)

    def MulNew(self, op0: Ciphertext, op1) -> Ciphertext:
        B = op0.Value[0].batch
        if isinstance(op1, Ciphertext):
            out = self._new_result(op0.Degree() + op1.Degree(), min(op0.level, op1.level), B)
        else:
            out = self._new_result(op0.Degree(), op0.level, B)
        self.Mul(op0, op1, out)
        return out

    def MulRelinNew(self, op0: Ciphertext, op1) -> Ciphertext:
        B = op0.Value[0].batch
        out = self._new_result(1, min(op0.level, op1.level) if isinstance(op1, Ciphertext) else op0.level, B)
        self.MulRelin(op0, op1, out)
        return out

    # -- MulThenAdd with an integer operand (:1108-1140)
    def MulThenAdd(self, op0: Ciphertext, op1: int, opOut: Ciphertext):
        level = min(op0.level, opOut.level)
        self._resize(opOut, op0.Degree(), opOut.level)  # (sic) the reference resizes to op0's degree
        v = int(op1)
        if op0.Scale != opOut.Scale:
            v *= pow(op0.Scale, self.t - 2, self.t) * opOut.Scale % self.t
        v = self._centered(v)
        r = self.ringQ.AtLevel(level)
        for i in range(op0.Degree() + 1):
            r.MulScalarBigintThenAdd(op0.Value[i], v, opOut.Value[i])

    # -- Relinearize (rlwe.Evaluator.Relinearize) / Rescale (:1363-1393)
    def Relinearize(self, op0: Ciphertext, opOut: Ciphertext):
        if self.rlk is None:
            raise KeyError("cannot Relinearize: RelinearizationKey is nil")
        level = min(op0.level, opOut.level)
        B = op0.Value[0].batch
        out = [Poly(self.ringQ, level + 1, B), Poly(self.ringQ, level + 1, B)] if opOut is op0 else None
        tgt = out if out is not None else opOut.Value[:2]
        if out is None:
            self._resize(opOut, 1, level)
            tgt = opOut.Value
        self.eval.Relinearize(level, op0.Value, self.rlk, tgt)
        if out is not None:
            opOut.Value = out
            opOut.level = level
        opOut.Scale = op0.Scale

    def Rescale(self, op0: Ciphertext, opOut: Ciphertext):
        if op0.level == 0:
            raise ValueError("cannot rescale: op0 already at level 0")
        level = op0.level
        r = self.ringQ.AtLevel(level)
        for a, o in zip(op0.Value, opOut.Value):
            r.DivRoundByLastModulusNTT(a, o)
        self._resize(opOut, op0.Degree(), level - 1)
        opOut.Scale = op0.Scale * pow(self.Q[level], -1, self.t) % self.t


# ----------------------------------------------------------------------------------------------------------------
# ckks.Evaluator on rlwe.Ciphertext objects.  A constant c is encoded exactly as bigComplexToRNSScalar does
# (schemes/ckks/scaling.go:10-43): c is first held at EncodingPrecision bits (bignum.ToComplex, evaluator.go:77,168,639), the
# product with the scale is a big.Float of 128 bits (rlwe.ScalePrecision, round to nearest even), +-0.5 is added in the same
# precision and the result truncated (_big_float_scalar below; round 4 -- rounds 1-3 rounded the exact product).  Scales are
# carried as exact rationals where the reference re-rounds its 128-bit float after every Mul / Div (core/rlwe/scale.go:77-113);
# the value used to encode a constant is that rational rounded to 128 bits.
# ----------------------------------------------------------------------------------------------------------------
def _round_bits(x, prec: int):
    """x (a Fraction) rounded to `prec` significant bits, ties to even: the value a big.Float of that precision holds"""
    from fractions import Fraction
    x = Fraction(x)
    if x == 0:
        return x
    neg, x = x < 0, abs(x)
    e = x.numerator.bit_length() - x.denominator.bit_length() + 1  # x < 2^e
    if x < Fraction(2) ** (e - 1):
        e -= 1                                                     # now 2^(e-1) <= x < 2^e
    ulp = Fraction(2) ** (e - prec)
    m, rem = divmod(x / ulp, 1)
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and int(m) & 1):
        m += 1
    x = int(m) * ulp
    return -x if neg else x


def _big_float_scalar(c, scale, enc_prec: int = 53) -> int:
    """One component of bigComplexToRNSScalar (schemes/ckks/scaling.go:16-26 / :30-40): r = Mul(c, scale) in a fresh big.Float --
    precision max(prec(c), prec(scale)) = max(EncodingPrecision, 128) --, r +- 0.5 in that precision, r.Int() (toward zero)."""
    import math
    from fractions import Fraction
    c = _round_bits(c, enc_prec)  # bignum.ToComplex(op1, EncodingPrecision)
    if c == 0:
        return 0
    prec = max(enc_prec, 128)
    r = _round_bits(c * _round_bits(scale, 128), prec)
    r = _round_bits(r + Fraction(1, 2) if c > 0 else r - Fraction(1, 2), prec)
    return math.trunc(r)


def _round_half_away(x):
    from fractions import Fraction
    x = Fraction(x)
    if x > 0:
        return (x + Fraction(1, 2)).__floor__()
    if x < 0:
        return -((-x + Fraction(1, 2)).__floor__())
    return 0


def _as_complex_fraction(c):
    from fractions import Fraction
    if isinstance(c, tuple):
        return Fraction(c[0]), Fraction(c[1])
    if isinstance(c, complex):
        return Fraction(c.real), Fraction(c.imag)
    return Fraction(c), Fraction(0)


def encoding_precision(default_scale) -> int:
    """Parameters.EncodingPrecision (schemes/ckks/params.go:185-195): `log2scale := math.Log2(DefaultScale().Float64())`; 53 when
    that is <= 53, else uint(log2scale) (truncation).  default_scale None: parameters whose scale is at most 2^53."""
    import math
    if default_scale is None:
        return 53
    l2 = math.log2(float(default_scale))
    return 53 if l2 <= 53 else int(l2)


class CKKSCiphertextEvaluator:
    """schemes/ckks Evaluator at the rlwe.Ciphertext level (one level per rescaling): Add / Sub (ciphertext or constant),
    Mul / MulRelin (ciphertext or constant), MulThenAdd (constant), Relinearize, Rescale
    (schemes/ckks/evaluator.go:42-135, 221-424, 477-515, 570-760, 875-940)."""

    def __init__(self, evaluator: Evaluator, rlk: EvaluationKey | None = None, default_scale=None):
        self.eval, self.rlk, self.ringQ = evaluator, rlk, evaluator.ringQ
        self.Q = [int(q) for q in self.ringQ.ModuliChain()]
        self.t = None
        # RootsForward[1] of every limb as a plain integer: the square root of -1 that evaluateWithScalar uses (:417)
        self.imag_unit = [int(self.ringQ.roots(i)[1]) * pow(1 << 64, -1, q) % q for i, q in enumerate(self.Q)]
        # Parameters.EncodingPrecision (schemes/ckks/params.go:185-195): 53, or floor(log2(DefaultScale)) when that is larger
        self.EncodingPrecision = encoding_precision(default_scale)

    NewCiphertext = BGVCiphertextEvaluator.NewCiphertext
    _new_result = BGVCiphertextEvaluator._new_result
    CopyNew = BGVCiphertextEvaluator.CopyNew
    _resize = BGVCiphertextEvaluator._resize

    def _rns(self, level, scale, c):
        """bigComplexToRNSScalar + the (a + b i, a - b i) pair of evaluateWithScalar (schemes/ckks/scaling.go:10, evaluator.go:410)"""
        re, im = _as_complex_fraction(c)
        real, imag = _big_float_scalar(re, scale, self.EncodingPrecision), _big_float_scalar(im, scale, self.EncodingPrecision)
        s0, s1 = [], []
        for i, q in enumerate(self.Q[: level + 1]):
            r, m = real % q, (imag % q) * self.imag_unit[i] % q
            s0.append((r + m) % q)
            s1.append((r - m) % q)
        return np.array(s0, dtype=np.uint64), np.array(s1, dtype=np.uint64)

    def _is_int(self, c):
        """bignum.Complex.IsInt of the constant as held at EncodingPrecision (evaluator.go:645)"""
        re, im = _as_complex_fraction(c)
        re, im = _round_bits(re, self.EncodingPrecision), _round_bits(im, self.EncodingPrecision)
        return re.denominator == 1 and im.denominator == 1

    def _addsub_ct(self, op0, op1, opOut, sub):
        from fractions import Fraction
        level = min(op0.level, op1.level, opOut.level)
        r = self.ringQ.AtLevel(level)
        a, b = op0, op1
        scale = op0.Scale
        if op0.Scale != op1.Scale:  # evaluateInPlace (:221-400): the smaller scale is brought up by the integer ratio
            if op0.Scale > op1.Scale:
                ratio = int(Fraction(op0.Scale) / Fraction(op1.Scale))
                if ratio > 0:
                    b = self.NewCiphertext(op1.Degree(), level, op1.Value[0].batch)
                    self.Mul(op1, ratio, b)
            else:
                ratio = int(Fraction(op1.Scale) / Fraction(op0.Scale))
                if ratio > 0:
                    a = self.NewCiphertext(op0.Degree(), level, op0.Value[0].batch)
                    self.Mul(op0, ratio, a)
                    scale = op1.Scale
        self._resize(opOut, max(op0.Degree(), op1.Degree()), level)
        lo = min(a.Degree(), b.Degree())
        for i in range(lo + 1):
            (r.Sub if sub else r.Add)(a.Value[i], b.Value[i], opOut.Value[i])
        for i in range(lo + 1, a.Degree() + 1):
            if opOut.Value[i] is not a.Value[i]:
                opOut.Value[i].CopyLvl(level, a.Value[i])
        for i in range(lo + 1, b.Degree() + 1):
            if sub:
                r.Neg(b.Value[i], opOut.Value[i])
            elif opOut.Value[i] is not b.Value[i]:
                opOut.Value[i].CopyLvl(level, b.Value[i])
        opOut.Scale = scale

    def Add(self, op0, op1, opOut):
        if isinstance(op1, Ciphertext):
            return self._addsub_ct(op0, op1, opOut, False)
        level = min(op0.level, opOut.level)  # constant (:68-86)
        self._resize(opOut, op0.Degree(), level)
        s0, s1 = self._rns(level, op0.Scale, op1)
        self.ringQ.AtLevel(level).AddDoubleRNSScalar(op0.Value[0], s0, s1, opOut.Value[0])
        if op0 is not opOut:
            for i in range(1, op0.Degree() + 1):
                opOut.Value[i].CopyLvl(level, op0.Value[i])
            opOut.Scale = op0.Scale

    def Sub(self, op0, op1, opOut):
        if isinstance(op1, Ciphertext):
            return self._addsub_ct(op0, op1, opOut, True)
        re, im = _as_complex_fraction(op1)
        self.Add(op0, (-re, -im), opOut)

    def _tensor(self, op0, op1, relin, opOut):
        level = min(op0.level, op1.level, opOut.level)
        scale = op0.Scale * op1.Scale
        if op0.Degree() == 0 or op1.Degree() == 0:  # plaintext (x) ciphertext (schemes/ckks/evaluator.go:842-870)
            pt, ct = (op0, op1) if op0.Degree() == 0 else (op1, op0)
            r = self.ringQ.AtLevel(level)
            c0 = Poly(self.ringQ, level + 1, pt.Value[0].batch)
            r.MForm(pt.Value[0], c0)
            src = list(ct.Value)
            self._resize(opOut, max(op0.Degree(), op1.Degree()), level)
            for a, o in zip(src, opOut.Value):
                r.MulCoeffsMontgomery(c0, a, o)
            opOut.Scale = scale
            return
        self._resize(opOut, 1 if relin else 2, level)
        if relin and self.rlk is None:
            raise KeyError("cannot relinearize: RelinearizationKey is nil")
        self.eval.CKKSMulRelin(level, op0.Value, op1.Value, self.rlk if relin else None, opOut.Value)
        opOut.Scale = scale

    def Mul(self, op0, op1, opOut):
        if isinstance(op1, Ciphertext):
            return self._tensor(op0, op1, False, opOut)
        level = min(op0.level, opOut.level)  # constant (:625-660)
        self._resize(opOut, op0.Degree(), level)
        scale = 1 if self._is_int(op1) else self.Q[level]
        s0, s1 = self._rns(level, scale, op1)
        r = self.ringQ.AtLevel(level)
        for i in range(op0.Degree() + 1):
            r.MulDoubleRNSScalar(op0.Value[i], s0, s1, opOut.Value[i])
        opOut.Scale = op0.Scale * scale

    def MulRelin(self, op0, op1, opOut):
        if isinstance(op1, Ciphertext):
            return self._tensor(op0, op1, True, opOut)
        self.Mul(op0, op1, opOut)

    MulNew = BGVCiphertextEvaluator.MulNew
    MulRelinNew = BGVCiphertextEvaluator.MulRelinNew

    def MulThenAdd(self, op0, op1, opOut):
        """constant operand (:893-935)"""
        from fractions import Fraction
        level = min(op0.level, opOut.level)
        self._resize(opOut, op0.Degree(), opOut.level)
        if op0.Scale == opOut.Scale:
            if self._is_int(op1):
                scale = 1
            else:
                scale = self.Q[level]
                self.Mul(opOut, scale, opOut)  # an integer: the scale factor of Mul is 1, so set the new scale by hand
                opOut.Scale = opOut.Scale * scale
        elif op0.Scale < opOut.Scale:
            scale = Fraction(opOut.Scale) / Fraction(op0.Scale)
        else:
            raise ValueError("cannot MulThenAdd: op0.Scale > opOut.Scale is not supported")
        s0, s1 = self._rns(level, scale, op1)
        r = self.ringQ.AtLevel(level)
        for i in range(op0.Degree() + 1):
            r.MulDoubleRNSScalarThenAdd(op0.Value[i], s0, s1, opOut.Value[i])

    def Relinearize(self, op0, opOut):
        BGVCiphertextEvaluator.Relinearize(self, op0, opOut)

    def Rescale(self, op0, opOut):
        """:477-515 with LevelsConsumedPerRescaling = 1"""
        from fractions import Fraction
        if op0.level <= 0:
            raise ValueError("cannot Rescale: input Ciphertext level is too low")
        level = op0.level
        r = self.ringQ.AtLevel(level)
        for a, o in zip(op0.Value, opOut.Value):
            r.DivRoundByLastModulusManyNTT(1, a, o)
        self._resize(opOut, op0.Degree(), level - 1)
        opOut.Scale = Fraction(op0.Scale) / self.Q[level]

"""Host-side mirror of circuits/ckks/mod1: the homomorphic evaluation of x mod 1 (bootstrapping's EvalMod) -- a scaled
sine evaluated as a Chebyshev polynomial of cos(2 pi (x - 1/4) / 2^r) followed by r double-angle steps -- as a driver over
the polynomial evaluator (polyeval.py) and the device-resident ckks.Evaluator mirror (schemes.py).

The approximation polynomial is the truncated Chebyshev series of the same function on the same interval with the same
degree, computed in 60-digit arithmetic (the reference interpolates at Chebyshev nodes in arbitrary precision,
circuits/ckks/mod1/mod1_parameters.go:150-215; the two differ by the aliased tail of the series, < 1e-40 here); only the
continuous sine / cosine types, the Han-Ki discrete cosine (cosine.py) and the optional arcsine are provided."""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

from .polyeval import Polynomial, PolynomialEvaluator

CosDiscrete, SinContinuous, CosContinuous = 0, 1, 2  # mod1.Type (mod1_parameters.go:19-23)


def _pi(prec: int):
    """pi to `prec` decimal digits (Machin: 16 atan(1/5) - 4 atan(1/239))"""
    import decimal
    with decimal.localcontext() as c:
        c.prec = prec + 10

        def atan_inv(x):
            x2, term, total, k = decimal.Decimal(x * x), decimal.Decimal(1) / x, decimal.Decimal(0), 0
            while abs(term) > decimal.Decimal(10) ** -(prec + 8):
                total += term / (2 * k + 1) if k % 2 == 0 else -term / (2 * k + 1)
                term /= x2
                k += 1
            return total

        return +(16 * atan_inv(5) - 4 * atan_inv(239))


def _bessel_j(k: int, a, prec: int):
    """J_k(a) = sum_m (-1)^m (a/2)^(2m+k) / (m! (m+k)!) in `prec`-digit decimal arithmetic"""
    import decimal
    with decimal.localcontext() as c:
        c.prec = prec + 20  # the alternating series cancels ~e^a / value digits
        h = a / 2
        term = h ** k
        for i in range(2, k + 1):
            term /= i
        total, m, h2 = decimal.Decimal(0), 0, h * h
        while abs(term) > decimal.Decimal(10) ** -(prec + 15) or m < 5:
            total += term
            m += 1
            term = -term * h2 / (m * (m + k))
        return +total


class Mod1Parameters:
    """mod1.Parameters from a ParametersLiteral (mod1_parameters.go:98-217).  The Chebyshev coefficients of
    cos(a u) = J_0(a) + 2 sum_{k even} (-1)^(k/2) J_k(a) T_k(u) and sin(a u) = 2 sum_{k odd} (-1)^((k-1)/2) J_k(a) T_k(u)
    (a = 2 pi K / 2^r) are computed in 60-digit arithmetic: the reduced value x mod 1 is ~2^-8 of the function's range, so
    double-precision coefficients would cap the precision of the whole bootstrap at ~30 bits relative to it."""

    PREC = 60

    def __init__(self, Q0: int, LevelQ: int, LogScale: int, Mod1Type: int, K: int, Mod1Degree: int, DoubleAngle: int = 0,
                 LogMessageRatio: int = 8, Scaling: float = 1.0, Mod1InvDegree: int = 0):
        import decimal
        self.LevelQ, self.LogDefaultScale, self.Mod1Type, self.LogMessageRatio = LevelQ, LogScale, Mod1Type, LogMessageRatio
        self.DoubleAngle = 0 if Mod1Type == SinContinuous else DoubleAngle
        self.K = float(K)
        self.QDiff = float(Q0) / 2.0 ** round(math.log2(float(Q0)))
        pi = _pi(self.PREC)
        with decimal.localcontext() as c:
            c.prec = self.PREC
            qdiff = decimal.Decimal(Q0) / (decimal.Decimal(2) ** round(math.log2(float(Q0))))
            self.Mod1InvPoly = None
            if Mod1InvDegree > 0:  # arcsine series of (1 / 2 pi) asin(x) scaled by qDiff * scaling (:117-137); then sqrt2pi = 1
                inv = [None] * (Mod1InvDegree + 1)
                cur = qdiff * decimal.Decimal(Scaling or 1.0) / (2 * pi)
                inv[1] = (Fraction(cur), Fraction(0))
                for i in range(3, Mod1InvDegree + 1, 2):
                    cur = cur * (i * i - 4 * i + 4) / (i * i - i)
                    inv[i] = (Fraction(cur), Fraction(0))
                self.Mod1InvPoly = Polynomial(inv, Basis="Monomial")
                self.Mod1InvPoly.IsEven = False
                s2p = decimal.Decimal(1)
            else:
                s2p = qdiff * decimal.Decimal(Scaling or 1.0) / (2 * pi)
                for _ in range(self.DoubleAngle):
                    s2p = s2p.sqrt()
            a = 2 * pi * decimal.Decimal(K) / (decimal.Decimal(2) ** self.DoubleAngle)
            coeffs = []
            if Mod1Type == CosDiscrete:  # Han-Ki interpolation around the integers; the odd coefficients are dropped (:186-196)
                from .cosine import ApproximateCos
                hk = ApproximateCos(K, Mod1Degree, float(1 << LogMessageRatio), self.DoubleAngle)
                coeffs = [None if (k & 1) else (Fraction(v * s2p), Fraction(0)) for k, v in enumerate(hk)]
            for k in range(Mod1Degree + 1 if Mod1Type != CosDiscrete else 0):
                odd = k & 1
                if (Mod1Type == SinContinuous) != bool(odd):
                    coeffs.append(None)
                    continue
                j = _bessel_j(k, a, self.PREC)
                sign = -1 if ((k - odd) // 2) & 1 else 1
                v = (j if k == 0 else 2 * j) * sign * s2p
                coeffs.append((Fraction(v), Fraction(0)))
            self.Sqrt2Pi = Fraction(s2p)
        self.Mod1Poly = Polynomial(coeffs, Basis="Chebyshev")
        if Mod1Type == SinContinuous:
            self.Mod1Poly.IsEven = False
        else:
            self.Mod1Poly.IsOdd = False

    def IntervalShrinkFactor(self) -> float:
        return 2.0 ** self.DoubleAngle

    def ScalingFactor(self) -> Fraction:
        return Fraction(1 << self.LogDefaultScale)

    def MessageRatio(self) -> float:
        return float(1 << self.LogMessageRatio)

    def Depth(self) -> int:
        inv = self.Mod1InvPoly.Degree().bit_length() if self.Mod1InvPoly is not None else 0
        return self.Mod1Poly.Degree().bit_length() + self.DoubleAngle + inv


def _bigfloat_round(x: Fraction, prec: int = 128) -> Fraction:
    """x rounded to `prec` significant bits, ties to even: what a big.Float of that precision keeps of an exact product
    (rlwe.Scale.Mul, core/rlwe/scale.go:77-93, ScalePrecision = 128)"""
    if x == 0:
        return Fraction(0)
    e = x.numerator.bit_length() - x.denominator.bit_length()  # 2^(e-1) < x < 2^(e+1)
    sh = prec - e
    n = x * Fraction(2) ** sh
    while n >= 1 << prec:
        sh -= 1
        n = x * Fraction(2) ** sh
    while n < 1 << (prec - 1):
        sh += 1
        n = x * Fraction(2) ** sh
    m, rem = divmod(n.numerator, n.denominator)
    twice = 2 * rem
    if twice > n.denominator or (twice == n.denominator and m & 1):
        m += 1
    return Fraction(m) / Fraction(2) ** sh


def _bigfloat_sqrt(x: Fraction, prec: int = 128) -> Fraction:
    """sqrt(x) correctly rounded to `prec` bits (big.Float.Sqrt on the 128-bit scale, mod1_evaluator.go:57; Go's Newton iteration
    carries 32 guard bits, so it returns this value except with probability ~2^-32)"""
    e = x.numerator.bit_length() - x.denominator.bit_length()
    k = prec - e // 2 + 1  # sqrt(x) 2^k has prec + 1 or prec + 2 bits
    t = x * Fraction(4) ** k
    r = math.isqrt(t.numerator // t.denominator)  # floor(sqrt(x) 2^k)
    exact = Fraction(r * r) == t
    # r carries at least one bit below the result's last place: sticky-extend and round once
    return _bigfloat_round(Fraction(2 * r + (0 if exact else 1), 2) / Fraction(2) ** k, prec)


class Mod1Evaluator:
    """mod1.Evaluator (circuits/ckks/mod1/mod1_evaluator.go:17-144)"""

    def __init__(self, evaluator, params: Mod1Parameters):
        self.eval, self.Parameters = evaluator, params
        self.PolynomialEvaluator = PolynomialEvaluator(evaluator)

    def EvaluateNew(self, ct):
        """the input slots are x / K with |x| < K; the result is QDiff / (2 pi) * sin(2 pi x) at the scale of ct"""
        ev, evm = self.eval, self.Parameters
        if ct.Level() < evm.LevelQ:
            raise ValueError("cannot Evaluate: ct.Level() < Mod1Parameters.LevelQ")
        res = ev.CopyNew(ct)
        if res.Level() > evm.LevelQ:  # DropLevel (:38-40)
            ev._resize(res, res.Degree(), evm.LevelQ) if hasattr(ev, "_resize") else ev._set(res, res.Value, evm.LevelQ)
        res.Scale = evm.ScalingFactor()  # normalise the reduction to mod 1 (:45)
        Qi = ev.Q
        targetScale = Fraction(res.Scale)
        depth = evm.Mod1Poly.Depth()
        for i in range(evm.DoubleAngle):  # :54-58
            targetScale = _bigfloat_round(targetScale * Qi[res.Level() - depth - evm.DoubleAngle + i + 1])
            targetScale = _bigfloat_sqrt(targetScale)
        if evm.Mod1Type in (CosContinuous, CosDiscrete):  # change of variable x -> x - 1/4 (:61-68)
            Kp = evm.K / evm.IntervalShrinkFactor()
            offset = Fraction(-0.5) / (Fraction(2 * Kp) * Fraction(evm.IntervalShrinkFactor()))
            ev.Add(res, (offset, 0), res)
        sqrt2pi = evm.Sqrt2Pi
        res = self.PolynomialEvaluator.Evaluate(res, evm.Mod1Poly, targetScale)  # Chebyshev evaluation (:96)
        for _ in range(evm.DoubleAngle):  # cos(2a) = 2 cos(a)^2 - 1 (:100-118)
            sqrt2pi *= sqrt2pi
            ev.MulRelin(res, res, res)
            ev.Add(res, res, res)
            ev.Add(res, (-sqrt2pi, 0), res)
            ev.Rescale(res, res)
        if evm.Mod1InvPoly is not None:  # arcsine (:121-138)
            res = self.PolynomialEvaluator.Evaluate(res, evm.Mod1InvPoly, res.Scale)
        res.Scale = ct.Scale  # multiplies back by q (:141)
        return res

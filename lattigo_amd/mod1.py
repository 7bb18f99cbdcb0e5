"""Host-side mirror of circuits/ckks/mod1: the homomorphic evaluation of x mod 1 (bootstrapping's EvalMod) -- a scaled
sine evaluated as a Chebyshev polynomial of cos(2 pi (x - 1/4) / 2^r) followed by r double-angle steps -- as a driver over
the polynomial evaluator (polyeval.py) and the device-resident ckks.Evaluator mirror (schemes.py).

The approximation polynomial is generated here by Chebyshev interpolation in double precision (the reference uses
arbitrary-precision interpolation, circuits/ckks/mod1/mod1_parameters.go:150-215): same function, same interval, same
degree; only the continuous sine / cosine types are provided (no Han-Ki discrete cosine, no arcsine)."""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

from .polyeval import Polynomial, PolynomialEvaluator

SinContinuous, CosContinuous = 1, 2  # mod1.Type (mod1_parameters.go:19-23)


class Mod1Parameters:
    """mod1.Parameters from a ParametersLiteral (mod1_parameters.go:98-217)"""

    def __init__(self, Q0: int, LevelQ: int, LogScale: int, Mod1Type: int, K: int, Mod1Degree: int, DoubleAngle: int = 0,
                 LogMessageRatio: int = 8, Scaling: float = 1.0):
        self.LevelQ, self.LogDefaultScale, self.Mod1Type, self.LogMessageRatio = LevelQ, LogScale, Mod1Type, LogMessageRatio
        self.DoubleAngle = 0 if Mod1Type == SinContinuous else DoubleAngle
        scFac = 2.0 ** self.DoubleAngle
        Kp = K / scFac
        self.K = float(K)
        self.QDiff = float(Q0) / 2.0 ** round(math.log2(float(Q0)))
        self.Sqrt2Pi = (0.15915494309189535 * self.QDiff * (Scaling or 1.0)) ** (1.0 / scFac)
        f = (lambda u: np.sin(2 * np.pi * Kp * u)) if Mod1Type == SinContinuous else (lambda u: np.cos(2 * np.pi * Kp * u))
        coeffs = np.polynomial.chebyshev.chebinterpolate(f, Mod1Degree) * self.Sqrt2Pi
        drop = 0 if Mod1Type == SinContinuous else 1  # sine: odd polynomial, cosine: even polynomial
        self.Mod1Poly = Polynomial([None if (i & 1) == drop else (Fraction(float(c)), Fraction(0)) for i, c in enumerate(coeffs)],
                                   Basis="Chebyshev")
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
        return self.Mod1Poly.Degree().bit_length() + self.DoubleAngle


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
            targetScale = targetScale * Qi[res.Level() - depth - evm.DoubleAngle + i + 1]
            targetScale = Fraction(math.sqrt(float(targetScale)))
        if evm.Mod1Type == CosContinuous:  # change of variable x -> x - 1/4 (:61-68)
            Kp = evm.K / evm.IntervalShrinkFactor()
            offset = Fraction(-0.5) / (Fraction(2 * Kp) * Fraction(evm.IntervalShrinkFactor()))
            ev.Add(res, (offset, 0), res)
        sqrt2pi = evm.Sqrt2Pi
        res = self.PolynomialEvaluator.Evaluate(res, evm.Mod1Poly, targetScale)  # Chebyshev evaluation (:96)
        for _ in range(evm.DoubleAngle):  # cos(2a) = 2 cos(a)^2 - 1 (:100-118)
            sqrt2pi *= sqrt2pi
            ev.MulRelin(res, res, res)
            ev.Add(res, res, res)
            ev.Add(res, (Fraction(-sqrt2pi), 0), res)
            ev.Rescale(res, res)
        res.Scale = ct.Scale  # multiplies back by q (:141)
        return res

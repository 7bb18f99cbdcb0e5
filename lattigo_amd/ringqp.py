"""Host-side mirror of ringqp.Ring (ring/ringqp/ring.go:15, operations.go:8-349): the pair (RingQ, RingP); every method is
the RingQ operation at levelQ followed by the RingP operation at levelP on ``(Q, P)`` pairs of device polynomials."""
from __future__ import annotations

from ._lib import check, load
from .ring import Poly, Ring
from .rlwe import Evaluator

_BIN = ("Add", "AddLazy", "Sub", "MulCoeffsMontgomery", "MulCoeffsMontgomeryLazy", "MulCoeffsMontgomeryLazyThenAddLazy",
        "MulCoeffsMontgomeryThenSub", "MulCoeffsMontgomeryLazyThenSubLazy", "MulCoeffsMontgomeryThenAdd")
_UN = ("Neg", "NTT", "INTT", "NTTLazy", "INTTLazy", "MForm", "IMForm", "Reduce")


class RingQP:
    def __init__(self, ringQ: Ring, ringP: Ring, evaluator: Evaluator | None = None):
        self.RingQ, self.RingP, self._eval = ringQ, ringP, evaluator

    def AtLevel(self, levelQ: int, levelP: int) -> "RingQP":
        return RingQP(self.RingQ.AtLevel(levelQ), self.RingP.AtLevel(levelP), self._eval)

    def NewPoly(self, batch: int = 1):
        return (Poly(self.RingQ, self.RingQ.Level() + 1, batch), Poly(self.RingP, self.RingP.Level() + 1, batch))

    def MulScalar(self, p1, scalar: int, p2):
        self.RingQ.MulScalar(p1[0], scalar, p2[0])
        self.RingP.MulScalar(p1[1], scalar, p2[1])

    def MulRNSScalarMontgomery(self, p1, scalarQ, scalarP, p2):
        """operations.go: MulRNSScalarMontgomery with an RNSScalar{Q, P}"""
        self.RingQ.MulRNSScalarMontgomery(p1[0], scalarQ, p2[0])
        self.RingP.MulRNSScalarMontgomery(p1[1], scalarP, p2[1])

    def AutomorphismNTT(self, p1, galEl: int, p2):
        self.AutomorphismNTTWithIndex(p1, self.RingQ.AutomorphismNTTIndex(galEl), p2)

    def AutomorphismNTTWithIndex(self, p1, index, p2):
        self.RingQ.AutomorphismNTTWithIndex(p1[0], index, p2[0])
        self.RingP.AutomorphismNTTWithIndex(p1[1], index, p2[1])

    def AutomorphismNTTWithIndexThenAddLazy(self, p1, index, p2):
        self.RingQ.AutomorphismNTTWithIndexThenAddLazy(p1[0], index, p2[0])
        self.RingP.AutomorphismNTTWithIndexThenAddLazy(p1[1], index, p2[1])

    def Automorphism(self, p1, galEl: int, p2):
        self.RingQ.Automorphism(p1[0], galEl, p2[0])
        self.RingP.Automorphism(p1[1], galEl, p2[1])

    def ExtendBasisSmallNormAndCenter(self, polyInQ: Poly, levelP: int, polyOutQ: Poly, polyOutP: Poly):
        """operations.go:325-349: the small-norm polynomial (read from limb 0 of Q) extended to P limbs 0..levelP"""
        if self._eval is None:
            raise ValueError("ExtendBasisSmallNormAndCenter needs the evaluator the rings belong to")
        if polyOutQ is not polyInQ:
            polyOutQ.CopyLvl(min(polyInQ.Level(), polyOutQ.Level()), polyInQ)
        lq = polyOutQ.Level()
        check(load().he_centered_lift(self._eval.h, 3, polyInQ.h, lq + 1, lq, polyOutQ.h, levelP, polyOutP.h))


def _bin(name):
    def f(self, p1, p2, p3):
        getattr(self.RingQ, name)(p1[0], p2[0], p3[0])
        getattr(self.RingP, name)(p1[1], p2[1], p3[1])
    f.__name__ = name
    return f


def _un(name):
    def f(self, p1, p2):
        getattr(self.RingQ, name)(p1[0], p2[0])
        getattr(self.RingP, name)(p1[1], p2[1])
    f.__name__ = name
    return f


for _n in _BIN:
    setattr(RingQP, _n, _bin(_n))
for _n in _UN:
    setattr(RingQP, _n, _un(_n))

"""Host-side mirror of the ring-level part of CKKS bootstrapping's first step, bootstrapping.Evaluator.ModUp
(circuits/ckks/bootstrapping/evaluator.go:612-769): raise a level-0 ciphertext from q = Q[0] to Q by centred lifts
(optionally through the sparse-secret encapsulation keys), rescale the message, apply the Trace.  A driver over the
device-resident operators; the homomorphic DFTs sit on lintrans.py, the modular reduction (polynomial evaluation) is a
host-driven circuit that is not part of this package."""
from __future__ import annotations

from ._lib import check, load
from .ring import Poly
from .rlwe import Decomposition, EvaluationKey, Evaluator, InnerSumEvaluator


def ApplyEvaluationKey(ev: Evaluator, level: int, ctIn, evk: EvaluationKey, opOut):
    """rlwe.Evaluator.ApplyEvaluationKey, same ring degree (core/rlwe/evaluator_evaluationkey.go:36,98-106)"""
    B = ctIn[0].batch
    tmp = [Poly(ev.ringQ, level + 1, B), Poly(ev.ringQ, level + 1, B)]
    ev.GadgetProduct(level, ctIn[1], evk, tmp)
    ev.ringQ.AtLevel(level).Add(ctIn[0], tmp[0], opOut[0])
    opOut[1].CopyLvl(level, tmp[1])


def centered_lift(ev: Evaluator, strict: bool, src: Poly, first_q: int, levelQ: int, dstQ: Poly, levelP: int = -1,
                  dstP: Poly | None = None):
    """the coefficient loops of ModUp (:654-667 `>=`, :677-696 `>`, :742-755 `>=`)"""
    check(load().he_centered_lift(ev.h, int(strict), src.h, first_q, levelQ, dstQ.h, levelP, dstP.h if dstP is not None else 0))


def ModUp(ev: Evaluator, ise: InnerSumEvaluator, levelIn: int, ct, scale: float, logSlots: int,
          EvkDenseToSparse: EvaluationKey | None = None, EvkSparseToDense: EvaluationKey | None = None):
    """bootstrapping.Evaluator.ModUp.  ct = [c0, c1]: NTT-domain polynomials allocated at the maximum level of the
    bootstrapping ring, meaningful on limbs 0..levelIn; modified in place (as the reference) and returned at the top
    level.  `scale` = (Mod1Parameters.ScalingFactor / MessageRatio) / ct.Scale (:711, :759); logSlots parametrises the
    final Trace (:768)."""
    ringQ, ringP = ev.ringQ, ev.ringP
    levelQ, levelP = ringQ.MaxLevel(), ringP.MaxLevel()
    B = ct[0].batch
    if EvkDenseToSparse is not None:  # switch to the sparse key (:615-619)
        ApplyEvaluationKey(ev, levelIn, ct, EvkDenseToSparse, ct)
    rIn, rQ, rP = ringQ.AtLevel(levelIn), ringQ.AtLevel(levelQ), ringP.AtLevel(levelP)
    for c in ct:
        rIn.INTT(c, c)
    centered_lift(ev, False, ct[0], 1, levelQ, ct[0])  # ModUp q->Q for ct[0] centred around q (:654-667)
    scalar = int(round(scale)) if scale > 1 else None
    if EvkSparseToDense is not None:
        liftQ, liftP = Poly(ringQ, levelQ + 1, B), Poly(ringP, levelP + 1, B)
        centered_lift(ev, True, ct[1], 0, levelQ, liftQ, levelP, liftP)  # q->QP for ct[1] (:677-696)
        rQ.NTT(liftQ, liftQ)
        rP.NTT(liftP, liftP)
        rQ.NTT(ct[0], ct[0])
        if scalar is not None:  # :711-723
            rQ.MulScalar(liftQ, scalar, liftQ)
            rP.MulScalar(liftP, scalar, liftP)
            rQ.MulScalar(ct[0], scalar, ct[0])
        decomp = Decomposition(ev, B)
        check(load().he_decomp_fill(decomp.h, levelQ, levelP, liftQ.h, liftP.h))  # every digit = the lifted poly (:699-705)
        tmp0 = Poly(ringQ, levelQ + 1, B)
        ev.GadgetProductHoisted(levelQ, decomp, EvkSparseToDense, [tmp0, ct[1]])  # back to the dense key (:733)
        rQ.Add(ct[0], tmp0, ct[0])
    else:
        centered_lift(ev, False, ct[1], 1, levelQ, ct[1])  # :742-755
        for c in ct:
            rQ.NTT(c, c)
            if scalar is not None:
                rQ.MulScalar(c, scalar, c)
    ise.Trace(levelQ, ct, logSlots, ct)  # SubSum X -> (N/dslots) * Y^dslots (:768)
    return ct

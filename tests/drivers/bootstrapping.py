"""Host-side mirror of the ring-level part of CKKS bootstrapping's first step, bootstrapping.Evaluator.ModUp
(circuits/ckks/bootstrapping/evaluator.go:612-769): raise a level-0 ciphertext from q = Q[0] to Q by centred lifts
(optionally through the sparse-secret encapsulation keys), rescale the message, apply the Trace.  A driver over the
device-resident operators; the homomorphic DFTs sit on lintrans.py, the modular reduction (polynomial evaluation) is a
host-driven circuit that is not part of this package."""
from __future__ import annotations

from lattigo_amd._lib import check, load
from lattigo_amd.ring import Poly
from lattigo_amd.rlwe import Decomposition, EvaluationKey, Evaluator, InnerSumEvaluator


def ApplyEvaluationKey(ev: Evaluator, level: int, ctIn, evk: EvaluationKey, opOut):
    """rlwe.Evaluator.ApplyEvaluationKey, same ring degree (core/rlwe/evaluator_evaluationkey.go:36,98-106)"""
    B = ctIn[0].batch
    tmp = [Poly(ev.ringQ, level + 1, B, zero=False), Poly(ev.ringQ, level + 1, B, zero=False)]
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
        tmp0 = Poly(ringQ, levelQ + 1, B, zero=False)
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


# ----------------------------------------------------------------------------------------------------------------
# The bootstrapping circuit proper: bootstrapping.Evaluator.bootstrap (circuits/ckks/bootstrapping/evaluator.go:518-560)
# with the homomorphic DFT steps of circuits/ckks/dft/dft.go:240-340, as control flow over backend adapters.
# ----------------------------------------------------------------------------------------------------------------
class DeviceBootstrapBackend:
    """Adapters binding the driver below to the device-resident operators: `ckks` a schemes.CKKSCiphertextEvaluator,
    `lte` a lintrans.LinTransEvaluator, `ise` an rlwe.InnerSumEvaluator, Galois keys in `lte.gks`."""

    def __init__(self, ckks, lte, ise, EvkDenseToSparse=None, EvkSparseToDense=None):
        self.ckks, self.lte, self.ise = ckks, lte, ise
        self.d2s, self.s2d = EvkDenseToSparse, EvkSparseToDense

    def modup(self, ct, scale, logSlots):
        from .schemes import Ciphertext
        ev = self.ckks.eval
        B, top = ct.Value[0].batch, ev.ringQ.MaxLevel()
        full = [Poly(ev.ringQ, top + 1, B) for _ in range(2)]
        for a, b in zip(ct.Value, full):
            b.CopyLvl(ct.level, a)
        ModUp(ev, self.ise, ct.level, full, scale, logSlots, self.d2s, self.s2d)
        return Ciphertext(full, top, ct.Scale * (int(round(scale)) if scale > 1 else 1))

    def lintrans(self, ct, matrix, matrix_scale):
        from .schemes import Ciphertext
        level = min(ct.level, matrix.LevelQ)
        out = [Poly(self.ckks.ringQ, level + 1, ct.Value[0].batch) for _ in range(2)]
        self.lte.EvaluateMany(ct.level, ct.Value, [matrix], [out])
        return Ciphertext(out, level, ct.Scale * matrix_scale)

    def conjugate(self, ct):
        from .schemes import Ciphertext
        g = self.ckks.ringQ.NthRoot() - 1
        out = [Poly(self.ckks.ringQ, ct.level + 1, ct.Value[0].batch, zero=False) for _ in range(2)]
        self.ckks.eval.Automorphism(ct.level, ct.Value, g, self.lte.gks.GetGaloisKey(g), out)
        return Ciphertext(out, ct.level, ct.Scale)


    def stack(self, cts):
        """independent ciphertexts of equal level / scale / batch -> one ciphertext whose batch is their concatenation"""
        from .schemes import Ciphertext
        a = cts[0]
        B, lv = a.Value[0].batch, a.level
        out = [Poly(self.ckks.ringQ, lv + 1, B * len(cts), zero=False) for _ in range(a.Degree() + 1)]
        for i, c in enumerate(cts):
            if (c.level, c.Scale, c.Degree(), c.Value[0].batch) != (lv, a.Scale, a.Degree(), B):
                raise ValueError("stack: ciphertexts must share level, scale, degree and batch")
            for o, v in zip(out, c.Value):
                o.CopyBatch(lv, i * B, v, 0, B)
        return Ciphertext(out, lv, a.Scale)

    def unstack(self, ct, parts: int):
        from .schemes import Ciphertext
        B = ct.Value[0].batch // parts
        res = []
        for i in range(parts):
            vals = [Poly(self.ckks.ringQ, ct.level + 1, B, zero=False) for _ in ct.Value]
            for o, v in zip(vals, ct.Value):
                o.CopyBatch(ct.level, 0, v, i * B, B)
            res.append(Ciphertext(vals, ct.level, ct.Scale))
        return res


class Bootstrapper:
    """bootstrapping.Evaluator.bootstrap for fully packed ciphertexts: ModUp -> CoeffsToSlots -> EvalMod (real and
    imaginary halves) -> SlotsToCoeffs.  `backend` provides modup / lintrans / conjugate and the ckks evaluator; the encoded
    DFT matrices (one each here; the reference factorises them, circuits/ckks/dft) and their plaintext scales are inputs."""

    def __init__(self, backend, mod1_evaluator, cts_matrix, cts_scale, stc_matrix, stc_scale, modup_scale: float = 1.0,
                 logSlots: int | None = None):
        """cts_matrix / stc_matrix: one encoded matrix or a list of them (the factors of the homomorphic DFT, applied in
        order, one level each); cts_scale / stc_scale: the matching plaintext scale(s)"""
        self.be, self.mod1 = backend, mod1_evaluator
        as_list = lambda m, sc: (list(m), list(sc)) if isinstance(m, (list, tuple)) else ([m], [sc])
        self.cts, self.cts_scale = as_list(cts_matrix, cts_scale)
        self.stc, self.stc_scale = as_list(stc_matrix, stc_scale)
        self.modup_scale, self.logSlots = modup_scale, logSlots

    def _dft(self, ct, matrices, scales):
        """dft.Evaluator.dft (circuits/ckks/dft/dft.go:336-366): one linear transformation + Rescale per level"""
        ev = self.be.ckks
        for m, sc in zip(matrices, scales):
            ct = self.be.lintrans(ct, m, sc)
            ev.Rescale(ct, ct)
        return ct

    def CoeffsToSlots(self, ct):
        """dft.Evaluator.CoeffsToSlots, SplitRealAndImag (circuits/ckks/dft/dft.go:240-277)"""
        ev = self.be.ckks
        zV = self._dft(ct, self.cts, self.cts_scale)
        ctReal = self.be.conjugate(zV)
        ctImag = ev.NewCiphertext(1, zV.level, getattr(zV.Value[0], "batch", 1))
        ev.Sub(zV, ctReal, ctImag)
        ev.Mul(ctImag, (0, -1), ctImag)  # * -i (a Gaussian integer: no scale change)
        ev.Add(ctReal, zV, ctReal)
        return ctReal, ctImag

    def SlotsToCoeffs(self, ctReal, ctImag):
        """dft.Evaluator.SlotsToCoeffs (:313-333)"""
        ev = self.be.ckks
        out = ev.NewCiphertext(1, min(ctReal.level, ctImag.level), getattr(ctReal.Value[0], "batch", 1))
        ev.Mul(ctImag, (0, 1), out)
        ev.Add(out, ctReal, out)
        return self._dft(out, self.stc, self.stc_scale)

    def ScaleDown(self, ct, message_ratio: float):
        """bootstrapping.Evaluator.ScaleDown (circuits/ckks/bootstrapping/evaluator.go:566-610) for a ciphertext that can be
        brought to level 0: drop the unnecessary primes, multiply by round((Q_level / scale) / MessageRatio) so that the message
        sits MessageRatio below Q[0]; returns the ciphertext and the scale error factor"""
        from fractions import Fraction
        ev = self.be.ckks
        Q = ev.Q
        res = ev.CopyNew(ct)

        def modulus(level):
            m = 1
            for x in Q[: level + 1]:
                m *= x
            return m

        while res.level != 0 and Fraction(modulus(res.level)) / Fraction(res.Scale) >= Fraction(Q[res.level]) * Fraction(message_ratio):
            (ev._resize(res, res.Degree(), res.level - 1) if hasattr(ev, "_resize") else ev._set(res, res.Value, res.level - 1))
        if res.level != 0:
            raise ValueError("ScaleDown: the message is too large to be brought to level 0 (RescaleTo path not built)")
        scale_up = Fraction(modulus(res.level)) / Fraction(res.Scale) / Fraction(message_ratio)
        if scale_up < Fraction(1, 2):
            raise ValueError("initial Q/Scale < 0.5*Q[0]/MessageRatio")
        k = int(scale_up)  # Scale.BigInt() truncates
        ev.Mul(res, k, res)
        res.Scale = Fraction(res.Scale) * k
        target = Fraction(Q[0]) / Fraction(message_ratio)
        return res, Fraction(res.Scale) / target

    def Bootstrap(self, ct, work_scale):
        """ct: level-0 ciphertext.  `work_scale` is the scale metadata given to the raised ciphertext (the reference sets it
        through ScaleDown / the ModUp message scaling and Mod1Parameters.ScalingFactor)."""
        logSlots = self.logSlots if self.logSlots is not None else self.be.ckks.ringQ.N.bit_length() - 2
        up = self.be.modup(ct, self.modup_scale, logSlots)
        up.Scale = work_scale
        ctReal, ctImag = self.CoeffsToSlots(up)
        if hasattr(self.be, "stack"):  # the two halves are independent: evaluate EvalMod on them as one batch
            ctReal, ctImag = self.be.unstack(self.mod1.EvaluateNew(self.be.stack([ctReal, ctImag])), 2)
        else:
            ctReal = self.mod1.EvaluateNew(ctReal)
            ctImag = self.mod1.EvaluateNew(ctImag)
        return self.SlotsToCoeffs(ctReal, ctImag)

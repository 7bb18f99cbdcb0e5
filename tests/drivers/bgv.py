"""Host-side mirror of the BFV-style (scale-invariant) multiplication of the reference's bgv package:
Evaluator.tensorScaleInvariant / modUpAndNTT / tensorLowDeg / quantize (schemes/bgv/evaluator.go:898-1071), as a driver
over the device-resident ring and basis-extension operators.  The standard BGV tensoring is the fused
``Evaluator.BGVMulRelin`` (rlwe.py)."""
from __future__ import annotations

from lattigo_amd.ring import BasisExtender, Poly, Ring
from lattigo_amd.rlwe import EvaluationKey, Evaluator


class ScaleInvariantEvaluator:
    """bgv.Evaluator with ScaleInvariant = true, restricted to ct x ct MulRelinScaleInvariant.
    ringQMul is Parameters.RingQMul() (schemes/bgv/params.go:98-109), t the plaintext modulus."""

    def __init__(self, evaluator: Evaluator, ringQMul: Ring, t: int):
        self.eval, self.t = evaluator, int(t)
        self.ringQ, self.ringQMul = evaluator.ringQ, ringQMul
        self.be = BasisExtender(self.ringQ, ringQMul)  # basisExtenderQ1toQ2 (evaluator.go:58)
        logN = self.ringQ.N.bit_length() - 1
        self.levelQMul, Q = [], 1
        for m in self.ringQ.ModuliChain():  # evaluator.go:43-48
            Q *= int(m)
            self.levelQMul.append(-(-(Q.bit_length() + logN) // 61) - 1)
        if self.levelQMul[-1] > ringQMul.MaxLevel():
            raise ValueError("ringQMul has too few moduli for this Q")

    def _mod_up_and_ntt(self, level, lm, ct, B):
        """modUpAndNTT (:991)"""
        rQ, rM = self.ringQ.AtLevel(level), self.ringQMul.AtLevel(lm)
        buff, out = Poly(self.ringQ, level + 1, B), []
        for c in ct:
            rQ.INTT(c, buff)
            o = Poly(self.ringQMul, lm + 1, B)
            self.be.ModUpQtoP(level, lm, buff, o)
            rM.NTTLazy(o, o)
            out.append(o)
        return out

    def _quantize(self, level, lm, c2Q1: Poly, c2Q2: Poly):
        """quantize (:1050): scale by t/Q and return to basis Q"""
        rQ, rM = self.ringQ.AtLevel(level), self.ringQMul.AtLevel(lm)
        rQ.INTTLazy(c2Q1, c2Q1)
        rM.INTTLazy(c2Q2, c2Q2)
        self.be.ModDownQPtoP(level, lm, c2Q1, c2Q2, c2Q2)  # QP / Q -> P
        self.be.ModUpPtoQ(lm, level, c2Q2, c2Q1)            # centred, back to Q
        rQ.MulScalar(c2Q1, self.t, c2Q1)                    # (ct/Q) * T
        rQ.NTT(c2Q1, c2Q1)

    def MulRelinScaleInvariant(self, level, op0, op1, rlk: EvaluationKey | None, opOut):
        """tensorScaleInvariant (:898): op0, op1 = [c0, c1] NTT at `level` (op1 is op0: squaring branch); opOut = [c0, c1]
        with a relinearisation key, [c0, c1, c2] without."""
        square = op1 is op0
        lm = self.levelQMul[level]
        rQ, rM = self.ringQ.AtLevel(level), self.ringQMul.AtLevel(lm)
        B = op0[0].batch
        t0M = self._mod_up_and_ntt(level, lm, op0, B)
        t1M = t0M if square else self._mod_up_and_ntt(level, lm, op1, B)
        c2 = Poly(self.ringQ, level + 1, B) if rlk is not None else opOut[2]
        outQ = [opOut[0], opOut[1], c2]
        outM = [Poly(self.ringQMul, lm + 1, B) for _ in range(3)]
        # tensorLowDeg (:1003)
        c00, c01 = Poly(self.ringQ, level + 1, B), Poly(self.ringQ, level + 1, B)
        c00M, c01M = Poly(self.ringQMul, lm + 1, B), Poly(self.ringQMul, lm + 1, B)
        rQ.MForm(op0[0], c00)
        rQ.MForm(op0[1], c01)
        rM.MForm(t0M[0], c00M)
        rM.MForm(t0M[1], c01M)
        for r, a0, a1, b, o in ((rQ, c00, c01, op0 if square else op1, outQ), (rM, c00M, c01M, t1M, outM)):
            r.MulCoeffsMontgomery(a0, b[0], o[0])      # c0 = c0[0]*c1[0]
            r.MulCoeffsMontgomery(a1, b[1], o[2])      # c2 = c0[1]*c1[1]
            r.MulCoeffsMontgomery(a0, b[1], o[1])      # c1 = c0[0]*c1[1] + c0[1]*c1[0]
            if square:
                r.AddLazy(o[1], o[1], o[1])
            else:
                r.MulCoeffsMontgomeryThenAddLazy(a1, b[0], o[1])
        for k in range(3):
            self._quantize(level, lm, outQ[k], outM[k])
        if rlk is not None:
            tmp = [Poly(self.ringQ, level + 1, B), Poly(self.ringQ, level + 1, B)]
            self.eval.GadgetProduct(level, c2, rlk, tmp)  # :966
            rQ.Add(opOut[0], tmp[0], opOut[0])
            rQ.Add(opOut[1], tmp[1], opOut[1])

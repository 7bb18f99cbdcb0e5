"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's host-side drivers that sit on
rlwe.EvaluatorProvider: core/rlwe/inner_sum.go (Trace, PartialTracesSum, InnerSum, Replicate) and
circuits/common/lintrans/lintrans_evaluator.go (MultiplyByDiagMatrix, MultiplyByDiagMatrixBSGS, EvaluateMany).
Built on the C oracle's primitives (oracle.py); functional style: arrays in, arrays out.  A ciphertext is an array
[2][limbs][N], a QP element a pair (Q [limbs_q][N], P [limbs_p][N]).  Nothing under lattigo_amd/ imports this."""
from __future__ import annotations

import numpy as np

from . import oracle as O

GaloisGen = 5  # core/rlwe/params.go:33


def GaloisElement(nth_root: int, k: int) -> int:
    """core/rlwe/params.go:580"""
    return O.ModExp(GaloisGen, (k & 0xFFFFFFFFFFFFFFFF) & (nth_root - 1), nth_root)


def _prod(ms):
    r = 1
    for m in ms:
        r *= int(m)
    return r


class InnerSumEvaluator:
    def __init__(self, evaluator: O.Evaluator, gks: dict):
        self.eval, self.gks = evaluator, gks
        self.ringQ, self.ringP = evaluator.ringQ, evaluator.ringP
        self.be = O.BasisExtender(self.ringQ, self.ringP)
        self.nth_root = self.ringQ.NthRoot()
        self.logN = self.ringQ.N.bit_length() - 1

    def GaloisElement(self, k):
        return GaloisElement(self.nth_root, k)

    def Trace(self, ctIn, logN, isNTT=True):
        """core/rlwe/inner_sum.go:36-117 (both ring types: :60-62, :97)"""
        ctIn = np.asarray(ctIn, dtype=np.uint64)
        level = ctIn.shape[1] - 1
        rQ = self.ringQ
        gap = 1 << (self.logN - logN - 1)
        if logN == 0:
            gap <<= 1
        if gap <= 1:
            return ctIn.copy()
        ci = getattr(rQ, "conjugate_invariant", False)
        if ci:
            gap >>= 1  # :60-62: the last step, phi(5^-1), is skipped
        ninv = pow(gap, -1, _prod(rQ.moduli[: level + 1]))
        out = np.stack([rQ.MulScalarBigint(ctIn[k], ninv) for k in range(2)])  # :68-70
        if not isNTT:
            out = np.stack([rQ.NTT(out[k]) for k in range(2)])
        steps = [self.GaloisElement(1 << i) for i in range(logN, self.logN - 1)]  # :82
        if logN == 0 and not ci:
            steps.append(self.nth_root - 1)  # :97 (ringQ.Type() == ring.Standard)
        for galEl in steps:
            buff = self.eval.Automorphism(out, galEl, self.gks[galEl])
            out = np.stack([rQ.binop("Add", out[k], buff[k]) for k in range(2)])
        if not isNTT:
            out = np.stack([rQ.INTT(out[k]) for k in range(2)])
        return out

    def PartialTracesSum(self, ctIn, offset, n, isNTT=True):
        """core/rlwe/inner_sum.go:147-294"""
        if n == 0 or offset == 0:
            raise ValueError("partialtrace: invalid parameter (n = 0 or batchSize = 0)")
        ctIn = np.asarray(ctIn, dtype=np.uint64)
        levelQ, levelP = ctIn.shape[1] - 1, self.ringP.MaxLevel()
        rQ, rP = self.ringQ, self.ringP
        ctInNTT = ctIn.copy() if isNTT else np.stack([rQ.NTT(ctIn[k]) for k in range(2)])
        opOut = None
        if n == 1:
            opOut = ctIn.copy()
        else:
            accQ = accP = None
            state, copy = False, True
            i, j = 0, n
            while j > 0:  # :216
                dq, dp = self.eval.DecomposeNTT(levelQ, levelP, levelP + 1, ctInNTT[1], True)
                if j & 1:
                    k = (n - (n & ((2 << i) - 1))) * offset
                    if k != 0:
                        rot = self.GaloisElement(k)
                        cQ, cP = self.eval.AutomorphismHoistedLazy(levelQ, ctInNTT[0], dq, dp, rot, self.gks[rot])
                        if copy:
                            accQ, accP, copy = cQ, cP, False
                        else:  # ringQP.Add (:243-244)
                            accQ = np.stack([rQ.binop("Add", accQ[t], cQ[t]) for t in range(2)])
                            accP = np.stack([rP.binop("Add", accP[t], cP[t]) for t in range(2)])
                    else:
                        state = True
                        if n & (n - 1):  # :255-262
                            opOut = np.stack([rQ.binop("Add", self.be.ModDownQPtoQNTT(levelQ, levelP, accQ[t], accP[t]),
                                                       ctInNTT[t]) for t in range(2)])
                        else:
                            opOut = ctInNTT.copy()
                if not state:  # :271-281
                    rot = self.GaloisElement((1 << i) * offset)
                    cQ = self.eval.AutomorphismHoisted(ctInNTT, dq, dp, rot, self.gks[rot])
                    ctInNTT = np.stack([rQ.binop("Add", ctInNTT[t], cQ[t]) for t in range(2)])
                i, j = i + 1, j >> 1
        if not isNTT:
            opOut = np.stack([rQ.INTT(opOut[k]) for k in range(2)])
        return opOut

    def InnerSum(self, ctIn, batchSize, n, isNTT=True):
        return self.PartialTracesSum(ctIn, batchSize, n, isNTT)

    def Replicate(self, ctIn, batchSize, n, isNTT=True):
        """core/rlwe/inner_sum.go:475"""
        return self.PartialTracesSum(ctIn, -batchSize, n, isNTT)


def BSGSIndex(nonZeroDiags, slots, N1):
    """circuits/common/lintrans/lintrans.go:344-366"""
    index, r1, r2 = {}, set(), set()
    for rot in nonZeroDiags:
        rot &= slots - 1
        a = ((rot // N1) * N1) & (slots - 1)
        b = rot & (N1 - 1)
        index.setdefault(a, []).append(b)
        r1.add(a)
        r2.add(b)
    for k in index:
        index[k].sort()
    return index, sorted(r1), sorted(r2)


class LinearTransformation:
    """Vec[k] = (Q [LevelQ+1][N], P [LevelP+1][N]) encoded diagonals (NTT, Montgomery)"""

    def __init__(self, Vec, LevelQ, LevelP, slots, N1=0):
        self.Vec, self.LevelQ, self.LevelP, self.slots, self.N1 = Vec, LevelQ, LevelP, slots, N1


def _margin(moduli, level):
    """core/rlwe/params.go:554-568"""
    return int(2.0 ** 64 / float(max(int(m) for m in moduli[: level + 1])))


class LinTransEvaluator:
    def __init__(self, evaluator: O.Evaluator, gks: dict):
        self.eval, self.gks = evaluator, gks
        self.ringQ, self.ringP = evaluator.ringQ, evaluator.ringP
        self.be = O.BasisExtender(self.ringQ, self.ringP)
        self.nth_root = 2 * self.ringQ.N

    def GaloisElement(self, k):
        return GaloisElement(self.nth_root, k)

    def EvaluateMany(self, ctIn, lts):
        """lintrans_evaluator.go:27-79"""
        ctIn = np.asarray(ctIn, dtype=np.uint64)
        levelP = lts[0].LevelP
        levelQ = min(max(lt.LevelQ for lt in lts), ctIn.shape[1] - 1)
        dq, dp = self.eval.DecomposeNTT(levelQ, levelP, levelP + 1, ctIn[1][: levelQ + 1], True)
        pre, outs = {}, []
        for lt in lts:
            if lt.N1 == 0:
                outs.append(self.MultiplyByDiagMatrix(ctIn, lt, dq, dp))
            else:
                _, _, rotN2 = BSGSIndex(list(lt.Vec.keys()), lt.slots, lt.N1)
                self.PreRotated(levelQ, levelP, ctIn, dq, dp, rotN2, pre)
                outs.append(self.MultiplyByDiagMatrixBSGS(ctIn, lt, pre))
        return outs

    def PreRotated(self, levelQ, levelP, ctIn, dq, dp, rots, pre):
        """lintrans_evaluator.go:82-110"""
        for i in list(pre):
            if i not in rots:
                del pre[i]
        for i in rots:
            if i != 0 and i not in pre:
                g = self.GaloisElement(i)
                pre[i] = self.eval.AutomorphismHoistedLazy(levelQ, ctIn[0][: levelQ + 1], dq, dp, g, self.gks[g])

    def MultiplyByDiagMatrix(self, ctIn, matrix, dq, dp, out_level=None):
        """lintrans_evaluator.go:142-274"""
        ctIn = np.asarray(ctIn, dtype=np.uint64)
        lv = ctIn.shape[1] - 1
        levelQ = min(lv if out_level is None else out_level, lv, matrix.LevelQ)
        levelP = matrix.LevelP
        rQ, rP = self.ringQ, self.ringP
        QiOverF, PiOverF = _margin(rQ.moduli, levelQ), _margin(rP.moduli, levelP)
        c0, c1 = ctIn[0][: levelQ + 1].copy(), ctIn[1][: levelQ + 1].copy()
        ct0TimesP = rQ.MulScalarBigint(c0, _prod(rP.moduli[: levelP + 1]))
        keys = sorted(matrix.Vec.keys())
        state = False
        if keys[0] == 0:
            state, keys = True, keys[1:]
        outQ = [None, None]
        outP = [None, None]
        for i, k in enumerate(keys):
            k &= matrix.slots - 1
            g = self.GaloisElement(k)
            evk = self.gks[g]
            index = rQ.AutomorphismNTTIndex(g)
            cQ, cP = self.eval.GadgetProductHoistedLazy(levelQ, dq, dp, evk)
            cQ[0] = rQ.binop("Add", cQ[0], ct0TimesP)
            ptQ, ptP = matrix.Vec[k]
            for t in range(2):
                tq = rQ.AutomorphismNTTWithIndex(cQ[t], index)
                tp = rP.AutomorphismNTTWithIndex(cP[t], index)
                if i == 0:
                    outQ[t] = rQ.binop("MulCoeffsMontgomery", ptQ[: levelQ + 1], tq)
                    outP[t] = rP.binop("MulCoeffsMontgomery", ptP[: levelP + 1], tp)
                else:
                    outQ[t] = rQ.binop("MulCoeffsMontgomeryThenAdd", ptQ[: levelQ + 1], tq, outQ[t])
                    outP[t] = rP.binop("MulCoeffsMontgomeryThenAdd", ptP[: levelP + 1], tp, outP[t])
            if i % QiOverF == QiOverF - 1:
                outQ = [rQ.unop("Reduce", x) for x in outQ]
            if i % PiOverF == PiOverF - 1:
                outP = [rP.unop("Reduce", x) for x in outP]
        if len(keys) % QiOverF == 0:
            outQ = [rQ.unop("Reduce", x) for x in outQ]
        if len(keys) % PiOverF == 0:
            outP = [rP.unop("Reduce", x) for x in outP]
        res = [self.be.ModDownQPtoQNTT(levelQ, levelP, outQ[t], outP[t]) for t in range(2)]
        if state:
            pt0 = matrix.Vec[0][0][: levelQ + 1]
            res[0] = rQ.binop("MulCoeffsMontgomeryThenAdd", pt0, c0, res[0])
            res[1] = rQ.binop("MulCoeffsMontgomeryThenAdd", pt0, c1, res[1])
        return np.stack(res)

    def MultiplyByDiagMatrixBSGS(self, ctIn, matrix, pre, out_level=None):
        """lintrans_evaluator.go:280-469"""
        ctIn = np.asarray(ctIn, dtype=np.uint64)
        lv = ctIn.shape[1] - 1
        levelQ = min(lv if out_level is None else out_level, lv, matrix.LevelQ)
        levelP = matrix.LevelP
        rQ, rP = self.ringQ, self.ringP
        QiOverF, PiOverF = _margin(rQ.moduli, levelQ) >> 1, _margin(rP.moduli, levelP) >> 1
        index, _, _ = BSGSIndex(list(matrix.Vec.keys()), matrix.slots, matrix.N1)
        Pm = _prod(rP.moduli[: levelP + 1])
        cin = [rQ.MulScalarBigint(ctIn[t][: levelQ + 1], Pm) for t in range(2)]  # P*c0, P*c1 (:332-333)
        N = rQ.N
        outQ, outP = [None, None], [None, None]
        cnt0 = 0
        for j in sorted(index.keys()):
            cnt1 = 0
            tQ, tP = [None, None], [None, None]
            for i in index[j]:
                ptQ, ptP = matrix.Vec[j + i]
                ptQ, ptP = ptQ[: levelQ + 1], ptP[: levelP + 1]
                for t in range(2):
                    if i == 0:
                        if cnt1 == 0:
                            tQ[t] = rQ.binop("MulCoeffsMontgomeryLazy", ptQ, cin[t])
                            tP[t] = np.zeros((levelP + 1, N), dtype=np.uint64)
                        else:
                            tQ[t] = rQ.binop("MulCoeffsMontgomeryLazyThenAddLazy", ptQ, cin[t], tQ[t])
                    else:
                        cQ, cP = pre[i]
                        if cnt1 == 0:
                            tQ[t] = rQ.binop("MulCoeffsMontgomeryLazy", ptQ, cQ[t])
                            tP[t] = rP.binop("MulCoeffsMontgomeryLazy", ptP, cP[t])
                        else:
                            tQ[t] = rQ.binop("MulCoeffsMontgomeryLazyThenAddLazy", ptQ, cQ[t], tQ[t])
                            tP[t] = rP.binop("MulCoeffsMontgomeryLazyThenAddLazy", ptP, cP[t], tP[t])
                if cnt1 % QiOverF == QiOverF - 1:
                    tQ = [rQ.unop("Reduce", x) for x in tQ]
                if cnt1 % PiOverF == PiOverF - 1:
                    tP = [rP.unop("Reduce", x) for x in tP]
                cnt1 += 1
            if cnt1 % QiOverF != 0:
                tQ = [rQ.unop("Reduce", x) for x in tQ]
            if cnt1 % PiOverF != 0:
                tP = [rP.unop("Reduce", x) for x in tP]
            if j != 0:
                t1 = self.be.ModDownQPtoQNTT(levelQ, levelP, tQ[1], tP[1])  # :397
                g = self.GaloisElement(j)
                rot = rQ.AutomorphismNTTIndex(g)
                cQ, cP = self.eval.GadgetProductLazy(levelQ, t1, self.gks[g])
                cQ[0] = rQ.binop("Add", cQ[0], tQ[0])
                cP[0] = rP.binop("Add", cP[0], tP[0])
                for t in range(2):
                    if cnt0 == 0:
                        outQ[t] = rQ.AutomorphismNTTWithIndex(cQ[t], rot)
                        outP[t] = rP.AutomorphismNTTWithIndex(cP[t], rot)
                    else:
                        outQ[t] = rQ.AutomorphismNTTWithIndexThenAddLazy(cQ[t], rot, outQ[t])
                        outP[t] = rP.AutomorphismNTTWithIndexThenAddLazy(cP[t], rot, outP[t])
            else:
                for t in range(2):
                    if cnt0 == 0:
                        outQ[t], outP[t] = tQ[t].copy(), tP[t].copy()
                    else:
                        outQ[t] = rQ.binop("AddLazy", outQ[t], tQ[t])
                        outP[t] = rP.binop("AddLazy", outP[t], tP[t])
            if cnt0 % QiOverF == QiOverF - 1:
                outQ = [rQ.unop("Reduce", x) for x in outQ]
            if cnt0 % PiOverF == PiOverF - 1:
                outP = [rP.unop("Reduce", x) for x in outP]
            cnt0 += 1
        if cnt0 % QiOverF != 0:
            outQ = [rQ.unop("Reduce", x) for x in outQ]
        if cnt0 % PiOverF != 0:
            outP = [rP.unop("Reduce", x) for x in outP]
        return np.stack([self.be.ModDownQPtoQNTT(levelQ, levelP, outQ[t], outP[t]) for t in range(2)])


class ScaleInvariantEvaluator:
    """BFV-style multiplication of the bgv package: Evaluator.tensorScaleInvariant + modUpAndNTT + tensorLowDeg +
    quantize (schemes/bgv/evaluator.go:898-1071), precomputation newEvaluatorPrecomp (:38-70)."""

    def __init__(self, evaluator: O.Evaluator, ringQMul: O.Ring, t: int):
        self.eval, self.t = evaluator, int(t)
        self.ringQ, self.ringQMul = evaluator.ringQ, ringQMul
        self.be = O.BasisExtender(self.ringQ, ringQMul)  # basisExtenderQ1toQ2 (:58)
        logN = self.ringQ.N.bit_length() - 1
        self.levelQMul = []
        Q = 1
        for m in self.ringQ.moduli:  # :43-48
            Q *= int(m)
            self.levelQMul.append(-(-(Q.bit_length() + logN) // 61) - 1)

    def _mod_up_and_ntt(self, level, lm, ct):
        """:991-1001"""
        out = []
        for c in ct:
            buff = self.ringQ.INTT(c[: level + 1])
            out.append(self.ringQMul.NTTLazy(self.be.ModUpQtoP(level, lm, buff)))
        return out

    def _quantize(self, level, lm, c2Q1, c2Q2):
        """:1050-1071"""
        c2Q1 = self.ringQ.INTTLazy(c2Q1)
        c2Q2 = self.ringQMul.INTTLazy(c2Q2)
        c2Q2 = self.be.ModDownQPtoP(level, lm, c2Q1, c2Q2)       # QP / Q -> P
        c2Q1 = self.be.ModUpPtoQ(lm, level, c2Q2)                # centred, back to Q
        c2Q1 = self.ringQ.scalarop("MulScalar", c2Q1, self.t)    # (ct/Q) * T
        return self.ringQ.NTT(c2Q1)

    def MulRelinScaleInvariant(self, ct0, ct1, rlk=None, square=False):
        """tensorScaleInvariant (:898-972); ct0, ct1 degree-1 NTT ciphertexts at the same level; `square` is the
        reference's ct0 == ct1 (same object) branch."""
        ct0 = np.asarray(ct0, dtype=np.uint64)
        ct1 = ct0 if square else np.asarray(ct1, dtype=np.uint64)
        level = ct0.shape[1] - 1
        lm = self.levelQMul[level]
        rQ, rM = self.ringQ, self.ringQMul
        t0M = self._mod_up_and_ntt(level, lm, ct0)
        t1M = t0M if square else self._mod_up_and_ntt(level, lm, ct1)
        # tensorLowDeg (:1003-1048)
        c00, c01 = rQ.unop("MForm", ct0[0]), rQ.unop("MForm", ct0[1])
        c00M, c01M = rM.unop("MForm", t0M[0]), rM.unop("MForm", t0M[1])
        if square:
            q = [rQ.binop("MulCoeffsMontgomery", c00, ct0[0]), rQ.binop("MulCoeffsMontgomery", c00, ct0[1]),
                 rQ.binop("MulCoeffsMontgomery", c01, ct0[1])]
            q[1] = rQ.binop("AddLazy", q[1], q[1])
            m = [rM.binop("MulCoeffsMontgomery", c00M, t0M[0]), rM.binop("MulCoeffsMontgomery", c00M, t0M[1]),
                 rM.binop("MulCoeffsMontgomery", c01M, t0M[1])]
            m[1] = rM.binop("AddLazy", m[1], m[1])
        else:
            q = [rQ.binop("MulCoeffsMontgomery", c00, ct1[0]), rQ.binop("MulCoeffsMontgomery", c00, ct1[1]),
                 rQ.binop("MulCoeffsMontgomery", c01, ct1[1])]
            q[1] = rQ.binop("MulCoeffsMontgomeryThenAddLazy", c01, ct1[0], q[1])
            m = [rM.binop("MulCoeffsMontgomery", c00M, t1M[0]), rM.binop("MulCoeffsMontgomery", c00M, t1M[1]),
                 rM.binop("MulCoeffsMontgomery", c01M, t1M[1])]
            m[1] = rM.binop("MulCoeffsMontgomeryThenAddLazy", c01M, t1M[0], m[1])
        out = [self._quantize(level, lm, q[k], m[k]) for k in range(3)]
        if rlk is None:
            return np.stack(out)
        tmp = self.eval.GadgetProduct(level, out[2], rlk)  # :966-969
        return np.stack([rQ.binop("Add", out[0], tmp[0]), rQ.binop("Add", out[1], tmp[1])])


def ApplyEvaluationKey(ev: O.Evaluator, ct, evk):
    """core/rlwe/evaluator_evaluationkey.go:36,98-106 (same ring degree)"""
    ct = np.asarray(ct, dtype=np.uint64)
    level = ct.shape[1] - 1
    tmp = ev.GadgetProduct(level, ct[1], evk)
    return np.stack([ev.ringQ.binop("Add", ct[0], tmp[0]), tmp[1]])


def _centered_lift(c0, q, moduli, strict):
    """circuits/ckks/bootstrapping/evaluator.go:654-667 / :677-696: rows for `moduli` from the limb-0 coefficients"""
    c0 = np.asarray(c0, dtype=np.uint64)
    half = np.uint64(q >> 1)
    neg = (c0 > half) if strict else (c0 >= half)
    coeff = np.where(neg, np.uint64(q) - c0, c0)
    out = np.empty((len(moduli), c0.shape[0]), dtype=np.uint64)
    for i, m in enumerate(moduli):
        tmp = coeff % np.uint64(int(m))  # ring.BRedAdd: the canonical residue
        out[i] = np.where(neg, np.uint64(int(m)) - tmp, tmp)
    return out


def BootstrappingModUp(ev: O.Evaluator, ise: InnerSumEvaluator, ct, scale, logSlots, EvkDenseToSparse=None,
                       EvkSparseToDense=None):
    """bootstrapping.Evaluator.ModUp (circuits/ckks/bootstrapping/evaluator.go:612-769); ct: [2][levelIn+1][N] NTT."""
    ct = np.asarray(ct, dtype=np.uint64)
    rQ, rP = ev.ringQ, ev.ringP
    Q, P = rQ.moduli, rP.moduli
    levelQ, levelP = len(Q) - 1, len(P) - 1
    if EvkDenseToSparse is not None:
        ct = ApplyEvaluationKey(ev, ct, EvkDenseToSparse)
    c = [rQ.INTT(ct[k]) for k in range(2)]
    q = int(Q[0])
    c0 = np.concatenate([c[0][:1], _centered_lift(c[0][0], q, Q[1:], False)])  # :654-667
    scalar = int(round(scale)) if scale > 1 else None
    if EvkSparseToDense is not None:
        dQ = rQ.NTT(_centered_lift(c[1][0], q, Q, True))  # :677-696, :699-705
        dP = rP.NTT(_centered_lift(c[1][0], q, P, True))
        c0 = rQ.NTT(c0)
        if scalar is not None:  # :711-723
            dQ, dP = rQ.scalarop("MulScalar", dQ, scalar), rP.scalarop("MulScalar", dP, scalar)
            c0 = rQ.scalarop("MulScalar", c0, scalar)
        beta = O.BaseRNSDecompositionVectorSize(levelQ, levelP)
        tmp = ev.GadgetProductHoisted(levelQ, np.stack([dQ] * beta), np.stack([dP] * beta), EvkSparseToDense)  # :733
        out = np.stack([rQ.binop("Add", c0, tmp[0]), tmp[1]])
    else:
        c1 = np.concatenate([c[1][:1], _centered_lift(c[1][0], q, Q[1:], False)])  # :742-755
        out = np.stack([rQ.NTT(c0), rQ.NTT(c1)])
        if scalar is not None:
            out = np.stack([rQ.scalarop("MulScalar", out[k], scalar) for k in range(2)])
    return ise.Trace(out, logSlots)  # :768


# ---------------------------------------------------------------------------------------------------------------
# ring-level call sites of schemes.Evaluator for CKKS / BGV (ct x ct and ct x pt), restated op for op
# ---------------------------------------------------------------------------------------------------------------
def _ct_ct_then_add(ev: O.Evaluator, c00, c01, op1, rlk, opOut):
    """schemes/ckks/evaluator.go:1131-1155 == schemes/bgv/evaluator.go:1288-1314"""
    rQ = ev.ringQ
    level = op1.shape[1] - 1
    out = [np.asarray(x, dtype=np.uint64).copy() for x in opOut]
    out[0] = rQ.binop("MulCoeffsMontgomeryThenAdd", c00, op1[0], out[0])
    out[1] = rQ.binop("MulCoeffsMontgomeryThenAdd", c00, op1[1], out[1])
    out[1] = rQ.binop("MulCoeffsMontgomeryThenAdd", c01, op1[0], out[1])
    if rlk is not None:
        c2 = rQ.binop("MulCoeffsMontgomery", c01, op1[1])
        tmp = ev.GadgetProduct(level, c2, rlk)
        out[0], out[1] = rQ.binop("Add", out[0], tmp[0]), rQ.binop("Add", out[1], tmp[1])
    else:
        out[2] = rQ.binop("MulCoeffsMontgomeryThenAdd", c01, op1[1], out[2])
    return np.stack(out)


def ckks_mul_relin_then_add(ev, op0, op1, rlk, opOut):
    """schemes/ckks/evaluator.go:1081-1155 (ct x ct, scales already matched)"""
    op0, op1 = np.asarray(op0, dtype=np.uint64), np.asarray(op1, dtype=np.uint64)
    return _ct_ct_then_add(ev, ev.ringQ.unop("MForm", op0[0]), ev.ringQ.unop("MForm", op0[1]), op1, rlk, opOut)


def ckks_mul_plaintext(ev, op0, pt, opOut=None):
    """schemes/ckks/evaluator.go:842-870 (opOut None) and :1158-1170 (MulThenAdd)"""
    rQ = ev.ringQ
    c0 = rQ.unop("MForm", np.asarray(pt, dtype=np.uint64))
    if opOut is None:
        return np.stack([rQ.binop("MulCoeffsMontgomery", c0, a) for a in op0])
    return np.stack([rQ.binop("MulCoeffsMontgomeryThenAdd", a, c0, o) for a, o in zip(op0, opOut)])


def bgv_t_montgomery(ringQ, t):
    """schemes/bgv/evaluator.go:60-62"""
    return np.array([O.MForm((int(t) << 64) % int(q), int(q)) for q in ringQ.moduli], dtype=np.uint64)


def bgv_match_scales_binary(scale0, scale1, t):
    """schemes/bgv/evaluator.go:1569-1615"""
    from math import gcd
    assert gcd(scale0, t) == 1
    thalf = t >> 1
    center = lambda x: t - x if x >= thalf else x
    a, b = t, 0
    A, B = O.BRed(O.ModExp(scale0, t - 2, t), scale1, t), 1
    r0, r1, e = A, B, center(A) + 1
    while A != 0:
        qq = a // A
        a, A = A, a % A
        b, B = B, (t + b - O.BRed(B, qq, t)) % t
        if A != 0 and gcd(A, t) == 1:
            tmp = center(A) + center(B)
            if tmp < e:
                e, r0, r1 = tmp, A, B
    return r0, r1, e


def bgv_mul_plaintext(ev, t, op0, pt):
    """schemes/bgv/evaluator.go:669-683"""
    rQ = ev.ringQ
    level = np.asarray(pt).shape[0] - 1
    c00 = rQ.MulRNSScalarMontgomery(np.asarray(pt, dtype=np.uint64), bgv_t_montgomery(rQ, t)[: level + 1])
    return np.stack([rQ.binop("MulCoeffsMontgomery", a, c00) for a in op0])


def bgv_mul_relin_then_add(ev, t, op0, op1, rlk, opOut, scales=(1, 1, 1)):
    """schemes/bgv/evaluator.go:1230-1314 (ct x ct); returns (opOut, new opOut scale)"""
    rQ = ev.ringQ
    op0, op1 = np.asarray(op0, dtype=np.uint64), np.asarray(op1, dtype=np.uint64)
    level = op0.shape[1] - 1
    out = [np.asarray(x, dtype=np.uint64).copy() for x in opOut]
    s0, s1, so = (int(x) % t for x in scales)
    r0, target = 1, O.BRed(s0, s1, t)
    if so != target:
        r0, r1, _ = bgv_match_scales_binary(target, so, t)
        out = [rQ.scalarop("MulScalar", o, r1) for o in out]
        so = so * r1 % t
    tm = bgv_t_montgomery(rQ, t)[: level + 1]
    c00, c01 = rQ.MulRNSScalarMontgomery(op0[0], tm), rQ.MulRNSScalarMontgomery(op0[1], tm)
    if r0 != 1:
        c00, c01 = rQ.scalarop("MulScalar", c00, r0), rQ.scalarop("MulScalar", c01, r0)
    return _ct_ct_then_add(ev, c00, c01, op1, rlk, out), so


def ExtendBasisSmallNormAndCenter(ringQ, ringP, polyInQ, levelP):
    """ringqp.Ring.ExtendBasisSmallNormAndCenter (ring/ringqp/operations.go:325-349), 64-bit wrapping arithmetic"""
    c = np.asarray(polyInQ, dtype=np.uint64)[0]
    Q = np.uint64(ringQ.moduli[0])
    neg = c > (Q >> np.uint64(1))
    coeff = np.where(neg, Q - c, c)
    out = np.empty((levelP + 1, c.shape[0]), dtype=np.uint64)
    for i, pi in enumerate(ringP.moduli[: levelP + 1]):
        out[i] = np.where(neg, np.uint64(pi) - coeff, coeff)
    return out


# ---------------------------------------------------------------------------------------------------------------
# bgv.Evaluator at the rlwe.Ciphertext level (numpy arrays), the backend the polynomial evaluator is checked against
# ---------------------------------------------------------------------------------------------------------------
class Ct:
    """Value: list of [limbs][N] arrays (limbs = level + 1); Scale in Z_t (BGV) or an exact Fraction (CKKS)"""

    def __init__(self, value, scale=1):
        from fractions import Fraction
        # BGV scales are residues mod t (ints); CKKS scales are exact rationals and must survive CopyNew unrounded
        self.Value = [np.asarray(v, dtype=np.uint64) for v in value]
        self.Scale = scale if isinstance(scale, Fraction) else int(scale)

    @property
    def level(self):
        return self.Value[0].shape[0] - 1

    def Degree(self):
        return len(self.Value) - 1

    def Level(self):
        return self.level


class BGVCtEvaluator:
    """schemes/bgv/evaluator.go:122-260, 384-470, 500-560, 592-685, 1056-1140, 1363-1393 on numpy ciphertexts"""

    def __init__(self, ev: O.Evaluator, t: int, rlk=None):
        self.ev, self.t, self.rlk, self.ringQ = ev, int(t), rlk, ev.ringQ
        self.Q = [int(q) for q in ev.ringQ.moduli]
        self.tInvModQ, Qi = [], 1
        for q in self.Q:
            Qi *= q
            self.tInvModQ.append(pow(self.t, -1, Qi))

    def NewCiphertext(self, degree, level, batch=1):
        return Ct([np.zeros((level + 1, self.ringQ.N), dtype=np.uint64) for _ in range(degree + 1)], 1)

    def CopyNew(self, ct):
        return Ct([v.copy() for v in ct.Value], ct.Scale)

    def _set(self, ct, values, level):
        ct.Value = [np.asarray(v, dtype=np.uint64)[: level + 1].copy() for v in values]

    def _centered(self, x):
        x %= self.t
        return x - self.t if x > (self.t >> 1) else x

    def Add(self, op0, op1, opOut):
        rQ = self.ringQ
        if isinstance(op1, Ct):
            level = min(op0.level, op1.level, opOut.level)
            cut = lambda v: v[: level + 1]
            if op0.Scale == op1.Scale:
                small, large = (op0, op1) if op0.Degree() <= op1.Degree() else (op1, op0)
                vals = [rQ.binop("Add", cut(op0.Value[i]), cut(op1.Value[i])) for i in range(small.Degree() + 1)]
                vals += [cut(large.Value[i]) for i in range(small.Degree() + 1, large.Degree() + 1)]
                scale = max(op0.Scale, op1.Scale)
            else:
                r0, r1, _ = bgv_match_scales_binary(op0.Scale, op1.Scale, self.t)
                deg = max(op0.Degree(), op1.Degree())
                vals = [rQ.scalarop("MulScalar", cut(op0.Value[i]), r0) if i <= op0.Degree()
                        else np.zeros((level + 1, rQ.N), dtype=np.uint64) for i in range(deg + 1)]
                for i in range(op1.Degree() + 1):
                    vals[i] = rQ.scalarop("MulScalarThenAdd", cut(op1.Value[i]), r1, vals[i])
                scale = op0.Scale * r0 % self.t
            self._set(opOut, vals, level)
            opOut.Scale = scale
            return
        level = min(op0.level, opOut.level)
        v = self._centered(int(op1) * op0.Scale) * self.tInvModQ[level]
        vals = [rQ.AddScalarBigint(op0.Value[0][: level + 1], v)] + [x[: level + 1] for x in op0.Value[1:]]
        scale = op0.Scale
        self._set(opOut, vals, level)
        opOut.Scale = scale

    def _tensor(self, op0, op1, relin, opOut):
        level = min(op0.level, op1.level, opOut.level)
        a = np.stack([v[: level + 1] for v in op0.Value])
        b = np.stack([v[: level + 1] for v in op1.Value])
        out = self.ev.BGVMulRelin(self.t, a, b, self.rlk if relin else None, relin)
        self._set(opOut, list(out), level)
        opOut.Scale = op0.Scale * op1.Scale % self.t

    def Mul(self, op0, op1, opOut):
        if isinstance(op1, Ct):
            return self._tensor(op0, op1, False, opOut)
        level = min(op0.level, opOut.level)
        v = self._centered(int(op1))
        vals = [self.ringQ.MulScalarBigint(x[: level + 1], v) for x in op0.Value]
        scale = op0.Scale
        self._set(opOut, vals, level)
        opOut.Scale = scale

    def MulRelin(self, op0, op1, opOut):
        if isinstance(op1, Ct):
            return self._tensor(op0, op1, True, opOut)
        self.Mul(op0, op1, opOut)

    def MulNew(self, op0, op1):
        lv = min(op0.level, op1.level) if isinstance(op1, Ct) else op0.level
        out = self.NewCiphertext(1, lv)
        self.Mul(op0, op1, out)
        return out

    def MulRelinNew(self, op0, op1):
        lv = min(op0.level, op1.level) if isinstance(op1, Ct) else op0.level
        out = self.NewCiphertext(1, lv)
        self.MulRelin(op0, op1, out)
        return out

    def MulThenAdd(self, op0, op1, opOut):
        level = min(op0.level, opOut.level)
        v = int(op1)
        if op0.Scale != opOut.Scale:
            v *= O.BRed(O.ModExp(op0.Scale, self.t - 2, self.t), opOut.Scale, self.t)
        v = self._centered(v)
        vals = list(opOut.Value[: op0.Degree() + 1])  # Resize(op0.Degree(), ...)
        while len(vals) < op0.Degree() + 1:
            vals.append(np.zeros((opOut.level + 1, self.ringQ.N), dtype=np.uint64))
        for i in range(op0.Degree() + 1):
            head = self.ringQ.MulScalarBigintThenAdd(op0.Value[i][: level + 1], v, vals[i][: level + 1])
            vals[i] = np.concatenate([head, vals[i][level + 1:]])
        opOut.Value = vals

    def Relinearize(self, op0, opOut):
        level = min(op0.level, opOut.level)
        out = self.ev.Relinearize(np.stack([v[: level + 1] for v in op0.Value]), self.rlk)
        scale = op0.Scale
        self._set(opOut, list(out), level)
        opOut.Scale = scale

    def Rescale(self, op0, opOut):
        if op0.level == 0:
            raise ValueError("cannot rescale: op0 already at level 0")
        level = op0.level
        vals = [self.ringQ.DivRoundByLastModulusNTT(v) for v in op0.Value]
        scale = op0.Scale * pow(self.Q[level], -1, self.t) % self.t
        self._set(opOut, vals, level - 1)
        opOut.Scale = scale


# ---------------------------------------------------------------------------------------------------------------
# ckks.Evaluator at the rlwe.Ciphertext level (numpy arrays); scales are exact rationals, constants are encoded as
# bigComplexToRNSScalar encodes them (schemes/ckks/scaling.go:10-43): with the reference's big.Float roundings restated on
# integers (_keep / _const_to_int below; written apart from the product driver's tests/drivers/schemes.py)
# ---------------------------------------------------------------------------------------------------------------
def _keep(num: int, den: int, bits: int):
    """(m, e): the nearest-even `bits`-bit approximation m 2^e of num / den (num, den > 0) -- a big.Float of that precision"""
    e = num.bit_length() - den.bit_length() - bits  # num / den in (2^(e+bits-1), 2^(e+bits+1))
    for _ in range(3):
        n, d = (num, den << e) if e >= 0 else (num << -e, den)
        m, r = divmod(n, d)
        if m >> bits:
            e += 1
            continue
        if not m >> (bits - 1):
            e -= 1
            continue
        if 2 * r > d or (2 * r == d and m & 1):
            m += 1  # (a carry to bits + 1 bits is still m 2^e exactly)
        return m, e
    raise AssertionError("unreachable")


def _const_to_int(c, scale, enc_prec=53):
    """scaling.go:16-26: big.Float(c at enc_prec bits) * big.Float(scale at 128 bits) -> 128 bits; +-0.5 -> 128 bits; Int()"""
    from fractions import Fraction
    c, scale = Fraction(c), Fraction(scale)
    if c == 0:
        return 0
    sign = 1 if c > 0 else -1
    cm, ce = _keep(abs(c.numerator), c.denominator, enc_prec)
    if cm == 0:
        return 0
    sm, se = _keep(scale.numerator, scale.denominator, 128)
    bits = max(enc_prec, 128)
    pm, pe = _keep(cm * sm, 1, bits)
    pe += ce + se                                   # |c| * scale ~ pm 2^pe
    # + 0.5 in the same precision: align, add, keep
    if pe >= 0:
        num, den = (pm << pe) * 2 + 1, 2
    else:
        num, den = pm * 2 + (1 << -pe), 2 << -pe
    rm, re_ = _keep(num, den, bits)
    mag = rm << re_ if re_ >= 0 else rm >> -re_     # toward zero
    return sign * mag


def _ckks_is_int(c, enc_prec=53):
    from fractions import Fraction
    for part in _cfrac(c):
        part = Fraction(part)
        if part == 0:
            continue
        m, e = _keep(abs(part.numerator), part.denominator, enc_prec)
        if e < 0 and m & ((1 << -e) - 1):
            return False
    return True



def _rha(x):
    from fractions import Fraction
    x = Fraction(x)
    if x > 0:
        return (x + Fraction(1, 2)).__floor__()
    if x < 0:
        return -((-x + Fraction(1, 2)).__floor__())
    return 0


def _cfrac(c):
    from fractions import Fraction
    if isinstance(c, tuple):
        return Fraction(c[0]), Fraction(c[1])
    if isinstance(c, complex):
        return Fraction(c.real), Fraction(c.imag)
    return Fraction(c), Fraction(0)


class CKKSCtEvaluator:
    """schemes/ckks/evaluator.go:42-135, 221-424, 477-515, 570-760, 875-940 on numpy ciphertexts"""

    def __init__(self, ev: O.Evaluator, rlk=None, default_scale=None):
        self.ev, self.rlk, self.ringQ, self.t = ev, rlk, ev.ringQ, None
        self.Q = [int(q) for q in ev.ringQ.moduli]
        # Parameters.EncodingPrecision (schemes/ckks/params.go:185-195): log2scale := math.Log2(DefaultScale().Float64());
        # 53 when log2scale <= 53, else uint(log2scale)
        self.EncodingPrecision = 53
        if default_scale is not None:
            import math
            l2 = math.log2(float(default_scale))
            self.EncodingPrecision = 53 if l2 <= 53 else int(l2)

    NewCiphertext = BGVCtEvaluator.NewCiphertext
    CopyNew = BGVCtEvaluator.CopyNew
    _set = BGVCtEvaluator._set
    MulNew = BGVCtEvaluator.MulNew
    MulRelinNew = BGVCtEvaluator.MulRelinNew
    Relinearize = BGVCtEvaluator.Relinearize

    def _rns(self, level, scale, c):
        re, im = _cfrac(c)
        real, imag = _const_to_int(re, scale, self.EncodingPrecision), _const_to_int(im, scale, self.EncodingPrecision)
        s0, s1 = [], []
        for i, q in enumerate(self.Q[: level + 1]):
            root = int(self.ringQ.roots_forward(i)[1])  # Montgomery form of sqrt(-1) mod q_i (evaluator.go:417)
            r, m = real % q, O.MRed(imag % q, root, q)
            s0.append((r + m) % q)
            s1.append((r + q - m) % q)
        return np.array(s0, dtype=np.uint64), np.array(s1, dtype=np.uint64)

    def _is_int(self, c):
        return _ckks_is_int(c, self.EncodingPrecision)

    def _addsub_ct(self, op0, op1, opOut, sub):
        from fractions import Fraction
        rQ = self.ringQ
        level = min(op0.level, op1.level, opOut.level)
        a, b, scale = op0, op1, op0.Scale
        if op0.Scale != op1.Scale:
            if op0.Scale > op1.Scale:
                ratio = int(Fraction(op0.Scale) / Fraction(op1.Scale))
                if ratio > 0:
                    b = self.NewCiphertext(op1.Degree(), level)
                    self.Mul(op1, ratio, b)
            else:
                ratio = int(Fraction(op1.Scale) / Fraction(op0.Scale))
                if ratio > 0:
                    a = self.NewCiphertext(op0.Degree(), level)
                    self.Mul(op0, ratio, a)
                    scale = op1.Scale
        cut = lambda v: v[: level + 1]
        lo = min(a.Degree(), b.Degree())
        vals = [rQ.binop("Sub" if sub else "Add", cut(a.Value[i]), cut(b.Value[i])) for i in range(lo + 1)]
        vals += [cut(a.Value[i]) for i in range(lo + 1, a.Degree() + 1)]
        vals += [rQ.unop("Neg", cut(b.Value[i])) if sub else cut(b.Value[i]) for i in range(lo + 1, b.Degree() + 1)]
        self._set(opOut, vals, level)
        opOut.Scale = scale

    def Add(self, op0, op1, opOut):
        if isinstance(op1, Ct):
            return self._addsub_ct(op0, op1, opOut, False)
        level = min(op0.level, opOut.level)
        s0, s1 = self._rns(level, op0.Scale, op1)
        vals = [self.ringQ.AddDoubleRNSScalar(op0.Value[0][: level + 1], s0, s1)] + [v[: level + 1] for v in op0.Value[1:]]
        scale = op0.Scale
        self._set(opOut, vals, level)
        opOut.Scale = scale

    def Sub(self, op0, op1, opOut):
        if isinstance(op1, Ct):
            return self._addsub_ct(op0, op1, opOut, True)
        re, im = _cfrac(op1)
        self.Add(op0, (-re, -im), opOut)

    def _tensor(self, op0, op1, relin, opOut):
        level = min(op0.level, op1.level, opOut.level)
        if op0.Degree() == 0 or op1.Degree() == 0:  # plaintext (x) ciphertext (schemes/ckks/evaluator.go:842-870)
            pt, ct = (op0, op1) if op0.Degree() == 0 else (op1, op0)
            c0 = self.ringQ.unop("MForm", pt.Value[0][: level + 1])
            vals = [self.ringQ.binop("MulCoeffsMontgomery", c0, v[: level + 1]) for v in ct.Value]
            scale = op0.Scale * op1.Scale
            self._set(opOut, vals, level)
            opOut.Scale = scale
            return
        a = np.stack([v[: level + 1] for v in op0.Value])
        b = np.stack([v[: level + 1] for v in op1.Value])
        out = self.ev.CKKSMulRelin(a, b, self.rlk if relin else None, relin)
        scale = op0.Scale * op1.Scale
        self._set(opOut, list(out), level)
        opOut.Scale = scale

    def Mul(self, op0, op1, opOut):
        if isinstance(op1, Ct):
            return self._tensor(op0, op1, False, opOut)
        level = min(op0.level, opOut.level)
        scale = 1 if self._is_int(op1) else self.Q[level]
        s0, s1 = self._rns(level, scale, op1)
        vals = [self.ringQ.MulDoubleRNSScalar(v[: level + 1], s0, s1) for v in op0.Value]
        new_scale = op0.Scale * scale
        self._set(opOut, vals, level)
        opOut.Scale = new_scale

    def MulRelin(self, op0, op1, opOut):
        if isinstance(op1, Ct):
            return self._tensor(op0, op1, True, opOut)
        self.Mul(op0, op1, opOut)

    def MulThenAdd(self, op0, op1, opOut):
        from fractions import Fraction
        level = min(op0.level, opOut.level)
        vals = list(opOut.Value[: op0.Degree() + 1])
        while len(vals) < op0.Degree() + 1:
            vals.append(np.zeros((opOut.level + 1, self.ringQ.N), dtype=np.uint64))
        opOut.Value = vals
        if op0.Scale == opOut.Scale:
            if self._is_int(op1):
                scale = 1
            else:
                scale = self.Q[level]
                self.Mul(opOut, scale, opOut)  # at opOut's own level (:912-916)
                opOut.Scale = opOut.Scale * scale
        elif op0.Scale < opOut.Scale:
            scale = Fraction(opOut.Scale) / Fraction(op0.Scale)
        else:
            raise ValueError("cannot MulThenAdd: op0.Scale > opOut.Scale is not supported")
        s0, s1 = self._rns(level, scale, op1)
        for i in range(op0.Degree() + 1):
            head = self.ringQ.MulDoubleRNSScalarThenAdd(op0.Value[i][: level + 1], s0, s1, opOut.Value[i][: level + 1])
            opOut.Value[i] = np.concatenate([head, opOut.Value[i][level + 1:]])

    def Rescale(self, op0, opOut):
        from fractions import Fraction
        if op0.level <= 0:
            raise ValueError("cannot Rescale: input Ciphertext level is too low")
        level = op0.level
        vals = [self.ringQ.DivRoundByLastModulusManyNTT(1, v) for v in op0.Value]
        scale = Fraction(op0.Scale) / self.Q[level]
        self._set(opOut, vals, level - 1)
        opOut.Scale = scale


class OracleBootstrapBackend:
    """adapters for drivers.bootstrapping.Bootstrapper over the oracle (test infrastructure)"""

    def __init__(self, ckks: CKKSCtEvaluator, lte: LinTransEvaluator, ise: InnerSumEvaluator, EvkDenseToSparse=None,
                 EvkSparseToDense=None):
        self.ckks, self.lte, self.ise, self.d2s, self.s2d = ckks, lte, ise, EvkDenseToSparse, EvkSparseToDense

    def modup(self, ct, scale, logSlots):
        out = BootstrappingModUp(self.ckks.ev, self.ise, np.stack(ct.Value), scale, logSlots, self.d2s, self.s2d)
        return Ct(list(out), ct.Scale * (int(round(scale)) if scale > 1 else 1))

    def lintrans(self, ct, matrix, matrix_scale):
        (out,) = self.lte.EvaluateMany(np.stack(ct.Value), [matrix])
        return Ct(list(out), ct.Scale * matrix_scale)

    def conjugate(self, ct):
        g = 2 * self.ckks.ringQ.N - 1
        return Ct(list(self.ckks.ev.Automorphism(np.stack(ct.Value), g, self.lte.gks[g])), ct.Scale)

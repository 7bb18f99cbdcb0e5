"""Host-side mirror of circuits/common/lintrans (homomorphic plaintext-matrix x ciphertext-vector products by
diagonals), device-resident: every polynomial stays in HBM and every step is one of the operators of
``rlwe.EvaluatorProvider`` (core/rlwe/rlwe.go:10-18) or a ringqp coefficient-wise kernel, exactly as the reference's own
implementation (circuits/common/lintrans/lintrans_evaluator.go) is written.  Encoding the diagonals (a scheme encoder
job, circuits/common/lintrans/lintrans.go:205) stays on the host: a ``LinearTransformation`` here holds already-encoded
diagonals as QP polynomials in the NTT + Montgomery domain."""
from __future__ import annotations

import os

import ctypes as C

from lattigo_amd._lib import H, check, load
from lattigo_amd.ring import BasisExtender, Poly
from lattigo_amd.rlwe import Decomposition, Evaluator, GaloisElement, GaloisKeySet

MAX_TERMS = 64  # terms per he_lintrans_mul_sum call


def BSGSIndex(nonZeroDiags, slots: int, N1: int):
    """circuits/common/lintrans/lintrans.go:344"""
    index, rotN1, rotN2 = {}, set(), set()
    for rot in nonZeroDiags:
        rot &= slots - 1
        idxN1 = ((rot // N1) * N1) & (slots - 1)
        idxN2 = rot & (N1 - 1)
        index.setdefault(idxN1, []).append(idxN2)
        rotN1.add(idxN1)
        rotN2.add(idxN2)
    for k in index:
        index[k].sort()
    return index, sorted(rotN1), sorted(rotN2)


def FindBestBSGSRatio(nonZeroDiags, maxN: int, logMaxRatio: int) -> int:
    """circuits/common/lintrans/lintrans.go:321"""
    maxRatio = float(1 << logMaxRatio)
    N1 = 1
    while N1 < maxN:
        _, rotN1, rotN2 = BSGSIndex(nonZeroDiags, maxN, N1)
        nbN1, nbN2 = len(rotN1) - 1, len(rotN2) - 1
        ratio = float("inf") if nbN1 == 0 and nbN2 > 0 else (float("nan") if nbN1 == 0 else nbN2 / nbN1)
        if ratio == maxRatio:
            return N1
        if ratio > maxRatio:
            return N1 // 2
        N1 <<= 1
    return 1


def GaloisElements(nth_root: int, diags, slots: int, logBSGSRatio: int):
    """circuits/common/lintrans/lintrans.go:297"""
    if logBSGSRatio < 0:
        _, _, rotN2 = BSGSIndex(diags, slots, slots)
        return [GaloisElement(nth_root, r) for r in rotN2]
    N1 = FindBestBSGSRatio(diags, slots, logBSGSRatio)
    _, rotN1, rotN2 = BSGSIndex(diags, slots, N1)
    return [GaloisElement(nth_root, r) for r in sorted(set(rotN1) | set(rotN2))]


class LinearTransformation:
    """lintrans.LinearTransformation (circuits/common/lintrans/lintrans.go:117): Vec[k] = (Q, P) device polynomials of
    the k-th non-zero diagonal (NTT + Montgomery), LevelQ/LevelP, N1 (0 = naive evaluation), slots = 2^LogDimensions.Cols."""

    def __init__(self, Vec: dict, LevelQ: int, LevelP: int, slots: int, N1: int = 0):
        self.Vec, self.LevelQ, self.LevelP, self.slots, self.N1 = Vec, LevelQ, LevelP, slots, N1

    def BSGSIndex(self):
        return BSGSIndex(list(self.Vec.keys()), self.slots, self.N1)


def _overflow_margin(moduli, level: int) -> int:
    """Parameters.QiOverflowMargin / PiOverflowMargin (core/rlwe/params.go:554,563)"""
    return int(2.0 ** 64 / float(max(int(m) for m in moduli[: level + 1])))


class LinTransEvaluator:
    """lintrans.Evaluator (circuits/common/lintrans/lintrans_evaluator.go:12)"""

    def __init__(self, evaluator: Evaluator, gks: GaloisKeySet):
        self.eval, self.gks = evaluator, gks
        self.ringQ, self.ringP = evaluator.ringQ, evaluator.ringP
        self.be = BasisExtender(self.ringQ, self.ringP)
        self.nth_root = self.ringQ.NthRoot()
        self._index = {}
        # the giant step of the BSGS product through he_lintrans_giant_step (one call) or through the reference's own sequence of
        # calls (HERING_DRIVER_NO_GIANT=1, and the parity test of the two)
        self.fuse_giant = os.environ.get("HERING_DRIVER_NO_GIANT", "0") in ("", "0")

    def GaloisElement(self, k: int) -> int:
        return GaloisElement(self.nth_root, k)

    def AutomorphismIndex(self, galEl: int):
        if galEl not in self._index:
            self._index[galEl] = self.ringQ.AutomorphismNTTIndex(galEl)
        return self._index[galEl]

    def _qp(self, levelQ, levelP, B):
        # scratch: every user below has the next operation overwrite both parts in full
        return (Poly(self.ringQ, levelQ + 1, B, zero=False), Poly(self.ringP, levelP + 1, B, zero=False))

    def _mul_sum(self, levelQ, levelP, terms, out0, out1, accumulate=False):
        """out_k = Reduce([out_k +] sum_i MulCoeffsMontgomeryLazy(pt_i, phi_i(ct_i[k]))) on QP in one pass per ring
        (he_lintrans_mul_sum): the canonical value of the reference's per-diagonal MulCoeffsMontgomeryLazy[ThenAddLazy] +
        Reduce chain.  terms = [(ptQP, ct0QP, ct1QP, index or None)]; a ciphertext P part of None means "no P part"."""
        for lo in range(0, max(len(terms), 1), MAX_TERMS):
            chunk = terms[lo: lo + MAX_TERMS]
            n = len(chunk)
            arr = lambda f: (H * max(n, 1))(*[f(t) for t in chunk])
            hp = lambda p: p.h if p is not None else 0
            check(load().he_lintrans_mul_sum(
                self.eval.h, levelQ, levelP, n, arr(lambda t: t[0][0].h), arr(lambda t: t[0][1].h),
                arr(lambda t: t[1][0].h), arr(lambda t: hp(t[1][1])), arr(lambda t: t[2][0].h), arr(lambda t: hp(t[2][1])),
                arr(lambda t: t[3].h if t[3] is not None else 0), int(accumulate or lo > 0),
                out0[0].h, out0[1].h, out1[0].h, out1[1].h))

    # Evaluator.EvaluateMany (:27); ctIn = [c0, c1] NTT at level ctLevel; opOut = list of [c0, c1]
    def EvaluateMany(self, ctLevel, ctIn, linearTransformations, opOut):
        if len(opOut) < len(linearTransformations):
            raise ValueError("output *rlwe.Ciphertext slice is too small")
        levelP = linearTransformations[0].LevelP
        levelQ = 0
        for lt in linearTransformations:
            levelQ = max(levelQ, lt.LevelQ)
            if lt.LevelP != levelP:
                raise ValueError("all linearTransformations must have the same levelP")
        levelQ = min(levelQ, ctLevel)
        decomp = Decomposition(self.eval, ctIn[0].batch)
        self.eval.DecomposeNTT(levelQ, levelP, levelP + 1, ctIn[1], True, decomp)
        ctPreRot = {}
        for lt, out in zip(linearTransformations, opOut):
            if lt.N1 == 0:
                self.MultiplyByDiagMatrix(ctLevel, ctIn, lt, decomp, out)
            else:
                _, _, rotN2 = lt.BSGSIndex()
                self.PreRotatedCiphertextForDiagonalMatrixMultiplication(levelQ, levelP, ctIn, decomp, rotN2, ctPreRot)
                self.MultiplyByDiagMatrixBSGS(ctLevel, ctIn, lt, ctPreRot, out)

    # :82
    def PreRotatedCiphertextForDiagonalMatrixMultiplication(self, levelQ, levelP, ctIn, decomp, rots, ctPreRot):
        keep = set(rots)
        for i in list(ctPreRot):
            if i not in keep:
                del ctPreRot[i]
        B = ctIn[0].batch
        for i in rots:
            if i != 0 and i not in ctPreRot:
                ctPreRot[i] = [self._qp(levelQ, levelP, B), self._qp(levelQ, levelP, B)]
                galEl = self.GaloisElement(i)
                self.eval.AutomorphismHoistedLazy(levelQ, ctIn, decomp, galEl, self.gks.GetGaloisKey(galEl), ctPreRot[i])

    # Evaluator.MultiplyByDiagMatrix (:142): single hoisting, no baby-step giant-step
    def MultiplyByDiagMatrix(self, ctLevel, ctIn, matrix: LinearTransformation, decomp: Decomposition, opOut):
        levelQ, levelP = min(opOut[0].Level(), ctLevel, matrix.LevelQ), matrix.LevelP
        rQ, rP = self.ringQ.AtLevel(levelQ), self.ringP.AtLevel(levelP)
        B = ctIn[0].batch
        c0P, c1P = Poly(self.ringP, levelP + 1, B), Poly(self.ringP, levelP + 1, B)
        c0OutQP, c1OutQP = (opOut[0], c0P), (opOut[1], c1P)
        ct0TimesP = Poly(self.ringQ, levelQ + 1, B, zero=False)
        cQP = [self._qp(levelQ, levelP, B), self._qp(levelQ, levelP, B)]
        ctInTmp0, ctInTmp1 = Poly(self.ringQ, levelQ + 1, B, zero=False), Poly(self.ringQ, levelQ + 1, B, zero=False)
        ctInTmp0.CopyLvl(levelQ, ctIn[0])
        ctInTmp1.CopyLvl(levelQ, ctIn[1])
        P = 1
        for m in self.ringP.ModuliChain()[: levelP + 1]:
            P *= int(m)
        rQ.MulScalarBigint(ctInTmp0, P, ct0TimesP)  # P*c0 (:190)
        keys = sorted(matrix.Vec.keys())
        state = False
        if keys[0] == 0:
            state, keys = True, keys[1:]
        for i, k0 in enumerate(keys):
            k = k0 & (matrix.slots - 1)
            galEl = self.GaloisElement(k)
            evk = self.gks.GetGaloisKey(galEl)
            if evk.LevelP() != levelP:
                raise ValueError(f"LinearTransformation.LevelP = {levelP} != GaloiKey[{galEl}].LevelP() = {evk.LevelP()}")
            index = self.AutomorphismIndex(galEl)
            self.eval.GadgetProductHoistedLazy(levelQ, decomp, evk, cQP)
            rQ.Add(cQP[0][0], ct0TimesP, cQP[0][0])
            # AutomorphismNTTWithIndex + MulCoeffsMontgomery[ThenAdd] on QP (:224-241) in one fused pass per ring; the
            # accumulator stays canonical, so the reference's periodic Reduce calls (:243-262) are no-ops here
            self._mul_sum(levelQ, levelP, [(matrix.Vec[k], cQP[0], cQP[1], index)], c0OutQP, c1OutQP, accumulate=i > 0)
        self.eval.ModDownQPtoQNTT(levelQ, levelP, c0OutQP[0], c0OutQP[1], c0OutQP[0])  # sum(phi(c0*P + d0_QP))/P
        self.eval.ModDownQPtoQNTT(levelQ, levelP, c1OutQP[0], c1OutQP[1], c1OutQP[0])  # sum(phi(d1_QP))/P
        if state:  # rotation by zero
            rQ.MulCoeffsMontgomeryThenAdd(matrix.Vec[0][0], ctInTmp0, c0OutQP[0])
            rQ.MulCoeffsMontgomeryThenAdd(matrix.Vec[0][0], ctInTmp1, c1OutQP[0])

    # Evaluator.MultiplyByDiagMatrixBSGS (:280): double hoisting with baby-step giant-step
    def MultiplyByDiagMatrixBSGS(self, ctLevel, ctIn, matrix: LinearTransformation, ctInPreRot: dict, opOut):
        levelQ, levelP = min(opOut[0].Level(), ctLevel, matrix.LevelQ), matrix.LevelP
        rQ, rP = self.ringQ.AtLevel(levelQ), self.ringP.AtLevel(levelP)
        B = ctIn[0].batch
        QiOverF = _overflow_margin(self.ringQ.ModuliChain(), levelQ) >> 1
        PiOverF = _overflow_margin(self.ringP.ModuliChain(), levelP) >> 1
        index, _, _ = matrix.BSGSIndex()
        ctInTmp0, ctInTmp1 = Poly(self.ringQ, levelQ + 1, B, zero=False), Poly(self.ringQ, levelQ + 1, B, zero=False)
        ctInTmp0.CopyLvl(levelQ, ctIn[0])
        ctInTmp1.CopyLvl(levelQ, ctIn[1])
        tmp0QP, tmp1QP = self._qp(levelQ, levelP, B), self._qp(levelQ, levelP, B)   # accumulator, inner loop
        cQP = [self._qp(levelQ, levelP, B), self._qp(levelQ, levelP, B)]             # accumulator, outer loop
        c0OutQP = (opOut[0], Poly(self.ringP, levelP + 1, B, zero=False))  # written by the first giant step
        c1OutQP = (opOut[1], Poly(self.ringP, levelP + 1, B, zero=False))
        P = 1
        for m in self.ringP.ModuliChain()[: levelP + 1]:
            P *= int(m)
        rQ.MulScalarBigint(ctInTmp0, P, ctInTmp0)  # P*c0
        rQ.MulScalarBigint(ctInTmp1, P, ctInTmp1)  # P*c1
        cnt0 = 0
        for j in sorted(index.keys()):  # outer loop
            # inner loop (:342-394): tmp_k = Reduce(sum_i pt_{j+i} (.) ct_i[k]), one fused pass instead of one
            # MulCoeffsMontgomeryLazy[ThenAddLazy] launch per diagonal plus the periodic Reduce calls
            terms = []
            for i in index[j]:
                pt = matrix.Vec[j + i]
                if i == 0:
                    terms.append((pt, (ctInTmp0, None), (ctInTmp1, None), None))  # P*ct, no P part (:349-352)
                else:
                    ct = ctInPreRot[i]
                    terms.append((pt, ct[0], ct[1], None))
            self._mul_sum(levelQ, levelP, terms, tmp0QP, tmp1QP)
            if j != 0:
                # hoisting of the ModDown of sum(sum(phi(d1) * plaintext)) (:397)
                self.eval.ModDownQPtoQNTT(levelQ, levelP, tmp1QP[0], tmp1QP[1], tmp1QP[0])
                galEl = self.GaloisElement(j)
                evk = self.gks.GetGaloisKey(galEl)
                if evk.LevelP() != levelP:
                    raise ValueError(f"LinearTransformation.LevelP = {levelP} != GaloiKey[{galEl}].LevelP() = {evk.LevelP()}")
                if self.fuse_giant:
                    # :412-423 as one native call (he_lintrans_giant_step): the key inner products store c0 + tmp0 and c1 through
                    # the rotation into the outer accumulators; cQP is not materialised
                    self.eval.LinTransGiantStep(levelQ, tmp1QP[0], evk, galEl, tmp0QP, (c0OutQP, c1OutQP), cnt0 != 0)
                else:
                    rotIndex = self.AutomorphismIndex(galEl)
                    self.eval.GadgetProductLazy(levelQ, tmp1QP[0], evk, cQP)
                    rQ.Add(cQP[0][0], tmp0QP[0], cQP[0][0])
                    rP.Add(cQP[0][1], tmp0QP[1], cQP[0][1])
                    for src, dst in ((cQP[0], c0OutQP), (cQP[1], c1OutQP)):
                        if cnt0 == 0:
                            rQ.AutomorphismNTTWithIndex(src[0], rotIndex, dst[0])
                            rP.AutomorphismNTTWithIndex(src[1], rotIndex, dst[1])
                        else:
                            rQ.AutomorphismNTTWithIndexThenAddLazy(src[0], rotIndex, dst[0])
                            rP.AutomorphismNTTWithIndexThenAddLazy(src[1], rotIndex, dst[1])
            else:
                for src, dst in ((tmp0QP, c0OutQP), (tmp1QP, c1OutQP)):
                    if cnt0 == 0:
                        dst[0].CopyLvl(levelQ, src[0])
                        dst[1].CopyLvl(levelP, src[1])
                    else:
                        rQ.AddLazy(dst[0], src[0], dst[0])
                        rP.AddLazy(dst[1], src[1], dst[1])
            if cnt0 % QiOverF == QiOverF - 1:
                rQ.Reduce(opOut[0], opOut[0])
                rQ.Reduce(opOut[1], opOut[1])
            if cnt0 % PiOverF == PiOverF - 1:
                rP.Reduce(c0OutQP[1], c0OutQP[1])
                rP.Reduce(c1OutQP[1], c1OutQP[1])
            cnt0 += 1
        if cnt0 % QiOverF != 0:
            rQ.Reduce(opOut[0], opOut[0])
            rQ.Reduce(opOut[1], opOut[1])
        if cnt0 % PiOverF != 0:
            rP.Reduce(c0OutQP[1], c0OutQP[1])
            rP.Reduce(c1OutQP[1], c1OutQP[1])
        self.eval.ModDownQPtoQNTT(levelQ, levelP, opOut[0], c0OutQP[1], opOut[0])  # sum(phi(c0 * P + d0_QP))/P
        self.eval.ModDownQPtoQNTT(levelQ, levelP, opOut[1], c1OutQP[1], opOut[1])  # sum(phi(d1_QP))/P

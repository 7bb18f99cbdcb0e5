"""Host-side mirror of the rlwe.Evaluator hot path (core/rlwe/evaluator*.go,
operator interface core/rlwe/rlwe.go:10-18) and of the CKKS/BGV call sites that
sit on it, backed by libhering's HIP kernels.  NTT-domain ciphertexts; a
ciphertext is a list of ``Poly`` (rlwe.Ciphertext.Value), a QP element a
``(Q, P)`` pair of ``Poly`` (ringqp.Poly)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import H, check, load, u64p
from .ring import BasisExtender, Poly, Ring, _p


def BaseRNSDecompositionVectorSize(levelQ: int, levelP: int) -> int:
    """core/rlwe/params.go:543-550"""
    if levelP == -1:
        return levelQ + 1
    return (levelQ + levelP + 1) // (levelP + 1)


class EvaluationKey:
    """rlwe.GadgetCiphertext / EvaluationKey with BaseTwoDecomposition = 0
    (core/rlwe/gadgetciphertext.go:19): q [beta,2,nQk,N], p [beta,2,nPk,N], NTT + Montgomery."""

    def __init__(self, evaluator: "Evaluator", q: np.ndarray, p: np.ndarray, BaseTwoDecomposition: int = 0,
                 BaseTwoDecompositionVectorSize=None, shape=None):
        if shape is not None:  # (beta, nQk, nPk): a zeroed key to be filled on the device (DeviceBuffer / Commit)
            self.beta, self.nQk, self.nPk = shape
            q = p = None
        else:
            q = np.ascontiguousarray(q, dtype=np.uint64)
            # p = None: a key without P part (parameters without special primes, levelP = -1; base-2 gadgets only)
            p = np.zeros(q.shape[:2] + (0, q.shape[3]), dtype=np.uint64) if p is None else np.ascontiguousarray(p, dtype=np.uint64)
            assert q.ndim == 4 and p.ndim == 4 and q.shape[:2] == p.shape[:2] and q.shape[1] == 2
            self.beta, self.nQk, self.nPk = q.shape[0], q.shape[2], p.shape[2]
        self.N = evaluator.ringQ.N
        self.BaseTwoDecomposition = BaseTwoDecomposition
        self.BaseTwoDecompositionVectorSize = list(BaseTwoDecompositionVectorSize) if BaseTwoDecomposition else None
        h = H()
        pq, pp = (None, None) if q is None else (_p(q), _p(p) if p.size else None)
        if BaseTwoDecomposition:
            nj = (C.c_int * len(BaseTwoDecompositionVectorSize))(*BaseTwoDecompositionVectorSize)
            check(load().he_evk_create_base2(evaluator.h, BaseTwoDecomposition, nj, len(BaseTwoDecompositionVectorSize),
                                             self.nQk, self.nPk, pq, pp, C.byref(h)))
        else:
            check(load().he_evk_create(evaluator.h, self.beta, self.nQk, self.nPk, pq, pp, C.byref(h)))
        self.h = h.value

    def LevelQ(self):
        return self.nQk - 1

    def LevelP(self):
        return self.nPk - 1

    def Shape(self):
        """What a peer needs to allocate the same key: (beta, nQk, nPk, BaseTwoDecomposition, vector sizes)."""
        return (self.beta, self.nQk, self.nPk, self.BaseTwoDecomposition, self.BaseTwoDecompositionVectorSize)

    def DeviceBuffer(self):
        """(device pointer, bytes) of the key words [beta][2][nQk + nPk][N]; the context's stream is drained first."""
        ptr, nb = C.c_void_p(), C.c_size_t()
        check(load().he_evk_device_buffer(self.h, C.byref(ptr), C.byref(nb)))
        return ptr.value, nb.value

    def Commit(self):
        """After an external device-side write of the key words (RCCL broadcast) has completed."""
        check(load().he_evk_commit(self.h))

    def download(self) -> np.ndarray:
        """The key words as [beta, 2, nQk + nPk, N] (tests / wire export)."""
        out = np.empty((self.beta, 2, self.nQk + self.nPk, self.N), dtype=np.uint64)
        check(load().he_evk_download(self.h, _p(out), out.size))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                load().he_evk_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Decomposition:
    """BuffDecompQP []ringqp.Poly of Evaluator.DecomposeNTT, device resident."""

    def __init__(self, evaluator: "Evaluator", batch: int = 1):
        self.ev = evaluator
        self.batch = batch
        h = H()
        check(load().he_decomp_create(evaluator.h, batch, C.byref(h)))
        self.h = h.value

    def limb(self, b, digit, is_p, limb) -> np.ndarray:
        out = np.empty(self.ev.ringQ.N, dtype=np.uint64)
        check(load().he_decomp_download_limb(self.h, b, digit, int(is_p), limb, _p(out)))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                load().he_decomp_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Evaluator:
    """rlwe.Evaluator (core/rlwe/evaluator.go:12) restricted to the key-switch path."""

    def __init__(self, ringQ: Ring, ringP: Ring | None):
        """ringP = None: parameters without special primes (rlwe.ParametersLiteral.P = nil): levelP = -1 in every call, keys
        are base-2 gadgets without a P part and a QP element is a (Q, None) pair."""
        self.ringQ, self.ringP = ringQ, ringP
        h = H()
        check(load().he_evaluator_create(ringQ.h, ringP.h if ringP is not None else 0, C.byref(h)))
        self.h = h.value

    def __del__(self):
        try:
            if getattr(self, "h", None):
                load().he_evaluator_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def NewEvaluationKey(self, q, p, BaseTwoDecomposition=0, BaseTwoDecompositionVectorSize=None) -> EvaluationKey:
        return EvaluationKey(self, q, p, BaseTwoDecomposition, BaseTwoDecompositionVectorSize)

    def SetCoalescing(self, max_batch: int = 64, window_us: int = 30):
        """he_evaluator_set_coalescing (include/hering.h): concurrent single-ciphertext MulRelin calls on this evaluator -- one
        OS thread per ciphertext, the reference's own parallel mode (b.RunParallel over shallow copies of one evaluator,
        schemes/ckks/ckks_benchmarks_test.go:116-207, core/rlwe/evaluator.go:200-227) -- are gathered into batched launches
        over the callers' own polynomials.  max_batch <= 1 switches it off."""
        check(load().he_evaluator_set_coalescing(self.h, max_batch, window_us))

    def CoalescingStats(self) -> dict:
        out = (C.c_uint64 * 4)()
        check(load().he_evaluator_coalescing_stats(self.h, out))
        return {"calls": int(out[0]), "launches": int(out[1]), "largest_batch": int(out[2]), "one_by_one": int(out[3])}

    def EvaluationKeyFromBinary(self, data: bytes) -> EvaluationKey:
        """rlwe.EvaluationKey.UnmarshalBinary (core/rlwe/keys.go:443 -> gadgetciphertext.go:134): load a key the
        reference serialised straight into a device handle."""
        from . import wire
        kq, kp, base_two, nj = wire.gadget_ciphertext_unmarshal(data)
        return EvaluationKey(self, kq, kp, base_two, nj if base_two else None)

    def GaloisKeyFromBinary(self, data: bytes):
        """rlwe.GaloisKey.UnmarshalBinary (core/rlwe/keys.go:659) -> (GaloisElement, NthRoot, EvaluationKey)"""
        from . import wire
        g, nth, kq, kp, base_two, nj = wire.galois_key_unmarshal(data)
        return g, nth, EvaluationKey(self, kq, kp, base_two, nj if base_two else None)

    # ring.Decomposer.DecomposeAndSplit (ring/basis_extension.go:381)
    def DecomposeAndSplit(self, levelQ, levelP, nbPi, digit, p0Q: Poly, p1Q: Poly, p1P: Poly):
        check(load().he_decompose_and_split(self.h, levelQ, levelP, nbPi, digit, p0Q.h, p1Q.h, p1P.h))

    # EvaluatorProvider.DecomposeNTT (core/rlwe/evaluator_gadget_product.go:459)
    def DecomposeNTT(self, levelQ, levelP, nbPi, c2: Poly, c2IsNTT: bool, decomp: Decomposition):
        check(load().he_decompose_ntt(self.h, levelQ, levelP, nbPi, c2.h, int(c2IsNTT), decomp.h))

    # EvaluatorProvider.GadgetProductLazy (:108); ctQP = [(Q0,P0),(Q1,P1)].  isNTT is the IsNTT flag of ctQP, i.e. the
    # domain of cx and of the result (:121-125, :142-152)
    def GadgetProductLazy(self, levelQ, cx: Poly, evk: EvaluationKey, ctQP, isNTT: bool = True):
        (q0, p0), (q1, p1) = ctQP
        hp = lambda p: p.h if p is not None else 0  # levelP = -1: no P part
        if isNTT:
            check(load().he_gadget_product_lazy(self.h, levelQ, cx.h, evk.h, q0.h, hp(p0), q1.h, hp(p1)))
            return
        rQ = self.ringQ.AtLevel(levelQ)
        rP = self.ringP.AtLevel(evk.LevelP()) if evk.LevelP() >= 0 else None
        cxNTT = Poly(self.ringQ, levelQ + 1, cx.batch, zero=False)
        rQ.NTT(cx, cxNTT)
        check(load().he_gadget_product_lazy(self.h, levelQ, cxNTT.h, evk.h, q0.h, hp(p0), q1.h, hp(p1)))
        for q, p in ctQP:  # ringQP.INTT (:121-125)
            rQ.INTT(q, q)
            if rP is not None:
                rP.INTT(p, p)

    # EvaluatorProvider.GadgetProductHoistedLazy (:379)
    def GadgetProductHoistedLazy(self, levelQ, decomp: Decomposition, evk: EvaluationKey, ctQP):
        (q0, p0), (q1, p1) = ctQP
        check(load().he_gadget_product_hoisted_lazy(self.h, levelQ, decomp.h, evk.h, q0.h, p0.h, q1.h, p1.h))

    def GadgetProductHoistedLazyDigits(self, levelQ, decomp: Decomposition, evk: EvaluationKey, digit_begin: int, digit_end: int, ctQP):
        """the inner product of GadgetProductHoistedLazy over the digits [digit_begin, digit_end) only (canonical): the
        per-rank share when one key switch is split over several GPUs by digit (dist.SplitGadgetProductHoisted)"""
        (q0, p0), (q1, p1) = ctQP
        check(load().he_gadget_product_hoisted_lazy_digits(self.h, levelQ, decomp.h, evk.h, digit_begin, digit_end, q0.h, p0.h, q1.h, p1.h))

    # Evaluator.ModDown (:39-97), the four domain combinations of (ctQP.IsNTT, ct.IsNTT); ctQP is modified in place in the
    # NTT -> INTT case, as the reference
    # BasisExtender.ModDownQPtoQNTT (ring/basis_extension.go:235) on the evaluator's fused pipeline; p2Q may alias p1Q
    def ModDownQPtoQNTT(self, levelQ, levelP, p1Q: Poly, p1P: Poly, p2Q: Poly):
        check(load().he_eval_moddown_qp_to_q_ntt(self.h, levelQ, levelP, p1Q.h, p1P.h, p2Q.h))

    def ModDown(self, levelQ, levelP, ctQP, ct, ctQPIsNTT: bool = True, ctIsNTT: bool = True):
        (q0, p0), (q1, p1) = ctQP
        if levelP == -1:  # no special primes (:74-96): a copy, or one NTT / INTT of the Q part
            rQ = self.ringQ.AtLevel(levelQ)
            if ctQPIsNTT == ctIsNTT:
                check(load().he_moddown(self.h, levelQ, -1, q0.h, 0, q1.h, 0, ct[0].h, ct[1].h))
            else:
                for q, o in zip((q0, q1), ct):
                    (rQ.INTT if ctQPIsNTT else rQ.NTT)(q, o)
            return
        if ctQPIsNTT and ctIsNTT:
            check(load().he_moddown(self.h, levelQ, levelP, q0.h, p0.h, q1.h, p1.h, ct[0].h, ct[1].h))
            return
        from .ring import BasisExtender
        if getattr(self, "_be", None) is None:
            self._be = BasisExtender(self.ringQ, self.ringP)
        rQ, rP = self.ringQ.AtLevel(levelQ), self.ringP.AtLevel(levelP)
        if ctQPIsNTT:  # NTT -> INTT (:51-58)
            for q, p in ctQP:
                rQ.INTTLazy(q, q)
                rP.INTTLazy(p, p)
        for (q, p), o in zip(ctQP, ct):
            self._be.ModDownQPtoQ(levelQ, levelP, q, p, o)
        if not ctQPIsNTT and ctIsNTT:  # INTT -> NTT (:62-67)
            for o in ct:
                rQ.NTT(o, o)

    # Evaluator.GadgetProduct (:16) / GadgetProductHoisted (:348); isNTT = ct.IsNTT (domain of cx and of the result)
    def GadgetProduct(self, levelQ, cx: Poly, evk: EvaluationKey, ct, isNTT: bool = True):
        if isNTT:
            check(load().he_gadget_product(self.h, levelQ, cx.h, evk.h, ct[0].h, ct[1].h))
            return
        B = cx.batch
        ctQP = [(Poly(self.ringQ, levelQ + 1, B, zero=False),
                 Poly(self.ringP, evk.LevelP() + 1, B, zero=False) if evk.LevelP() >= 0 else None) for _ in range(2)]
        self.GadgetProductLazy(levelQ, cx, evk, ctQP, isNTT=False)
        self.ModDown(levelQ, evk.LevelP(), ctQP, ct, ctQPIsNTT=False, ctIsNTT=False)

    def GadgetProductHoisted(self, levelQ, decomp: Decomposition, evk: EvaluationKey, ct):
        check(load().he_gadget_product_hoisted(self.h, levelQ, decomp.h, evk.h, ct[0].h, ct[1].h))

    # Evaluator.Relinearize (core/rlwe/evaluator_evaluationkey.go:117)
    def Relinearize(self, level, ctIn, rlk: EvaluationKey, opOut):
        check(load().he_relinearize(self.h, level, ctIn[0].h, ctIn[1].h, ctIn[2].h, rlk.h, opOut[0].h, opOut[1].h))

    # Evaluator.Automorphism (core/rlwe/evaluator_automorphism.go:13)
    def Automorphism(self, level, ctIn, galEl: int, gk: EvaluationKey, opOut):
        check(load().he_automorphism_ct(self.h, level, ctIn[0].h, ctIn[1].h, galEl, gk.h, opOut[0].h, opOut[1].h))

    # Evaluator.AutomorphismHoisted (:60)
    def AutomorphismHoisted(self, level, ctIn, c1DecompQP: Decomposition, galEl: int, gk: EvaluationKey, opOut):
        check(load().he_automorphism_hoisted(self.h, level, ctIn[0].h, c1DecompQP.h, galEl, gk.h, opOut[0].h, opOut[1].h))

    # EvaluatorProvider.AutomorphismHoistedLazy (:104)
    def AutomorphismHoistedLazy(self, levelQ, ctIn, c1DecompQP: Decomposition, galEl: int, gk: EvaluationKey, ctQP):
        (q0, p0), (q1, p1) = ctQP
        check(load().he_automorphism_hoisted_lazy(self.h, levelQ, ctIn[0].h, c1DecompQP.h, galEl, gk.h,
                                                  q0.h, p0.h, q1.h, p1.h))

    # the giant step of lintrans.Evaluator.MultiplyByDiagMatrixBSGS (circuits/common/lintrans/lintrans_evaluator.go:397-441):
    # GadgetProductLazy(cx) -> cQP; cQP[0] += addQP; outQP[k] (+)= phi_galEl(cQP[k]) -- he_lintrans_giant_step
    def LinTransGiantStep(self, levelQ, cx: Poly, gk: EvaluationKey, galEl: int, addQP, outQP, accumulate: bool):
        (q0, p0), (q1, p1) = outQP
        check(load().he_lintrans_giant_step(self.h, levelQ, cx.h, gk.h, galEl, addQP[0].h, addQP[1].h, q0.h, p0.h, q1.h, p1.h,
                                            1 if accumulate else 0))

    # schemes/ckks Evaluator.Mul / MulRelin (schemes/ckks/evaluator.go:613,742 -> mulRelin :764)
    def CKKSMulRelin(self, level, op0, op1, rlk: EvaluationKey | None, opOut):
        o2 = opOut[2].h if rlk is None else 0
        check(load().he_ckks_mul_relin(self.h, level, op0[0].h, op0[1].h, op1[0].h, op1[1].h, rlk.h if rlk else 0,
                                       opOut[0].h, opOut[1].h, o2))

    # schemes/bgv Evaluator.Mul / MulRelin (schemes/bgv/evaluator.go:529 -> tensorStandard :592)
    def BGVMulRelin(self, level, t: int, op0, op1, rlk: EvaluationKey | None, opOut):
        o2 = opOut[2].h if rlk is None else 0
        check(load().he_bgv_mul_relin(self.h, level, t, op0[0].h, op0[1].h, op1[0].h, op1[1].h, rlk.h if rlk else 0,
                                      opOut[0].h, opOut[1].h, o2))

    # schemes/{ckks,bgv} Evaluator.Rescale (ckks :477, bgv :1363)
    def Rescale(self, level, nbRescales, op0, opOut):
        n = min(len(op0), len(opOut))
        hs = lambda ps: (H * n)(*[p.h for p in ps[:n]])
        check(load().he_rescale_polys(self.ringQ.h, level, nbRescales, n, hs(op0), hs(opOut)))


# ----------------------------------------------------------------------------------------------------
# core/rlwe/inner_sum.go: Trace / PartialTracesSum / InnerSum / Replicate.  Host-side drivers over the
# device-resident operators above, exactly as the reference's own code sits on rlwe.EvaluatorProvider.
# ----------------------------------------------------------------------------------------------------
GaloisGen = 5  # core/rlwe/params.go:33


def GaloisElement(nth_root: int, k: int) -> int:
    """Parameters.GaloisElement (core/rlwe/params.go:580): GaloisGen^k mod NthRoot, k reduced as a uint64."""
    return pow(GaloisGen, (k & 0xFFFFFFFFFFFFFFFF) & (nth_root - 1), nth_root)


class GaloisKeySet:
    """rlwe.MemEvaluationKeySet restricted to Galois keys (core/rlwe/evaluationkeyset.go): galEl -> EvaluationKey."""

    def __init__(self, keys: dict | None = None):
        self.keys = dict(keys or {})

    def GetGaloisKey(self, galEl: int) -> EvaluationKey:
        if galEl not in self.keys:
            raise KeyError(f"GaloisKey[{galEl}] is nil")
        return self.keys[galEl]


class InnerSumEvaluator:
    """The inner_sum.go methods of rlwe.Evaluator, bound to an Evaluator and a Galois key set."""

    def __init__(self, evaluator: Evaluator, gks: GaloisKeySet):
        self.eval, self.gks = evaluator, gks
        self.ringQ, self.ringP = evaluator.ringQ, evaluator.ringP
        self.be = BasisExtender(self.ringQ, self.ringP)
        self.nth_root = self.ringQ.NthRoot()
        self.logN = self.ringQ.N.bit_length() - 1

    def GaloisElement(self, k: int) -> int:
        return GaloisElement(self.nth_root, k)

    # Evaluator.Trace (core/rlwe/inner_sum.go:36); ct = [c0, c1] at `level`
    def Trace(self, level, ctIn, logN: int, opOut, isNTT: bool = True):
        rQ = self.ringQ.AtLevel(level)
        gap = 1 << (self.logN - logN - 1)
        if logN == 0:
            gap <<= 1
        if gap <= 1:
            if ctIn is not opOut:
                for a, b in zip(opOut, ctIn):
                    a.CopyLvl(level, b)
            return
        ci = getattr(self.ringQ, "conjugate_invariant", False)
        if ci:
            gap >>= 1  # the last step, phi(5^-1), is skipped on Z[X + X^-1] (core/rlwe/inner_sum.go:60-62)
        Q = 1
        for m in self.ringQ.ModuliChain()[: level + 1]:
            Q *= int(m)
        ninv = pow(gap, -1, Q)
        for a, b in zip(ctIn, opOut):
            rQ.MulScalarBigint(a, ninv, b)  # pre-multiplication by (N/n)^-1 (:68-70)
            if not isNTT:
                rQ.NTT(b, b)
        buff = [Poly(self.ringQ, level + 1, opOut[0].batch, zero=False) for _ in range(2)]
        steps = [self.GaloisElement(1 << i) for i in range(logN, self.logN - 1)]
        if logN == 0 and not ci:
            steps.append(self.nth_root - 1)  # X -> X^-1, standard ring only (:97-105)
        for galEl in steps:
            self.eval.Automorphism(level, opOut, galEl, self.gks.GetGaloisKey(galEl), buff)
            for a, b in zip(opOut, buff):
                rQ.Add(a, b, a)
        if not isNTT:
            for b in opOut:
                rQ.INTT(b, b)

    # Evaluator.PartialTracesSum (core/rlwe/inner_sum.go:147)
    def PartialTracesSum(self, level, ctIn, offset: int, n: int, opOut, isNTT: bool = True):
        if n == 0 or offset == 0:
            raise ValueError("partialtrace: invalid parameter (n = 0 or batchSize = 0)")
        levelQ, levelP = level, self.ringP.MaxLevel()
        rQ, rP = self.ringQ.AtLevel(levelQ), self.ringP.AtLevel(levelP)
        B = ctIn[0].batch
        ctInNTT = [Poly(self.ringQ, levelQ + 1, B, zero=False) for _ in range(2)]
        for a, b in zip(ctIn, ctInNTT):
            if not isNTT:
                rQ.NTT(a, b)
            else:
                b.CopyLvl(levelQ, a)
        if n == 1:
            if ctIn is not opOut:
                for a, b in zip(ctIn, opOut):
                    b.CopyLvl(levelQ, a)
        else:
            newQP = lambda: (Poly(self.ringQ, levelQ + 1, B), Poly(self.ringP, levelP + 1, B))
            accQP, cQP = [newQP(), newQP()], [newQP(), newQP()]
            cQ = [cQP[0][0], cQP[1][0]]
            decomp = Decomposition(self.eval, B)
            state, copy = False, True
            i, j = 0, n
            while j > 0:  # binary reading of n (:216)
                self.eval.DecomposeNTT(levelQ, levelP, levelP + 1, ctInNTT[1], True, decomp)
                if j & 1:
                    k = (n - (n & ((2 << i) - 1))) * offset
                    if k != 0:
                        rot = self.GaloisElement(k)
                        gk = self.gks.GetGaloisKey(rot)
                        if copy:
                            self.eval.AutomorphismHoistedLazy(levelQ, ctInNTT, decomp, rot, gk, accQP)
                            copy = False
                        else:
                            self.eval.AutomorphismHoistedLazy(levelQ, ctInNTT, decomp, rot, gk, cQP)
                            for a, c in zip(accQP, cQP):
                                rQ.Add(a[0], c[0], a[0])
                                rP.Add(a[1], c[1], a[1])
                    else:
                        state = True
                        if n & (n - 1):
                            for a, o, c in zip(accQP, opOut, ctInNTT):
                                self.eval.ModDownQPtoQNTT(levelQ, levelP, a[0], a[1], o)
                                rQ.Add(o, c, o)
                        else:
                            for o, c in zip(opOut, ctInNTT):
                                o.CopyLvl(levelQ, c)
                if not state:
                    rot = self.GaloisElement((1 << i) * offset)
                    self.eval.AutomorphismHoisted(levelQ, ctInNTT, decomp, rot, self.gks.GetGaloisKey(rot), cQ)
                    for c, t in zip(ctInNTT, cQ):
                        rQ.Add(c, t, c)
                i, j = i + 1, j >> 1
        if not isNTT:
            for o in opOut:
                rQ.INTT(o, o)

    # InnerSum / Replicate (core/rlwe/inner_sum.go:475 and the scheme-level InnerSum wrappers)
    def InnerSum(self, level, ctIn, batchSize: int, n: int, opOut, isNTT: bool = True):
        self.PartialTracesSum(level, ctIn, batchSize, n, opOut, isNTT)

    def Replicate(self, level, ctIn, batchSize: int, n: int, opOut, isNTT: bool = True):
        self.PartialTracesSum(level, ctIn, -batchSize, n, opOut, isNTT)


def GaloisElementsForInnerSum(nth_root: int, batch: int, n: int):
    """core/rlwe/inner_sum.go:442"""
    rots = set()
    i = 1
    while i < n:
        rots.add(i * batch)
        rots.add((n - (n & ((i << 1) - 1))) * batch)
        i <<= 1
    return sorted({GaloisElement(nth_root, k) for k in rots})


def GaloisElementsForTrace(nth_root: int, logN_ring: int, logN: int):
    """core/rlwe/inner_sum.go:120 (standard ring)"""
    g = [GaloisElement(nth_root, 1 << i) for i in range(logN, logN_ring - 1)]
    if logN == 0:
        g.append(nth_root - 1)
    return g


class CKKSRotations:
    """schemes/ckks Evaluator.Rotate / Conjugate / RotateHoisted / RotateHoistedLazyNew / InnerSum
    (schemes/ckks/evaluator.go:1197-1262, 1283-1300), bound to an Evaluator and a Galois key set."""

    def __init__(self, evaluator: Evaluator, gks: GaloisKeySet):
        self.eval, self.gks = evaluator, gks
        self.ise = InnerSumEvaluator(evaluator, gks)
        self.nth_root = evaluator.ringQ.NthRoot()

    def Rotate(self, level, op0, k: int, opOut):
        g = GaloisElement(self.nth_root, k)
        self.eval.Automorphism(level, op0, g, self.gks.GetGaloisKey(g), opOut)

    def Conjugate(self, level, op0, opOut):
        g = self.nth_root - 1  # GaloisElementOrderTwoOrthogonalSubgroup (core/rlwe/params.go:592)
        self.eval.Automorphism(level, op0, g, self.gks.GetGaloisKey(g), opOut)

    def RotateHoisted(self, level, ctIn, rotations, opOut: dict):
        levelP = self.eval.ringP.MaxLevel()
        decomp = Decomposition(self.eval, ctIn[0].batch)
        self.eval.DecomposeNTT(level, levelP, levelP + 1, ctIn[1], True, decomp)
        for i in rotations:
            g = GaloisElement(self.nth_root, i)
            self.eval.AutomorphismHoisted(level, ctIn, decomp, g, self.gks.GetGaloisKey(g), opOut[i])

    def RotateHoistedLazyNew(self, level, rotations, ct, decomp: Decomposition) -> dict:
        levelP, B, out = self.eval.ringP.MaxLevel(), ct[0].batch, {}
        for i in rotations:
            if i != 0:
                out[i] = [(Poly(self.eval.ringQ, level + 1, B, zero=False), Poly(self.eval.ringP, levelP + 1, B, zero=False)) for _ in range(2)]
                g = GaloisElement(self.nth_root, i)
                self.eval.AutomorphismHoistedLazy(level, ct, decomp, g, self.gks.GetGaloisKey(g), out[i])
        return out

    def InnerSum(self, level, ctIn, batchSize: int, n: int, opOut, slots: int | None = None):
        N = slots if slots is not None else self.eval.ringQ.N // 2
        l = n * batchSize
        if n <= 0 or batchSize <= 0:
            raise ValueError("innersum: invalid parameter (n <= 0 or batchSize <= 0)")
        if l > N:
            raise ValueError(f"innersum: invalid parameters (n*batchSize={l} > #slots={N})")
        if l & (l - 1):
            raise ValueError(f"innersum: invalid parameters (n*batchSize={l} does not divide #slots={N})")
        self.ise.PartialTracesSum(level, ctIn, batchSize, n, opOut)


def ConcurrentCalls(op: str, callers, level: int, iters: int, t: int = 0, sync_each: bool = False) -> float:
    """The reference's parallel benchmark shape (b.RunParallel, schemes/ckks/ckks_benchmarks_test.go:116-207) on OS threads inside
    the library (he_debug_concurrent_mul_relin, include/hering_debug.h).  op "ckks_mulrelin" / "bgv_mulrelin" (t = plaintext
    modulus): callers = [(ctx, evaluator, op0, op1, rlk, opOut), ...]; "rotate" (t = Galois element): [(ctx, evaluator, ct, None,
    galois key, ctOut)]; "relinearize": [(ctx, evaluator, ct3, None, rlk, ctOut)]; "gadget_product": [(ctx, evaluator, [cx], None,
    key, ctOut)] -- all with batch-1 polynomials.  Caller i makes `iters` calls on its own handles, synchronising its context
    after every call (sync_each) or once at the end.  Returns the wall time in seconds."""
    code = {"ckks_mulrelin": 0, "bgv_mulrelin": 1, "rotate": 2, "relinearize": 3, "gadget_product": 4}[op]
    n = len(callers)
    arr = lambda f: (H * n)(*[f(c) for c in callers])
    pick = lambda c, which, i: (c[which][i].h if c[which] is not None and len(c[which]) > i else c[2][0].h)
    wall = C.c_double()
    check(load().he_debug_concurrent_mul_relin(
        n, iters, int(sync_each), code, level, t, arr(lambda c: c[0].h), arr(lambda c: c[1].h), arr(lambda c: pick(c, 2, 0)),
        arr(lambda c: pick(c, 2, 1)), arr(lambda c: pick(c, 3, 0) if code < 2 else pick(c, 2, 2)), arr(lambda c: pick(c, 3, 1)),
        arr(lambda c: c[4].h), arr(lambda c: c[5][0].h), arr(lambda c: c[5][1].h), C.byref(wall)))
    return float(wall.value)


def ConcurrentMulRelin(callers, level: int, iters: int, t: int = 0, sync_each: bool = False) -> float:
    """ConcurrentCalls for BGVMulRelin (t != 0) / CKKSMulRelin"""
    return ConcurrentCalls("bgv_mulrelin" if t != 0 else "ckks_mulrelin", callers, level, iters, t, sync_each)

"""GPU parity of the host-side drivers that sit on the device-resident rlwe.EvaluatorProvider operators --
core/rlwe/inner_sum.go (Trace, PartialTracesSum / InnerSum / Replicate) and circuits/common/lintrans
(MultiplyByDiagMatrix, MultiplyByDiagMatrixBSGS, EvaluateMany) -- against the oracle's restatement
(oracle/circuits.py, itself pinned semantically by tests/test_oracle_circuits.py).  Bit-exact."""
import numpy as np
import pytest

import lattigo_amd as la
from drivers import lintrans as LT
from lattigo_amd import rlwe as R
from oracle import circuits as OC
from oracle import oracle as O
from tests.gpu_common import Pair, ctx  # noqa: F401
from tests.helpers import rng_for, uniform_poly

pytestmark = pytest.mark.gpu


class Rig:
    """rings + evaluators on both sides, with (random, arithmetic-parity only) Galois keys created on demand"""

    def __init__(self, ctx, logN, logq, logp, seed):
        q, p = O.GenModuli(logN + 1, logq, logp)
        self.pr = Pair(ctx, logN, len(q), len(p), qmods=q, pmods=p)
        self.q, self.p, self.N = q, p, 1 << logN
        self.rng = rng_for(seed)
        self.gev, self.oev = la.Evaluator(self.pr.gQ, self.pr.gP), O.Evaluator(self.pr.oQ, self.pr.oP)
        self.beta = O.BaseRNSDecompositionVectorSize(len(q) - 1, len(p) - 1)
        self.ogks, self.ggks = {}, R.GaloisKeySet()

    def keys(self, galels):
        for g in galels:
            g = int(g)
            if g in self.ogks:
                continue
            kq = np.stack([np.stack([uniform_poly(self.rng, self.q, self.N) for _ in range(2)]) for _ in range(self.beta)])
            kp = np.stack([np.stack([uniform_poly(self.rng, self.p, self.N) for _ in range(2)]) for _ in range(self.beta)])
            self.ogks[g] = O.EvaluationKey(kq, kp)
            self.ggks.keys[g] = self.gev.NewEvaluationKey(kq, kp)

    def ct(self, level, batch=1):
        return np.stack([np.stack([uniform_poly(self.rng, self.q[: level + 1], self.N) for _ in range(2)])
                         for _ in range(batch)])  # [batch][2][limbs][N]

    def up(self, ct):
        B, _, nl, _ = ct.shape
        return [la.Poly(self.pr.gQ, nl, B).upload(ct[:, k]) for k in range(2)]

    def new_ct(self, level, batch=1):
        return [la.Poly(self.pr.gQ, level + 1, batch) for _ in range(2)]

    @staticmethod
    def down(polys):
        a = np.stack([p.download() for p in polys], axis=1)  # [batch][2][limbs][N]
        return a


@pytest.mark.parametrize("logn", [0, 4, 8, 9])
def test_trace(ctx, logn):
    rg = Rig(ctx, 10, [55, 45, 45, 50], [55, 46], 4100 + logn)
    nth = 2 * rg.N
    rg.keys(R.GaloisElementsForTrace(nth, 10, logn))
    gi, oi = R.InnerSumEvaluator(rg.gev, rg.ggks), OC.InnerSumEvaluator(rg.oev, rg.ogks)
    for level, isntt in ((3, True), (2, True), (3, False)):
        ct = rg.ct(level, 2)
        out = rg.new_ct(level, 2)
        gi.Trace(level, rg.up(ct), logn, out, isNTT=isntt)
        got = Rig.down(out)
        for b in range(2):
            assert np.array_equal(got[b], oi.Trace(ct[b], logn, isNTT=isntt)), (level, isntt, b)


@pytest.mark.parametrize("n,offset", [(1, 1), (2, 1), (3, 2), (5, 1), (8, 1), (12, 3), (7, -1), (4, -2)])
def test_partial_traces_sum(ctx, n, offset):
    rg = Rig(ctx, 10, [55, 45, 45, 50], [55, 46], 4200 + n)
    nth = 2 * rg.N
    rg.keys(R.GaloisElementsForInnerSum(nth, offset, n))
    gi, oi = R.InnerSumEvaluator(rg.gev, rg.ggks), OC.InnerSumEvaluator(rg.oev, rg.ogks)
    for level, isntt, B in ((3, True, 2), (1, True, 1), (3, False, 1)):
        ct = rg.ct(level, B)
        out = rg.new_ct(level, B)
        gi.PartialTracesSum(level, rg.up(ct), offset, n, out, isNTT=isntt)
        got = Rig.down(out)
        for b in range(B):
            assert np.array_equal(got[b], oi.PartialTracesSum(ct[b], offset, n, isNTT=isntt)), (level, isntt, b)
    with pytest.raises(ValueError):
        gi.PartialTracesSum(3, rg.up(rg.ct(3)), 0, n, rg.new_ct(3))
    with pytest.raises(KeyError):
        R.InnerSumEvaluator(rg.gev, R.GaloisKeySet()).PartialTracesSum(3, rg.up(rg.ct(3)), 1, 16, rg.new_ct(3))


def make_lt(rg, diags, levelQ, slots, N1):
    """the same random 'encoded diagonals' on both sides"""
    oVec, gVec = {}, {}
    for d in diags:
        dq, dp = uniform_poly(rg.rng, rg.q[: levelQ + 1], rg.N), uniform_poly(rg.rng, rg.p, rg.N)
        oVec[d] = (dq, dp)
        gVec[d] = (la.Poly(rg.pr.gQ, levelQ + 1).upload(dq), la.Poly(rg.pr.gP, len(rg.p)).upload(dp))
    return (OC.LinearTransformation(oVec, levelQ, len(rg.p) - 1, slots, N1),
            LT.LinearTransformation(gVec, levelQ, len(rg.p) - 1, slots, N1))


@pytest.mark.parametrize("logN,logq,logp", [(10, [55, 45, 45, 50], [55, 46]), (12, [60, 45, 45], [61])])
@pytest.mark.parametrize("diags,N1", [([0, 1, 2, 5], 0), ([1, 3, 130], 0), ([0, 1, 2, 3, 4, 5, 6, 7], 4),
                                      ([1, 2, 9, 17, 18], 8), ([3, 4, 5], 4)])
def test_lintrans(ctx, logN, logq, logp, diags, N1):
    rg = Rig(ctx, logN, logq, logp, 4300 + logN + N1 + len(diags))
    slots, nth = rg.N // 2, 2 * rg.N
    rg.keys(LT.GaloisElements(nth, diags, slots, -1) if N1 == 0 else
            [R.GaloisElement(nth, r) for r in sum(LT.BSGSIndex(diags, slots, N1)[1:], []) if r])
    ge, oe = LT.LinTransEvaluator(rg.gev, rg.ggks), OC.LinTransEvaluator(rg.oev, rg.ogks)
    top = len(rg.q) - 1
    for ct_level, lt_level, B in ((top, top, 2), (top, top - 1, 1), (top - 1, top, 1)):
        olt, glt = make_lt(rg, diags, lt_level, slots, N1)
        ct = rg.ct(ct_level, B)
        lv = min(ct_level, lt_level)
        out = rg.new_ct(lv, B)
        ge.EvaluateMany(ct_level, rg.up(ct), [glt], [out])
        got = Rig.down(out)
        for b in range(B):
            (want,) = oe.EvaluateMany(ct[b], [olt])
            assert np.array_equal(got[b], want), (ct_level, lt_level, b)


def test_evaluate_many_mixed_and_overflow_margins(ctx):
    """two BSGS matrices sharing pre-rotations + one naive matrix in one EvaluateMany; 61-bit moduli make the lazy
    accumulation margins (QiOverflowMargin = 7, halved in BSGS) actually trigger the intermediate Reduce calls."""
    rg = Rig(ctx, 10, [61, 61, 60], [61, 61], 4400)
    slots, nth = rg.N // 2, 2 * rg.N
    d1, d2, d3 = list(range(16)), [0, 2, 4, 6, 9, 11, 33], [1, 2, 3, 4, 5, 6, 7, 8, 9]
    rots = set(sum(LT.BSGSIndex(d1, slots, 8)[1:], [])) | set(sum(LT.BSGSIndex(d2, slots, 8)[1:], [])) | set(d3)
    rg.keys([R.GaloisElement(nth, r) for r in rots if r])
    ge, oe = LT.LinTransEvaluator(rg.gev, rg.ggks), OC.LinTransEvaluator(rg.oev, rg.ogks)
    o1, g1 = make_lt(rg, d1, 2, slots, 8)
    o2, g2 = make_lt(rg, d2, 2, slots, 8)
    o3, g3 = make_lt(rg, d3, 2, slots, 0)
    ct = rg.ct(2)
    outs = [rg.new_ct(2) for _ in range(3)]
    ge.EvaluateMany(2, rg.up(ct), [g1, g2, g3], outs)
    wants = oe.EvaluateMany(ct[0], [o1, o2, o3])
    for k in range(3):
        assert np.array_equal(Rig.down(outs[k])[0], wants[k]), k
    with pytest.raises(ValueError):
        ge.EvaluateMany(2, rg.up(ct), [g1, g2], [outs[0]])


def test_lintrans_mul_sum_kernel(ctx):
    """he_lintrans_mul_sum against the reference's op chain (MulCoeffsMontgomeryLazy[ThenAddLazy] + Reduce, resp.
    AutomorphismNTTWithIndex + MulCoeffsMontgomery[ThenAdd]): gather, accumulate, broadcast and per-entry plaintexts,
    terms without a P part, more terms than one launch takes, lazy (non-canonical) ciphertext words."""
    rg = Rig(ctx, 10, [61, 45, 58], [61, 46], 4500)
    N, B, lq, lp = rg.N, 3, 2, 1
    rng, oQ, oP = rg.rng, rg.pr.oQ, rg.pr.oP
    lte = LT.LinTransEvaluator(rg.gev, rg.ggks)
    for n, with_index, per_entry_pt in ((1, True, False), (5, False, True), (70, True, False)):
        terms, oterms = [], []
        for i in range(n):
            ptq, ptp = uniform_poly(rng, rg.q, N), uniform_poly(rng, rg.p, N)
            if per_entry_pt:
                ptq, ptp = np.stack([ptq] * B), np.stack([ptp] * B)
                ptq[1] = uniform_poly(rng, rg.q, N)
                ptp[1] = uniform_poly(rng, rg.p, N)
            cq = rng.integers(0, 1 << 64, size=(2, B, lq + 1, N), dtype=np.uint64)  # arbitrary 64-bit words
            cp = rng.integers(0, 1 << 64, size=(2, B, lp + 1, N), dtype=np.uint64)
            no_p = (i % 4 == 3)
            gal = pow(5, i + 1, 2 * N) if with_index and i % 2 == 0 else None
            gpt = (la.Poly(rg.pr.gQ, lq + 1, B if per_entry_pt else 1).upload(ptq),
                   la.Poly(rg.pr.gP, lp + 1, B if per_entry_pt else 1).upload(ptp))
            g0 = (la.Poly(rg.pr.gQ, lq + 1, B).upload(cq[0]), None if no_p else la.Poly(rg.pr.gP, lp + 1, B).upload(cp[0]))
            g1 = (la.Poly(rg.pr.gQ, lq + 1, B).upload(cq[1]), None if no_p else la.Poly(rg.pr.gP, lp + 1, B).upload(cp[1]))
            terms.append((gpt, g0, g1, lte.AutomorphismIndex(gal) if gal else None))
            oterms.append((ptq, ptp, cq, cp, no_p, gal))
        out0 = (la.Poly(rg.pr.gQ, lq + 1, B), la.Poly(rg.pr.gP, lp + 1, B))
        out1 = (la.Poly(rg.pr.gQ, lq + 1, B), la.Poly(rg.pr.gP, lp + 1, B))
        prevq = np.stack([np.stack([uniform_poly(rng, rg.q, N) for _ in range(B)]) for _ in range(2)])
        prevp = np.stack([np.stack([uniform_poly(rng, rg.p, N) for _ in range(B)]) for _ in range(2)])
        for acc in (False, True):
            for k, o in enumerate((out0, out1)):
                o[0].upload(prevq[k])
                o[1].upload(prevp[k])
            lte._mul_sum(lq, lp, terms, out0, out1, accumulate=acc)
            got = [(out0[0].download(), out0[1].download()), (out1[0].download(), out1[1].download())]
            for k in range(2):
                for b in range(B):
                    wq = prevq[k][b].copy() if acc else np.zeros((lq + 1, N), dtype=np.uint64)
                    wp = prevp[k][b].copy() if acc else np.zeros((lp + 1, N), dtype=np.uint64)
                    for ptq, ptp, cq, cp, no_p, gal in oterms:
                        tq, tp = (ptq[b], ptp[b]) if per_entry_pt else (ptq, ptp)
                        xq, xp = cq[k][b], cp[k][b]
                        if gal:
                            xq = oQ.AutomorphismNTTWithIndex(xq, oQ.AutomorphismNTTIndex(gal))
                            xp = oP.AutomorphismNTTWithIndex(xp, oP.AutomorphismNTTIndex(gal))
                        wq = oQ.binop("MulCoeffsMontgomeryThenAdd", tq, xq, wq)
                        if not no_p:
                            wp = oP.binop("MulCoeffsMontgomeryThenAdd", tp, xp, wp)
                    assert np.array_equal(got[k][0][b], wq) and np.array_equal(got[k][1][b], wp), (n, acc, k, b)
    with pytest.raises(la.HeringError):  # an input aliasing the output
        lte._mul_sum(lq, lp, [(terms[0][0], out0, terms[0][2], None)], out0, out1)


@pytest.mark.parametrize("logN,logq,logp", [(10, [55, 45, 45], [55, 55]), (12, [60, 50, 40, 40], [61])])
def test_scale_invariant_multiplication(ctx, logN, logq, logp):
    """BFV-style MulRelinScaleInvariant (schemes/bgv/evaluator.go:898-1071) on the device vs the oracle: with and without
    relinearisation, the squaring branch, a lower level, a batch."""
    from drivers import bgv as BGV
    from tests.rlwe_fixtures import downstream_primes
    rg = Rig(ctx, logN, logq, logp, 4600 + logN)
    N, t = rg.N, 65537
    nb = -(-(sum(int(x).bit_length() for x in rg.q) + logN) // 61)
    qm = downstream_primes(61, 2 * N, nb + 1, set(rg.q) | set(rg.p))
    gM, oM = la.Ring(ctx, N, qm), O.Ring(N, qm)
    gs, os_ = BGV.ScaleInvariantEvaluator(rg.gev, gM, t), OC.ScaleInvariantEvaluator(rg.oev, oM, t)
    assert gs.levelQMul == os_.levelQMul
    rg.keys([1])  # any key material: arithmetic parity
    grlk, orlk = rg.ggks.keys[1], rg.ogks[1]
    top = len(rg.q) - 1
    for level, B in ((top, 2), (top - 1, 1)):
        ct0, ct1 = rg.ct(level, B), rg.ct(level, B)
        g0, g1 = rg.up(ct0), rg.up(ct1)
        out = rg.new_ct(level, B)
        gs.MulRelinScaleInvariant(level, g0, g1, grlk, out)
        got = Rig.down(out)
        out3 = rg.new_ct(level, B) + [la.Poly(rg.pr.gQ, level + 1, B)]
        gs.MulRelinScaleInvariant(level, g0, g1, None, out3)
        got3 = Rig.down(out3)
        outs = rg.new_ct(level, B)
        gs.MulRelinScaleInvariant(level, g0, g0, grlk, outs)
        gots = Rig.down(outs)
        for b in range(B):
            assert np.array_equal(got[b], os_.MulRelinScaleInvariant(ct0[b], ct1[b], orlk)), (level, b)
            assert np.array_equal(got3[b], os_.MulRelinScaleInvariant(ct0[b], ct1[b], None)), (level, b)
            assert np.array_equal(gots[b], os_.MulRelinScaleInvariant(ct0[b], None, orlk, square=True)), (level, b)


@pytest.mark.parametrize("sparse,scale,logSlots,levelIn", [(False, 1000.3, 9, 0), (True, 37.6, 6, 0), (True, 0.5, 9, 1),
                                                             (False, 1.0, 3, 2)])
def test_bootstrapping_modup(ctx, sparse, scale, logSlots, levelIn):
    """bootstrapping.Evaluator.ModUp (circuits/ckks/bootstrapping/evaluator.go:612-769): centred lifts with both sign
    conventions, the hoisting buffer filled with the lifted polynomial, message rescaling, Trace; bit-exact, batch 2.
    The first two coefficients are forced to q/2 and q/2 + 1 (where `>=` and `>` differ)."""
    from drivers import bootstrapping as BS
    rg = Rig(ctx, 10, [55, 45, 45, 50], [55, 46], 4700 + logSlots)
    nth, top = 2 * rg.N, len(rg.q) - 1
    rg.keys(R.GaloisElementsForTrace(nth, 10, logSlots) + [3, 7])
    gi, oi = R.InnerSumEvaluator(rg.gev, rg.ggks), OC.InnerSumEvaluator(rg.oev, rg.ogks)
    B = 2
    coeff = rg.ct(levelIn, B)  # build in the coefficient domain so that the edge values can be planted
    q0 = int(rg.q[0])
    for b in range(B):
        for k in range(2):
            coeff[b, k, 0, :4] = [q0 >> 1, (q0 >> 1) + 1, (q0 >> 1) - 1, 0]
    sub = O.Ring(rg.N, rg.q[: levelIn + 1])
    ct = np.stack([np.stack([sub.NTT(coeff[b, k]) for k in range(2)]) for b in range(B)])
    full = np.zeros((B, 2, top + 1, rg.N), dtype=np.uint64)
    full[:, :, : levelIn + 1] = ct
    g = [la.Poly(rg.pr.gQ, top + 1, B).upload(full[:, k]) for k in range(2)]
    kw_g, kw_o = {}, {}
    if sparse:
        kw_g = dict(EvkDenseToSparse=rg.ggks.keys[3], EvkSparseToDense=rg.ggks.keys[7])
        kw_o = dict(EvkDenseToSparse=rg.ogks[3], EvkSparseToDense=rg.ogks[7])
    BS.ModUp(rg.gev, gi, levelIn, g, scale, logSlots, **kw_g)
    got = Rig.down(g)
    for b in range(B):
        assert np.array_equal(got[b], OC.BootstrappingModUp(rg.oev, oi, ct[b], scale, logSlots, **kw_o)), b


def test_ckks_rotation_call_sites(ctx):
    """schemes/ckks Evaluator.Rotate / Conjugate / RotateHoisted / RotateHoistedLazyNew / InnerSum
    (schemes/ckks/evaluator.go:1197-1300) against the oracle's Automorphism family."""
    rg = Rig(ctx, 11, [55, 45, 45, 50], [55, 46], 4800)
    nth, top = 2 * rg.N, 3
    rots = [1, 5, -3, 0, 64]
    rg.keys([R.GaloisElement(nth, k) for k in rots] + [nth - 1] + R.GaloisElementsForInnerSum(nth, 2, 8))
    cr = R.CKKSRotations(rg.gev, rg.ggks)
    B = 2
    ct = rg.ct(top, B)
    g = rg.up(ct)
    outs = {k: rg.new_ct(top, B) for k in rots}
    cr.RotateHoisted(top, g, rots, outs)
    single, conj = rg.new_ct(top, B), rg.new_ct(top, B)
    cr.Rotate(top, g, 5, single)
    cr.Conjugate(top, g, conj)
    dec = la.Decomposition(rg.gev, B)
    rg.gev.DecomposeNTT(top, 1, 2, g[1], True, dec)
    lazy = cr.RotateHoistedLazyNew(top, rots, g, dec)
    assert sorted(lazy) == sorted(k for k in rots if k)
    isum = rg.new_ct(top, B)
    cr.InnerSum(top, g, 2, 8, isum)
    oi = OC.InnerSumEvaluator(rg.oev, rg.ogks)
    for b in range(B):
        dq, dp = rg.oev.DecomposeNTT(top, 1, 2, ct[b][1], True)
        for k in rots:
            ge = R.GaloisElement(nth, k)
            want = rg.oev.Automorphism(ct[b], ge, rg.ogks[ge])
            assert np.array_equal(Rig.down(outs[k])[b], want), k
            if k:
                wQ, wP = rg.oev.AutomorphismHoistedLazy(top, ct[b][0], dq, dp, ge, rg.ogks[ge])
                for c in range(2):
                    assert np.array_equal(lazy[k][c][0].download()[b], wQ[c]) and np.array_equal(lazy[k][c][1].download()[b], wP[c])
        assert np.array_equal(Rig.down(single)[b], rg.oev.Automorphism(ct[b], R.GaloisElement(nth, 5), rg.ogks[R.GaloisElement(nth, 5)]))
        assert np.array_equal(Rig.down(conj)[b], rg.oev.Automorphism(ct[b], nth - 1, rg.ogks[nth - 1]))
        assert np.array_equal(Rig.down(isum)[b], oi.PartialTracesSum(ct[b], 2, 8))
    for bad in ((0, 4), (3, 8), (2, 3)):
        with pytest.raises(ValueError):
            cr.InnerSum(top, g, bad[0], bad[1], isum, slots=32 if bad == (3, 8) else None)


def test_scheme_evaluator_call_sites(ctx):
    """schemes.Evaluator ring-level call sites for CKKS and BGV (MulRelinThenAdd with and without relinearisation, the
    BGV scale-matching branch, ct x pt products, Add/Sub of unequal degrees) on the device vs the oracle, batch 2."""
    from drivers import schemes as S
    rg = Rig(ctx, 11, [55, 45, 45, 50], [55, 46], 4900)
    rg.keys([1])
    grlk, orlk = rg.ggks.keys[1], rg.ogks[1]
    t, B, top = 65537, 2, 3
    ck, bg = S.CKKSEvaluator(rg.gev), S.BGVEvaluator(rg.gev, t)
    for level in (top, 1):
        a, b = rg.ct(level, B), rg.ct(level, B)
        acc = np.stack([np.stack([uniform_poly(rg.rng, rg.q[: level + 1], rg.N) for _ in range(3)]) for _ in range(B)])
        pt = np.stack([uniform_poly(rg.rng, rg.q[: level + 1], rg.N) for _ in range(B)])
        ga, gb = rg.up(a), rg.up(b)
        gpt = la.Poly(rg.pr.gQ, level + 1, B).upload(pt)

        def accs(n):
            return [la.Poly(rg.pr.gQ, level + 1, B).upload(acc[:, k]) for k in range(n)]

        sub = O.Evaluator(O.Ring(rg.N, rg.q[: level + 1]), rg.pr.oP) if level != top else rg.oev
        cases = []
        o = accs(2); ck.MulRelinThenAdd(level, ga, gb, grlk, o)
        cases.append((o, lambda i: OC.ckks_mul_relin_then_add(rg.oev, a[i], b[i], orlk, acc[i][:2])))
        o = accs(3); ck.MulRelinThenAdd(level, ga, gb, None, o)
        cases.append((o, lambda i: OC.ckks_mul_relin_then_add(rg.oev, a[i], b[i], None, acc[i])))
        o = accs(2); so = bg.MulRelinThenAdd(level, ga, gb, grlk, o, scales=(3, 5, 7))
        assert so == OC.bgv_mul_relin_then_add(rg.oev, t, a[0], b[0], orlk, acc[0][:2], scales=(3, 5, 7))[1]
        cases.append((o, lambda i: OC.bgv_mul_relin_then_add(rg.oev, t, a[i], b[i], orlk, acc[i][:2], scales=(3, 5, 7))[0]))
        o = accs(3); bg.MulRelinThenAdd(level, ga, gb, None, o)
        cases.append((o, lambda i: OC.bgv_mul_relin_then_add(rg.oev, t, a[i], b[i], None, acc[i])[0]))
        o = rg.new_ct(level, B); ck.MulPlaintext(level, ga, gpt, o)
        cases.append((o, lambda i: OC.ckks_mul_plaintext(sub, a[i], pt[i])))
        o = accs(2); ck.MulPlaintextThenAdd(level, ga, gpt, o)
        cases.append((o, lambda i: OC.ckks_mul_plaintext(sub, a[i], pt[i], acc[i][:2])))
        o = rg.new_ct(level, B); bg.MulPlaintext(level, ga, gpt, o)
        cases.append((o, lambda i: OC.bgv_mul_plaintext(sub, t, a[i], pt[i])))
        o = accs(3); ck.Sub(level, ga, accs(3), o)  # degree 1 - degree 2
        rs = O.Ring(rg.N, rg.q[: level + 1])
        cases.append((o, lambda i: np.stack([rs.binop("Sub", a[i][0], acc[i][0]), rs.binop("Sub", a[i][1], acc[i][1]),
                                            rs.unop("Neg", acc[i][2])])))
        o = accs(3); ck.Add(level, accs(3), gb, o)  # degree 2 + degree 1
        cases.append((o, lambda i: np.stack([rs.binop("Add", acc[i][0], b[i][0]), rs.binop("Add", acc[i][1], b[i][1]), acc[i][2]])))
        for n, (o, want) in enumerate(cases):
            got = Rig.down(o)
            for i in range(B):
                assert np.array_equal(got[i], want(i)), (level, n, i)


def test_ringqp_mirror(ctx):
    """ringqp.Ring (ring/ringqp/operations.go): Q-op then P-op, and ExtendBasisSmallNormAndCenter (:325) word for word,
    also on coefficients that violate its small-norm precondition."""
    from lattigo_amd.ringqp import RingQP
    rg = Rig(ctx, 10, [55, 45, 45], [55, 46], 5000)
    N, B = rg.N, 2
    qp = RingQP(rg.pr.gQ, rg.pr.gP, rg.gev).AtLevel(2, 1)
    q0 = int(rg.q[0])
    small = rg.rng.integers(-5, 6, size=(B, N))
    a = np.zeros((B, 3, N), dtype=np.uint64)
    for i, m in enumerate(rg.q):
        a[:, i] = np.where(small < 0, int(m) + small, small).astype(np.uint64)
    a[0, 0, :4] = [q0 >> 1, (q0 >> 1) + 1, q0 - 1, 1 << 50]  # edge / large values
    pin = la.Poly(rg.pr.gQ, 3, B).upload(a)
    outQ, outP = la.Poly(rg.pr.gQ, 3, B), la.Poly(rg.pr.gP, 2, B)
    qp.ExtendBasisSmallNormAndCenter(pin, 1, outQ, outP)
    for b in range(B):
        assert np.array_equal(outQ.download()[b], a[b])
        assert np.array_equal(outP.download()[b], OC.ExtendBasisSmallNormAndCenter(rg.pr.oQ, rg.pr.oP, a[b], 1)), b
    x = (la.Poly(rg.pr.gQ, 3, B).upload(rg.ct(2, B)[:, 0]), la.Poly(rg.pr.gP, 2, B).upload(
        np.stack([uniform_poly(rg.rng, rg.p, N) for _ in range(B)])))
    y, z = qp.NewPoly(B), qp.NewPoly(B)
    qp.NTT(x, y)
    qp.MulCoeffsMontgomery(x, y, z)
    qp.AutomorphismNTT(z, 5, y)
    xq, xp = x[0].download(), x[1].download()
    for b in range(B):
        wq = rg.pr.oQ.AutomorphismNTT(rg.pr.oQ.binop("MulCoeffsMontgomery", xq[b], rg.pr.oQ.NTT(xq[b])), 5)
        wp = rg.pr.oP.AutomorphismNTT(rg.pr.oP.binop("MulCoeffsMontgomery", xp[b], rg.pr.oP.NTT(xp[b])), 5)
        assert np.array_equal(y[0].download()[b], wq) and np.array_equal(y[1].download()[b], wp)


@pytest.mark.parametrize("deg", [1, 2, 3, 7, 8, 17, 33])
def test_bgv_polynomial_evaluation(ctx, deg):
    """circuits/bgv/polynomial Evaluator.Evaluate: the product driver on the device-resident bgv.Evaluator mirror vs the
    oracle's own restatement of the evaluator (oracle/polyeval_ref.py, written from the Go sources) on the oracle backend:
    the same primitive sequence with the same (level, scale, degree) after every call, and the same words; batch 2."""
    from drivers import polyeval as PE
    from drivers import schemes as S
    from oracle import polyeval_ref as PR
    rg = Rig(ctx, 10, [55, 45, 45, 45, 45, 45, 45, 45], [55, 55], 5100 + deg)
    rg.keys([1])
    t, B, top = 65537, 2, 7
    gbe = S.BGVCiphertextEvaluator(rg.gev, t, rg.ggks.keys[1])
    obe = OC.BGVCtEvaluator(rg.oev, t, rg.ogks[1])
    ct = rg.ct(top, B)
    coeffs = [int(x) for x in rg.rng.integers(0, t, size=deg + 1)]
    coeffs[-1] = coeffs[-1] or 1
    gct = S.Ciphertext(rg.up(ct), top, 3)
    gtr = PR.Trace(gbe)
    res = PE.PolynomialEvaluator(gtr).Evaluate(gct, coeffs, 11)
    got = np.stack([p.download() for p in res.Value], axis=1)  # [B][2][limbs][N]
    for b in range(B):
        otr = PR.Trace(obe)
        want = PR.evaluate_polynomial(otr, OC.Ct(list(ct[b]), 3), coeffs, 11)
        assert gtr.log == otr.log
        assert (res.Scale, res.level, res.Degree()) == (want.Scale, want.level, want.Degree()) == (11, top - deg.bit_length(), 1)
        assert np.array_equal(got[b][:, : res.level + 1], np.stack(want.Value)), (deg, b)
    assert np.array_equal(gct.Value[0].download(), ct[:, 0])  # the input ciphertext is left untouched


@pytest.mark.parametrize("deg,basis", [(1, "Monomial"), (7, "Monomial"), (12, "Monomial"), (5, "Chebyshev"), (31, "Chebyshev")])
def test_ckks_polynomial_evaluation(ctx, deg, basis):
    """circuits/ckks/polynomial Evaluator.Evaluate (complex coefficients, monomial / Chebyshev power bases, exact rational
    scale planning): the product driver on the device-resident ckks.Evaluator mirror vs oracle/polyeval_ref.py on the oracle
    backend: the same primitive sequence and level / scale schedule, bit-exact words, batch 2."""
    from fractions import Fraction
    from drivers import polyeval as PE
    from drivers import schemes as S
    from oracle import polyeval_ref as PR
    rg = Rig(ctx, 10, [55] + [45] * 7, [55, 55], 5200 + deg)
    rg.keys([1])
    B, top = 2, 7
    gce = S.CKKSCiphertextEvaluator(rg.gev, rg.ggks.keys[1])
    oce = OC.CKKSCtEvaluator(rg.oev, rg.ogks[1])
    ct = rg.ct(top, B)
    rr = rg.rng
    coeffs = [complex(a, b if basis == "Monomial" else 0.0) for a, b in zip(rr.uniform(-1, 1, size=deg + 1), rr.uniform(-1, 1, size=deg + 1))]
    scale = Fraction(1 << 45)
    pol = lambda: PE.Polynomial([PE._cpair(c) for c in coeffs], Basis=basis)
    gtr = PR.Trace(gce)
    res = PE.PolynomialEvaluator(gtr).Evaluate(S.Ciphertext(rg.up(ct), top, scale), pol(), scale)
    got = np.stack([p.download() for p in res.Value], axis=1)
    for b in range(B):
        otr = PR.Trace(oce)
        want = PR.evaluate_polynomial(otr, OC.Ct(list(ct[b]), scale), coeffs, scale, basis)
        assert gtr.log == otr.log
        assert (res.Scale, res.level, res.Degree()) == (want.Scale, want.level, want.Degree()) == (scale, top - deg.bit_length(), 1)
        assert np.array_equal(got[b][:, : res.level + 1], np.stack(want.Value)), (deg, basis, b)


@pytest.mark.parametrize("kind,K,deg,r", [("cos", 8, 30, 2), ("cos", 12, 40, 3), ("sin", 3, 31, 0), ("hanki", 16, 30, 3), ("asin", 8, 30, 1)])
def test_mod1(ctx, kind, K, deg, r):
    """circuits/ckks/mod1 Evaluator.EvaluateNew (EvalMod: even / odd Chebyshev polynomial incl. a degree-0 baby step,
    double-angle steps): the product driver on the device-resident ckks.Evaluator mirror vs the oracle's restatement
    (oracle/polyeval_ref.py evaluate_mod1, fed the same approximation coefficients) on the oracle backend: the same primitive
    sequence and level / scale schedule, bit-exact words, batch 2."""
    from fractions import Fraction
    from drivers import mod1 as M1
    from drivers import schemes as S
    from oracle import polyeval_ref as PR
    rg = Rig(ctx, 10, [55] + [45] * 10, [55, 55], 5300 + K)
    rg.keys([1])
    B, top = 2, 10
    gce = S.CKKSCiphertextEvaluator(rg.gev, rg.ggks.keys[1])
    oce = OC.CKKSCtEvaluator(rg.oev, rg.ogks[1])
    typ = {"cos": M1.CosContinuous, "sin": M1.SinContinuous, "hanki": M1.CosDiscrete, "asin": M1.CosContinuous}[kind]
    pm = M1.Mod1Parameters(int(rg.q[0]), LevelQ=top, LogScale=45, Mod1Type=typ, K=K, Mod1Degree=deg, DoubleAngle=r,
                           Mod1InvDegree=7 if kind == "asin" else 0)
    ct = rg.ct(top, B)
    scale = Fraction(1 << 45)
    gtr = PR.Trace(gce)
    res = M1.Mod1Evaluator(gtr, pm).EvaluateNew(S.Ciphertext(rg.up(ct), top, scale))
    got = np.stack([p.download() for p in res.Value], axis=1)
    for b in range(B):
        otr = PR.Trace(oce)
        want = PR.evaluate_mod1(otr, OC.Ct(list(ct[b]), scale), level_q=pm.LevelQ, log_scale=pm.LogDefaultScale,
                                cosine=typ != M1.SinContinuous, K=pm.K, double_angle=pm.DoubleAngle, sqrt2pi=pm.Sqrt2Pi,
                                poly_coeffs=pm.Mod1Poly.Coeffs, poly_even=pm.Mod1Poly.IsEven, poly_odd=pm.Mod1Poly.IsOdd,
                                inv_coeffs=None if pm.Mod1InvPoly is None else pm.Mod1InvPoly.Coeffs)
        assert gtr.log == otr.log
        assert (res.Scale, res.level, res.Degree()) == (want.Scale, want.level, want.Degree()) == (scale, top - pm.Depth(), 1)
        assert np.array_equal(got[b][:, : res.level + 1], np.stack(want.Value)), (kind, b)


def test_toy_bootstrapping_end_to_end(ctx):
    """bootstrapping.Evaluator.bootstrap (ModUp -> CoeffsToSlots -> EvalMod x2 -> SlotsToCoeffs, circuits/ckks/bootstrapping/
    evaluator.go:518-560) fully device-resident on the toy instance of tests/bootstrap_fixtures.py: every polynomial of the
    refreshed ciphertext equals the oracle-backed run bit for bit, and it decrypts to the input slots (BASELINE config 5's
    pipeline at toy size; batch 2)."""
    from drivers import bootstrapping as BS
    from drivers import mod1 as M1
    from drivers import schemes as S
    from tests.bootstrap_fixtures import ToyBootstrap
    rng = rng_for(5400)
    tb = ToyBootstrap(rng)
    N, B = tb.N, 2
    gQ, gP = la.Ring(ctx, N, tb.q), la.Ring(ctx, N, tb.p)
    gev = la.Evaluator(gQ, gP)
    ggks = R.GaloisKeySet({g: gev.NewEvaluationKey(k.q, k.p) for g, k in tb.gks.items()})
    grlk = gev.NewEvaluationKey(tb.rlk.q, tb.rlk.p)

    def up_lt(olt):
        vec = {k: (la.Poly(gQ, olt.LevelQ + 1).upload(v[0]), la.Poly(gP, len(tb.p)).upload(v[1])) for k, v in olt.Vec.items()}
        return LT.LinearTransformation(vec, olt.LevelQ, olt.LevelP, olt.slots, olt.N1)

    gce = S.CKKSCiphertextEvaluator(gev, grlk)
    be = BS.DeviceBootstrapBackend(gce, LT.LinTransEvaluator(gev, ggks), R.InnerSumEvaluator(gev, ggks))
    boot = BS.Bootstrapper(be, M1.Mod1Evaluator(gce, tb.mod1_params), [up_lt(m) for m in tb.cts], tb.cts_scale,
                           [up_lt(m) for m in tb.stc], tb.stc_scale)
    zs = [rng.uniform(-1, 1, size=N // 2) + 1j * rng.uniform(-1, 1, size=N // 2) for _ in range(B)]
    ct0 = np.stack([tb.encrypt_level0(rng, z) for z in zs])  # [B][2][1][N]
    gct = S.Ciphertext([la.Poly(gQ, 1, B).upload(ct0[:, k]) for k in range(2)], 0, 1)
    res = boot.Bootstrap(gct, tb.Se)
    got = np.stack([p.download() for p in res.Value], axis=1)
    oboot = tb.oracle_bootstrapper()
    for b in range(B):
        want = oboot.Bootstrap(OC.Ct(list(ct0[b]), 1), tb.Se)
        assert (res.level, res.Scale) == (want.level, want.Scale) and res.level >= 1
        assert np.array_equal(got[b][:, : res.level + 1], np.stack(want.Value)), b
        assert np.max(np.abs(tb.decode(want) - zs[b])) < 1e-5


@pytest.mark.gpu
def test_ckks_encoder_and_dft_factors(ctx):
    """drivers.dft: Encode is bit-identical to the oracle's NTT of the same rounded coefficients, Decode inverts it, and the
    factor lists multiply back to the special FFT (without its bit-reversal) and its inverse."""
    from fractions import Fraction
    from drivers import dft as DFT
    logN = 9
    N, n = 1 << logN, 1 << (logN - 1)
    q, p = O.GenModuli(logN + 1, [55, 45, 45], [56])
    gQ, gP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
    oQ, oP = O.Ring(N, q), O.Ring(N, p)
    rng = rng_for(7300)
    z = rng.uniform(-1, 1, size=n) + 1j * rng.uniform(-1, 1, size=n)
    enc = DFT.Encoder(gQ, gP)
    scale = Fraction(1 << 40)
    pt = enc.Encode(z, 2, scale)
    assert np.array_equal(pt.download()[0], oQ.NTT(DFT.fast_encode_rns(z, N, scale, q)))
    assert np.max(np.abs(enc.Decode(pt, scale) - z)) < 2.0 ** -30
    dq, dp = enc.EncodeQP(z, 1, scale)
    sub = O.Ring(N, q[:2])
    assert np.array_equal(dq.download()[0], sub.unop("MForm", sub.NTT(DFT.fast_encode_rns(z, N, scale, q[:2]))))
    assert np.array_equal(dp.download()[0], oP.unop("MForm", oP.NTT(DFT.fast_encode_rns(z, N, scale, p))))

    def apply(diags, v):
        return sum(d * np.roll(v, -k) for k, d in diags.items())

    w = rng.uniform(-1, 1, size=n) + 1j * rng.uniform(-1, 1, size=n)
    br = DFT.bitrev_indices(n)
    fwd = DFT.factor_diagonals(N, [(0, 4), (4, logN - 1)], False)
    v = w[br]
    for f in fwd:
        v = apply(f, v)
    assert np.max(np.abs(v - DFT.special_fft(w, N))) < 1e-9
    inv = DFT.factor_diagonals(N, [(4, logN - 1), (0, 4)], True)
    for f in inv:
        v = apply(f, v)
    assert np.max(np.abs(v[br] - w)) < 1e-9  # bitrev is an involution

"""GPU parity, basis extension and key-switch: libhering vs the
CPU oracle, bit-exact on every output limb, plus decrypt-and-check semantics
(core/rlwe/rlwe_test.go:666-779 style) on the GPU outputs."""
import os

import numpy as np
import pytest

import lattigo_amd as la
from lattigo_amd import rlwe as R
from oracle import oracle as O
from tests.conftest import Pi60, Qi60
from tests.gpu_common import Pair, ctx  # noqa: F401
from tests.helpers import prod, rand_bigints, rng_for, set_coefficients_bigint, uniform_poly
from tests.rlwe_fixtures import (SecretKey, automorphism_secret, gen_evaluation_key, gen_evaluation_key_base2,
                                 noise_log2, phase)

pytestmark = pytest.mark.gpu


def _uploadQ(pr, arr, batch=1):
    return la.Poly(pr.gQ, arr.shape[-2], batch).upload(arr)


def _uploadP(pr, arr, batch=1):
    return la.Poly(pr.gP, arr.shape[-2], batch).upload(arr)


@pytest.mark.parametrize("nq,np_", [(6, 3), (14, 14), (4, 1)])
def test_modup_moddown_word_exact(ctx, nq, np_):
    """ring/basis_extension.go ModUp/ModDown: exact words (incl. the lazy ModUp representative)."""
    logN = 10
    pr = Pair(ctx, logN, nq, np_, qmods=Qi60[-nq:], pmods=Pi60[-np_:])
    be, obe = la.BasisExtender(pr.gQ, pr.gP), O.BasisExtender(pr.oQ, pr.oP)
    rng = rng_for(2000 + nq)
    for levelQ, levelP in {(nq - 1, np_ - 1), (max(0, nq - 2), max(0, np_ - 2)), (0, 0)}:
        Qm, Pm = pr.q[: levelQ + 1], pr.p[: levelP + 1]
        xq, xp = uniform_poly(rng, Qm, pr.N), uniform_poly(rng, Pm, pr.N)
        pq, pp = _uploadQ(pr, xq), _uploadP(pr, xp)
        out = la.Poly(pr.gP, levelP + 1)
        be.ModUpQtoP(levelQ, levelP, pq, out)
        assert np.array_equal(out.get(), obe.ModUpQtoP(levelQ, levelP, xq)), ("QtoP", levelQ, levelP)
        out = la.Poly(pr.gQ, levelQ + 1)
        be.ModUpPtoQ(levelP, levelQ, pp, out)
        assert np.array_equal(out.get(), obe.ModUpPtoQ(levelP, levelQ, xp)), ("PtoQ", levelQ, levelP)
        for name, ring_, nl in (("ModDownQPtoQ", pr.gQ, levelQ + 1), ("ModDownQPtoQNTT", pr.gQ, levelQ + 1),
                                ("ModDownQPtoP", pr.gP, levelP + 1)):
            out = la.Poly(ring_, nl)
            getattr(be, name)(levelQ, levelP, pq, pp, out)
            assert np.array_equal(out.get(), getattr(obe, name)(levelQ, levelP, xq, xp)), (name, levelQ, levelP)


def test_modup_against_bigint_and_adversarial_float(ctx):
    """ModUp of centred big integers (ring/ring_test.go:714) plus inputs engineered so that the
    float64 sum v sits next to an integer boundary (coefficients 0, +-1, Q/2 and neighbours)."""
    pr = Pair(ctx, 10, 6, 4, qmods=Qi60[-6:], pmods=Pi60[-4:])
    be, obe = la.BasisExtender(pr.gQ, pr.gP), O.BasisExtender(pr.oQ, pr.oP)
    Q = prod(pr.q)
    half = Q >> 1
    coeffs = [c - half for c in rand_bigints(rng_for(2100), Q, pr.N)]
    special = [0, 1, -1, half, -half, half - 1, -half + 1, 2, -2, Q // 3, -(Q // 3)]
    for k in range(6):  # exact multiples of Q/q_k +- 1: the y_i/q_i sum is within 1 ulp of an integer
        special += [(Q // pr.q[k]) * j + d for j in (1, 2, pr.q[k] // 2) for d in (-1, 0, 1)]
    for i, s in enumerate(special):
        s = ((s + half) % Q) - half
        coeffs[i] = s
    xq = set_coefficients_bigint(coeffs, pr.q)
    out = la.Poly(pr.gP, 4)
    be.ModUpQtoP(5, 3, _uploadQ(pr, xq), out)
    got = out.get()
    assert np.array_equal(got, obe.ModUpQtoP(5, 3, xq))  # word-exact, adversarial columns included
    # big-integer ground truth on the random columns only: next to +-Q/2 the reference's own
    # float64 method (HPS) is allowed to be off by Q, and the GPU must reproduce exactly that.
    ns = len(special)
    assert np.array_equal(pr.oP.unop("Reduce", got)[:, ns:], set_coefficients_bigint(coeffs, pr.p)[:, ns:])


def _setup(ctx, logN, nq, np_, seed):
    pr = Pair(ctx, logN, nq, np_)
    rng = rng_for(seed)
    oev, gev = O.Evaluator(pr.oQ, pr.oP), la.Evaluator(pr.gQ, pr.gP)
    sk = SecretKey(rng, pr.oQ, pr.oP)
    return pr, rng, oev, gev, sk


@pytest.mark.parametrize("nq,np_", [(7, 3), (6, 2), (5, 1), (8, 4)])
def test_decompose_and_gadget_product(ctx, nq, np_):
    pr, rng, oev, gev, sk = _setup(ctx, 10, nq, np_, 2200 + nq)
    sk2 = SecretKey(rng, pr.oQ, pr.oP)
    oevk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, sk2)
    gevk = gev.NewEvaluationKey(oevk.q, oevk.p)
    odec = O.Decomposer(pr.oQ, pr.oP)
    levelP = np_ - 1
    for levelQ in (nq - 1, nq - 2, max(0, nq - 4), 0):
        Qm = pr.q[: levelQ + 1]
        cx = uniform_poly(rng, Qm, pr.N)
        pcx = _uploadQ(pr, cx)
        beta = O.BaseRNSDecompositionVectorSize(levelQ, levelP)
        # DecomposeAndSplit on the coefficient-domain input, every digit
        for d in range(beta):
            p1Q, p1P = la.Poly(pr.gQ, levelQ + 1), la.Poly(pr.gP, levelP + 1)
            gev.DecomposeAndSplit(levelQ, levelP, levelP + 1, d, pcx, p1Q, p1P)
            wQ, wP = odec.DecomposeAndSplit(levelQ, levelP, levelP + 1, d, cx)
            lo, hi = d * np_, min(d * np_ + np_, levelQ + 1)
            own = set(range(lo, hi)) if hi - lo > 1 else set()
            gQ = p1Q.get()
            for l in range(levelQ + 1):
                if l not in own:
                    assert np.array_equal(gQ[l], wQ[l]), (levelQ, d, l)
            assert np.array_equal(p1P.get(), wP), (levelQ, d)
        # DecomposeNTT (hoisting buffer)
        dq, dp = oev.DecomposeNTT(levelQ, levelP, levelP + 1, cx, True)
        gdec = la.Decomposition(gev)
        if beta <= O.BaseRNSDecompositionVectorSize(nq - 1, levelP):
            gev.DecomposeNTT(levelQ, levelP, levelP + 1, pcx, True, gdec)
            for d in range(beta):
                for l in range(levelQ + 1):
                    assert np.array_equal(gdec.limb(0, d, False, l), dq[d, l]), (levelQ, d, l)
                for l in range(levelP + 1):
                    assert np.array_equal(gdec.limb(0, d, True, l), dp[d, l]), (levelQ, d, l)
        # GadgetProductLazy / Hoisted / ModDown / GadgetProduct
        wQ, wP = oev.GadgetProductLazy(levelQ, cx, oevk)
        qp = [(la.Poly(pr.gQ, levelQ + 1), la.Poly(pr.gP, levelP + 1)) for _ in range(2)]
        gev.GadgetProductLazy(levelQ, pcx, gevk, qp)
        for k in range(2):
            assert np.array_equal(qp[k][0].get(), wQ[k]) and np.array_equal(qp[k][1].get(), wP[k]), (levelQ, k)
        if beta <= O.BaseRNSDecompositionVectorSize(nq - 1, levelP):
            qp2 = [(la.Poly(pr.gQ, levelQ + 1), la.Poly(pr.gP, levelP + 1)) for _ in range(2)]
            gev.GadgetProductHoistedLazy(levelQ, gdec, gevk, qp2)
            for k in range(2):
                assert np.array_equal(qp2[k][0].get(), wQ[k]) and np.array_equal(qp2[k][1].get(), wP[k])
        want = oev.GadgetProduct(levelQ, cx, oevk)
        ct = [la.Poly(pr.gQ, levelQ + 1), la.Poly(pr.gQ, levelQ + 1)]
        gev.ModDown(levelQ, levelP, qp, ct)
        assert np.array_equal(np.stack([c.get() for c in ct]), want)
        ct2 = [la.Poly(pr.gQ, levelQ + 1), la.Poly(pr.gQ, levelQ + 1)]
        gev.GadgetProduct(levelQ, pcx, gevk, ct2)
        got = np.stack([c.get() for c in ct2])
        assert np.array_equal(got, want)
        # semantic: <ct,(1,sk2)> = cx*sk + small
        sub = O.Ring(pr.N, Qm)
        noise = noise_log2(pr.oQ, sub.binop("Sub", phase(pr.oQ, got, sk2.Q),
                                            sub.binop("MulCoeffsMontgomery", cx, sk.Q[: levelQ + 1])))
        assert noise <= 16, noise


def test_batched_gadget_product_matches_single(ctx):
    pr, rng, oev, gev, sk = _setup(ctx, 11, 6, 3, 2300)
    oevk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, sk)
    gevk = gev.NewEvaluationKey(oevk.q, oevk.p)
    for B in (2, 3, 5):
        cx = np.stack([uniform_poly(rng, pr.q, pr.N) for _ in range(B)])
        pcx = la.Poly(pr.gQ, 6, B).upload(cx)
        ct = [la.Poly(pr.gQ, 6, B), la.Poly(pr.gQ, 6, B)]
        gev.GadgetProduct(5, pcx, gevk, ct)
        g0, g1 = ct[0].get(), ct[1].get()
        for b in range(B):
            want = oev.GadgetProduct(5, cx[b], oevk)
            assert np.array_equal(g0[b], want[0]) and np.array_equal(g1[b], want[1]), (B, b)


@pytest.mark.parametrize("scheme", ["ckks", "bgv"])
@pytest.mark.parametrize("nq,np_", [(5, 2), (4, 1)])
def test_mul_relin_rescale(ctx, scheme, nq, np_):
    pr, rng, oev, gev, sk = _setup(ctx, 11, nq, np_, 2400 + nq)
    sk_sq = pr.oQ.binop("MulCoeffsMontgomery", sk.Q, sk.Q)
    orlk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk_sq, sk)
    grlk = gev.NewEvaluationKey(orlk.q, orlk.p)
    level, t = nq - 1, 65537
    ct0 = np.stack([uniform_poly(rng, pr.q, pr.N) for _ in range(2)])
    ct1 = np.stack([uniform_poly(rng, pr.q, pr.N) for _ in range(2)])
    a, b = [_uploadQ(pr, c) for c in ct0], [_uploadQ(pr, c) for c in ct1]
    omul = (lambda r, k: oev.CKKSMulRelin(ct0, ct1, k, r)) if scheme == "ckks" else (lambda r, k: oev.BGVMulRelin(t, ct0, ct1, k, r))
    gmul = (lambda k, o: gev.CKKSMulRelin(level, a, b, k, o)) if scheme == "ckks" else (lambda k, o: gev.BGVMulRelin(level, t, a, b, k, o))
    out3 = [pr.gQ.NewPoly() for _ in range(3)]
    gmul(None, out3)
    deg2 = np.stack([o.get() for o in out3])
    assert np.array_equal(deg2, omul(False, None))
    out2 = [pr.gQ.NewPoly() for _ in range(2)]
    gmul(grlk, out2)
    rel = np.stack([o.get() for o in out2])
    assert np.array_equal(rel, omul(True, orlk))
    # Relinearize of the degree-2 result gives the same ciphertext
    out2b = [pr.gQ.NewPoly() for _ in range(2)]
    gev.Relinearize(level, out3, grlk, out2b)
    assert np.array_equal(np.stack([o.get() for o in out2b]), rel)
    # Rescale (1 and 2 levels), also in place
    for nb in (1, 2):
        res = [la.Poly(pr.gQ, nq - nb) for _ in range(2)]
        gev.Rescale(level, nb, out2, res)
        assert np.array_equal(np.stack([r.get() for r in res]), oev.Rescale(rel, nb))
    # semantic check: decrypts to the product of the phases (times t for BGV)
    p0, p1 = phase(pr.oQ, ct0, sk.Q), phase(pr.oQ, ct1, sk.Q)
    want = pr.oQ.binop("MulCoeffsBarrett", p0, p1)
    if scheme == "bgv":
        want = pr.oQ.scalarop("MulScalar", want, t)
    assert np.array_equal(phase(pr.oQ, deg2, sk.Q), want)
    assert noise_log2(pr.oQ, pr.oQ.binop("Sub", phase(pr.oQ, rel, sk.Q), want)) <= 18


@pytest.mark.parametrize("nq,np_", [(5, 2), (4, 1)])
def test_rotate(ctx, nq, np_):
    pr, rng, oev, gev, sk = _setup(ctx, 11, nq, np_, 2500 + nq)
    N = pr.N
    level = nq - 1
    ct = np.stack([uniform_poly(rng, pr.q, N) for _ in range(2)])
    pct = [_uploadQ(pr, c) for c in ct]
    gdec = la.Decomposition(gev)
    gev.DecomposeNTT(level, np_ - 1, np_, pct[1], True, gdec)
    for galel in (5, 2 * N - 1, pow(5, 9, 2 * N)):
        sk_out = automorphism_secret(rng, pr.oQ, pr.oP, sk, pow(galel, 2 * N - 1, 2 * N))
        ogk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, sk_out)
        ggk = gev.NewEvaluationKey(ogk.q, ogk.p)
        want = oev.Automorphism(ct, galel, ogk)
        out = [pr.gQ.NewPoly(), pr.gQ.NewPoly()]
        gev.Automorphism(level, pct, galel, ggk, out)
        got = np.stack([o.get() for o in out])
        assert np.array_equal(got, want), galel
        out2 = [pr.gQ.NewPoly(), pr.gQ.NewPoly()]
        gev.AutomorphismHoisted(level, pct, gdec, galel, ggk, out2)
        assert np.array_equal(np.stack([o.get() for o in out2]), want), galel
        # EvaluatorProvider.AutomorphismHoistedLazy: QP output, not divided by P
        dq, dp = oev.DecomposeNTT(level, np_ - 1, np_, ct[1], True)
        wQ, wP = oev.AutomorphismHoistedLazy(level, ct[0], dq, dp, galel, ogk)
        qp = [(pr.gQ.NewPoly(), pr.gP.NewPoly()) for _ in range(2)]
        gev.AutomorphismHoistedLazy(level, pct, gdec, galel, ggk, qp)
        for k in range(2):
            assert np.array_equal(qp[k][0].get(), wQ[k]) and np.array_equal(qp[k][1].get(), wP[k]), (galel, k)
        idx = pr.oQ.AutomorphismNTTIndex(galel)
        wantp = pr.oQ.AutomorphismNTTWithIndex(phase(pr.oQ, ct, sk.Q), idx)
        assert noise_log2(pr.oQ, pr.oQ.binop("Sub", phase(pr.oQ, got, sk.Q), wantp)) <= 18


def test_full_size_properties_logN15(ctx):
    """BASELINE config 3 shape (logN=15, 12 Q-limbs, 3 P-limbs): hoisted == plain, batch == single,
    linearity of the gadget product in cx, and every limb of both outputs checked against the oracle."""
    logN, nq, np_ = 15, 12, 3
    q, p = O.GenModuli(logN + 1, [55] + [45] * 11, [55] * 3)
    pr = Pair(ctx, logN, nq, np_, qmods=q, pmods=p)
    rng = rng_for(3)
    gev = la.Evaluator(pr.gQ, pr.gP)
    kq = np.stack([np.stack([uniform_poly(rng, q, pr.N) for _ in range(2)]) for _ in range(4)])
    kp = np.stack([np.stack([uniform_poly(rng, p, pr.N) for _ in range(2)]) for _ in range(4)])
    gevk = gev.NewEvaluationKey(kq, kp)
    cx = np.stack([uniform_poly(rng, q, pr.N) for _ in range(2)])
    pcx = la.Poly(pr.gQ, nq, 2).upload(cx)
    ct = [la.Poly(pr.gQ, nq, 2), la.Poly(pr.gQ, nq, 2)]
    gev.GadgetProduct(nq - 1, pcx, gevk, ct)
    g0 = ct[0].get()
    # single == batched
    p1 = la.Poly(pr.gQ, nq).upload(cx[1])
    ct1 = [pr.gQ.NewPoly(), pr.gQ.NewPoly()]
    gev.GadgetProduct(nq - 1, p1, gevk, ct1)
    assert np.array_equal(ct1[0].get(), g0[1])
    # hoisted == plain
    dec = la.Decomposition(gev)
    gev.DecomposeNTT(nq - 1, np_ - 1, np_, p1, True, dec)
    ct2 = [pr.gQ.NewPoly(), pr.gQ.NewPoly()]
    gev.GadgetProductHoisted(nq - 1, dec, gevk, ct2)
    assert np.array_equal(ct2[0].get(), g0[1]) and np.array_equal(ct2[1].get(), ct1[1].get())
    # oracle on the whole op (seconds at this size)
    oev = O.Evaluator(pr.oQ, pr.oP)
    want = oev.GadgetProduct(nq - 1, cx[1], O.EvaluationKey(kq, kp))
    assert np.array_equal(g0[1], want[0]) and np.array_equal(ct1[1].get(), want[1])


def _full_size_check(ctx, logN, logq, logp, seed, do_rotate):
    """Full-size config check: limb-complete oracle comparison of GadgetProduct (+ Rotate) for every batch entry on a
    random key, batch of 2, plus hoisted == plain."""
    q, p = O.GenModuli(logN + 1, logq, logp)
    nq, np_ = len(q), len(p)
    pr = Pair(ctx, logN, nq, np_, qmods=q, pmods=p)
    rng = rng_for(seed)
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    beta = O.BaseRNSDecompositionVectorSize(nq - 1, np_ - 1)
    kq = np.stack([np.stack([uniform_poly(rng, q, pr.N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(rng, p, pr.N) for _ in range(2)]) for _ in range(beta)])
    gevk, oevk = gev.NewEvaluationKey(kq, kp), O.EvaluationKey(kq, kp)
    for level in (nq - 1, nq - 2):
        Qm = q[: level + 1]
        ct = np.stack([np.stack([uniform_poly(rng, Qm, pr.N) for _ in range(2)]) for _ in range(2)])  # [b][k]
        pc = [la.Poly(pr.gQ, level + 1, 2).upload(ct[:, k]) for k in range(2)]
        out = [la.Poly(pr.gQ, level + 1, 2), la.Poly(pr.gQ, level + 1, 2)]
        gev.GadgetProduct(level, pc[1], gevk, out)
        g = [o.get() for o in out]
        for b in range(2):  # every batch entry, every limb
            want = oev.GadgetProduct(level, ct[b, 1], oevk)
            assert np.array_equal(g[0][b], want[0]) and np.array_equal(g[1][b], want[1]), ("GadgetProduct", level, b)
        if do_rotate:
            galel = pow(5, 3, 2 * pr.N)
            out2 = [la.Poly(pr.gQ, level + 1, 2), la.Poly(pr.gQ, level + 1, 2)]
            gev.Automorphism(level, pc, galel, gevk, out2)
            for b in range(2):
                wantr = oev.Automorphism(ct[b], galel, oevk)
                assert np.array_equal(out2[0].get()[b], wantr[0]) and np.array_equal(out2[1].get()[b], wantr[1]), ("Rotate", level, b)
            dec = la.Decomposition(gev, 2)
            gev.DecomposeNTT(level, np_ - 1, np_, pc[1], True, dec)
            out3 = [la.Poly(pr.gQ, level + 1, 2), la.Poly(pr.gQ, level + 1, 2)]
            gev.AutomorphismHoisted(level, pc, dec, galel, gevk, out3)
            assert np.array_equal(out3[0].get(), out2[0].get()) and np.array_equal(out3[1].get(), out2[1].get())
        # MulRelin (CKKS tensor) + Rescale at this level
        ct2 = np.stack([uniform_poly(rng, Qm, pr.N) for _ in range(2)])
        a = [la.Poly(pr.gQ, level + 1).upload(c) for c in ct[0]]
        b = [la.Poly(pr.gQ, level + 1).upload(c) for c in ct2]
        o2 = [la.Poly(pr.gQ, level + 1), la.Poly(pr.gQ, level + 1)]
        gev.CKKSMulRelin(level, a, b, gevk, o2)
        wantm = oev.CKKSMulRelin(ct[0], ct2, oevk, True)
        assert np.array_equal(np.stack([o.get() for o in o2]), wantm), ("MulRelin", level)
        if level == 0:
            continue  # (nothing to rescale into)
        res = [la.Poly(pr.gQ, level), la.Poly(pr.gQ, level)]
        gev.Rescale(level, 1, o2, res)
        assert np.array_equal(np.stack([r.get() for r in res]), oev.Rescale(wantm, 1)), ("Rescale", level)


def test_full_size_config2_ckks_logN14(ctx):
    """BASELINE config 2: CKKS LogN=14, LogQ=[50,40x7], LogP=[60] (alpha = 1 path)."""
    _full_size_check(ctx, 14, [50] + [40] * 7, [60], 2, True)


def test_full_size_config4_ckks_logN16(ctx):
    """BASELINE config 4: CKKS LogN=16, LogQ=[60,45x19], LogP=[61x4] (alpha = 4, beta = 5): Rotate."""
    _full_size_check(ctx, 16, [60] + [45] * 19, [61] * 4, 4, True)


def test_full_size_config5_shape_logN16(ctx):
    """Bootstrapping-sized chain (logN=16, 25 Q-limbs, 5 P-limbs of 61 bits, alpha = 5)."""
    _full_size_check(ctx, 16, [60] + [45] * 10 + [60] * 6 + [40] * 8, [61] * 5, 5, False)


@pytest.mark.parametrize("logN", [18, 19])
def test_key_switch_beyond_the_fused_pipelines(ctx, logN):
    """logN > 17 (the reference accepts up to MaxLogN = 20): no fused basis extension / NTT + MAC there -- GadgetProduct, Rotate,
    the hoisted forms, MulRelin and Rescale run through the generic passes and must give the same words as everywhere else."""
    _full_size_check(ctx, logN, [55, 45, 45, 58], [61, 55], 18, True)


@pytest.mark.parametrize("pw2", [12, 20, 31])
def test_base2_gadget_product(ctx, pw2):
    """gadgetProductSinglePAndBitDecompLazy with BaseTwoDecomposition != 0 (core/rlwe/evaluator_gadget_product.go:203-338)."""
    pr, rng, oev, gev, sk = _setup(ctx, 11, 4, 1, 2600 + pw2)
    sk2 = SecretKey(rng, pr.oQ, pr.oP)
    oevk = gen_evaluation_key_base2(rng, pr.oQ, pr.oP, sk.Q, sk2, pw2)
    gevk = gev.NewEvaluationKey(oevk.q, oevk.p, pw2, oevk.nj[:4])
    for levelQ in (3, 2, 0):
        Qm = pr.q[: levelQ + 1]
        cx = np.stack([uniform_poly(rng, Qm, pr.N) for _ in range(2)])
        pcx = la.Poly(pr.gQ, levelQ + 1, 2).upload(cx)
        qp = [(la.Poly(pr.gQ, levelQ + 1, 2), la.Poly(pr.gP, 1, 2)) for _ in range(2)]
        gev.GadgetProductLazy(levelQ, pcx, gevk, qp)
        wQ, wP = oev.GadgetProductLazy(levelQ, cx[1], oevk)
        for k in range(2):
            assert np.array_equal(qp[k][0].get()[1], wQ[k]) and np.array_equal(qp[k][1].get()[1], wP[k]), (levelQ, k)
        ct = [la.Poly(pr.gQ, levelQ + 1, 2), la.Poly(pr.gQ, levelQ + 1, 2)]
        gev.GadgetProduct(levelQ, pcx, gevk, ct)
        want = oev.GadgetProduct(levelQ, cx[0], oevk)
        got = np.stack([c.get()[0] for c in ct])
        assert np.array_equal(got, want), levelQ
        sub = O.Ring(pr.N, Qm)
        noise = noise_log2(pr.oQ, sub.binop("Sub", phase(pr.oQ, got, sk2.Q), sub.binop("MulCoeffsMontgomery", cx[0], sk.Q[: levelQ + 1])))
        assert noise <= 11 + pw2 + 6, noise
    with pytest.raises(la.HeringError):  # hoisted forms reject base-2 keys, as the reference (:381-383)
        gev.GadgetProductHoisted(3, la.Decomposition(gev, 2), gevk, ct)


@pytest.mark.parametrize("logN,np_", [(12, 6), (13, 7), (14, 8), (13, 2), (13, 5)])
def test_fused_pipeline_wide_digits_and_logN13(ctx, logN, np_):
    """The fused key-switch pipeline at the shapes round 1 sent down the unfused path: digits of 6, 7 and 8 limbs (the
    reference allows up to 32 source limbs, ring/basis_extension.go:285) and logN = 13 (one fused column stage); moduli of
    all three arithmetic classes; full and partial trailing digit; GadgetProduct, hoisted form and MulRelin against the oracle."""
    logq = ([55, 45, 61, 40, 58, 36] * 4)[: 2 * np_ + 1]
    logp = ([61, 46, 55, 60, 40, 58, 45, 61])[:np_]
    q, p = O.GenModuli(logN + 1, logq, logp)
    pr = Pair(ctx, logN, len(q), len(p), qmods=q, pmods=p)
    rng = rng_for(2800 + logN * 10 + np_)
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    beta = O.BaseRNSDecompositionVectorSize(len(q) - 1, np_ - 1)
    kq = np.stack([np.stack([uniform_poly(rng, q, pr.N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(rng, p, pr.N) for _ in range(2)]) for _ in range(beta)])
    gevk, oevk = gev.NewEvaluationKey(kq, kp), O.EvaluationKey(kq, kp)
    for level in (len(q) - 1, len(q) - 2, np_ - 1):
        Qm = q[: level + 1]
        cx = np.stack([uniform_poly(rng, Qm, pr.N) for _ in range(2)])
        pcx = la.Poly(pr.gQ, level + 1, 2).upload(cx)
        out = [la.Poly(pr.gQ, level + 1, 2), la.Poly(pr.gQ, level + 1, 2)]
        gev.GadgetProduct(level, pcx, gevk, out)
        for b in range(2):
            want = oev.GadgetProduct(level, cx[b], oevk)
            assert np.array_equal(out[0].get()[b], want[0]) and np.array_equal(out[1].get()[b], want[1]), (level, b)
        dec = la.Decomposition(gev, 2)
        gev.DecomposeNTT(level, np_ - 1, np_, pcx, True, dec)
        dq, dp = oev.DecomposeNTT(level, np_ - 1, np_, cx[1], True)
        for d in range(dq.shape[0]):
            for l in range(level + 1):
                assert np.array_equal(dec.limb(1, d, False, l), dq[d, l]), (level, d, l)
            for l in range(np_):
                assert np.array_equal(dec.limb(1, d, True, l), dp[d, l]), (level, d, l)
        out2 = [la.Poly(pr.gQ, level + 1, 2), la.Poly(pr.gQ, level + 1, 2)]
        gev.GadgetProductHoisted(level, dec, gevk, out2)
        assert np.array_equal(out2[0].get(), out[0].get()) and np.array_equal(out2[1].get(), out[1].get())
    level = len(q) - 1
    ct0 = np.stack([uniform_poly(rng, q, pr.N) for _ in range(2)])
    ct1 = np.stack([uniform_poly(rng, q, pr.N) for _ in range(2)])
    a, b = [pr.gQ.NewPoly().upload(c) for c in ct0], [pr.gQ.NewPoly().upload(c) for c in ct1]
    o2 = [pr.gQ.NewPoly(), pr.gQ.NewPoly()]
    gev.CKKSMulRelin(level, a, b, gevk, o2)
    assert np.array_equal(np.stack([o.get() for o in o2]), oev.CKKSMulRelin(ct0, ct1, oevk, True))


@pytest.mark.parametrize("logN", [10, 13])
def test_conjugate_invariant_basis_extender_and_evaluator(ctx, logN):
    """RingType = ConjugateInvariant (Z[X + X^-1]/(X^2N + 1), NthRoot = 4N) through ring.BasisExtender and rlwe.Evaluator:
    every ModUp / ModDown form, DecomposeNTT, GadgetProduct plain and hoisted, rotations by 5^k (plain, hoisted, hoisted-lazy,
    coefficient-domain), the automorphism index table, CKKS MulRelin + Rescale and Trace, bit for bit against the oracle
    (ring/ntt.go:716-1311, ring/automorphism.go:12-34,:122-151, core/rlwe/inner_sum.go:60-62)."""
    q, p = O.GenModuli(logN + 2, [55, 45, 58, 40, 61], [55, 46])   # = 1 mod 4N; every arithmetic class
    pr = Pair(ctx, logN, len(q), len(p), qmods=q, pmods=p, ci=True)
    N, nq, np_ = pr.N, len(q), len(p)
    rng = rng_for(2900 + logN)
    assert pr.gQ.NthRoot() == 4 * N
    obe, gbe = O.BasisExtender(pr.oQ, pr.oP), la.BasisExtender(pr.gQ, pr.gP)
    for levelQ, levelP in ((nq - 1, np_ - 1), (2, 0)):
        xq, xp = uniform_poly(rng, q[: levelQ + 1], N), uniform_poly(rng, p[: levelP + 1], N)
        pq, pp = la.Poly(pr.gQ, levelQ + 1).upload(xq), la.Poly(pr.gP, levelP + 1).upload(xp)
        oq, op = la.Poly(pr.gQ, levelQ + 1), la.Poly(pr.gP, levelP + 1)
        gbe.ModUpQtoP(levelQ, levelP, pq, op)
        assert np.array_equal(op.get(), obe.ModUpQtoP(levelQ, levelP, xq))
        gbe.ModUpPtoQ(levelP, levelQ, pp, oq)
        assert np.array_equal(oq.get(), obe.ModUpPtoQ(levelP, levelQ, xp))
        gbe.ModDownQPtoQ(levelQ, levelP, pq, pp, oq)
        assert np.array_equal(oq.get(), obe.ModDownQPtoQ(levelQ, levelP, xq, xp))
        gbe.ModDownQPtoQNTT(levelQ, levelP, pq, pp, oq)
        assert np.array_equal(oq.get(), obe.ModDownQPtoQNTT(levelQ, levelP, xq, xp))
        gbe.ModDownQPtoP(levelQ, levelP, pq, pp, op)
        assert np.array_equal(op.get(), obe.ModDownQPtoP(levelQ, levelP, xq, xp))
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    sk = SecretKey(rng, pr.oQ, pr.oP)
    level = nq - 1
    ct = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(2)])   # [b][k]
    pct = [la.Poly(pr.gQ, nq, 2).upload(ct[:, k]) for k in range(2)]
    okey = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, SecretKey(rng, pr.oQ, pr.oP))
    gkey = gev.NewEvaluationKey(okey.q, okey.p)
    out = [la.Poly(pr.gQ, nq, 2), la.Poly(pr.gQ, nq, 2)]
    gev.GadgetProduct(level, pct[1], gkey, out)
    for b in range(2):
        want = oev.GadgetProduct(level, ct[b, 1], okey)
        assert np.array_equal(out[0].get()[b], want[0]) and np.array_equal(out[1].get()[b], want[1]), b
    dec = la.Decomposition(gev, 2)
    gev.DecomposeNTT(level, np_ - 1, np_, pct[1], True, dec)
    dq, dp = oev.DecomposeNTT(level, np_ - 1, np_, ct[1, 1], True)
    for d in range(dq.shape[0]):
        for l in range(nq):
            assert np.array_equal(dec.limb(1, d, False, l), dq[d, l]), (d, l)
        for l in range(np_):
            assert np.array_equal(dec.limb(1, d, True, l), dp[d, l]), (d, l)
    out2 = [la.Poly(pr.gQ, nq, 2), la.Poly(pr.gQ, nq, 2)]
    gev.GadgetProductHoisted(level, dec, gkey, out2)
    assert np.array_equal(out2[0].get(), out[0].get()) and np.array_equal(out2[1].get(), out[1].get())
    for k in (1, 5):
        galel = pow(5, k, 4 * N)
        # the index table itself (ring.AutomorphismNTTIndex with NthRoot = 4N)
        assert np.array_equal(pr.gQ.AutomorphismNTTIndex(galel).download(), pr.oQ.AutomorphismNTTIndex(galel))
        ogk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, automorphism_secret(rng, pr.oQ, pr.oP, sk, pow(galel, 4 * N - 1, 4 * N)))
        ggk = gev.NewEvaluationKey(ogk.q, ogk.p)
        gev.Automorphism(level, pct, galel, ggk, out)
        want = oev.Automorphism(ct[0], galel, ogk)
        got = np.stack([o.get()[0] for o in out])
        assert np.array_equal(got, want), k
        idx = pr.oQ.AutomorphismNTTIndex(galel)
        wantp = pr.oQ.AutomorphismNTTWithIndex(phase(pr.oQ, ct[0], sk.Q), idx)
        assert noise_log2(pr.oQ, pr.oQ.binop("Sub", phase(pr.oQ, got, sk.Q), wantp)) <= logN + 7
        gev.AutomorphismHoisted(level, pct, dec, galel, ggk, out2)
        assert np.array_equal(out2[0].get(), out[0].get()) and np.array_equal(out2[1].get(), out[1].get())
        wQ, wP = oev.AutomorphismHoistedLazy(level, ct[1, 0], dq, dp, galel, ogk)
        qp = [(la.Poly(pr.gQ, nq, 2), la.Poly(pr.gP, np_, 2)) for _ in range(2)]
        gev.AutomorphismHoistedLazy(level, pct, dec, galel, ggk, qp)
        for c in range(2):
            assert np.array_equal(qp[c][0].get()[1], wQ[c]) and np.array_equal(qp[c][1].get()[1], wP[c]), (k, c)
        # coefficient domain (ring.Automorphism on Z[X + X^-1])
        x = uniform_poly(rng, q, N)
        po = pr.gQ.NewPoly()
        pr.gQ.Automorphism(pr.gQ.NewPoly().upload(x), galel, po)
        assert np.array_equal(po.get(), pr.oQ.Automorphism(x, galel)), k
    o2 = [la.Poly(pr.gQ, nq), la.Poly(pr.gQ, nq)]
    a, b = [la.Poly(pr.gQ, nq).upload(c) for c in ct[0]], [la.Poly(pr.gQ, nq).upload(c) for c in ct[1]]
    gev.CKKSMulRelin(level, a, b, gkey, o2)
    wantm = oev.CKKSMulRelin(ct[0], ct[1], okey, True)
    assert np.array_equal(np.stack([o.get() for o in o2]), wantm)
    res = [la.Poly(pr.gQ, nq - 1), la.Poly(pr.gQ, nq - 1)]
    gev.Rescale(level, 1, o2, res)
    assert np.array_equal(np.stack([r.get() for r in res]), oev.Rescale(wantm, 1))
    # Trace: on this ring the last step (phi(5^-1)) and the conjugation do not exist
    from oracle import circuits as OC
    nth = 4 * N
    gks, ogks = R.GaloisKeySet(), {}
    for i in range(0, logN - 1):
        g = R.GaloisElement(nth, 1 << i)
        ogks[g] = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, automorphism_secret(rng, pr.oQ, pr.oP, sk, pow(g, nth - 1, nth)))
        gks.keys[g] = gev.NewEvaluationKey(ogks[g].q, ogks[g].p)
    gi, oi = R.InnerSumEvaluator(gev, gks), OC.InnerSumEvaluator(oev, ogks)
    for logn in (0, logN - 3):
        tout = [la.Poly(pr.gQ, nq), la.Poly(pr.gQ, nq)]
        gi.Trace(level, [la.Poly(pr.gQ, nq).upload(c) for c in ct[0]], logn, tout)
        assert np.array_equal(np.stack([o.get() for o in tout]), oi.Trace(ct[0], logn)), logn


@pytest.mark.parametrize("pw2", [2, 16])
def test_gadget_product_without_special_primes(ctx, pw2):
    """levelP = -1 (rlwe.ParametersLiteral.P = nil; the reference's P-less test set, core/rlwe/test_params.go:36-46):
    GadgetProductLazy / GadgetProduct / ModDown / Relinearize / Automorphism with a base-2 key that has no P part, bit for bit
    against the oracle (core/rlwe/evaluator_gadget_product.go:74-96,:203-338 with ringP == nil)."""
    q = [0x200000440001, 0x7fff80001, 0x800280001, 0x7ffd80001, 0x7ffc80001]
    logN = 10
    pr = Pair(ctx, logN, len(q), qmods=q)
    rng = rng_for(2700 + pw2)
    oev, gev = O.Evaluator(pr.oQ, None), la.Evaluator(pr.gQ, None)
    sk, sk2 = SecretKey(rng, pr.oQ, None), SecretKey(rng, pr.oQ, None)
    oevk = gen_evaluation_key_base2(rng, pr.oQ, None, sk.Q, sk2, pw2)
    gevk = gev.NewEvaluationKey(oevk.q, None, pw2, oevk.nj[: len(q)])
    assert gevk.LevelP() == -1
    for levelQ in (4, 3, 0):
        Qm = q[: levelQ + 1]
        cx = np.stack([uniform_poly(rng, Qm, pr.N) for _ in range(3)])
        pcx = la.Poly(pr.gQ, levelQ + 1, 3).upload(cx)
        qp = [(la.Poly(pr.gQ, levelQ + 1, 3), None) for _ in range(2)]
        gev.GadgetProductLazy(levelQ, pcx, gevk, qp)
        ct = [la.Poly(pr.gQ, levelQ + 1, 3), la.Poly(pr.gQ, levelQ + 1, 3)]
        gev.GadgetProduct(levelQ, pcx, gevk, ct)
        md = [la.Poly(pr.gQ, levelQ + 1, 3), la.Poly(pr.gQ, levelQ + 1, 3)]
        gev.ModDown(levelQ, -1, qp, md)
        for b in range(3):
            wQ, _ = oev.GadgetProductLazy(levelQ, cx[b], oevk)
            want = oev.GadgetProduct(levelQ, cx[b], oevk)
            for k in range(2):
                assert np.array_equal(qp[k][0].get()[b], wQ[k]), (levelQ, b, k)
                assert np.array_equal(ct[k].get()[b], want[k]) and np.array_equal(md[k].get()[b], want[k]), (levelQ, b, k)
        sub = O.Ring(pr.N, Qm)
        got = np.stack([c.get()[0] for c in ct])
        noise = noise_log2(pr.oQ, sub.binop("Sub", phase(pr.oQ, got, sk2.Q), sub.binop("MulCoeffsMontgomery", cx[0], sk.Q[: levelQ + 1])))
        assert noise <= logN + pw2 + 6, noise
        # coefficient-domain forms of GadgetProduct / ModDown (ct.IsNTT = false)
        cxc = pr.oQ.INTT(np.concatenate([cx[0], np.zeros((len(q) - levelQ - 1, pr.N), dtype=np.uint64)]))[: levelQ + 1]
        pc = la.Poly(pr.gQ, levelQ + 1).upload(cxc)
        ctc = [la.Poly(pr.gQ, levelQ + 1), la.Poly(pr.gQ, levelQ + 1)]
        gev.GadgetProduct(levelQ, pc, gevk, ctc, isNTT=False)
        wantc = np.stack([sub.INTT(w) for w in oev.GadgetProduct(levelQ, cx[0], oevk)])
        assert np.array_equal(np.stack([c.get() for c in ctc]), wantc), levelQ
    # Relinearize and Automorphism sit on the same path
    level = len(q) - 1
    ct3 = np.stack([uniform_poly(rng, q, pr.N) for _ in range(3)])
    p3 = [pr.gQ.NewPoly().upload(c) for c in ct3]
    out = [pr.gQ.NewPoly(), pr.gQ.NewPoly()]
    gev.Relinearize(level, p3, gevk, out)
    assert np.array_equal(np.stack([o.get() for o in out]), oev.Relinearize(ct3, oevk))
    galel = pow(5, 3, 2 * pr.N)
    gev.Automorphism(level, p3[:2], galel, gevk, out)
    assert np.array_equal(np.stack([o.get() for o in out]), oev.Automorphism(ct3[:2], galel, oevk))
    # MulRelin with the P-less key (schemes/ckks/evaluator.go:764-872, schemes/bgv/evaluator.go:592-667): no ModDown to fuse the
    # tensor into -- the three-output tensor kernel followed by the copy branch of the gadget product
    a, b = ct3[:2], np.stack([uniform_poly(rng, q, pr.N) for _ in range(2)])
    pa, pb = [pr.gQ.NewPoly().upload(c) for c in a], [pr.gQ.NewPoly().upload(c) for c in b]
    gev.CKKSMulRelin(level, pa, pb, gevk, out)
    assert np.array_equal(np.stack([o.get() for o in out]), oev.CKKSMulRelin(a, b, oevk, True))
    gev.BGVMulRelin(level, 65537, pa, pb, gevk, out)
    assert np.array_equal(np.stack([o.get() for o in out]), oev.BGVMulRelin(65537, a, b, oevk, True))
    # what the reference cannot do without special primes is rejected: RNS-only keys, hoisted decompositions
    with pytest.raises(la.HeringError):
        gev.NewEvaluationKey(oevk.q[: len(q)], None)
    with pytest.raises(la.HeringError):
        la.Decomposition(gev)


@pytest.mark.parametrize("logN", [9, 10, 12, 13])
def test_mixed_modulus_sizes_all_kernel_classes(ctx, logN):
    """Chains mixing moduli below 2^47 (double-precision kernels, fused NTT+MAC), below 2^58 (correction-free
    integer butterflies) and 60/61-bit ones (Harvey form), in Q and in P: every class must give the oracle's bits."""
    q, p = O.GenModuli(logN + 1, [55, 40, 61, 45, 40, 60, 36], [40, 60, 46])
    pr = Pair(ctx, logN, len(q), len(p), qmods=q, pmods=p)
    rng = rng_for(2700 + logN)
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    sk = SecretKey(rng, pr.oQ, pr.oP)
    sk2 = SecretKey(rng, pr.oQ, pr.oP)
    oevk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, sk2)
    gevk = gev.NewEvaluationKey(oevk.q, oevk.p)
    for levelQ in (6, 5, 3):
        Qm = q[: levelQ + 1]
        cx = np.stack([uniform_poly(rng, Qm, pr.N) for _ in range(3)])
        pcx = la.Poly(pr.gQ, levelQ + 1, 3).upload(cx)
        x = la.Poly(pr.gQ, levelQ + 1, 3)
        pr.gQ.AtLevel(levelQ).NTT(pcx, x)
        sub = O.Ring(pr.N, Qm)
        assert np.array_equal(x.get()[2], sub.NTT(cx[2]))
        pr.gQ.AtLevel(levelQ).INTT(x, x)
        assert np.array_equal(x.get(), cx)
        qp = [(la.Poly(pr.gQ, levelQ + 1, 3), la.Poly(pr.gP, 3, 3)) for _ in range(2)]
        gev.GadgetProductLazy(levelQ, pcx, gevk, qp)
        wQ, wP = oev.GadgetProductLazy(levelQ, cx[1], oevk)
        for k in range(2):
            assert np.array_equal(qp[k][0].get()[1], wQ[k]) and np.array_equal(qp[k][1].get()[1], wP[k]), (levelQ, k)
        ct = [la.Poly(pr.gQ, levelQ + 1, 3), la.Poly(pr.gQ, levelQ + 1, 3)]
        gev.GadgetProduct(levelQ, pcx, gevk, ct)
        want = oev.GadgetProduct(levelQ, cx[0], oevk)
        got = np.stack([c.get()[0] for c in ct])
        assert np.array_equal(got, want), levelQ
        noise = noise_log2(pr.oQ, sub.binop("Sub", phase(pr.oQ, got, sk2.Q), sub.binop("MulCoeffsMontgomery", cx[0], sk.Q[: levelQ + 1])))
        assert noise <= logN + 8, noise


@pytest.mark.parametrize("logN,logq,logp", [(14, [55, 45, 45, 45, 45, 45, 45], [55, 55, 55]), (12, [60, 45, 45, 61, 40, 45], [61, 46])])
def test_fused_decomposition_adversarial_float(ctx, logN, logq, logp):
    """The fused basis extension estimates v = trunc(sum fl(y_i/q_i)) with a reciprocal and redoes the exact IEEE
    divisions only near integers: feed digits whose CRT value sits at 0, +-1, +-Q_d/2 and at multiples of Q_d/q_k +- 1
    (the y_i/q_i sum is then within an ulp of an integer) and require the oracle's bits."""
    q, p = O.GenModuli(logN + 1, logq, logp)
    nq, np_ = len(q), len(p)
    pr = Pair(ctx, logN, nq, np_, qmods=q, pmods=p)
    rng = rng_for(2800 + logN)
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    beta = O.BaseRNSDecompositionVectorSize(nq - 1, np_ - 1)
    kq = np.stack([np.stack([uniform_poly(rng, q, pr.N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(rng, p, pr.N) for _ in range(2)]) for _ in range(beta)])
    gevk, oevk = gev.NewEvaluationKey(kq, kp), O.EvaluationKey(kq, kp)
    coeff = uniform_poly(rng, q, pr.N)  # coefficient domain
    col = 0
    for d in range(beta):
        lo, hi = d * np_, min(d * np_ + np_, nq)
        mods = q[lo:hi]
        Qd = prod(mods)
        half = Qd >> 1
        special = [0, 1, Qd - 1, half, half + 1, half - 1, 2, Qd - 2]
        for k, m in enumerate(mods):
            for j in (1, 2, m // 2, m - 1):
                for dd in (-1, 0, 1):
                    special.append(((Qd // m) * j + dd) % Qd)
        for v in special:  # the reference adds Qd/2 before reconstructing: place the special value AFTER that shift
            x = (v - half) % Qd
            for k, m in enumerate(mods):
                coeff[lo + k, col] = x % m
            col += 1
    assert col < pr.N
    cx = pr.oQ.NTT(coeff)
    pcx = la.Poly(pr.gQ, nq).upload(cx)
    qp = [(la.Poly(pr.gQ, nq), la.Poly(pr.gP, np_)) for _ in range(2)]
    gev.GadgetProductLazy(nq - 1, pcx, gevk, qp)
    wQ, wP = oev.GadgetProductLazy(nq - 1, cx, oevk)
    for k in range(2):
        assert np.array_equal(qp[k][0].get(), wQ[k]) and np.array_equal(qp[k][1].get(), wP[k]), k
    ct = [la.Poly(pr.gQ, nq), la.Poly(pr.gQ, nq)]
    gev.GadgetProduct(nq - 1, pcx, gevk, ct)
    assert np.array_equal(np.stack([c.get() for c in ct]), oev.GadgetProduct(nq - 1, cx, oevk))


def test_keys_loaded_from_the_wire_format(ctx):
    """rlwe.EvaluationKey / GaloisKey bytes (core/rlwe/keys.go:443,628) -> device handle: same gadget product."""
    from lattigo_amd import wire
    pr, rng, oev, gev, sk = _setup(ctx, 10, 4, 1, 2900)
    sk2 = SecretKey(rng, pr.oQ, pr.oP)
    cx = uniform_poly(rng, pr.q, pr.N)
    pcx = _uploadQ(pr, cx)
    for oevk, pw2 in ((gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, sk2), 0),
                      (gen_evaluation_key_base2(rng, pr.oQ, pr.oP, sk.Q, sk2, 20), 20)):
        nj = list(oevk.nj[:4]) if pw2 else None
        blob = wire.galois_key_marshal(5, 2 * pr.N, oevk.q, oevk.p, pw2, nj)
        g, nth, gevk = gev.GaloisKeyFromBinary(blob)
        assert (g, nth) == (5, 2 * pr.N) and gevk.BaseTwoDecomposition == pw2
        gevk2 = gev.EvaluationKeyFromBinary(blob[16:])
        want = oev.GadgetProduct(3, cx, oevk)
        for k in (gevk, gevk2):
            ct = [pr.gQ.NewPoly(), pr.gQ.NewPoly()]
            gev.GadgetProduct(3, pcx, k, ct)
            assert np.array_equal(np.stack([c.get() for c in ct]), want)


def test_concurrent_callers_share_an_evaluator(ctx):
    """Since 6.2.0 every evaluator method of the reference is safe for concurrent callers (core/rlwe/evaluator.go:200-227);
    here: 6 threads issue key-switches, NTTs and buffer churn on ONE context/evaluator/key, plus a second context
    running alongside; every result must equal the oracle's."""
    import threading
    pr, rng, oev, gev, sk = _setup(ctx, 11, 5, 2, 3000)
    oevk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, SecretKey(rng, pr.oQ, pr.oP))
    gevk = gev.NewEvaluationKey(oevk.q, oevk.p)
    inputs = [uniform_poly(rng, pr.q, pr.N) for _ in range(6)]
    wants = [(oev.GadgetProduct(4, x, oevk), pr.oQ.NTT(x)) for x in inputs]
    ctx2 = la.Context(0)
    q2 = la.Ring(ctx2, pr.N, pr.q)
    errs = []

    def work(i):
        try:
            for rep in range(8):
                if i == 5:  # a second context (own stream) running alongside
                    p = la.Poly(q2, 5).upload(inputs[i])
                    q2.NTT(p, p)
                    assert np.array_equal(p.get(), wants[i][1])
                    continue
                pcx = _uploadQ(pr, inputs[i])
                ct = [pr.gQ.NewPoly(), pr.gQ.NewPoly()]
                gev.GadgetProduct(4, pcx, gevk, ct)
                assert np.array_equal(np.stack([c.get() for c in ct]), wants[i][0]), (i, rep)
                t = pr.gQ.NewPoly()
                pr.gQ.NTT(pcx, t)
                assert np.array_equal(t.get(), wants[i][1]), (i, rep)
        except BaseException as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not errs, errs
    assert not any(t.is_alive() for t in th)


def test_gadget_product_and_moddown_domain_flags(ctx):
    """The IsNTT branches of GadgetProductLazy (core/rlwe/evaluator_gadget_product.go:121-125, 142-152) and the four
    (ctQP.IsNTT, ct.IsNTT) cases of Evaluator.ModDown (:39-71): every result must equal the NTT-domain path up to the
    strict transforms."""
    pr, rng, oev, gev, sk = _setup(ctx, 10, 5, 2, 3100)
    oevk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, SecretKey(rng, pr.oQ, pr.oP))
    gevk = gev.NewEvaluationKey(oevk.q, oevk.p)
    for levelQ in (4, 2):
        sub = O.Ring(pr.N, pr.q[: levelQ + 1])
        cx = uniform_poly(rng, pr.q[: levelQ + 1], pr.N)         # NTT domain
        cxc = sub.INTT(cx)                                       # the same polynomial, coefficient domain
        want = oev.GadgetProduct(levelQ, cx, oevk)               # NTT result
        wantc = np.stack([sub.INTT(want[k]) for k in range(2)])  # coefficient-domain result
        wQ, wP = oev.GadgetProductLazy(levelQ, cx, oevk)
        # coefficient-domain input -> coefficient-domain QP output
        qp = [(la.Poly(pr.gQ, levelQ + 1), la.Poly(pr.gP, 2)) for _ in range(2)]
        gev.GadgetProductLazy(levelQ, _uploadQ(pr, cxc), gevk, qp, isNTT=False)
        for k in range(2):
            assert np.array_equal(qp[k][0].get(), sub.INTT(wQ[k])) and np.array_equal(qp[k][1].get(), pr.oP.INTT(wP[k]))
        ct = [la.Poly(pr.gQ, levelQ + 1), la.Poly(pr.gQ, levelQ + 1)]
        gev.GadgetProduct(levelQ, _uploadQ(pr, cxc), gevk, ct, isNTT=False)
        assert np.array_equal(np.stack([c.get() for c in ct]), wantc)
        for qp_ntt, ct_ntt in ((True, False), (False, True), (False, False), (True, True)):
            qp = [(la.Poly(pr.gQ, levelQ + 1).upload(wQ[k] if qp_ntt else sub.INTT(wQ[k])),
                   la.Poly(pr.gP, 2).upload(wP[k] if qp_ntt else pr.oP.INTT(wP[k]))) for k in range(2)]
            out = [la.Poly(pr.gQ, levelQ + 1), la.Poly(pr.gQ, levelQ + 1)]
            gev.ModDown(levelQ, 1, qp, out, ctQPIsNTT=qp_ntt, ctIsNTT=ct_ntt)
            assert np.array_equal(np.stack([c.get() for c in out]), want if ct_ntt else wantc), (levelQ, qp_ntt, ct_ntt)


_DEVICE_BUFFER_WORKER = """
import os, sys
import torch                              # before libhering: one HIP runtime in the process (lattigo_amd/dist.py)
sys.path.insert(0, %r)
import numpy as np
import lattigo_amd as la
from lattigo_amd import rlwe as R
from lattigo_amd.dist import ControlPlane
from oracle import oracle as O
from tests.helpers import rng_for, uniform_poly
from tests.rlwe_fixtures import SecretKey, gen_evaluation_key
ctx = la.Context(0)
N = 1 << 11
q, p = O.GenModuli(12, [55, 45, 45, 45, 45], [46, 46])  # limbs below 2^47: the key also carries its derived f64 copy
gQ, gP, oQ, oP = la.Ring(ctx, N, q), la.Ring(ctx, N, p), O.Ring(N, q), O.Ring(N, p)
rng = rng_for(2950)
oev, gev = O.Evaluator(oQ, oP), la.Evaluator(gQ, gP)
oevk = gen_evaluation_key(rng, oQ, oP, SecretKey(rng, oQ, oP).Q, SecretKey(rng, oQ, oP))
src = gev.NewEvaluationKey(oevk.q, oevk.p)
words = src.download()
assert np.array_equal(words[:, :, : src.nQk], oevk.q) and np.array_equal(words[:, :, src.nQk:], oevk.p)
beta, nQk, nPk, base_two, nj = src.Shape()
peer = R.EvaluationKey(gev, None, None, base_two, nj, shape=(beta, nQk, nPk))
assert not peer.download().any()

def view(key):
    ptr, nbytes = key.DeviceBuffer()
    class W:
        __cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<i8", "data": (ptr, False), "version": 2}
    t = torch.as_tensor(W(), device="cuda:0")
    assert t.data_ptr() == ptr
    return t

view(peer).copy_(view(src))
torch.cuda.synchronize()
peer.Commit()
cx = uniform_poly(rng, q, N)
want = oev.GadgetProduct(4, cx, oevk)

def check(key):
    ct = [la.Poly(gQ, 5), la.Poly(gQ, 5)]
    gev.GadgetProduct(4, la.Poly(gQ, 5).upload(cx), key, ct)
    assert np.array_equal(np.stack([c.get() for c in ct]), want)

check(src); check(peer)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
cp = ControlPlane(init_single=True)       # a one-rank group drives the same broadcast call the multi-GPU path uses
for transport in ("rccl", "host"):
    check(cp.ReplicateEvaluationKey(gev, src, src=0, transport=transport))
cp.close()
print("DEVICE_BUFFER_OK")
"""


def _run_worker(tmp_path, text, nproc, port):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(text % root)
    cmd = [sys.executable, str(script)] if nproc == 1 else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
        "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


def test_key_replication_device_buffer(tmp_path):
    """SURVEY.md section 8e key replication: an empty key of the same shape, filled device-side through
    he_evk_device_buffer (the target of the RCCL broadcast) and committed, gives the same gadget product; a one-rank
    process group then drives ReplicateEvaluationKey over both transports.  In a process of its own: torch's HIP runtime
    has to be the one libhering binds to."""
    assert "DEVICE_BUFFER_OK" in _run_worker(tmp_path, _DEVICE_BUFFER_WORKER, 1, 0)


_REPLICA_WORKER = """
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
import lattigo_amd as la
from lattigo_amd.dist import ControlPlane
from oracle import oracle as O
from tests.helpers import rng_for, uniform_poly
from tests.rlwe_fixtures import SecretKey, gen_evaluation_key
cp = ControlPlane()
assert cp.world == 2
ctx = la.Context(0)                      # both ranks on the one GPU of the test box
q, p = O.GenModuli(11, [55, 45, 45], [46])
N = 1 << 10
gQ, gP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
gev = la.Evaluator(gQ, gP)
key = None
rng = rng_for(2960)
cx = uniform_poly(rng, q, N)
if cp.rank == 0:                         # only rank 0 ever holds the key material on the host
    oQ, oP = O.Ring(N, q), O.Ring(N, p)
    key = gev.NewEvaluationKey(*(lambda k: (k.q, k.p))(gen_evaluation_key(rng, oQ, oP, SecretKey(rng, oQ, oP).Q, SecretKey(rng, oQ, oP))))
key = cp.ReplicateEvaluationKey(gev, key, src=0, transport="host")
ct = [la.Poly(gQ, 3), la.Poly(gQ, 3)]
gev.GadgetProduct(2, la.Poly(gQ, 3).upload(cx), key, ct)
digest = int(hashlib.sha256(np.stack([c.get() for c in ct]).tobytes()).hexdigest()[:12], 16)
assert cp.max_over_ranks(digest) == digest == -cp.max_over_ranks(-digest)
if cp.rank == 0:
    print("REPLICATED", digest)
cp.close()
"""


def test_key_replication_two_ranks_host_transport(tmp_path):
    """Two processes (sharing the box's one GPU): rank 1 receives the key from rank 0 and computes the same gadget product."""
    assert "REPLICATED" in _run_worker(tmp_path, _REPLICA_WORKER, 2, 29543)


@pytest.mark.parametrize("logN,batch", [(10, 1), (12, 8), (13, 32), (14, 16)])
def test_evaluator_moddown_matches_basis_extender(ctx, logN, batch):
    """he_eval_moddown_qp_to_q_ntt (fused three-launch pipeline when the grid is wide enough, six-launch form otherwise)
    returns ModDownQPtoQNTT's words (ring/basis_extension.go:235-256), also in place and into a scratch polynomial."""
    q, p = O.GenModuli(logN + 1, [55, 45, 45, 58, 40], [55, 46])  # every kernel class on both sides
    pr = Pair(ctx, logN, 5, 2, qmods=q, pmods=p)
    obe = O.BasisExtender(pr.oQ, pr.oP)
    gev = la.Evaluator(pr.gQ, pr.gP)
    rng = rng_for(3100 + logN)
    for levelQ, levelP in ((4, 1), (2, 0)):
        xq = np.stack([uniform_poly(rng, q[: levelQ + 1], pr.N) for _ in range(batch)])
        xp = np.stack([uniform_poly(rng, p[: levelP + 1], pr.N) for _ in range(batch)])
        want = np.stack([obe.ModDownQPtoQNTT(levelQ, levelP, xq[b], xp[b]) for b in range(batch)])
        pq = la.Poly(pr.gQ, levelQ + 1, batch).upload(xq)
        pp = la.Poly(pr.gP, levelP + 1, batch).upload(xp)
        out = la.Poly(pr.gQ, levelQ + 1, batch, zero=False)
        gev.ModDownQPtoQNTT(levelQ, levelP, pq, pp, out)
        assert np.array_equal(out.download(), want), (levelQ, levelP)
        gev.ModDownQPtoQNTT(levelQ, levelP, pq, pp, pq)  # in place
        assert np.array_equal(pq.download(), want), ("in place", levelQ, levelP)


@pytest.mark.parametrize("logN,logq,logp", [(13, [55, 45, 45, 45], [55, 55, 55]), (13, [60, 45, 40, 36], [61, 61, 61, 61]),
                                            (12, [45, 45, 58], [61, 55])])
def test_moddown_split_residues_at_their_extremes(ctx, logN, logq, logp):
    """ModDown from special primes of 2^51 and above into double-precision limbs: the residues y_i = (x_i + P/2)(P/p_i)^-1 do
    not fit a double, the kernel splits them at 29 bits and sums the products exactly in one binade before a single reduction.
    Drive every source's y_i through 0, 1, p_i - 1 and the values around the split point and the top of its high half, in all
    combinations across the sources, and require ModDownQPtoQNTT's words (ring/basis_extension.go:235-256)."""
    import itertools
    q, p = O.GenModuli(logN + 1, logq, logp)
    pr = Pair(ctx, logN, len(q), len(p), qmods=q, pmods=p)
    obe = O.BasisExtender(pr.oQ, pr.oP)
    gev = la.Evaluator(pr.gQ, pr.gP)
    rng = rng_for(3300 + logN)
    levelQ, levelP = len(q) - 1, len(p) - 1
    P = prod(p)
    half = P >> 1
    xp = uniform_poly(rng, p, pr.N)  # coefficient domain
    cands = [[0, 1, m - 1, (1 << 29) - 1, 1 << 29, m - (1 << 29), ((m >> 29) << 29) - 1] for m in p]
    for col, ys in enumerate(itertools.islice(itertools.product(*cands), pr.N)):
        for i, (m, y) in enumerate(zip(p, ys)):
            xp[i, col] = (y * (P // m) - half) % m
    xpn = pr.oP.NTT(xp)
    xq = uniform_poly(rng, q, pr.N)
    want = obe.ModDownQPtoQNTT(levelQ, levelP, xq, xpn)
    batch = 8  # wide enough for the fused three-launch pipeline (he_eval_moddown_qp_to_q_ntt)
    pq = la.Poly(pr.gQ, levelQ + 1, batch).upload(np.stack([xq] * batch))
    pp = la.Poly(pr.gP, levelP + 1, batch).upload(np.stack([xpn] * batch))
    out = la.Poly(pr.gQ, levelQ + 1, batch, zero=False)
    gev.ModDownQPtoQNTT(levelQ, levelP, pq, pp, out)
    got = out.download()
    for b in range(batch):
        assert np.array_equal(got[b], want), b


def test_captured_graph_replays_a_call_sequence(ctx):
    """he_graph_begin / he_graph_end / he_graph_launch (include/hering.h): a MulRelin -> Rescale -> Automorphism chain, with a
    temporary created inside the sequence, is recorded once and replayed on NEW input data uploaded into the same polynomials;
    every replay must give the oracle's bits.  Host transfers and sync are refused while capturing."""
    pr, rng, oev, gev, sk = _setup(ctx, 12, 5, 2, 3400)
    sk_sq = pr.oQ.binop("MulCoeffsMontgomery", sk.Q, sk.Q)
    orlk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk_sq, sk)
    grlk = gev.NewEvaluationKey(orlk.q, orlk.p)
    galel = 5
    sk_out = automorphism_secret(rng, pr.oQ, pr.oP, sk, pow(galel, 2 * pr.N - 1, 2 * pr.N))
    ogk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, sk_out)
    ggk = gev.NewEvaluationKey(ogk.q, ogk.p)
    level = 4
    a = [pr.gQ.NewPoly() for _ in range(2)]
    b = [pr.gQ.NewPoly() for _ in range(2)]
    res = [la.Poly(pr.gQ, level) for _ in range(2)]
    rot = [la.Poly(pr.gQ, level) for _ in range(2)]

    def chain():
        tmp = [pr.gQ.NewPoly() for _ in range(2)]  # lives only inside the call sequence
        gev.CKKSMulRelin(level, a, b, grlk, tmp)
        gev.Rescale(level, 1, tmp, res)
        gev.Automorphism(level - 1, res, galel, ggk, rot)

    def fill():
        ct0 = np.stack([uniform_poly(rng, pr.q, pr.N) for _ in range(2)])
        ct1 = np.stack([uniform_poly(rng, pr.q, pr.N) for _ in range(2)])
        for k in range(2):
            a[k].upload(ct0[k])
            b[k].upload(ct1[k])
        want_res = oev.Rescale(oev.CKKSMulRelin(ct0, ct1, orlk, True), 1)
        return want_res, oev.Automorphism(want_res, galel, ogk)

    def outputs():
        return np.stack([r.get() for r in res]), np.stack([r.get() for r in rot])

    def clear():
        for r in res + rot:
            r.upload(np.zeros((level, pr.N), dtype=np.uint64))

    want = fill()
    chain()  # first run: builds plans, index tables, scratch
    got = outputs()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    with ctx.capture() as g:
        chain()
        with pytest.raises(la.HeringError):
            a[0].get()
        with pytest.raises(la.HeringError):
            ctx.sync()
    assert g.nodes() >= 10
    # an eager call that needs a (much) larger scratch arena: the arena the graph's launches address must survive its growth
    big = [[la.Poly(pr.gQ, len(pr.q), 16) for _ in range(2)] for _ in range(3)]
    gev.CKKSMulRelin(level, big[0], big[1], grlk, big[2])
    ctx.sync()
    for _ in range(3):  # the first replay on the captured data, then new data through the same handles
        clear()
        g.launch()
        got = outputs()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        want = fill()
    g.close()
    # the buffers the graph held are back in the pool; eager calls go on working
    chain()
    got = outputs()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("logN", [13, 15])
def test_gadget_product_output_aliasing_its_input(ctx, logN):
    """GadgetProduct with an output polynomial that IS the input cx (core/rlwe/evaluator_gadget_product.go:26-56 reads cx completely
    before it writes ct): the pipeline with the ModDown epilogue inside the NTT + MAC kernel writes outputs while other workgroups
    still read cx, so the library must notice the aliasing and take the other pipeline -- same words either way."""
    q, p = O.GenModuli(logN + 1, [55, 45, 45, 45, 45, 45], [55, 55])
    pr = Pair(ctx, logN, len(q), len(p), qmods=q, pmods=p)
    rng = rng_for(3500 + logN)
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    sk, sk2 = SecretKey(rng, pr.oQ, pr.oP), SecretKey(rng, pr.oQ, pr.oP)
    oevk = gen_evaluation_key(rng, pr.oQ, pr.oP, sk.Q, sk2)
    gevk = gev.NewEvaluationKey(oevk.q, oevk.p)
    level, B = len(q) - 1, 8
    cx = np.stack([uniform_poly(rng, q, pr.N) for _ in range(B)])
    want = np.stack([oev.GadgetProduct(level, cx[b], oevk) for b in range(B)])  # [B][2][limbs][N]
    # separate outputs: the fused pipeline
    pcx = la.Poly(pr.gQ, len(q), B).upload(cx)
    out = [la.Poly(pr.gQ, len(q), B), la.Poly(pr.gQ, len(q), B)]
    gev.GadgetProduct(level, pcx, gevk, out)
    assert np.array_equal(out[0].get(), want[:, 0]) and np.array_equal(out[1].get(), want[:, 1])
    for k in range(2):  # output k is cx itself
        pcx = la.Poly(pr.gQ, len(q), B).upload(cx)
        other = la.Poly(pr.gQ, len(q), B)
        outs = [pcx, other] if k == 0 else [other, pcx]
        gev.GadgetProduct(level, pcx, gevk, outs)
        assert np.array_equal(outs[0].get(), want[:, 0]) and np.array_equal(outs[1].get(), want[:, 1]), k


def test_random_scheme_level_calls(ctx):
    """tools/fuzz_shapes.py's single-call generator, 150 draws of a fixed seed: operation, shape, modulus classes, level, a key
    that ends below the ring's top level, batch size and output / input aliasing all at random, every entry against the oracle
    (the tool itself ran 17 387 draws and 3 000 full-size shape checks on the GPU without a mismatch)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_shapes", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "fuzz_shapes.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.Generator(np.random.PCG64(20260924))
    seen = set()
    for _ in range(150):
        seen.add(fz.api_case(ctx, rng).split()[1])
    assert seen == {"op=bgv", "op=ckks", "op=relin", "op=rotate", "op=gadget", "op=giant"}  # (giant: he_lintrans_giant_step, round 6)
    # the ring-level generator of the same tool (every coefficient-wise formula, transforms, rescales, automorphisms; logN 4..16, level
    # below the top, batch, in place): 400 draws (71 681 on the GPU by the tool itself)
    ops = set()
    for _ in range(400):
        ops.add(fz.ring_case(ctx, rng).split()[1])
    assert len(ops) >= 30, ops
    # basis extension entry points, the evaluator's fused ModDownQPtoQNTT and DecomposeNTT at random (levelQ, levelP): 200 draws
    # (42 092 by the tool)
    for _ in range(200):
        fz.be_case(ctx, rng)


@pytest.mark.parametrize("deferred", [0, 6])
def test_random_calls_with_the_queue_on(ctx, deferred):
    """The same three generators (another seed) with the context's submission queue switched on: every draw over handles of fewer
    than max_batch entries is filed as a request and served as a lone caller's batch (closure over its own views, no entry table),
    every larger batch launches directly -- the dispatch every entry point takes since round 5 must not change a word.  deferred:
    the calls return once filed and the context's dispatcher thread launches them in the order this thread made them; the
    generators' downloads wait for the thread's pending requests."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_shapes", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "fuzz_shapes.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.Generator(np.random.PCG64(20260925))
    ctx.SetCoalescing(8, 50)
    ctx.SetDeferred(deferred)
    try:
        s0 = ctx.CoalescingStats()
        for _ in range(100):
            fz.api_case(ctx, rng)
        for _ in range(300):
            fz.ring_case(ctx, rng)
        for _ in range(150):
            fz.be_case(ctx, rng)
        ctx.sync()
        assert ctx.CoalescingStats()["calls"] - s0["calls"] > 50  # the batch-1 draws did go through the queue
    finally:
        ctx.SetCoalescing(0, 0)


@pytest.mark.parametrize("logN,logq,logp,B", [(13, [55, 45, 45, 60, 45, 45], [55, 45], 3),    # 4096-rows; a double-precision special prime too
                                               (16, [60, 45, 45, 45, 58], [61, 61], 2),       # 8192-rows
                                               (15, [55, 55, 56], [55], 1),                    # integer-class limbs only (no NTT + MAC launch)
                                               (11, [55, 45, 45], [55, 45], 2)])               # below the persistent kernel: the separate launches
def test_lintrans_giant_step(ctx, logN, logq, logp, B):
    """he_lintrans_giant_step (circuits/common/lintrans/lintrans_evaluator.go:397-441): GadgetProductLazy + ringQP.Add of the
    inner sum + AutomorphismNTTWithIndex[ThenAddLazy] into the outer accumulators as one call whose key inner products store
    through the automorphism -- against the oracle's separate calls, word for word, overwriting and accumulating (the
    accumulator's previous words are arbitrary 64-bit values: ...ThenAddLazy does not reduce), full level and one below."""
    q, p = O.GenModuli(logN + 1, logq, logp)
    pr = Pair(ctx, logN, len(q), len(p), qmods=q, pmods=p)
    N = pr.N
    rng = rng_for(6100 + logN)
    oev, gev = O.Evaluator(pr.oQ, pr.oP), la.Evaluator(pr.gQ, pr.gP)
    nq, np_ = len(q), len(p)
    beta = (nq + np_ - 1) // np_
    kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(beta)])
    okey, gkey = O.EvaluationKey(kq, kp), gev.NewEvaluationKey(kq, kp)
    for levelQ in (nq - 1, nq - 2):
        Qm = q[: levelQ + 1]
        for galel, acc in ((5, False), (pow(5, 77, 2 * N), True), (2 * N - 1, True)):
            cx = np.stack([uniform_poly(rng, Qm, N) for _ in range(B)])
            aq = np.stack([uniform_poly(rng, Qm, N) for _ in range(B)])
            ap = np.stack([uniform_poly(rng, p, N) for _ in range(B)])
            prev = [[rng.integers(0, 1 << 63, size=(B, n, N), dtype=np.uint64) * np.uint64(2) + np.uint64(1) for n in (levelQ + 1, np_)]
                    for _ in range(2)]
            outs = [(la.Poly(pr.gQ, levelQ + 1, B).upload(prev[k][0]), la.Poly(pr.gP, np_, B).upload(prev[k][1])) for k in range(2)]
            gev.LinTransGiantStep(levelQ, la.Poly(pr.gQ, levelQ + 1, B).upload(cx), gkey, galel,
                                  (la.Poly(pr.gQ, levelQ + 1, B).upload(aq), la.Poly(pr.gP, np_, B).upload(ap)), outs, acc)
            idx = pr.oQ.AutomorphismNTTIndex(galel)
            for b in range(B):
                wQ, wP = oev.GadgetProductLazy(levelQ, cx[b], okey)
                for k in range(2):
                    for part, (w, add, ring_, mods) in enumerate(((wQ[k], aq[b], pr.oQ, Qm), (wP[k], ap[b], pr.oP, p))):
                        v = w.copy()
                        if k == 0:  # ringQP.Add of canonical words: one conditional subtraction
                            for i, m in enumerate(mods):
                                s = v[i] + add[i]
                                v[i] = np.where(s >= np.uint64(m), s - np.uint64(m), s)
                        want = v[:, idx]
                        if acc:
                            want = prev[k][part][b] + want  # uint64 wrap-around, no reduction
                        got = outs[k][part].download()[b] if B > 1 else outs[k][part].download().reshape(want.shape)
                        assert np.array_equal(got, want), (logN, levelQ, galel, acc, b, k, part)

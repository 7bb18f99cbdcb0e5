"""Semantic checks of the oracle's key-switch (core/rlwe/rlwe_test.go:666-779,
897-1070 style: decrypt and bound the noise), plus internal consistency of the
hoisted / non-hoisted and lazy / strict forms.  No GPU needed."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import Pi60, Qi60
from tests.helpers import rng_for, uniform_poly
from tests.rlwe_fixtures import (SecretKey, automorphism_secret, gen_evaluation_key,
                                 gen_evaluation_key_base2, noise_log2, phase)

N = 1 << 10


def setup(nq, np_, seed):
    rng = rng_for(seed)
    ringQ, ringP = O.Ring(N, Qi60[:nq]), O.Ring(N, Pi60[:np_])
    ev = O.Evaluator(ringQ, ringP)
    sk = SecretKey(rng, ringQ, ringP)
    return rng, ringQ, ringP, ev, sk


@pytest.mark.parametrize("nq,np_", [(6, 3), (7, 3), (5, 2), (4, 1), (6, 4)])
@pytest.mark.parametrize("drop", [0, 1, 2])
def test_gadget_product_decrypts(nq, np_, drop):
    rng, ringQ, ringP, ev, sk = setup(nq, np_, 200 + nq * 10 + np_)
    sk2 = SecretKey(rng, ringQ, ringP)
    evk = gen_evaluation_key(rng, ringQ, ringP, sk.Q, sk2)
    levelQ = nq - 1 - drop
    if levelQ < 0:
        pytest.skip("level")
    sub = O.Ring(N, ringQ.moduli[: levelQ + 1])
    cx = uniform_poly(rng, sub.moduli, N)
    ct = ev.GadgetProduct(levelQ, cx, evk)
    # <ct, (1, sk2)> ~ cx * sk
    got = phase(ringQ, ct, sk2.Q)
    want = sub.binop("MulCoeffsMontgomery", cx, sk.Q[: levelQ + 1])
    noise = noise_log2(ringQ, sub.binop("Sub", got, want))
    assert noise <= 10 + 6, noise  # logN + a few bits (rlwe_test.go:725)


def test_hoisted_equals_plain_and_lazy_is_canonical():
    rng, ringQ, ringP, ev, sk = setup(6, 3, 300)
    evk = gen_evaluation_key(rng, ringQ, ringP, sk.Q, sk)
    for levelQ in (5, 4, 3):
        sub = O.Ring(N, ringQ.moduli[: levelQ + 1])
        cx = uniform_poly(rng, sub.moduli, N)
        dq, dp = ev.DecomposeNTT(levelQ, 2, 3, cx, True)
        ctQ, ctP = ev.GadgetProductLazy(levelQ, cx, evk)
        hQ, hP = ev.GadgetProductHoistedLazy(levelQ, dq, dp, evk)
        assert np.array_equal(ctQ, hQ) and np.array_equal(ctP, hP)
        q = np.array(sub.moduli, dtype=np.uint64)[None, :, None]
        assert np.all(ctQ < q)
        assert np.array_equal(ev.GadgetProduct(levelQ, cx, evk), ev.GadgetProductHoisted(levelQ, dq, dp, evk))
        assert np.array_equal(ev.ModDown(levelQ, 2, ctQ, ctP), ev.GadgetProduct(levelQ, cx, evk))


def test_decomposition_recombines():
    """sum_d decomp_d * (P * Qd~ ...) is not needed: check instead that each digit
    is congruent to the centred lift of the digit's residues on every limb."""
    rng, ringQ, ringP, ev, sk = setup(7, 3, 301)
    levelQ, levelP = 6, 2
    cx = uniform_poly(rng, ringQ.moduli, N)  # coefficient domain
    dec = O.Decomposer(ringQ, ringP)
    for d in range(3):
        lo, hi = d * 3, min(d * 3 + 3, levelQ + 1)
        mods = ringQ.moduli[lo:hi]
        Qd = 1
        for m in mods:
            Qd *= m
        # CRT-reconstruct the digit, centre it
        vals = []
        for j in range(8):
            x = 0
            for i, m in enumerate(mods):
                Mi = Qd // m
                x += int(cx[lo + i, j]) * pow(Mi, -1, m) % m * Mi
            x %= Qd
            # centring convention of the reference: +Qd/2 before, -Qd/2 after
            x = (x + (Qd >> 1)) % Qd - (Qd >> 1)
            vals.append(x)
        p1Q, p1P = dec.DecomposeAndSplit(levelQ, levelP, 3, d, cx)
        for l in range(levelQ + 1):
            if lo <= l < hi:
                continue
            m = ringQ.moduli[l]
            for j in range(8):
                assert int(p1Q[l, j]) % m == vals[j] % m
        for l in range(levelP + 1):
            m = ringP.moduli[l]
            for j in range(8):
                assert int(p1P[l, j]) % m == vals[j] % m


def test_relinearize_and_mul_decrypt():
    rng, ringQ, ringP, ev, sk = setup(5, 2, 302)
    sk_sq = ringQ.binop("MulCoeffsMontgomery", sk.Q, sk.Q)  # s^2, NTT + Montgomery
    rlk = gen_evaluation_key(rng, ringQ, ringP, sk_sq, sk)
    level = 4
    ct0 = np.stack([uniform_poly(rng, ringQ.moduli, N) for _ in range(2)])
    ct1 = np.stack([uniform_poly(rng, ringQ.moduli, N) for _ in range(2)])
    deg2 = ev.CKKSMulRelin(ct0, ct1, None, False)
    rel = ev.CKKSMulRelin(ct0, ct1, rlk, True)
    assert np.array_equal(ev.Relinearize(deg2, rlk), rel)
    # phase(deg2) == phase(ct0)*phase(ct1) exactly; phase(rel) == phase(deg2) + small
    p0, p1 = phase(ringQ, ct0, sk.Q), phase(ringQ, ct1, sk.Q)
    prod_ = ringQ.binop("MulCoeffsBarrett", p0, p1)
    assert np.array_equal(phase(ringQ, deg2, sk.Q), prod_)
    assert noise_log2(ringQ, ringQ.binop("Sub", phase(ringQ, rel, sk.Q), prod_)) <= 16
    # BGV tensor = CKKS tensor scaled by t
    t = 65537
    bgv = ev.BGVMulRelin(t, ct0, ct1, None, False)
    assert np.array_equal(bgv, np.stack([ringQ.scalarop("MulScalar", c, t) for c in deg2]))


def test_automorphism_decrypts():
    rng, ringQ, ringP, ev, sk = setup(5, 2, 303)
    for galel in (5, 2 * N - 1, pow(5, 3, 2 * N)):
        galinv = pow(galel, 2 * N - 1, 2 * N)  # core/rlwe/params.go:587
        sk_out = automorphism_secret(rng, ringQ, ringP, sk, galinv)
        gk = gen_evaluation_key(rng, ringQ, ringP, sk.Q, sk_out)
        ct = np.stack([uniform_poly(rng, ringQ.moduli, N) for _ in range(2)])
        out = ev.Automorphism(ct, galel, gk)
        idx = ringQ.AutomorphismNTTIndex(galel)
        want = ringQ.AutomorphismNTTWithIndex(phase(ringQ, ct, sk.Q), idx)
        assert noise_log2(ringQ, ringQ.binop("Sub", phase(ringQ, out, sk.Q), want)) <= 16
        dq, dp = ev.DecomposeNTT(4, 1, 2, ct[1], True)
        assert np.array_equal(ev.AutomorphismHoisted(ct, dq, dp, galel, gk), out)


def test_rescale_matches_ring_op():
    rng, ringQ, ringP, ev, sk = setup(5, 2, 304)
    ct = np.stack([uniform_poly(rng, ringQ.moduli, N) for _ in range(3)])
    out = ev.Rescale(ct, 1)
    for i in range(3):
        assert np.array_equal(out[i], ringQ.DivRoundByLastModulusNTT(ct[i]))
    out2 = ev.Rescale(ct, 2)
    for i in range(3):
        assert np.array_equal(out2[i], ringQ.DivRoundByLastModulusManyNTT(2, ct[i]))


@pytest.mark.parametrize("pw2", [12, 20, 31])
def test_base2_gadget_product_decrypts(pw2):
    """gadgetProductSinglePAndBitDecompLazy with BaseTwoDecomposition != 0 (core/rlwe/evaluator_gadget_product.go:203)."""
    rng, ringQ, ringP, ev, sk = setup(4, 1, 400 + pw2)
    sk2 = SecretKey(rng, ringQ, ringP)
    evk = gen_evaluation_key_base2(rng, ringQ, ringP, sk.Q, sk2, pw2)
    for levelQ in (3, 1):
        sub = O.Ring(N, ringQ.moduli[: levelQ + 1])
        cx = uniform_poly(rng, sub.moduli, N)
        ct = ev.GadgetProduct(levelQ, cx, evk)
        got = phase(ringQ, ct, sk2.Q)
        want = sub.binop("MulCoeffsMontgomery", cx, sk.Q[: levelQ + 1])
        assert noise_log2(ringQ, sub.binop("Sub", got, want)) <= 10 + pw2 + 6


@pytest.mark.parametrize("pw2", [2, 16])
def test_gadget_product_without_special_primes_decrypts(pw2):
    """Parameters without P (levelP = -1): the reference's third test set (core/rlwe/test_params.go:36-46, "No RNS
    decomposition, Pw2 decomposition") -- gadgetProductSinglePAndBitDecompLazy with ringP == nil and ModDown's levelP == -1
    copy (core/rlwe/evaluator_gadget_product.go:74-96,:284,:302,:330).  Noise bound as rlwe_test.go:679."""
    rng = rng_for(900 + pw2)
    q = [0x200000440001, 0x7fff80001, 0x800280001, 0x7ffd80001, 0x7ffc80001]  # core/rlwe/test_params.go:11
    ringQ = O.Ring(N, q)
    ev = O.Evaluator(ringQ, None)
    sk, sk2 = SecretKey(rng, ringQ, None), SecretKey(rng, ringQ, None)
    evk = gen_evaluation_key_base2(rng, ringQ, None, sk.Q, sk2, pw2)
    assert evk.LevelP() == -1
    for levelQ in (4, 2, 0):
        sub = O.Ring(N, q[: levelQ + 1])
        cx = uniform_poly(rng, sub.moduli, N)
        ctQ, ctP = ev.GadgetProductLazy(levelQ, cx, evk)
        assert ctP.shape == (2, 0, N)
        ct = ev.GadgetProduct(levelQ, cx, evk)
        assert np.array_equal(ct, ctQ)  # ModDown without P is a copy of the reduced Q accumulators
        got = phase(ringQ, ct, sk2.Q)
        want = sub.binop("MulCoeffsMontgomery", cx, sk.Q[: levelQ + 1])
        assert noise_log2(ringQ, sub.binop("Sub", got, want)) <= 10 + pw2 + 6


def test_conjugate_invariant_key_switch_decrypts():
    """The key-switch on the conjugate-invariant ring Z[X + X^-1]/(X^2N + 1) (RingType = ConjugateInvariant; real-valued
    CKKS): gadget product, relinearisation-style MulRelin and rotations by 5^k (NthRoot = 4N in the automorphism index,
    ring/ring.go:261, ring/automorphism.go:12-34) decrypt within the reference's noise bound (core/rlwe/rlwe_test.go:679,725)."""
    rng = rng_for(950)
    ringQ, ringP = O.Ring(N, Qi60[:5], True), O.Ring(N, Pi60[:2], True)
    assert ringQ.NthRoot() == 4 * N
    ev = O.Evaluator(ringQ, ringP)
    sk, sk2 = SecretKey(rng, ringQ, ringP), SecretKey(rng, ringQ, ringP)
    evk = gen_evaluation_key(rng, ringQ, ringP, sk.Q, sk2)
    for levelQ in (4, 2):
        sub = O.Ring(N, ringQ.moduli[: levelQ + 1], True)
        cx = uniform_poly(rng, sub.moduli, N)
        got = phase(ringQ, ev.GadgetProduct(levelQ, cx, evk), sk2.Q)
        want = sub.binop("MulCoeffsMontgomery", cx, sk.Q[: levelQ + 1])
        assert noise_log2(ringQ, sub.binop("Sub", got, want)) <= 10 + 6
    for k in (1, 3):  # rotations; the conjugation X -> X^-1 does not exist on this ring (core/rlwe/params.go:593)
        galel = pow(5, k, 4 * N)
        sk_out = automorphism_secret(rng, ringQ, ringP, sk, pow(galel, 4 * N - 1, 4 * N))
        gk = gen_evaluation_key(rng, ringQ, ringP, sk.Q, sk_out)
        ct = np.stack([uniform_poly(rng, ringQ.moduli, N) for _ in range(2)])
        out = ev.Automorphism(ct, galel, gk)
        want = ringQ.AutomorphismNTTWithIndex(phase(ringQ, ct, sk.Q), ringQ.AutomorphismNTTIndex(galel))
        assert noise_log2(ringQ, ringQ.binop("Sub", phase(ringQ, out, sk.Q), want)) <= 10 + 6
    # coefficient-domain automorphism agrees with the NTT-domain one
    x = uniform_poly(rng, ringQ.moduli, N)
    g = pow(5, 7, 4 * N)
    assert np.array_equal(ringQ.NTT(ringQ.Automorphism(x, g)), ringQ.AutomorphismNTT(ringQ.NTT(x), g))

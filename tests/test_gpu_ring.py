"""GPU parity, ring layer: libhering's HIP kernels vs the CPU
oracle on the same seeded inputs, bit-exact, plus the reference's own NTT known-answer vectors
(ring/ntt_test.go) and size-independent properties at logN = 15/16."""
import json
import os

import numpy as np
import pytest

import lattigo_amd as la
from oracle import oracle as O
from tests.conftest import Pi60, Qi60
from tests.gpu_common import Pair, ctx  # noqa: F401
from tests.helpers import rng_for, uniform_poly

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ntt_kat.json")


def test_ntt_known_answer_vectors(ctx):
    kat = json.load(open(GOLDEN))
    for c in kat["cases"]:
        r = la.Ring(ctx, c["N"], c["Qis"])
        x = np.array(c["poly"], dtype=np.uint64)
        y = np.array(c["polyNTT"], dtype=np.uint64)
        px, pz = r.NewPoly().upload(x), r.NewPoly()
        r.NTT(px, pz)
        assert np.array_equal(pz.get(), y), f"N={c['N']}"
        r.INTT(pz, pz)  # in place, as ring/ntt_test.go:113
        assert np.array_equal(pz.get(), x), f"N={c['N']}"


def test_config1_host_slice_transformer(ctx):
    """BASELINE config 1: ring.NTT forward+inverse, N=2^12, one prime Qi60[0] (ring/ntt_benchmark_test.go:29),
    through the ring.NumberTheoreticTransformer host-slice plug point (ring/ntt.go:17-22)."""
    N, q = 1 << 12, [0x1fffffffffe00001]
    g, o = la.Ring(ctx, N, q), O.Ring(N, q)
    x = uniform_poly(rng_for(0), q, N)
    y = g.Forward(0, x[0])
    assert np.array_equal(y, o.NTT(x)[0])
    assert np.array_equal(g.Backward(0, y), x[0])
    assert np.array_equal(o.unop("Reduce", g.ForwardLazy(0, x[0])[None])[0], y)
    assert np.array_equal(g.BackwardLazy(0, y), x[0])
    # wire format round trip through a device poly
    p = g.NewPoly().upload(x)
    p2 = g.NewPoly()
    p2.UnmarshalBinary(p.MarshalBinary())
    assert np.array_equal(p2.get(), x)


def test_tables_match_oracle(ctx):
    pr = Pair(ctx, 11, 3)
    for i in range(3):
        oc = pr.oQ.constants(i)
        assert pr.gQ.constant(i, 0) == oc["q"] and pr.gQ.constant(i, 1) == oc["qinv"]
        assert (pr.gQ.constant(i, 2), pr.gQ.constant(i, 3)) == oc["brc"]
        assert pr.gQ.constant(i, 4) == oc["ninv"] and pr.gQ.constant(i, 5) == oc["primroot"]
        assert np.array_equal(pr.gQ.roots(i), pr.oQ.roots_forward(i))
        assert np.array_equal(pr.gQ.roots(i, True), pr.oQ.roots_backward(i))


@pytest.mark.parametrize("logN", [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14])
def test_ntt_matches_oracle(ctx, logN):
    pr = Pair(ctx, logN, 3)
    rng = rng_for(1000 + logN)
    x = np.stack([uniform_poly(rng, pr.q, pr.N) for _ in range(2)])  # batch of 2
    px, py = pr.up(pr.gQ, x, 2), la.Poly(pr.gQ, 3, 2)
    pr.gQ.NTT(px, py)
    want = np.stack([pr.oQ.NTT(x[b]) for b in range(2)])
    assert np.array_equal(py.get(), want)
    pr.gQ.NTTLazy(px, py)
    lz = py.get()
    assert np.all(lz < 2 * np.array(pr.q, dtype=np.uint64)[None, :, None])
    assert np.array_equal(np.stack([pr.oQ.unop("Reduce", lz[b]) for b in range(2)]), want)
    pz = la.Poly(pr.gQ, 3, 2)
    pr.gQ.INTT(py, pz)
    assert np.array_equal(pz.get(), x)
    pr.gQ.INTTLazy(py, pz)
    assert np.array_equal(pz.get(), x)
    # at a lower level only the first limbs are touched
    pw = la.Poly(pr.gQ, 3, 2)
    pr.gQ.AtLevel(1).NTT(px, pw)
    got = pw.get()
    assert np.array_equal(got[:, :2], want[:, :2]) and not got[:, 2].any()


@pytest.mark.parametrize("logN", [15, 16])
def test_ntt_large_roundtrip_and_linearity(ctx, logN):
    """Full-size property checks, and every limb against the oracle (it takes milliseconds per limb)."""
    pr = Pair(ctx, logN, 4)
    rng = rng_for(1100 + logN)
    x, y = uniform_poly(rng, pr.q, pr.N), uniform_poly(rng, pr.q, pr.N)
    px, py, ps = pr.up(pr.gQ, x), pr.up(pr.gQ, y), pr.gQ.NewPoly()
    fx, fy, fs = pr.gQ.NewPoly(), pr.gQ.NewPoly(), pr.gQ.NewPoly()
    pr.gQ.NTT(px, fx)
    pr.gQ.NTT(py, fy)
    pr.gQ.Add(px, py, ps)
    pr.gQ.NTT(ps, fs)
    pr.gQ.Add(fx, fy, fy)
    assert np.array_equal(fs.get(), fy.get())  # linearity
    assert np.array_equal(fx.get(), pr.oQ.NTT(x))  # oracle, every limb
    pr.gQ.INTT(fx, fx)
    assert np.array_equal(fx.get(), x)  # round trip
    # negacyclic convolution theorem: NTT(x * X) = NTT(x) .* NTT(X)
    mono = np.zeros_like(x)
    mono[:, 1] = 1
    pm, fm = pr.up(pr.gQ, mono), pr.gQ.NewPoly()
    pr.gQ.NTT(pm, fm)
    pr.gQ.NTT(px, fx)
    pr.gQ.MulCoeffsBarrett(fx, fm, fm)
    pr.gQ.INTT(fm, fm)
    shifted = np.roll(x, 1, axis=1)
    shifted[:, 0] = (np.array(pr.q, dtype=np.uint64) - x[:, -1]) % np.array(pr.q, dtype=np.uint64)
    assert np.array_equal(fm.get(), shifted)


@pytest.mark.parametrize("logN", [17, 18, 19, 20])
def test_ntt_up_to_the_reference_max_logn(ctx, logN):
    """core/rlwe/params.go:21 MaxLogN = 20.  logN = 18 runs five column stages in one pass, 19 and 20 take an outer and an inner
    column pass before the rows (ntt_cols_kernel, NttArgs::cb).  All three modulus classes (double-precision, correction-free
    integer, [0, 4q) integer), batch of 2, every limb against the oracle; NTTLazy range; round trip; in place."""
    q, _ = O.GenModuli(logN + 1, [60, 45, 55, 40], [])
    pr = Pair(ctx, logN, len(q), qmods=q)
    rng = rng_for(1700 + logN)
    x = np.stack([uniform_poly(rng, pr.q, pr.N) for _ in range(2)])
    px, py = pr.up(pr.gQ, x, 2), la.Poly(pr.gQ, len(q), 2)
    pr.gQ.NTT(px, py)
    want = np.stack([pr.oQ.NTT(x[b]) for b in range(2)])
    assert np.array_equal(py.get(), want)
    pr.gQ.NTTLazy(px, py)
    lz = py.get()
    assert np.all(lz < 2 * np.array(pr.q, dtype=np.uint64)[None, :, None])
    assert np.array_equal(np.stack([pr.oQ.unop("Reduce", lz[b]) for b in range(2)]), want)
    pz = la.Poly(pr.gQ, len(q), 2)
    pr.gQ.INTT(py, pz)
    assert np.array_equal(pz.get(), x)
    pr.gQ.INTTLazy(py, pz)
    assert np.array_equal(pr.oQ.unop("Reduce", pz.get()[1]), x[1])
    pr.gQ.NTT(px, px)  # in place
    assert np.array_equal(px.get(), want)
    # non-canonical input words (the reference's NTT takes any word below overflow)
    big = x[0] + 3 * np.array(pr.q, dtype=np.uint64)[:, None]
    pb = pr.up(pr.gQ, big)
    pr.gQ.NTT(pb, pb)
    assert np.array_equal(pb.get(), want[0])
    po = la.Poly(pr.gQ, len(q), 2)
    pr.gQ.DivRoundByLastModulusNTT(py.upload(want), po)
    assert np.array_equal(po.get()[1, : len(q) - 1], pr.oQ.DivRoundByLastModulusNTT(want[1]))


def test_ntt_edge_inputs(ctx):
    """zeros, q-1 everywhere, and non-canonical inputs (the reference NTT accepts any word below overflow)."""
    pr = Pair(ctx, 10, 2)
    q = np.array(pr.q, dtype=np.uint64)[:, None]
    for x in (np.zeros((2, pr.N), dtype=np.uint64), np.broadcast_to(q - 1, (2, pr.N)).copy(),
              np.broadcast_to(3 * q + 5, (2, pr.N)).copy()):
        px, py = pr.up(pr.gQ, x), pr.gQ.NewPoly()
        pr.gQ.NTT(px, py)
        assert np.array_equal(py.get(), pr.oQ.NTT(x))


BIN = ["Add", "AddLazy", "Sub", "SubLazy", "MulCoeffsBarrett", "MulCoeffsBarrettLazy", "MulCoeffsBarrettThenAdd",
       "MulCoeffsBarrettThenAddLazy", "MulCoeffsMontgomery", "MulCoeffsMontgomeryLazy",
       "MulCoeffsMontgomeryLazyThenNeg", "MulCoeffsMontgomeryThenAdd", "MulCoeffsMontgomeryThenAddLazy",
       "MulCoeffsMontgomeryLazyThenAddLazy", "MulCoeffsMontgomeryThenSub", "MulCoeffsMontgomeryThenSubLazy",
       "MulCoeffsMontgomeryLazyThenSubLazy"]


def test_all_coefficient_wise_ops_word_exact(ctx):
    """Every ring/vec_ops.go formula, including the *Lazy representatives, word for word."""
    pr = Pair(ctx, 10, 4)
    rng = rng_for(1200)
    a, b, c = (uniform_poly(rng, pr.q, pr.N) for _ in range(3))
    # edge operands in the first columns
    q = np.array(pr.q, dtype=np.uint64)
    a[:, 0], b[:, 0] = 0, 0
    a[:, 1], b[:, 1] = q - 1, q - 1
    a[:, 2], b[:, 2] = 1, q - 1
    pa, pb = pr.up(pr.gQ, a), pr.up(pr.gQ, b)
    for name in BIN:
        pc = pr.up(pr.gQ, c)
        pr.gQ.binop(name, pa, pb, pc)
        assert np.array_equal(pc.get(), pr.oQ.binop(name, a, b, c)), name
    wide = a.copy()
    wide[:, 3] = 0xFFFFFFFFFFFFFFFF
    pw = pr.up(pr.gQ, wide)
    for name in ["Neg", "Reduce", "ReduceLazy", "MForm", "MFormLazy", "IMForm"]:
        src, psrc = (wide, pw) if "Reduce" in name else (a, pa)
        pc = pr.gQ.NewPoly()
        pr.gQ.unop(name, psrc, pc)
        assert np.array_equal(pc.get(), pr.oQ.unop(name, src)), name
    for name in ["AddScalar", "SubScalar", "MulScalar", "MulScalarThenAdd", "MulScalarThenSub"]:
        for scalar in (0, 1, 12345678901234567, int(q[0]) - 1):
            pc = pr.up(pr.gQ, c)
            pr.gQ.scalarop(name, pa, scalar, pc)
            assert np.array_equal(pc.get(), pr.oQ.scalarop(name, a, scalar, c)), (name, scalar)
    big = (1 << 300) + 987654321
    for name in ["AddScalarBigint", "SubScalarBigint", "MulScalarBigint"]:
        pc = pr.gQ.NewPoly()
        getattr(pr.gQ, name)(pa, big, pc)
        assert np.array_equal(pc.get(), getattr(pr.oQ, name)(a, big)), name
    sc = np.array([O.MForm(7 + i, m) for i, m in enumerate(pr.q)], dtype=np.uint64)
    pc = pr.gQ.NewPoly()
    pr.gQ.MulRNSScalarMontgomery(pa, sc, pc)
    assert np.array_equal(pc.get(), pr.oQ.MulRNSScalarMontgomery(a, sc))
    # in place + lower level
    pr.gQ.AtLevel(2).Add(pa, pb, pa)
    got = pa.get()
    assert np.array_equal(got[:3], pr.oQ.binop("Add", a, b)[:3]) and np.array_equal(got[3], a[3])


@pytest.mark.parametrize("mods", ["same", "mixed"])
def test_rescale_all_variants(ctx, mods):
    """ring/scaling.go: every DivRound/DivFloor variant, NTT and coefficient domain, nb = 1..3."""
    qm = Qi60[:5] if mods == "same" else None
    if mods == "mixed":  # CKKS-like chain: a large q0 and small rescaling primes (q_L << q_i and q_L >> q_i)
        qs, _ = O.GenModuli(12, [55, 36, 36, 60, 36], [])
        qm = qs
    pr = Pair(ctx, 11, 5, qmods=qm)
    rng = rng_for(1300)
    x = uniform_poly(rng, pr.q, pr.N)
    px = pr.up(pr.gQ, x)
    for name in ["DivRoundByLastModulusNTT", "DivRoundByLastModulus", "DivFloorByLastModulusNTT", "DivFloorByLastModulus"]:
        po = pr.gQ.NewPoly()
        getattr(pr.gQ, name)(px, po)
        assert np.array_equal(po.get()[:4], getattr(pr.oQ, name)(x)), name
    for name in ["DivRoundByLastModulusManyNTT", "DivRoundByLastModulusMany", "DivFloorByLastModulusManyNTT",
                 "DivFloorByLastModulusMany"]:
        for nb in (0, 1, 2, 3):
            po = pr.gQ.NewPoly()
            getattr(pr.gQ, name)(nb, px, po)
            assert np.array_equal(po.get()[: 5 - nb], getattr(pr.oQ, name)(nb, x)), (name, nb)
    # in place and at a lower level
    pin = pr.up(pr.gQ, x)
    pr.gQ.AtLevel(3).DivRoundByLastModulusNTT(pin, pin)
    sub = O.Ring(pr.N, pr.q[:4])
    assert np.array_equal(pin.get()[:3], sub.DivRoundByLastModulusNTT(x[:4]))


@pytest.mark.parametrize("logN", [13, 14, 15, 16])
def test_rescale_two_pass_rings_batched_in_place(ctx, logN):
    """DivRound/DivFloorByLastModulusNTT where the transform is column pass + row pass (logN > 12): the step is two fused
    transforms (scalar prologue, MRed epilogue through a scratch intermediate), on a chain that mixes the three modulus classes,
    with a batch, out of place and in place (p1 aliasing p0), down the whole chain."""
    qs, _ = O.GenModuli(logN + 1, [60, 45, 36, 58, 55, 40], [])
    pr = Pair(ctx, logN, len(qs), qmods=qs)
    rng = rng_for(1310 + logN)
    B = 3
    xs = np.stack([uniform_poly(rng, pr.q, pr.N) for _ in range(B)])
    for name in ("DivRoundByLastModulusNTT", "DivFloorByLastModulusNTT"):
        cur = xs.copy()
        pin = pr.up(pr.gQ, cur, batch=B)
        for level in range(len(qs) - 1, 0, -1):
            sub = O.Ring(pr.N, pr.q[: level + 1])
            want = np.stack([getattr(sub, name)(cur[b, : level + 1]) for b in range(B)])
            po = la.Poly(pr.gQ, len(qs), B)
            getattr(pr.gQ.AtLevel(level), name)(pin, po)
            assert np.array_equal(po.get()[:, :level], want), (name, level, "out of place")
            getattr(pr.gQ.AtLevel(level), name)(pin, pin)
            assert np.array_equal(pin.get()[:, :level], want), (name, level, "in place")
            cur = np.concatenate([want, cur[:, level:]], axis=1)


def test_automorphism(ctx):
    pr = Pair(ctx, 11, 3)
    rng = rng_for(1400)
    x, acc = uniform_poly(rng, pr.q, pr.N), uniform_poly(rng, pr.q, pr.N)
    px = pr.up(pr.gQ, x)
    for galel in (5, pow(5, 77, 2 * pr.N), 2 * pr.N - 1, 1):
        idx = pr.gQ.AutomorphismNTTIndex(galel)
        assert np.array_equal(idx.download(), pr.oQ.AutomorphismNTTIndex(galel))
        po = pr.gQ.NewPoly()
        pr.gQ.AutomorphismNTTWithIndex(px, idx, po)
        assert np.array_equal(po.get(), pr.oQ.AutomorphismNTTWithIndex(x, idx.download()))
        pa = pr.up(pr.gQ, acc)
        pr.gQ.AutomorphismNTTWithIndexThenAddLazy(px, idx, pa)
        assert np.array_equal(pa.get(), pr.oQ.AutomorphismNTTWithIndexThenAddLazy(x, idx.download(), acc))
        xz = x.copy()
        xz[:, :4] = 0  # negated zeros come out as q in the reference
        pz, po2 = pr.up(pr.gQ, xz), pr.gQ.NewPoly()
        pr.gQ.Automorphism(pz, galel, po2)
        assert np.array_equal(po2.get(), pr.oQ.Automorphism(xz, galel))
    with pytest.raises(la.HeringError):
        pr.gQ.AutomorphismNTTWithIndex(px, pr.gQ.AutomorphismNTTIndex(5), px)  # not in place (automorphism.go:37)


def test_error_behaviour(ctx):
    with pytest.raises(la.HeringError):
        la.Ring(ctx, 1 << 10, [Qi60[0], Qi60[0]])  # not distinct
    with pytest.raises(la.HeringError):
        la.Ring(ctx, 1 << 10, [0x1fffffffffe00001 + 2])  # not prime / not 1 mod 2N
    with pytest.raises(la.HeringError):
        la.Ring(ctx, 1 << 10, [])
    r = la.Ring(ctx, 1 << 10, Qi60[:2])
    small = la.Poly(r, 1, 1)
    with pytest.raises(la.HeringError):
        r.NTT(small, small)  # needs 2 limbs at level 1
    with pytest.raises(ValueError):
        r.AtLevel(5)


@pytest.mark.parametrize("logN", [4, 7, 10, 12, 13])
def test_conjugate_invariant_ntt(ctx, logN):
    """ring/ntt.go:716-1311: conjugate-invariant NTT vs the oracle, bit-exact on strict
    outputs, plus the reference's own property (ring/ring_test.go:85-126) against a standard 2N ring."""
    N, Q = 1 << logN, Qi60[:3]
    g, o = la.Ring(ctx, N, Q, conjugate_invariant=True), O.Ring(N, Q, conjugate_invariant=True)
    rng = rng_for(1500 + logN)
    x = np.stack([uniform_poly(rng, Q, N) for _ in range(2)])
    px, py = la.Poly(g, 3, 2).upload(x), la.Poly(g, 3, 2)
    g.NTT(px, py)
    want = np.stack([o.NTT(x[b]) for b in range(2)])
    assert np.array_equal(py.get(), want)
    g.NTTLazy(px, py)
    assert np.array_equal(np.stack([o.unop("Reduce", v) for v in py.get()]), want)
    pz = la.Poly(g, 3, 2)
    g.INTT(py, pz)
    assert np.array_equal(pz.get(), x)
    g.NTT(px, px)  # in place
    assert np.array_equal(px.get(), want)
    # squaring agrees with the unfolded polynomial in the standard ring of degree 2N
    g2 = la.Ring(ctx, 2 * N, Q)
    p2 = np.zeros((3, 2 * N), dtype=np.uint64)
    p2[:, :N] = x[0]
    for i, qi in enumerate(Q):
        p2[i, N + 1:] = (np.uint64(qi) - x[0][i, 1:][::-1]) % np.uint64(qi)
    q2 = la.Poly(g2, 3).upload(p2)
    g2.NTT(q2, q2)
    g2.MulCoeffsBarrett(q2, q2, q2)
    g2.INTT(q2, q2)
    t = la.Poly(g, 3).upload(x[0])
    g.NTT(t, t)
    g.MulCoeffsBarrett(t, t, t)
    g.INTT(t, t)
    assert np.array_equal(t.get(), q2.get()[:, :N])


def test_conjugate_invariant_rescale(ctx):
    """ring/scaling.go on a conjugate-invariant ring.  The NTT variants feed the LAZY words of INTTConjugateInvariantLazy
    (ring/ntt.go:1104-1152: a representative in [0, 2q), about one word in a thousand at or above q) into the other moduli, so
    the representative is observable: the device reproduces the reference's words (a value off by q_L moves the quotient by
    one).  Several polynomials so that such words are certain to occur."""
    N, Q = 1 << 10, Qi60[:4]
    g, o = la.Ring(ctx, N, Q, conjugate_invariant=True), O.Ring(N, Q, conjugate_invariant=True)
    rng = rng_for(1600)
    lazy_words = 0
    for trial in range(12):
        x = uniform_poly(rng, Q, N)
        lazy_words += int((o.INTTLazy(x)[3] >= np.uint64(Q[3])).sum())
        px = la.Poly(g, 4).upload(x)
        for name in ["DivRoundByLastModulusNTT", "DivFloorByLastModulusNTT", "DivRoundByLastModulus", "DivFloorByLastModulus"]:
            po = g.NewPoly()
            getattr(g, name)(px, po)
            assert np.array_equal(po.get()[:3], getattr(o, name)(x)), (name, trial)
        po = g.NewPoly()
        g.DivRoundByLastModulusManyNTT(2, px, po)
        assert np.array_equal(po.get()[:2], o.DivRoundByLastModulusManyNTT(2, x))
        g.DivFloorByLastModulusManyNTT(1, px, po)
        assert np.array_equal(po.get()[:3], o.DivFloorByLastModulusManyNTT(1, x))
    assert lazy_words > 0  # the case the exact representative is about did occur


def test_remaining_ring_operations(ctx):
    """The rest of ring/operations.go: {Add,Sub,Mul}DoubleRNSScalar[ThenAdd] (:166-184, 249-268), MulScalarBigintThenAdd
    (:240), EvalPolyScalar (:271), Shift (:279, incl. the reference's known answer ring_test.go:917), MultByMonomial
    (:307, every sign case, in place), MulByVectorMontgomery[ThenAddLazy] (:363-377), AutomorphismNTT; word-exact, batch 2."""
    logN, nq, B = 9, 4, 2
    pr = Pair(ctx, logN, nq, qmods=[Qi60[0], Qi60[1], Pi60[0], Qi60[5]])
    N, q = pr.N, pr.q
    rng = rng_for(1900)
    x = np.stack([uniform_poly(rng, q, N) for _ in range(B)])
    y = np.stack([uniform_poly(rng, q, N) for _ in range(B)])
    x[:, :, :3] = 0
    px = la.Poly(pr.gQ, nq, B).upload(x)
    s0 = np.array([int(rng.integers(0, int(m))) for m in q], dtype=np.uint64)
    s1 = np.array([int(rng.integers(0, int(m))) for m in q], dtype=np.uint64)
    for level in (nq - 1, 1):
        g, o = pr.gQ.AtLevel(level), O.Ring(N, q[: level + 1])
        xs, ys = x[:, : level + 1], y[:, : level + 1]

        def run(fn, *a, init=None):
            out = la.Poly(pr.gQ, nq, B)
            if init is not None:
                out.upload(init)
            fn(px, *a, out)
            return out.download()[:, : level + 1]

        for name in ("AddDoubleRNSScalar", "SubDoubleRNSScalar", "MulDoubleRNSScalar"):
            got = run(getattr(g, name), s0, s1)
            for b in range(B):
                assert np.array_equal(got[b], getattr(o, name)(xs[b], s0, s1)), (name, level, b)
        got = run(g.MulDoubleRNSScalarThenAdd, s0, s1, init=y)
        big = (1 << 130) + 987654321
        got2 = run(g.MulScalarBigintThenAdd, big, init=y)
        for b in range(B):
            assert np.array_equal(got[b], o.MulDoubleRNSScalarThenAdd(xs[b], s0, s1, ys[b])), (level, b)
            assert np.array_equal(got2[b], o.MulScalarBigintThenAdd(xs[b], big, ys[b])), (level, b)
        for k in (0, 3, N - 1, N, -5, 2 * N + 7):
            got = run(g.Shift, k)
            for b in range(B):
                assert np.array_equal(got[b], o.Shift(xs[b], k)), ("shift", k)
        for k in (0, 1, 9, N - 1, N, N + 3, 2 * N - 1, 2 * N, -1, -N - 2):
            got = run(g.MultByMonomial, k)
            for b in range(B):
                assert np.array_equal(got[b], o.MultByMonomial(xs[b], k)), ("monomial", k)
        tmp = la.Poly(pr.gQ, nq, B).upload(x)  # in place, as ring_test.go:897 (MultByMonomial(p3Test, 8, p3Test))
        g.MultByMonomial(tmp, 1, tmp)
        g.MultByMonomial(tmp, 8, tmp)
        g.Shift(tmp, 5, tmp)
        for b in range(B):
            assert np.array_equal(tmp.download()[b, : level + 1], o.Shift(o.MultByMonomial(xs[b], 9), 5))
        vec = rng.integers(0, 1 << 62, size=N, dtype=np.uint64)
        pv = la.Poly(pr.gQ, 1).upload(vec[None, :])
        got = run(g.MulByVectorMontgomery, pv)
        got2 = run(g.MulByVectorMontgomeryThenAddLazy, pv, init=y)
        for b in range(B):
            assert np.array_equal(got[b], o.MulByVectorMontgomery(xs[b], vec))
            assert np.array_equal(got2[b], o.MulByVectorMontgomery(xs[b], vec, ys[b]))
        polys = [np.stack([uniform_poly(rng, q, N) for _ in range(B)]) for _ in range(4)]
        gp = [la.Poly(pr.gQ, nq, B).upload(p) for p in polys]
        out = la.Poly(pr.gQ, nq, B)
        g.EvalPolyScalar(gp, 12345678901, out)
        got = out.download()[:, : level + 1]
        for b in range(B):
            assert np.array_equal(got[b], o.EvalPolyScalar([p[b, : level + 1] for p in polys], 12345678901))
        out = la.Poly(pr.gQ, nq, B)
        g.AutomorphismNTT(px, 2 * N - 1, out)
        assert np.array_equal(out.download()[0, : level + 1], o.AutomorphismNTT(xs[0], 2 * N - 1))
    r16 = la.Ring(ctx, 16, [97])
    p2 = r16.NewPoly()
    r16.Shift(r16.NewPoly().upload(np.arange(16, dtype=np.uint64)[None, :]), 3, p2)
    assert p2.get()[0].tolist() == [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2]  # ring_test.go:917
    with pytest.raises(la.HeringError):
        pr.gQ.MulByVectorMontgomery(px, px, la.Poly(pr.gQ, nq, B))  # the vector must be a batch-1 polynomial


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [2048, 2049])
def test_inverse_ntt_two_entries_per_workgroup(ctx, batch):
    """The inverse double-precision row kernel pairs batch entries per workgroup (second one prefetched) once a launch has
    at least 6144 workgroups: 3 limbs x `batch` rows here.  Odd batch = a last workgroup with a single entry.  Every entry
    against the oracle, and the forward transform back."""
    logN = 12
    q, _ = O.GenModuli(logN + 1, [45, 44, 40], [])
    N = 1 << logN
    gq, oq = la.Ring(ctx, N, q), O.Ring(N, q)
    rng = rng_for(4400 + batch)
    x = np.stack([uniform_poly(rng, q, N) for _ in range(batch)])
    px, py = la.Poly(gq, 3, batch).upload(x), la.Poly(gq, 3, batch, zero=False)
    gq.INTT(px, py)
    got = py.download()
    for b in (0, 1, 2, 3, batch // 2, batch - 2, batch - 1):
        assert np.array_equal(got[b], oq.INTT(x[b])), b
    # all entries: INTT is injective, so the round trip pins the ones not compared above
    gq.NTT(py, py)
    assert np.array_equal(py.download(), x)


@pytest.mark.gpu
def test_transforms_and_rescale_logN14_large_batch(ctx):
    """Stand-alone transforms and the fused rescale at logN = 14 with a large batch (96 entries x 4 limbs, all three modulus
    classes): NTT / INTT / NTTLazy and DivRound / DivFloorByLastModulusNTT, sampled entries against the oracle, the round trip
    and the in-place rescale on every entry.  (Written for the one-pass variant of these transforms -- the whole limb resident in
    a 1024-thread workgroup's LDS -- which measured slower than the two passes and was removed; kept as the large-batch case.)"""
    logN, B = 14, 96
    qs, _ = O.GenModuli(logN + 1, [60, 45, 55, 40], [])
    pr = Pair(ctx, logN, len(qs), qmods=qs)
    rng = rng_for(1320)
    x = np.stack([uniform_poly(rng, pr.q, pr.N) for _ in range(B)])
    px, py, pz = pr.up(pr.gQ, x, batch=B), la.Poly(pr.gQ, len(qs), B), la.Poly(pr.gQ, len(qs), B)
    pr.gQ.NTT(px, py)
    got = py.get()
    keep = (0, B // 2, B - 1)
    for b in keep:
        assert np.array_equal(got[b], pr.oQ.NTT(x[b])), b
    pr.gQ.INTT(py, pz)
    assert np.array_equal(pz.get(), x)
    pr.gQ.NTTLazy(px, pz)
    lz = pz.get()
    assert np.all(lz < 2 * np.array(pr.q, dtype=np.uint64)[None, :, None])
    for b in keep:
        assert np.array_equal(pr.oQ.unop("Reduce", lz[b]), got[b])
    for name in ("DivRoundByLastModulusNTT", "DivFloorByLastModulusNTT"):
        po = la.Poly(pr.gQ, len(qs), B)
        getattr(pr.gQ, name)(py, po)
        res = po.get()
        for b in keep:
            assert np.array_equal(res[b, : len(qs) - 1], getattr(pr.oQ, name)(got[b])), (name, b)
        pin = la.Poly(pr.gQ, len(qs), B).upload(got)
        getattr(pr.gQ, name)(pin, pin)  # in place
        assert np.array_equal(pin.get()[:, : len(qs) - 1], res[:, : len(qs) - 1]), name

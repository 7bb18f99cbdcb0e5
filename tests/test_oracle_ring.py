"""Pins the CPU oracle (oracle/lattigo_oracle.c) against the reference's own
known-answer vectors and big-integer property tests:

* ring/ntt_test.go:10-119   -- 12 NTT known-answer vectors + round trip
* ring/ring_test.go:537-673 -- BRed / MRed edge operands vs math/big
* ring/ring_test.go:675-687 -- MForm o IMForm
* ring/ring_test.go:245-334 -- DivFloor/DivRound ByLastModulusMany vs big-int division
* ring/ring_test.go:714-886 -- ModUp Q->P, P->Q, ModDown QP->Q, QP->P vs big-int
All comparisons are exact uint64 equality.  No GPU needed.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import (bitrev, div_round, prod, rand_bigints, rng_for,
                           set_coefficients_bigint, uniform_poly)

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ntt_kat.json")


@pytest.fixture(scope="module")
def rings(ring_test_params):
    N = 1 << ring_test_params["logN"]
    return O.Ring(N, ring_test_params["qi"]), O.Ring(N, ring_test_params["pi"])


def test_ntt_known_answer_vectors():
    kat = json.load(open(GOLDEN))
    assert len(kat["cases"]) == 6
    for c in kat["cases"]:
        r = O.Ring(c["N"], c["Qis"])
        x = np.array(c["poly"], dtype=np.uint64)
        y = np.array(c["polyNTT"], dtype=np.uint64)
        z = r.NTT(x)
        assert np.array_equal(z, y), f"N={c['N']}: NTT(poly) != polyNTT"
        assert np.array_equal(r.INTT(z), x), f"N={c['N']}: INTT(NTT(x)) != x"
        # lazy forms agree after Reduce (ring.Equal, ring/ring.go:512-524)
        assert np.array_equal(r.unop("Reduce", r.NTTLazy(x)), y)
        assert np.array_equal(r.unop("Reduce", r.INTTLazy(y)), x)


def test_tables_match_definition(rings):
    """RootsForward[bitrev(j)] = MForm(psi^j), ring/subring.go:142-156."""
    ringQ, _ = rings
    N = ringQ.N
    for i in (0, 5):
        c = ringQ.constants(i)
        q, g = c["q"], c["primroot"]
        assert c["qinv"] * q % (1 << 64) == 1
        assert c["brc"][0] * (1 << 64) + c["brc"][1] == (1 << 128) // q
        # g is the smallest generator >= 3
        fac = set()
        m, p = q - 1, 2
        while m % 2 == 0:
            fac.add(2); m //= 2
        p = 3
        while p * p <= m and p < 1 << 20:
            while m % p == 0:
                fac.add(p); m //= p
            p += 2
        if m > 1:
            fac.add(m)
        is_gen = lambda a: all(pow(a, (q - 1) // f, q) != 1 for f in fac)
        assert is_gen(g) and not any(is_gen(a) for a in range(3, g))
        psi = pow(g, (q - 1) // (2 * N), q)
        rf, rb = ringQ.roots_forward(i), ringQ.roots_backward(i)
        R = (1 << 64) % q
        logN = N.bit_length() - 1
        for j in (0, 1, 2, 3, N // 2, N - 1):
            assert int(rf[bitrev(j, logN)]) == pow(psi, j, q) * R % q
            assert int(rb[bitrev(j, logN)]) == pow(psi, -j, q) * R % q
        assert c["ninv"] == pow(N, -1, q) * R % q


def test_bred_mred_edge_cases(ring_test_params):
    F = 0xFFFFFFFFFFFFFFFF
    for q in ring_test_params["qi"]:
        for x, y in [(1, 1), (1, q - 1), (1, F), (q - 1, q - 1), (q - 1, F), (F, F)]:
            assert O.BRed(x, y, q) == x * y % q
            assert O.MRed(x, O.MForm(y, q), q) == x * y % q
            assert O.BRedLazy(x, y, q) % q == x * y % q and O.BRedLazy(x, y, q) < 2 * q
        for a in (0, 1, q - 1, q, 2 * q + 5, F):
            assert O.BRedAdd(a, q) == a % q
            assert O.BRedAddLazy(a, q) % q == a % q
            assert O.IMForm(O.MForm(a % q, q), q) == a % q
        assert O.MRedLazy(q - 1, q - 1, q) < 2 * q


def test_mform_imform_roundtrip(rings):
    ringQ, _ = rings
    x = uniform_poly(rng_for(100), ringQ.moduli, ringQ.N)
    assert np.array_equal(ringQ.unop("IMForm", ringQ.unop("MForm", x)), x)


def test_vec_ops_against_bigint(rings):
    ringQ, _ = rings
    rng = rng_for(101)
    a, b, c = (uniform_poly(rng, ringQ.moduli, ringQ.N) for _ in range(3))
    q = np.array(ringQ.moduli, dtype=object)[:, None]
    A, B, Cc = a.astype(object), b.astype(object), c.astype(object)
    Rinv = np.array([pow(1 << 64, -1, int(m)) for m in ringQ.moduli], dtype=object)[:, None]
    eq = lambda got, want: np.array_equal(got.astype(object), want)
    assert eq(ringQ.binop("Add", a, b), (A + B) % q)
    assert eq(ringQ.binop("Sub", a, b), (A - B) % q)
    assert eq(ringQ.unop("Neg", a) % np.array(ringQ.moduli, dtype=np.uint64)[:, None], (-A) % q)
    assert eq(ringQ.binop("MulCoeffsBarrett", a, b), A * B % q)
    assert eq(ringQ.binop("MulCoeffsMontgomery", a, b), A * B * Rinv % q)
    assert eq(ringQ.binop("MulCoeffsMontgomeryThenAdd", a, b, c), (Cc + A * B * Rinv) % q)
    assert eq(ringQ.binop("MulCoeffsMontgomeryThenSub", a, b, c), (Cc - A * B * Rinv) % q)
    assert eq(ringQ.binop("MulCoeffsBarrettThenAdd", a, b, c), (Cc + A * B) % q)
    lazy = ringQ.binop("MulCoeffsMontgomeryLazy", a, b).astype(object)
    assert np.all(lazy < 2 * q) and np.array_equal(lazy % q, A * B * Rinv % q)
    acc = ringQ.binop("MulCoeffsMontgomeryLazyThenAddLazy", a, b, c).astype(object)
    assert np.array_equal(acc % q, (Cc + A * B * Rinv) % q)
    assert eq(ringQ.scalarop("MulScalar", a, 12345), A * 12345 % q)
    assert eq(ringQ.scalarop("AddScalar", a, 12345), (A + 12345) % q)
    assert eq(ringQ.scalarop("SubScalar", a, 12345), (A - 12345) % q)
    assert eq(ringQ.scalarop("MulScalarThenAdd", a, 777, c), (Cc + A * 777) % q)
    assert eq(ringQ.scalarop("MulScalarThenSub", a, 777, c), (Cc - A * 777) % q)
    big = (1 << 200) + 12345
    assert eq(ringQ.MulScalarBigint(a, big), A * big % q)
    assert eq(ringQ.AddScalarBigint(a, big), (A + big) % q)


@pytest.mark.parametrize("kind", ["floor", "round"])
def test_div_by_last_modulus_many(rings, kind):
    """ring/ring_test.go:245-334."""
    ringQ, _ = rings
    N, level = ringQ.N, ringQ.MaxLevel()
    Q = prod(ringQ.moduli)
    coeffs = [c // 10 for c in rand_bigints(rng_for(102), Q, N)]
    nb = level
    want = []
    for c in coeffs:
        for j in range(nb):
            qj = ringQ.moduli[level - j]
            c = c // qj if kind == "floor" else div_round(c, qj)
        want.append(c)
    p0 = set_coefficients_bigint(coeffs, ringQ.moduli)
    pw = set_coefficients_bigint(want, ringQ.moduli[: level - nb + 1])
    fn = ringQ.DivFloorByLastModulusMany if kind == "floor" else ringQ.DivRoundByLastModulusMany
    got = fn(nb, p0)
    assert np.array_equal(got, pw)
    # NTT-domain variants agree with the coefficient-domain ones
    fn_ntt = ringQ.DivFloorByLastModulusManyNTT if kind == "floor" else ringQ.DivRoundByLastModulusManyNTT
    for nb2 in (1, 2, 3):
        ref = (ringQ.DivFloorByLastModulusMany if kind == "floor" else ringQ.DivRoundByLastModulusMany)(nb2, p0)
        got2 = fn_ntt(nb2, ringQ.NTT(p0))
        sub = O.Ring(N, ringQ.moduli[: level + 1 - nb2])
        assert np.array_equal(sub.INTT(got2), ref)


def _centered(rng, Q, N):
    half = Q >> 1
    return [c - half for c in rand_bigints(rng, Q, N)]


def test_modup_q_to_p_and_p_to_q(rings):
    """ring/ring_test.go:714-792: exact vs big-int, after Reduce."""
    ringQ, ringP = rings
    N = ringQ.N
    be = O.BasisExtender(ringQ, ringP)
    levelQ, levelP = ringQ.MaxLevel() - 1, ringP.MaxLevel() - 1
    Qm, Pm = ringQ.moduli[: levelQ + 1], ringP.moduli[: levelP + 1]
    coeffs = _centered(rng_for(103), prod(Qm), N)
    got = be.ModUpQtoP(levelQ, levelP, set_coefficients_bigint(coeffs, Qm))
    subP = O.Ring(N, Pm)
    assert np.array_equal(subP.unop("Reduce", got), set_coefficients_bigint(coeffs, Pm))
    coeffs = _centered(rng_for(104), prod(Pm), N)
    got = be.ModUpPtoQ(levelP, levelQ, set_coefficients_bigint(coeffs, Pm))
    subQ = O.Ring(N, Qm)
    assert np.array_equal(subQ.unop("Reduce", got), set_coefficients_bigint(coeffs, Qm))


def test_moddown_qp_to_q_and_p(rings):
    """ring/ring_test.go:794-886."""
    ringQ, ringP = rings
    N = ringQ.N
    be = O.BasisExtender(ringQ, ringP)
    levelQ, levelP = ringQ.MaxLevel() - 1, ringP.MaxLevel() - 1
    Qm, Pm = ringQ.moduli[: levelQ + 1], ringP.moduli[: levelP + 1]
    Q, P = prod(Qm), prod(Pm)
    coeffs = [c // 10 for c in rand_bigints(rng_for(105), Q * P, N)]
    pq, pp = set_coefficients_bigint(coeffs, Qm), set_coefficients_bigint(coeffs, Pm)
    subQ, subP = O.Ring(N, Qm), O.Ring(N, Pm)
    got = subQ.unop("Reduce", be.ModDownQPtoQ(levelQ, levelP, pq, pp))
    assert np.array_equal(got, set_coefficients_bigint([div_round(c, P) for c in coeffs], Qm))
    got = subP.unop("Reduce", be.ModDownQPtoP(levelQ, levelP, pq, pp))
    assert np.array_equal(got, set_coefficients_bigint([div_round(c, Q) for c in coeffs], Pm))
    # NTT variant == coefficient variant conjugated by the transforms
    got_ntt = be.ModDownQPtoQNTT(levelQ, levelP, subQ.NTT(pq), subP.NTT(pp))
    assert np.array_equal(subQ.INTT(got_ntt), subQ.unop("Reduce", be.ModDownQPtoQ(levelQ, levelP, pq, pp)))


def test_automorphism_ntt_matches_coefficient_domain(rings):
    ringQ, _ = rings
    N = ringQ.N
    x = uniform_poly(rng_for(106), ringQ.moduli, N)
    for galel in (5, pow(5, 7, 2 * N), 2 * N - 1, pow(5, N // 2 - 3, 2 * N)):
        idx = ringQ.AutomorphismNTTIndex(galel)
        assert sorted(idx.tolist()) == list(range(N))
        via_ntt = ringQ.INTT(ringQ.AutomorphismNTTWithIndex(ringQ.NTT(x), idx))
        direct = ringQ.unop("Reduce", ringQ.Automorphism(x, galel))
        assert np.array_equal(via_ntt, direct)


def test_gen_moduli_shapes():
    """core/rlwe/params.go:811: CKKS benchmark chain LogN=14, LogQ=[50,40x7], LogP=[60]."""
    q, p = O.GenModuli(15, [50] + [40] * 7, [60])
    assert len(set(q + p)) == 9
    for m, b in zip(q + p, [50] + [40] * 7 + [60]):
        assert O.IsPrime(m) and m % (1 << 15) == 1 and abs(np.log2(float(m)) - b) < 0.5
    # 61-bit primes walk downstream from 2^61
    q, p = O.GenModuli(17, [45, 45], [61, 61])
    assert p[0] > p[1] and p[0] < (1 << 61)


def test_conjugate_invariant_ntt_vs_standard_2n(ring_test_params):
    """ring/ring_test.go:85-126: squaring through the conjugate-invariant NTT of degree N equals
    squaring the unfolded polynomial through the standard NTT of degree 2N."""
    N = 1 << ring_test_params["logN"]
    Q = ring_test_params["qi"][:4]
    ring2n, ringci = O.Ring(2 * N, Q), O.Ring(N, Q, conjugate_invariant=True)
    p1 = uniform_poly(rng_for(107), Q, N)
    p2 = np.zeros((len(Q), 2 * N), dtype=np.uint64)
    p2[:, :N] = p1
    for i, qi in enumerate(Q):
        for j in range(1, N):
            p2[i, 2 * N - j] = qi - int(p1[i, j])
    p2 = ring2n.NTT(p2)
    p2 = ring2n.INTT(ring2n.binop("MulCoeffsBarrett", p2, p2))
    t = ringci.NTT(p1)
    got = ringci.INTT(ringci.binop("MulCoeffsBarrett", t, t))
    assert np.array_equal(got, p2[:, :N])
    # lazy forms are other representatives of the same residues; round trip is the identity
    assert np.array_equal(ringci.unop("Reduce", ringci.NTTLazy(p1)), t)
    assert np.array_equal(ringci.INTT(t), p1)
    assert np.array_equal(ringci.unop("Reduce", ringci.INTTLazy(t)), p1)


def test_shift_known_answer():
    """ring/ring_test.go:907-919 (testShift), the reference's literal answer"""
    r = O.Ring(16, [97])
    p1 = np.arange(16, dtype=np.uint64)[None, :]
    assert r.Shift(p1, 3)[0].tolist() == [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2]
    assert np.array_equal(r.Shift(p1, -13), r.Shift(p1, 3)) and np.array_equal(r.Shift(p1, 16), p1)


def test_mult_by_monomial(rings):
    """ring/ring_test.go:888-905 (X^1 then X^8 == X^9) + the definition in Z[X]/(X^N+1) for every sign case"""
    ringQ = rings[0]
    N, mods = ringQ.N, ringQ.moduli
    rng = rng_for(77)
    p1 = uniform_poly(rng, mods, N)
    p1[:, :3] = 0  # zeros expose the q - 0 = q representative
    assert np.array_equal(ringQ.MultByMonomial(ringQ.MultByMonomial(p1, 1), 8), ringQ.MultByMonomial(p1, 9))
    for k in (0, 1, 5, N - 1, N, N + 3, 2 * N - 1, 2 * N, -1, -N - 2):
        got = ringQ.unop("Reduce", ringQ.MultByMonomial(p1, k))
        for i, q in enumerate(mods):
            want = np.zeros(N, dtype=object)
            for j in range(N):
                e = (j + k) % (2 * N)
                sign = 1 if e < N else -1
                want[e % N] = (sign * int(p1[i, j])) % int(q)
            assert got[i].tolist() == want.tolist(), (k, i)
    if N >= 4:
        assert (ringQ.MultByMonomial(p1, N)[:, :3] == np.array(mods, dtype=np.uint64)[:, None]).all()  # q - 0


def test_double_rns_scalar_ops_and_friends(rings):
    """ring/operations.go:166-184, 240-277, 363-377 against their definitions on Python integers"""
    ringQ = rings[0]
    N, mods = ringQ.N, ringQ.moduli
    rng = rng_for(78)
    p1, p2 = uniform_poly(rng, mods, N), uniform_poly(rng, mods, N)
    s0 = np.array([int(rng.integers(0, int(q))) for q in mods], dtype=np.uint64)
    s1 = np.array([int(rng.integers(0, int(q))) for q in mods], dtype=np.uint64)
    h = N // 2
    outs = {"add": ringQ.AddDoubleRNSScalar(p1, s0, s1), "sub": ringQ.SubDoubleRNSScalar(p1, s0, s1),
            "mul": ringQ.MulDoubleRNSScalar(p1, s0, s1), "mad": ringQ.MulDoubleRNSScalarThenAdd(p1, s0, s1, p2)}
    big = int(rng.integers(1, 1 << 62)) ** 3 + 12345
    mba = ringQ.MulScalarBigintThenAdd(p1, big, p2)
    vec = rng.integers(0, 1 << 62, size=N, dtype=np.uint64)
    mv = ringQ.MulByVectorMontgomery(p1, vec)
    mva = ringQ.unop("Reduce", ringQ.MulByVectorMontgomery(p1, vec, p2))
    polys = [uniform_poly(rng, mods, N) for _ in range(4)]
    ev = ringQ.EvalPolyScalar(polys, 12345678901)
    for i, q in enumerate(mods):
        q = int(q)
        rinv = pow(1 << 64, -1, q)
        for j in (0, 1, h - 1, h, h + 1, N - 1):
            s = int(s0[i]) if j < h else int(s1[i])
            x, y = int(p1[i, j]), int(p2[i, j])
            assert int(outs["add"][i, j]) == (x + s) % q and int(outs["sub"][i, j]) == (x - s) % q
            assert int(outs["mul"][i, j]) == x * s % q and int(outs["mad"][i, j]) == (y + x * s) % q
            assert int(mba[i, j]) == (y + x * big) % q
            assert int(mv[i, j]) == x * int(vec[j]) * rinv % q and int(mva[i, j]) == (y + x * int(vec[j]) * rinv) % q
            assert int(ev[i, j]) == sum(int(polys[d][i, j]) * pow(12345678901, d, q) for d in range(4)) % q
    g = 5
    assert np.array_equal(ringQ.AutomorphismNTT(p1, g), ringQ.AutomorphismNTTWithIndex(p1, ringQ.AutomorphismNTTIndex(g)))

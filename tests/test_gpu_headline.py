"""Parity of the HEADLINE configuration itself and of the arithmetic-class boundaries (`-m gpu`).

* BASELINE config 3 exactly as bench.py runs it (same moduli, T, launch shapes): BGV ct x ct MulRelin at logN=15,
  12+3 limbs, at the batch sizes whose launch configurations differ (batch 9: odd, one entry per workgroup;
  batch 128: two-entries-per-workgroup inverse rows, XCD-swizzled NTT+MAC, wide fused ModDown; batch 256: the bench
  default -- persistent 512-workgroup NTT+MAC work list; batch 255: the same list without the XCD swizzle, ragged tail),
  every limb of EVERY batch entry against the oracle (schemes/bgv/evaluator.go:592-685; oracle.Evaluator.BatchOp on the
  host's cores), plus Relinearize(degree-2 result) == MulRelin.
* Boundary moduli of the two fast arithmetic classes: the largest NTT-friendly primes below 2^47 (double-precision
  kernels, exactness argument "34q + input < 2^53") and below 2^58 (correction-free integer butterflies,
  "34q / 36q < 2^64"; 42q at logN = 20) at logN = 16, 17 and 20, on worst-case inputs (all q-1, alternating 0 / q-1, non-canonical words up
  to 2^64-1, which this library reduces first), through Ring.NTT/INTT and through the whole key-switch pipeline.
"""
import numpy as np
import pytest

import lattigo_amd as la
from oracle import oracle as O
from tests.gpu_common import Pair, ctx  # noqa: F401
from tests.helpers import rng_for, uniform_poly

pytestmark = pytest.mark.gpu


def _bench_config():
    import bench
    q, p = bench.gen_moduli()
    return bench.LOGN, q, p, bench.T


def test_bench_moduli_are_the_restated_GenModuli():
    logN, q, p, _ = _bench_config()
    oq, op = O.GenModuli(logN + 1, [55] + [45] * 11, [55] * 3)
    assert list(oq) == q and list(op) == p


@pytest.mark.parametrize("B", [9, 128, 255, 256])
def test_full_size_config3_bgv_mulrelin_logN15(ctx, B):
    logN, q, p, t = _bench_config()
    N, L, alpha = 1 << logN, len(q), len(p)
    beta = (L + alpha - 1) // alpha
    pr = Pair(ctx, logN, L, alpha, qmods=q, pmods=p)
    rng = rng_for(3 + B)
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(beta)])
    grlk, orlk = gev.NewEvaluationKey(kq, kp), O.EvaluationKey(kq, kp)
    import bench
    ct0 = [bench.uniform(rng, q, N, (B,)) for _ in range(2)]   # [k][b][limb][N]
    ct1 = [bench.uniform(rng, q, N, (B,)) for _ in range(2)]
    a = [la.Poly(pr.gQ, L, B).upload(c) for c in ct0]
    b = [la.Poly(pr.gQ, L, B).upload(c) for c in ct1]
    out = [la.Poly(pr.gQ, L, B), la.Poly(pr.gQ, L, B)]
    gev.BGVMulRelin(L - 1, t, a, b, grlk, out)
    got = [o.get() for o in out]
    for b0 in range(0, B, 64):  # every entry, in chunks that bound the host copies
        b1 = min(B, b0 + 64)
        want = oev.BatchOp("bgv_mulrelin", np.stack([ct0[0][b0:b1], ct0[1][b0:b1]], axis=1),
                           np.stack([ct1[0][b0:b1], ct1[1][b0:b1]], axis=1), orlk, t=t)
        for k in range(2):
            assert np.array_equal(got[k][b0:b1], want[:, k]), (B, k, b0 + int(np.argwhere(got[k][b0:b1] != want[:, k])[0][0]))
    # the degree-2 result relinearised separately is the same ciphertext (core/rlwe/evaluator_evaluationkey.go:117-148)
    out3 = [la.Poly(pr.gQ, L, B) for _ in range(3)]
    gev.BGVMulRelin(L - 1, t, a, b, None, out3)
    e = B - 1
    want3 = oev.BGVMulRelin(t, np.stack([ct0[0][e], ct0[1][e]]), np.stack([ct1[0][e], ct1[1][e]]), None, False)
    assert np.array_equal(np.stack([o.get()[e] for o in out3]), want3)
    rel = [la.Poly(pr.gQ, L, B), la.Poly(pr.gQ, L, B)]
    gev.Relinearize(L - 1, out3, grlk, rel)
    assert np.array_equal(rel[0].get(), got[0]) and np.array_equal(rel[1].get(), got[1])
    # a second call on the same handles gives the same words (no state carried between launches)
    gev.BGVMulRelin(L - 1, t, a, b, grlk, out)
    assert np.array_equal(out[0].get(), got[0]) and np.array_equal(out[1].get(), got[1])


@pytest.mark.parametrize("scheme", ["bgv", "ckks"])
def test_full_size_mulrelin_aliasing_squaring_lazy_inputs(ctx, scheme):
    """The tensor-in-the-ModDown-epilogue path at the headline shape (logN = 15, 12 + 3 limbs, double-precision and integer
    limbs mixed) under the operand patterns the reference's callers use: the squaring branch (op0 is op1,
    schemes/bgv/evaluator.go:640-644, schemes/ckks/evaluator.go:812-816), outputs aliasing an input (MulRelin(res, res, res),
    circuits/ckks/mod1/mod1_evaluator.go:120-135), which must fall back to the three-output tensor kernel, and lazy input
    words in [0, 2q), which the double-precision limbs reduce before converting.  Every limb of every batch entry."""
    logN, q, p, t = _bench_config()
    N, L, alpha, B = 1 << logN, len(q), len(p), 3
    beta = (L + alpha - 1) // alpha
    pr = Pair(ctx, logN, L, alpha, qmods=q, pmods=p)
    rng = rng_for(77 + len(scheme))
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(beta)])
    grlk, orlk = gev.NewEvaluationKey(kq, kp), O.EvaluationKey(kq, kp)
    import bench
    if scheme == "bgv":
        gmul = lambda a, b, out: gev.BGVMulRelin(L - 1, t, a, b, grlk, out)
        omul = lambda a, b: oev.BGVMulRelin(t, a, b, orlk, True)
    else:
        gmul = lambda a, b, out: gev.CKKSMulRelin(L - 1, a, b, grlk, out)
        omul = lambda a, b: oev.CKKSMulRelin(a, b, orlk, True)
    lazy = lambda: np.stack([np.stack([rng.integers(0, 2 * int(m), size=N, dtype=np.uint64) for m in q]) for _ in range(B)])
    # any 64-bit word is a valid operand of MRed: x + m q for random m, up to 2^64 - 1 (the double-precision limbs take their
    # reduce-first path for these)
    wild = lambda: np.stack([np.stack([rng.integers(0, 1 << 63, size=N, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=N, dtype=np.uint64)
                                       for _ in q]) for _ in range(B)])
    ct0 = [bench.uniform(rng, q, N, (B,)), lazy()]   # [k][b][limb][N]; the second component holds words up to 2q - 1
    ct1 = [wild(), wild()]  # (b1 too: the product prologue of the inverse rows and component 1 of the epilogue meet such words)
    want = [omul(np.stack([ct0[0][e], ct0[1][e]]), np.stack([ct1[0][e], ct1[1][e]])) for e in range(B)]
    want_sq = [omul(np.stack([ct0[0][e], ct0[1][e]]), np.stack([ct0[0][e], ct0[1][e]])) for e in range(B)]

    def check(out, ref, what):
        got = [o.get() for o in out]
        for e in range(B):
            assert np.array_equal(got[0][e], ref[e][0]) and np.array_equal(got[1][e], ref[e][1]), (scheme, what, e)

    fresh = lambda ct: [la.Poly(pr.gQ, L, B).upload(c) for c in ct]
    # lazy words through the fused path (no aliasing)
    a, b, out = fresh(ct0), fresh(ct1), [la.Poly(pr.gQ, L, B), la.Poly(pr.gQ, L, B)]
    gmul(a, b, out)
    check(out, want, "lazy inputs")
    # squaring: both operands are the same handles, separate output
    gmul(a, a, out)
    check(out, want_sq, "op0 is op1")
    # output aliases the first operand / the second operand
    a, b = fresh(ct0), fresh(ct1)
    gmul(a, b, a)
    check(a, want, "out is op0")
    a, b = fresh(ct0), fresh(ct1)
    gmul(a, b, b)
    check(b, want, "out is op1")
    # crossed aliasing: out0 is op1[1], out1 is op0[0]
    a, b = fresh(ct0), fresh(ct1)
    gmul(a, b, [b[1], a[0]])
    check([b[1], a[0]], want, "crossed")
    # in-place squaring, as mod1's double-angle steps do
    a = fresh(ct0)
    gmul(a, a, a)
    check(a, want_sq, "MulRelin(res, res, res)")


def test_byte_accounting_matches_the_survey_formulas(ctx):
    """The two byte counters bench.py's roofline figures rest on (hering_debug.h): he_alg_bytes accumulates SURVEY.md section
    8(d)'s per-primitive formulas at the C ABI (NTT 2L, binary 3L, MulRelin 6L + 2 beta (L + alpha) limbs of N * 8 bytes, times
    the batch; the key charged per entry / once per call), he_prof_end_bytes sums what every launch of a primitive must read
    and write once."""
    logN, q, p, t = _bench_config()
    logN = 12  # the same chain on a small ring: one 4096-row per limb
    q, p = O.GenModuli(logN + 1, [55] + [45] * 11, [55] * 3)
    N, L, alpha, B = 1 << logN, len(q), len(p), 5
    beta = (L + alpha - 1) // alpha
    pr = Pair(ctx, logN, L, alpha, qmods=q, pmods=p)
    rng = rng_for(4242)
    gev = la.Evaluator(pr.gQ, pr.gP)
    kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(beta)])
    rlk = gev.NewEvaluationKey(kq, kp)
    import bench
    a = [la.Poly(pr.gQ, L, B).upload(bench.uniform(rng, q, N, (B,))) for _ in range(2)]
    b = [la.Poly(pr.gQ, L, B).upload(bench.uniform(rng, q, N, (B,))) for _ in range(2)]
    out = [la.Poly(pr.gQ, L, B), la.Poly(pr.gQ, L, B)]
    limb = N * 8
    ctx.alg_bytes(reset=True)
    pr.gQ.NTT(a[0], out[0])
    assert ctx.alg_bytes(reset=True) == (2 * L * limb * B, 2 * L * limb * B)
    pr.gQ.Add(a[0], a[1], out[0])
    pr.gQ.MulCoeffsMontgomeryThenAdd(a[0], a[1], out[0])
    assert ctx.alg_bytes(reset=True)[0] == (3 + 4) * L * limb * B
    gev.BGVMulRelin(L - 1, t, a, b, rlk, out)
    per_entry, per_call = ctx.alg_bytes(reset=True)
    assert per_entry == (6 * L + 2 * beta * (L + alpha)) * limb * B
    assert per_call == 6 * L * limb * B + 2 * beta * (L + alpha) * limb
    # launch accounting of the same call: the pipeline of DESIGN.md section 4 (12 + 3 limbs, 11 + 0 of them below 2^47)
    import os
    if any(k.startswith("HERING_NO_") and v not in ("", "0") for k, v in os.environ.items()):
        return  # a run of the suite with one of the fusions switched off: other launches, other bytes
    ctx.prof_begin()
    gev.BGVMulRelin(L - 1, t, a, b, rlk, out)
    prof = ctx.prof_end_bytes()
    small_q = sum(m < (1 << 47) for m in q)
    # c2 = T(a1, b1): the tensor kernel on the integer-class limbs, the inverse rows themselves on the others (two inputs in, c2
    # and the transform out)
    assert prof["tensor"][2] == 3 * (L - small_q) * limb * B
    assert prof["ntt_rows_inv_f64"][2] == 4 * small_q * limb * B
    assert prof["modup"][2] == (L + (beta * (L + alpha) - L) + 2 * alpha + 2 * L) * limb * B
    # the ModDown epilogue of the double-precision limbs runs inside the NTT + MAC kernel (no special prime is below 2^47): beta
    # digits in, two key rows per digit for the batch, then two extension rows + the four inputs of the product in, two outputs
    assert "ntt_rows_fwd_f64" not in prof
    assert prof["ntt_mac_f64"][2] == ((beta + 2 + 6) * B + 2 * beta) * small_q * limb
    total = sum(v[2] for v in prof.values())
    assert 6 * per_entry > total > per_entry  # the realised pipeline moves more than the ideal single pass, within a small factor


# ---------------------------------------------------------------------------------------------------------------
# class boundaries
# ---------------------------------------------------------------------------------------------------------------
def primes_below(bits: int, log_nth_root: int, count: int):
    """the `count` largest primes q < 2^bits with q = 1 mod 2^log_nth_root"""
    out, step = [], 1 << log_nth_root
    q = (1 << bits) - step + 1
    while len(out) < count:
        if O.IsPrime(q):
            out.append(q)
        q -= step
    return out


def _worst_case_inputs(rng, q, N):
    """worst cases inside the reference's input domain: canonical words and lazy words below 2q (ring/ntt.go:164-171 takes
    U, V in [0, 2q))"""
    alt = np.zeros(N, dtype=np.uint64)
    alt[::2] = q - 1
    alt2 = np.zeros(N, dtype=np.uint64)
    alt2[1::2] = 2 * q - 1
    half = np.zeros(N, dtype=np.uint64)
    half[: N // 2] = q - 1
    return [np.full(N, q - 1, dtype=np.uint64), alt, alt2, half, np.full(N, 2 * q - 1, dtype=np.uint64),
            rng.integers(0, q, size=N, dtype=np.uint64), rng.integers(0, 2 * q, size=N, dtype=np.uint64)]


@pytest.mark.parametrize("logN", [15, 16, 17, 20])
@pytest.mark.parametrize("bits", [47, 58, 61])
def test_ntt_class_boundary_moduli(ctx, logN, bits):
    """Largest primes of the double-precision (< 2^47) and correction-free (< 2^58) classes, plus the smallest primes
    just ABOVE each boundary (which must take the next class), worst-case inputs, forward and inverse, lazy forms.  2^61 is
    the largest modulus size the reference's butterflies support (ring/ntt.go:169) and this library accepts: its largest
    NTT-friendly primes sit at the edge of the word-serial Montgomery products' domain (5q < 2^64)."""
    N = 1 << logN
    below = primes_below(bits, logN + 1, 2)
    above, qq = [], (1 << bits) + 1
    while len(above) < (1 if bits < 61 else 0):
        if O.IsPrime(qq):
            above.append(qq)
        qq += 1 << (logN + 1)
    moduli = below + above
    assert below[0] < (1 << bits) and (not above or (1 << bits) <= above[0]) and (1 << bits) - below[0] < (1 << (logN + 8))
    pr = Pair(ctx, logN, len(moduli), qmods=moduli)
    rng = rng_for(4700 + bits + logN)
    cases = [np.stack(c) for c in zip(*[_worst_case_inputs(rng, q, N) for q in moduli])]
    for x in cases:
        px, py = pr.gQ.NewPoly().upload(x), pr.gQ.NewPoly()
        pr.gQ.NTT(px, py)
        want = pr.oQ.NTT(x)
        assert np.array_equal(py.get(), want)
        pr.gQ.NTTLazy(px, py)
        assert np.array_equal(pr.oQ.unop("Reduce", py.get()), want)
        for i, q in enumerate(moduli):
            assert int(py.get()[i].max()) < 2 * q  # this library's lazy range (the reference allows 6q-2)
        pr.gQ.INTT(px, py)
        assert np.array_equal(py.get(), pr.oQ.INTT(x))
        # round trip on the transform of the worst case (dense, full-range spectrum)
        pw = pr.gQ.NewPoly().upload(want)
        pr.gQ.INTT(pw, py)
        assert np.array_equal(py.get(), pr.oQ.unop("Reduce", x))
    # beyond the reference's domain (it assumes words below 2q): this library reduces arbitrary 64-bit words first, so the
    # transform of x and of x mod q agree
    for w in (36, None):
        x = np.stack([np.full(N, (min(w, ((1 << 64) - 1) // q) * q - 1) if w else (1 << 64) - 1, dtype=np.uint64) for q in moduli])
        px, py, pz = pr.gQ.NewPoly().upload(x), pr.gQ.NewPoly(), pr.gQ.NewPoly()
        xr = pr.oQ.unop("Reduce", x)
        pr.gQ.NTT(px, py)
        assert np.array_equal(py.get(), pr.oQ.NTT(xr))
        pr.gQ.INTT(px, pz)
        assert np.array_equal(pz.get(), pr.oQ.INTT(xr))


@pytest.mark.parametrize("logN,B", [(16, 2), (15, 3)])
def test_keyswitch_class_boundary_moduli(ctx, logN, B):
    """GadgetProduct / MulRelin with Q and P made of boundary primes of every class (largest below 2^47, largest below
    2^58, 61-bit), worst-case ciphertext words (all q-1, alternating) and worst-case key words (all q-1): every kernel of
    the key-switch pipeline (fused basis extension incl. its float step, row NTTs of the three classes, NTT+MAC in doubles,
    128-bit inner product, fused ModDown epilogue) runs at the edge of its exactness argument."""
    N = 1 << logN
    s47, s58 = primes_below(47, logN + 1, 5), primes_below(58, logN + 1, 4)
    s61 = primes_below(61, logN + 1, 2)
    q = [s58[0], s47[0], s47[1], s47[2], s61[0], s47[3]]
    p = [s58[1], s47[4], s61[1]]
    pr = Pair(ctx, logN, len(q), len(p), qmods=q, pmods=p)
    rng = rng_for(5800 + logN)
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    L, alpha = len(q), len(p)
    beta = (L + alpha - 1) // alpha
    worst_k = lambda mods: np.stack([np.full(N, m - 1, dtype=np.uint64) for m in mods])
    kq = np.stack([np.stack([worst_k(q), uniform_poly(rng, q, N)]) for _ in range(beta)])
    kp = np.stack([np.stack([worst_k(p), uniform_poly(rng, p, N)]) for _ in range(beta)])
    gk, ok = gev.NewEvaluationKey(kq, kp), O.EvaluationKey(kq, kp)
    for level in (L - 1, L - 2):
        Qm = q[: level + 1]
        cases = [np.stack(c) for c in zip(*[_worst_case_inputs(rng, m, N) for m in Qm])]
        for c0 in range(0, len(cases), B):
            cx = np.stack([cases[(c0 + i) % len(cases)] for i in range(B)])
            pcx = la.Poly(pr.gQ, level + 1, B).upload(cx)
            out = [la.Poly(pr.gQ, level + 1, B), la.Poly(pr.gQ, level + 1, B)]
            gev.GadgetProduct(level, pcx, gk, out)
            g = [o.get() for o in out]
            for i in range(B):
                want = oev.GadgetProduct(level, cx[i], ok)
                assert np.array_equal(g[0][i], want[0]) and np.array_equal(g[1][i], want[1]), (level, c0, i)
        # MulRelin on canonical worst cases (tensor inputs are canonical by contract)
        wa = np.stack([worst_k(Qm), cases[1]])
        wb = np.stack([worst_k(Qm), cases[2]])
        a = [la.Poly(pr.gQ, level + 1).upload(c) for c in wa]
        b = [la.Poly(pr.gQ, level + 1).upload(c) for c in wb]
        o2 = [la.Poly(pr.gQ, level + 1), la.Poly(pr.gQ, level + 1)]
        gev.CKKSMulRelin(level, a, b, gk, o2)
        assert np.array_equal(np.stack([o.get() for o in o2]), oev.CKKSMulRelin(wa, wb, ok, True)), level
        gev.BGVMulRelin(level, 65537, a, b, gk, o2)
        assert np.array_equal(np.stack([o.get() for o in o2]), oev.BGVMulRelin(65537, wa, wb, ok, True)), level


def test_hoisting_buffer_is_bound_to_its_evaluator_and_level(ctx):
    """A Decomposition from another evaluator, an unfilled one, or one filled at another levelP is rejected (it would index
    the device buffer with the wrong digit stride)."""
    pr = Pair(ctx, 11, 4, 2)
    ev1, ev2 = la.Evaluator(pr.gQ, pr.gP), la.Evaluator(pr.gQ, pr.gP)
    rng = rng_for(77)
    kq = np.stack([np.stack([uniform_poly(rng, pr.q, pr.N) for _ in range(2)]) for _ in range(2)])
    kp = np.stack([np.stack([uniform_poly(rng, pr.p, pr.N) for _ in range(2)]) for _ in range(2)])
    k1 = ev1.NewEvaluationKey(kq, kp)
    c = pr.gQ.NewPoly().upload(uniform_poly(rng, pr.q, pr.N))
    out = [pr.gQ.NewPoly(), pr.gQ.NewPoly()]
    d1, d2 = la.Decomposition(ev1), la.Decomposition(ev2)
    with pytest.raises(la.HeringError, match="never filled"):
        ev1.GadgetProductHoisted(3, d1, k1, out)
    ev2.DecomposeNTT(3, 1, 2, c, True, d2)
    with pytest.raises(la.HeringError, match="another evaluator"):
        ev1.GadgetProductHoisted(3, d2, k1, out)
    ev1.DecomposeNTT(2, 1, 2, c, True, d1)
    with pytest.raises(la.HeringError, match="requested"):
        ev1.GadgetProductHoisted(3, d1, k1, out)
    ev1.DecomposeNTT(3, 1, 2, c, True, d1)
    ev1.GadgetProductHoisted(3, d1, k1, out)


def test_polynomial_of_another_context_is_rejected(ctx):
    other = la.Context(0)
    pr = Pair(ctx, 10, 2)
    ro = la.Ring(other, pr.N, pr.q)
    po = ro.NewPoly()
    with pytest.raises(la.HeringError, match="another context"):
        pr.gQ.NTT(po, po)
    with pytest.raises(la.HeringError, match="context"):
        pr.gQ.NewPoly().CopyLvl(1, po)

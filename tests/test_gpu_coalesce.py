"""Coalescing of concurrent single-ciphertext calls (`-m gpu`): he_evaluator_set_coalescing (include/hering.h).

The reference's operator API is one ciphertext per call (schemes/schemes.go:14-28) and its parallel mode is many goroutines
making such calls at once on evaluators that share tables and keys (b.RunParallel, schemes/ckks/ckks_benchmarks_test.go:116-207;
core/rlwe/evaluator.go:200-227).  Here the same pattern -- one OS thread per ciphertext, batch-1 handles, one shared evaluator --
must (a) give every caller exactly the words the uncoalesced call gives (= the oracle's), whatever batches happened to form,
for both schemes, for outputs aliasing inputs, for callers at different levels and for shapes whose pipeline cannot take entry
tables, and (b) really batch: the queue's statistics show fewer launches than calls.
"""
import threading

import numpy as np
import pytest

import lattigo_amd as la
from lattigo_amd.rlwe import ConcurrentCalls, ConcurrentMulRelin
from oracle import oracle as O
from tests.gpu_common import Pair, ctx  # noqa: F401
from tests.helpers import rng_for, uniform_poly

pytestmark = pytest.mark.gpu
T = 65537


@pytest.fixture(autouse=True)
def _queue_off_afterwards(ctx):
    """the queue belongs to the context, which the session shares: every test leaves it switched off"""
    yield
    ctx.sync()
    ctx.SetCoalescing(0, 0)


def _chain(logN, nq, np_):
    """a chain mixing the double-precision class (45-bit) with the integer class (55-bit q0 and special primes), as the headline's"""
    q, p = O.GenModuli(logN + 1, [55] + [45] * (nq - 1), [55] * np_)
    return list(q), list(p)


def _setup(ctx, logN, nq, np_, ci=False):
    q, p = _chain(logN + (1 if ci else 0), nq, np_)
    pr = Pair(ctx, logN, nq, np_, qmods=q, pmods=p, ci=ci)
    N, beta = 1 << logN, (nq + np_ - 1) // np_
    rng = rng_for(9100 + logN + nq)
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(beta)])
    return pr, q, p, N, rng, gev, oev, gev.NewEvaluationKey(kq, kp), O.EvaluationKey(kq, kp)


def _stats(ctx_or_ev, before=None):
    """queue counters (they belong to the context, which the whole test session shares: tests look at differences)"""
    st = ctx_or_ev.CoalescingStats()
    if before is not None:
        st = {k: (st[k] - before[k] if k != "largest_batch" else st[k]) for k in st}
    return st


def _run_threads(fns):
    """every fn on its own OS thread, released together (ctypes drops the GIL inside the library)"""
    gate, errs = threading.Barrier(len(fns)), []

    def wrap(f):
        try:
            gate.wait()
            f()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=wrap, args=(f,)) for f in fns]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("logN,scheme", [(11, "bgv"), (13, "bgv"), (13, "ckks"), (15, "bgv")])
def test_concurrent_callers_get_the_uncoalesced_words(ctx, logN, scheme):
    """K threads x M calls each on their own batch-1 ciphertexts; logN = 11: the row-kernel epilogues carry the entry tables,
    13 / 15: the product prologue of the inverse rows and the NTT + MAC epilogue (production row sizes)."""
    nq, np_ = (12, 3) if logN == 15 else (5, 2)
    pr, q, p, N, rng, gev, oev, gk, ok = _setup(ctx, logN, nq, np_)
    K, M = (6, 2) if logN == 15 else (9, 3)
    gev.SetCoalescing(64, 3000)  # a wide window: Python threads arrive milliseconds apart
    st0 = _stats(gev)
    ins = [[np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(2)] for _ in range(K)]  # [k][op][comp]
    dev = [[[la.Poly(pr.gQ, nq).upload(c) for c in op] for op in ins[k]] for k in range(K)]
    outs = [[[la.Poly(pr.gQ, nq), la.Poly(pr.gQ, nq)] for _ in range(M)] for _ in range(K)]
    mul = (lambda a, b, o: gev.BGVMulRelin(nq - 1, T, a, b, gk, o)) if scheme == "bgv" else (lambda a, b, o: gev.CKKSMulRelin(nq - 1, a, b, gk, o))

    def caller(k):
        def f():
            for m in range(M):
                mul(dev[k][0], dev[k][1], outs[k][m])
        return f

    _run_threads([caller(k) for k in range(K)])
    ctx.sync()
    st = _stats(gev, st0)
    assert st["calls"] == K * M and st["one_by_one"] == 0
    assert st["launches"] < st["calls"] and st["largest_batch"] >= 2, st
    for k in range(K):
        want = oev.BGVMulRelin(T, ins[k][0], ins[k][1], ok, True) if scheme == "bgv" else oev.CKKSMulRelin(ins[k][0], ins[k][1], ok, True)
        for m in range(M):
            assert np.array_equal(np.stack([o.get() for o in outs[k][m]]), want), (k, m)
    gev.SetCoalescing(0, 0)
    before = gev.CoalescingStats()["calls"]
    mul(dev[0][0], dev[0][1], outs[0][0])  # off again: a direct launch
    assert gev.CoalescingStats()["calls"] == before


def test_aliasing_levels_and_squaring_are_kept_apart(ctx):
    """Requests are batched only with requests of the same (level, aliasing) key: in-place callers (MulRelin(ct, ct2, ct), which
    take the three-output tensor kernel), squaring callers and callers at a lower level run at the same time as plain ones."""
    logN, nq, np_ = 13, 5, 2
    pr, q, p, N, rng, gev, oev, gk, ok = _setup(ctx, logN, nq, np_)
    gev.SetCoalescing(16, 3000)
    st0 = _stats(gev)
    kinds = ["plain", "inplace0", "inplace1", "square", "lower", "plain", "inplace0", "lower", "square_inplace", "plain"]
    ins, dev, outs, want = [], [], [], []
    for kind in kinds:
        lv = nq - 2 if kind == "lower" else nq - 1
        a = np.stack([uniform_poly(rng, q[: lv + 1], N) for _ in range(2)])
        b = a if kind.startswith("square") else np.stack([uniform_poly(rng, q[: lv + 1], N) for _ in range(2)])
        da = [la.Poly(pr.gQ, lv + 1).upload(c) for c in a]
        db = da if kind.startswith("square") else [la.Poly(pr.gQ, lv + 1).upload(c) for c in b]
        o = {"inplace0": da, "inplace1": db, "square_inplace": da}.get(kind) or [la.Poly(pr.gQ, lv + 1), la.Poly(pr.gQ, lv + 1)]
        ins.append((lv, a, b)); dev.append((da, db)); outs.append(o)
        want.append(oev.BGVMulRelin(T, a, b, ok, True))
    _run_threads([(lambda i: lambda: gev.BGVMulRelin(ins[i][0], T, dev[i][0], dev[i][1], gk, outs[i]))(i) for i in range(len(kinds))])
    ctx.sync()
    for i, kind in enumerate(kinds):
        assert np.array_equal(np.stack([o.get() for o in outs[i]]), want[i]), (i, kind)
    st = _stats(gev, st0)
    assert st["calls"] == len(kinds) and st["launches"] >= 3  # three keys at least: plain / aliased at the top level, the lower level


def test_shapes_without_entry_tables_run_one_by_one(ctx):
    """A conjugate-invariant ring has no fused ModDown plan (its launches do not take entry tables): queued calls are served one
    by one, with the same words."""
    logN, nq, np_ = 11, 4, 2
    pr, q, p, N, rng, gev, oev, gk, ok = _setup(ctx, logN, nq, np_, ci=True)
    gev.SetCoalescing(8, 3000)
    st0 = _stats(gev)
    K = 4
    ins = [[np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(2)] for _ in range(K)]
    dev = [[[la.Poly(pr.gQ, nq).upload(c) for c in op] for op in ins[k]] for k in range(K)]
    outs = [[la.Poly(pr.gQ, nq), la.Poly(pr.gQ, nq)] for _ in range(K)]
    _run_threads([(lambda k: lambda: gev.CKKSMulRelin(nq - 1, dev[k][0], dev[k][1], gk, outs[k]))(k) for k in range(K)])
    ctx.sync()
    for k in range(K):
        assert np.array_equal(np.stack([o.get() for o in outs[k]]), oev.CKKSMulRelin(ins[k][0], ins[k][1], ok, True)), k
    st = _stats(gev, st0)
    assert st["calls"] == K and st["one_by_one"] >= 2 * (st["launches"] < K), st


@pytest.mark.parametrize("sync_each,deferred", [(False, 0), (True, 0), (False, 4), (True, 4)])
def test_library_side_thread_harness(ctx, sync_each, deferred):
    """he_debug_concurrent_mul_relin (bench.py's `concurrent_b1`): pthreads inside the library, every caller on its own handles;
    callers that wait for each result (sync_each) and callers that only enqueue."""
    logN, nq, np_ = 13, 5, 2
    pr, q, p, N, rng, gev, oev, gk, ok = _setup(ctx, logN, nq, np_)
    K, M = 24, 5
    gev.SetCoalescing(64, 50)
    ctx.SetDeferred(deferred)
    st0 = _stats(gev)
    ins = [[np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(2)] for _ in range(K)]
    dev = [[[la.Poly(pr.gQ, nq).upload(c) for c in op] for op in ins[k]] for k in range(K)]
    outs = [[la.Poly(pr.gQ, nq), la.Poly(pr.gQ, nq)] for _ in range(K)]
    wall = ConcurrentMulRelin([(ctx, gev, dev[k][0], dev[k][1], gk, outs[k]) for k in range(K)], nq - 1, M, t=T, sync_each=sync_each)
    assert wall > 0
    st = _stats(gev, st0)
    assert st["calls"] == K * M and st["launches"] < st["calls"], st
    for k in range(K):
        assert np.array_equal(np.stack([o.get() for o in outs[k]]), oev.BGVMulRelin(T, ins[k][0], ins[k][1], ok, True)), k


@pytest.mark.parametrize("logN", [13, 16])
def test_key_switches_coalesce_too(ctx, logN):
    """GadgetProduct, Relinearize (into fresh outputs and in place -- the reference's usual form) and Automorphism (Rotate) from
    concurrent single-ciphertext callers: the NTT-domain operand is then read through the entry table by the inverse rows and as
    the digits' own limbs, the addends / outputs by the epilogues and the final gathers.  logN = 16: 8192-coefficient rows."""
    nq, np_ = (6, 2) if logN == 16 else (5, 2)
    pr, q, p, N, rng, gev, oev, gk, ok = _setup(ctx, logN, nq, np_)
    K = 5 if logN == 16 else 8
    gev.SetCoalescing(64, 3000)
    st0 = _stats(gev)
    gal = 5
    cts = [np.stack([uniform_poly(rng, q, N) for _ in range(3)]) for _ in range(K)]       # [k][component 0..2]
    up = lambda k, n: [la.Poly(pr.gQ, nq).upload(c) for c in cts[k][:n]]
    fresh = lambda: [la.Poly(pr.gQ, nq), la.Poly(pr.gQ, nq)]
    # GadgetProduct(cx = component 1)
    cx, outs = [la.Poly(pr.gQ, nq).upload(cts[k][1]) for k in range(K)], [fresh() for _ in range(K)]
    _run_threads([(lambda k: lambda: gev.GadgetProduct(nq - 1, cx[k], gk, outs[k]))(k) for k in range(K)])
    for k in range(K):
        assert np.array_equal(np.stack([o.get() for o in outs[k]]), oev.GadgetProduct(nq - 1, cts[k][1], ok)), ("gadget", k)
    # Relinearize: even callers into fresh outputs, odd callers in place (out = the degree-2 ciphertext's first two components)
    ct3 = [up(k, 3) for k in range(K)]
    outs = [fresh() if k % 2 == 0 else ct3[k][:2] for k in range(K)]
    _run_threads([(lambda k: lambda: gev.Relinearize(nq - 1, ct3[k], gk, outs[k]))(k) for k in range(K)])
    for k in range(K):
        assert np.array_equal(np.stack([o.get() for o in outs[k]]), oev.Relinearize(cts[k], ok)), ("relinearize", k)
    # Relinearize with the output on the key switch's own operand (component 2): flagged, served one by one, same words
    ct3 = [up(k, 3) for k in range(K)]
    outs = [[ct3[k][2], la.Poly(pr.gQ, nq)] for k in range(K)]
    _run_threads([(lambda k: lambda: gev.Relinearize(nq - 1, ct3[k], gk, outs[k]))(k) for k in range(K)])
    for k in range(K):
        assert np.array_equal(np.stack([o.get() for o in outs[k]]), oev.Relinearize(cts[k], ok)), ("relinearize onto c2", k)
    # Automorphism
    ct2, outs = [up(k, 2) for k in range(K)], [fresh() for _ in range(K)]
    _run_threads([(lambda k: lambda: gev.Automorphism(nq - 1, ct2[k], gal, gk, outs[k]))(k) for k in range(K)])
    for k in range(K):
        assert np.array_equal(np.stack([o.get() for o in outs[k]]), oev.Automorphism(cts[k][:2], gal, ok)), ("automorphism", k)
    ctx.sync()
    st = _stats(gev, st0)
    assert st["calls"] == 4 * K and st["launches"] < st["calls"] and st["largest_batch"] >= 2, st
    # the library-side thread harness on rotations
    outs = [fresh() for _ in range(K)]
    ConcurrentCalls("rotate", [(ctx, gev, ct2[k], None, gk, outs[k]) for k in range(K)], nq - 1, 4, t=gal, sync_each=True)
    for k in range(K):
        assert np.array_equal(np.stack([o.get() for o in outs[k]]), oev.Automorphism(cts[k][:2], gal, ok)), ("harness rotate", k)



@pytest.mark.parametrize("deferred", [0, 3])
def test_mixed_dependent_chains_from_many_threads(ctx, deferred):
    """deferred > 0: he_ctx_set_deferred -- the calls return once filed and the context's dispatcher thread launches them; each
    thread's chain must still run in its own order.
    Every caller runs a CHAIN of dependent operations on its own ciphertext -- MulRelin (plain, squaring in place), two
    rotations (one in place: the flagged one-by-one path), GadgetProduct, and ring calls (Add) in between (queued as well since
    round 5) -- while eleven others do the same with the steps in another order.  What a step reads was written by a
    batch launched by ANOTHER thread (the leader of that moment) or by a direct launch of this one: stream order must hold across
    both.  Final ciphertexts: identical to the same chains run one after the other with the queue switched off, and caller 0's to
    the oracle's chain."""
    logN, nq, np_ = 13, 5, 2
    pr, q, p, N, rng, gev, oev, rlk, orlk = _setup(ctx, logN, nq, np_)
    beta, lv = (nq + np_ - 1) // np_, nq - 1
    g = [pow(5, 3, 2 * N), 2 * N - 1]
    gk, ogk = [], []
    for _ in g:
        kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(beta)])
        kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(beta)])
        gk.append(gev.NewEvaluationKey(kq, kp)); ogk.append(O.EvaluationKey(kq, kp))
    K, STEPS = 12, 9
    ops = ["mul", "rot0", "add", "square_inplace", "gadget", "rot1_inplace", "mul", "add", "rot0"]
    x0 = [np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(K)]
    y0 = [np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(K)]

    def run_chain(k, x, y, out):
        for s in range(STEPS):
            op = ops[(5 * k + s) % len(ops)]
            if op == "mul":
                gev.CKKSMulRelin(lv, x, y, rlk, out); x, out = out, x
            elif op == "square_inplace":
                gev.CKKSMulRelin(lv, x, x, rlk, x)
            elif op == "rot0":
                gev.Automorphism(lv, x, g[0], gk[0], out); x, out = out, x
            elif op == "rot1_inplace":
                gev.Automorphism(lv, x, g[1], gk[1], x)
            elif op == "gadget":
                gev.GadgetProduct(lv, x[1], rlk, out)
                pr.gQ.Add(out[0], x[0], out[0]); x, out = out, x
            else:
                pr.gQ.Add(x[0], y[0], x[0]); pr.gQ.Add(x[1], y[1], x[1])
        return x

    def fresh(k):
        return ([la.Poly(pr.gQ, nq).upload(c) for c in x0[k]], [la.Poly(pr.gQ, nq).upload(c) for c in y0[k]],
                [la.Poly(pr.gQ, nq), la.Poly(pr.gQ, nq)])

    gev.SetCoalescing(0, 0)
    ref = []
    for k in range(K):
        ref.append(np.stack([c.get() for c in run_chain(k, *fresh(k))]))
    gev.SetCoalescing(16, 500)
    ctx.SetDeferred(deferred)
    st0 = _stats(gev)
    state, res = [fresh(k) for k in range(K)], [None] * K

    def caller(k):
        def f():
            res[k] = run_chain(k, *state[k])
        return f

    _run_threads([caller(k) for k in range(K)])
    ctx.sync()
    for k in range(K):
        assert np.array_equal(np.stack([c.get() for c in res[k]]), ref[k]), k
    st = _stats(gev, st0)
    assert st["launches"] < st["calls"], st
    gev.SetCoalescing(0, 0)
    # caller 0's chain in the oracle
    sub = O.Ring(N, q)
    x, y = x0[0], y0[0]
    for s in range(STEPS):
        op = ops[s % len(ops)]
        if op == "mul":
            x = oev.CKKSMulRelin(x, y, orlk, True)
        elif op == "square_inplace":
            x = oev.CKKSMulRelin(x, x, orlk, True)
        elif op in ("rot0", "rot1_inplace"):
            i = 0 if op == "rot0" else 1
            x = np.stack(oev.Automorphism(x, g[i], ogk[i]))
        elif op == "gadget":
            w = oev.GadgetProduct(lv, x[1], orlk)
            x = np.stack([sub.binop("Add", w[0], x[0]), w[1]])
        else:
            x = np.stack([sub.binop("Add", x[0], y[0]), sub.binop("Add", x[1], y[1])])
    assert np.array_equal(ref[0], x)


# ---- round 5: the queue serves the WHOLE one-ciphertext interface ------------------------------------------------------------
def _program(pr, ctx, gev, be, keys, nq, np_, seed, N):
    """One caller's program: every operator entry point of include/hering.h on its own batch-1 polynomials, each result kept.
    Returns the list of result arrays (downloaded after the final sync by the caller of this function)."""
    rng = rng_for(seed)
    q, p = pr.q, pr.p
    lvQ, lvP = nq - 1, np_ - 1
    rQ, rP = pr.gQ, pr.gP
    rlk, gk, gal = keys
    keep = []

    def newq(nl=nq, arr=None):
        x = la.Poly(rQ, nl)
        if arr is not None:
            x.upload(arr)
        keep.append(x)
        return x

    def newp(arr=None):
        x = la.Poly(rP, np_)
        if arr is not None:
            x.upload(arr)
        keep.append(x)
        return x

    uq = lambda: uniform_poly(rng, q, N)
    up_ = lambda: uniform_poly(rng, p, N)
    a, b, c = newq(arr=uq()), newq(arr=uq()), newq(arr=uq())
    res = []
    # ring level
    t = newq(); rQ.NTT(a, t); res.append(t)
    t = newq(); rQ.INTT(a, t); res.append(t)
    t = newq(); rQ.NTTLazy(b, t); rQ.Reduce(t, t); res.append(t)
    t = newq(); rQ.Add(a, b, t); res.append(t)
    t = newq(); rQ.Sub(a, b, t); res.append(t)
    t = newq(); rQ.MulCoeffsMontgomery(a, b, t); rQ.MulCoeffsMontgomeryThenAdd(b, c, t); res.append(t)
    t = newq(); rQ.Neg(a, t); rQ.MForm(t, t); res.append(t)
    t = newq(); rQ.MulScalar(a, 0x1234567, t); rQ.AddScalar(t, 77, t); res.append(t)
    t = newq(); rQ.MulScalarBigint(a, (1 << 90) + 12345, t); res.append(t)
    t = newq(); rQ.MulDoubleRNSScalar(a, [3 + i for i in range(nq)], [5 + i for i in range(nq)], t); res.append(t)
    t = newq(); rQ.Shift(a, 37, t); rQ.Shift(t, 5, t); res.append(t)          # (out of place, then in place)
    t = newq(); rQ.MultByMonomial(a, N + 3, t); res.append(t)
    idx = rQ.AutomorphismNTTIndex(gal)
    t = newq(); rQ.AutomorphismNTTWithIndex(a, idx, t); rQ.AutomorphismNTTWithIndexThenAddLazy(b, idx, t); rQ.Reduce(t, t); res.append(t)
    t = newq(); rQ.Automorphism(a, gal, t); res.append(t)
    t = newq(nq - 1); rQ.DivRoundByLastModulusNTT(a, t); res.append(t)
    t = newq(nq - 2); rQ.DivRoundByLastModulusManyNTT(2, a, t); res.append(t)
    t = newq(nq - 1); rQ.DivFloorByLastModulus(a, t); res.append(t)
    t = newq(nq - 2); rQ.DivFloorByLastModulusManyNTT(2, b, t); res.append(t)
    t = newq(nq - 1); rQ.DivRoundByLastModulusNTT(c, c); res.append(c)          # in place (the schemes' Rescale form)
    c = newq(arr=uq())
    # basis extender
    pp = newp(arr=up_())
    t = newp(); be.ModUpQtoP(lvQ, lvP, a, t); res.append(t)
    t = newq(); be.ModUpPtoQ(lvP, lvQ, pp, t); res.append(t)
    t = newq(); be.ModDownQPtoQ(lvQ, lvP, a, pp, t); res.append(t)
    t = newq(); be.ModDownQPtoQNTT(lvQ, lvP, a, pp, t); res.append(t)
    t = newp(); be.ModDownQPtoP(lvQ, lvP, a, pp, t); res.append(t)
    # rlwe.EvaluatorProvider
    qp = lambda: [(newq(), newp()), (newq(), newp())]
    two = lambda: [newq(), newq()]
    t0, t1 = newq(), newp(); gev.DecomposeAndSplit(lvQ, lvP, np_, 1, a, t0, t1); res += [t0, t1]
    dec = la.Decomposition(gev); keep.append(dec)
    gev.DecomposeNTT(lvQ, lvP, np_, a, True, dec)
    acc = qp(); gev.GadgetProductHoistedLazy(lvQ, dec, gk, acc); res += [acc[0][0], acc[0][1], acc[1][0], acc[1][1]]
    o = two(); gev.ModDown(lvQ, lvP, acc, o); res += o
    o = two(); gev.GadgetProductHoisted(lvQ, dec, gk, o); res += o
    o = two(); gev.AutomorphismHoisted(lvQ, [b, a], dec, gal, gk, o); res += o
    acc2 = qp(); gev.AutomorphismHoistedLazy(lvQ, [b, a], dec, gal, gk, acc2); res += [acc2[0][0], acc2[0][1], acc2[1][0], acc2[1][1]]
    acc3 = qp(); gev.GadgetProductLazy(lvQ, b, rlk, acc3); res += [acc3[0][0], acc3[0][1], acc3[1][0], acc3[1][1]]
    t = newq(); gev.ModDownQPtoQNTT(lvQ, lvP, acc3[0][0], acc3[0][1], t); res.append(t)
    o = two(); gev.GadgetProduct(lvQ, c, rlk, o); res += o
    o = two(); gev.Relinearize(lvQ, [a, b, c], rlk, o); res += o
    o = two(); gev.Automorphism(lvQ, [a, b], gal, gk, o); res += o
    o = two(); gev.CKKSMulRelin(lvQ, [a, b], [b, c], rlk, o); res += o
    o = [newq(), newq(), newq()]; gev.BGVMulRelin(lvQ, T, [a, b], [b, c], None, o); res += o  # Mul without a key: degree 2
    o2 = two(); gev.Rescale(lvQ, 1, o[:2], o2); res += o2
    # the lintrans inner loop and the bootstrapping helpers
    import ctypes as C

    from lattigo_amd._lib import H, check, load
    out = qp()
    terms = [((a, pp), acc[0], acc[1], None), ((b, pp), acc2[0], acc2[1], idx), ((c, pp), acc3[0], acc3[1], None)]
    arr = lambda f: (H * len(terms))(*[f(t_) for t_ in terms])
    for accumulate in (0, 1):
        check(load().he_lintrans_mul_sum(gev.h, lvQ, lvP, len(terms), arr(lambda t_: t_[0][0].h), arr(lambda t_: t_[0][1].h),
                                         arr(lambda t_: t_[1][0].h), arr(lambda t_: t_[1][1].h), arr(lambda t_: t_[2][0].h),
                                         arr(lambda t_: t_[2][1].h), arr(lambda t_: t_[3].h if t_[3] is not None else 0), accumulate,
                                         out[0][0].h, out[0][1].h, out[1][0].h, out[1][1].h))
    res += [out[0][0], out[0][1], out[1][0], out[1][1]]
    # the giant step of the BSGS product (he_lintrans_giant_step): overwriting, then accumulating onto the same outer accumulators
    og = qp()
    gev.LinTransGiantStep(lvQ, a, gk, gal, out[0], og, False)
    gev.LinTransGiantStep(lvQ, b, gk, gal, out[1], og, True)
    res += [og[0][0], og[0][1], og[1][0], og[1][1]]
    lq, lp_ = newq(), newp()
    check(load().he_centered_lift(gev.h, 1, a.h, 0, lvQ, lq.h, lvP, lp_.h)); res += [lq, lp_]
    dec2 = la.Decomposition(gev); keep.append(dec2)
    check(load().he_decomp_fill(dec2.h, lvQ, lvP, lq.h, lp_.h))
    o = two(); gev.GadgetProductHoisted(lvQ, dec2, gk, o); res += o
    keep.append(idx)
    return res, keep


@pytest.mark.parametrize("logN,deferred", [(13, 0), (16, 0), (13, 8), (16, 2)])
def test_every_operator_entry_point_coalesces(ctx, logN, deferred):
    """deferred > 0: the same through deferred submission (he_ctx_set_deferred: calls return once filed, temporaries are freed
    while their requests are still pending, uploads of the same thread wait for its pending requests).
    Round 5: the queue belongs to the context and serves every entry point of the one-ciphertext interface.  K threads run the
    same program -- ring methods, all rescale variants, ModUp / ModDown, the seven rlwe.EvaluatorProvider methods, Mul with and
    without a key, Rescale, the lintrans inner loop, the bootstrapping helpers -- each on its own polynomials and hoisting
    buffers; every result must equal, word for word, the same program run alone with the queue off (whose words the rest of the
    suite pins on the oracle), the batches must really form (fewer launches than calls), and no request of this standard-ring
    shape may have been served one by one.  logN = 16: the 8192-row kernels."""
    nq, np_ = (6, 2) if logN == 16 else (5, 2)
    pr, q, p, N, rng, gev, oev, rlk, orlk = _setup(ctx, logN, nq, np_)
    beta = (nq + np_ - 1) // np_
    gal = pow(5, 7, 2 * N)
    kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(beta)])
    gk = gev.NewEvaluationKey(kq, kp)
    be = la.BasisExtender(pr.gQ, pr.gP)
    K = 6 if logN == 16 else 8
    ctx.SetCoalescing(0, 0)
    ref = []
    for k in range(K):
        res, keep = _program(pr, ctx, gev, be, (rlk, gk, gal), nq, np_, 7000 + k, N)
        ctx.sync()
        ref.append([r.get() for r in res])
        del res, keep
    ctx.SetCoalescing(64, 3000)
    ctx.SetDeferred(deferred)
    before = ctx.CoalescingStats()
    got = [None] * K

    def caller(k):
        def f():
            got[k] = _program(pr, ctx, gev, be, (rlk, gk, gal), nq, np_, 7000 + k, N)
        return f

    _run_threads([caller(k) for k in range(K)])
    ctx.sync()
    st = ctx.CoalescingStats()
    for k in range(K):
        res = [r.get() for r in got[k][0]]
        assert len(res) == len(ref[k])
        for i, (x, y) in enumerate(zip(res, ref[k])):
            assert np.array_equal(x, y), (k, i)
    calls, launches = st["calls"] - before["calls"], st["launches"] - before["launches"]
    assert calls >= K * 45 and launches < calls // 2, st
    assert st["one_by_one"] == before["one_by_one"], st
    ctx.SetCoalescing(0, 0)
    # and the oracle on two of the new paths directly (Rescale, the lazy gadget product + ModDown), caller 0's operands
    r0 = rng_for(7000)
    a, b, c = (uniform_poly(r0, q, N) for _ in range(3))
    assert np.array_equal(ref[0][14], pr.oQ.DivRoundByLastModulusNTT(a))


@pytest.mark.parametrize("deferred", [0, 4])
def test_handles_of_a_few_entries_join_the_queue(ctx, deferred):
    """The drivers stack independent ciphertexts into one handle (the real and imaginary halves of a bootstrap's EvalMod run as a
    batch-2 ciphertext): requests over handles of 1, 2 or 3 entries share batches -- a request of nb entries takes nb rows of the
    entry table -- and CopyBatch (stack / unstack) rides in the queue like CopyLvl.  Word for word the same program with the
    queue off."""
    logN, nq, np_ = 13, 5, 2
    pr, q, p, N, rng, gev, oev, rlk, orlk = _setup(ctx, logN, nq, np_)
    beta, lv = (nq + np_ - 1) // np_, nq - 1
    gal = pow(5, 5, 2 * N)
    kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(beta)])
    gk = gev.NewEvaluationKey(kq, kp)
    K = 9
    nbs = [1 + k % 3 for k in range(K)]
    ins = [[np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(nbs[k])]) for _ in range(2)]) for _ in range(2)] for k in range(K)]

    def program(k):
        nb = nbs[k]
        a, b = ([la.Poly(pr.gQ, nq, nb).upload(c) for c in ins[k][i]] for i in range(2))
        new = lambda nl=nq, n=nb: la.Poly(pr.gQ, nl, n)  # noqa: E731
        m = [new(), new()]
        gev.CKKSMulRelin(lv, a, b, rlk, m)                       # key switch over nb entries
        pr.gQ.Add(m[0], b[0], m[0]); pr.gQ.Add(m[1], b[1], m[1])  # ring calls
        r = [new(), new()]
        gev.Automorphism(lv, m, gal, gk, r)
        s = [new(nq - 1), new(nq - 1)]
        gev.Rescale(lv, 1, r, s)                                 # he_rescale_polys: two requests of nb entries in one call
        # stack two copies of the result into one handle of 2 nb entries, square it, unstack the second half
        st2 = [new(nq - 1, 2 * nb), new(nq - 1, 2 * nb)]
        for o, v in zip(st2, s):
            o.CopyBatch(lv - 1, 0, v, 0, nb); o.CopyBatch(lv - 1, nb, v, 0, nb)
        sq = [new(nq - 1, 2 * nb), new(nq - 1, 2 * nb)]
        gev.CKKSMulRelin(lv - 1, st2, st2, rlk, sq)
        half = [new(nq - 1), new(nq - 1)]
        for o, v in zip(half, sq):
            o.CopyBatch(lv - 1, 0, v, nb, nb)
        return [m, r, s, half]

    ctx.SetCoalescing(0, 0)
    ref = []
    for k in range(K):
        ref.append([[c.download() for c in ct] for ct in program(k)])
    ctx.SetCoalescing(32, 3000)
    ctx.SetDeferred(deferred)
    before = ctx.CoalescingStats()
    got = [None] * K

    def caller(k):
        def f():
            got[k] = program(k)
        return f

    _run_threads([caller(k) for k in range(K)])
    ctx.sync()
    st = ctx.CoalescingStats()
    for k in range(K):
        for i, (ct, want) in enumerate(zip(got[k], ref[k])):
            for c, w in zip(ct, want):
                assert np.array_equal(c.download(), w), (k, i)
    calls, launches = st["calls"] - before["calls"], st["launches"] - before["launches"]
    assert calls == K * 13 and launches < calls, (calls, launches)
    assert st["one_by_one"] == before["one_by_one"], st
    # the oracle on caller 1's first entry (nb = 2): MulRelin + Add
    x = [ins[1][0][c][0] for c in range(2)], [ins[1][1][c][0] for c in range(2)]
    w = oev.CKKSMulRelin(np.stack(x[0]), np.stack(x[1]), orlk, True)
    sub = O.Ring(N, q)
    assert np.array_equal(ref[1][0][0][0], sub.binop("Add", w[0], x[1][0]))


def test_where_a_failed_launch_surfaces(ctx):
    """The contract of include/hering.h: a launch that fails after its call was accepted is returned by the call itself in the queue's
    default mode, and by the next Sync() under deferred submission (the call has returned by then); the failure is reported once."""
    from lattigo_amd._lib import load, check, HeringError
    L = load()
    ctx.SetCoalescing(8, 100)
    with pytest.raises(HeringError, match="injected launch failure"):
        check(L.he_debug_queue_inject_failure(ctx.h))
    ctx.sync()
    ctx.SetDeferred(4)
    check(L.he_debug_queue_inject_failure(ctx.h))  # accepted: HE_OK
    with pytest.raises(HeringError, match="injected launch failure.*deferred"):
        ctx.sync()
    ctx.sync()  # reported once
    ctx.SetDeferred(0)
    ctx.sync()


def test_deferred_calls_survive_teardown_and_thread_churn():
    """A context of its own: deferred calls from short-lived threads (each leaves its record to the next new thread), polynomials
    freed while their requests are pending, and the context destroyed with requests still in the queue -- everything filed is
    launched first, nothing hangs, results are the direct calls' words."""
    import gc
    c2 = la.Context(0)
    logN, nq, np_ = 12, 4, 1
    q, p = _chain(logN, nq, np_)
    pr = Pair(c2, logN, nq, np_, qmods=q, pmods=p)
    N, rng = 1 << logN, rng_for(4242)
    c2.SetCoalescing(16, 200)
    c2.SetDeferred(4)
    a = [np.stack([uniform_poly(rng, q, N)]) for _ in range(6)]
    want = [pr.oQ.binop("Add", x[0], x[0]) for x in a]
    got = [None] * 6
    for wave in range(5):  # 30 threads over the context's life, six at a time
        def caller(k):
            def f():
                x = la.Poly(pr.gQ, nq).upload(a[k])
                y = la.Poly(pr.gQ, nq)
                for _ in range(12):
                    t = la.Poly(pr.gQ, nq)   # a temporary freed while its request may still be pending
                    pr.gQ.Add(x, x, t)
                    pr.gQ.Add(x, x, y)
                    del t
                got[k] = y
            return f
        _run_threads([caller(k) for k in range(6)])
        c2.sync()
        for k in range(6):
            assert np.array_equal(got[k].get(), want[k]), (wave, k)
    st = c2.CoalescingStats()
    assert st["calls"] == 5 * 6 * 24 and st["launches"] < st["calls"], st
    # file more and tear everything down without a sync
    x = la.Poly(pr.gQ, nq).upload(a[0])
    outs = [la.Poly(pr.gQ, nq) for _ in range(8)]
    for o in outs:
        pr.gQ.Add(x, x, o)
    del outs, x, got, pr, c2
    gc.collect()


def test_switching_modes_under_load(ctx):
    """Callers keep issuing dependent chains while the main thread switches deferred submission off and on: a call caught by the switch
    goes the other way (the dispatcher launches what it holds before it stops; calls made meanwhile wait for that), order per thread
    is kept, nothing hangs.  Every chain's result equals the direct calls' words."""
    import time
    from lattigo_amd._lib import HeringError
    logN, nq, np_ = 12, 4, 1
    q, p = _chain(logN, nq, np_)
    pr = Pair(ctx, logN, nq, np_, qmods=q, pmods=p)
    N, rng = 1 << logN, rng_for(777)
    K, M = 6, 150
    a = [uniform_poly(rng, q, N) for _ in range(K)]
    sub = O.Ring(N, q)

    def chain(k, x, y):
        # y <- x; then M times y <- y + x (in place: each call depends on the one before), a Neg in between every 10 calls
        y.CopyLvl(nq - 1, x)
        for i in range(M):
            pr.gQ.Add(y, x, y)
            if i % 10 == 9:
                pr.gQ.unop("Neg", y, y)
        return y

    want = []
    for k in range(K):
        w = a[k].copy()
        for i in range(M):
            w = sub.binop("Add", w, a[k])
            if i % 10 == 9:
                w = sub.unop("Neg", w)
        want.append(w)
    ctx.SetCoalescing(16, 100)
    xs = [la.Poly(pr.gQ, nq).upload(a[k]) for k in range(K)]
    ys = [la.Poly(pr.gQ, nq) for _ in range(K)]
    done = threading.Event()

    def toggler():
        d = 0
        while not done.is_set():
            d = 0 if d else 4
            try:
                ctx.SetDeferred(d)
            except HeringError:
                pass  # "calls are in flight": a leader of the default mode is launching right now -- try again next time
            time.sleep(0.002)

    t = threading.Thread(target=toggler)
    t.start()
    try:
        for rep in range(3):
            _run_threads([(lambda k=k: chain(k, xs[k], ys[k])) for k in range(K)])
            ctx.sync()
            for k in range(K):
                assert np.array_equal(ys[k].get(), want[k]), (rep, k)
    finally:
        done.set()
        t.join()

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
    st = gev.CoalescingStats()
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
    st = gev.CoalescingStats()
    assert st["calls"] == len(kinds) and st["launches"] >= 3  # three keys at least: plain / aliased at the top level, the lower level


def test_shapes_without_entry_tables_run_one_by_one(ctx):
    """A conjugate-invariant ring has no fused ModDown plan (its launches do not take entry tables): queued calls are served one
    by one, with the same words."""
    logN, nq, np_ = 11, 4, 2
    pr, q, p, N, rng, gev, oev, gk, ok = _setup(ctx, logN, nq, np_, ci=True)
    gev.SetCoalescing(8, 3000)
    K = 4
    ins = [[np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(2)] for _ in range(K)]
    dev = [[[la.Poly(pr.gQ, nq).upload(c) for c in op] for op in ins[k]] for k in range(K)]
    outs = [[la.Poly(pr.gQ, nq), la.Poly(pr.gQ, nq)] for _ in range(K)]
    _run_threads([(lambda k: lambda: gev.CKKSMulRelin(nq - 1, dev[k][0], dev[k][1], gk, outs[k]))(k) for k in range(K)])
    ctx.sync()
    for k in range(K):
        assert np.array_equal(np.stack([o.get() for o in outs[k]]), oev.CKKSMulRelin(ins[k][0], ins[k][1], ok, True)), k
    st = gev.CoalescingStats()
    assert st["calls"] == K and (st["largest_batch"] == 1 or st["one_by_one"] >= 2), st


@pytest.mark.parametrize("sync_each", [False, True])
def test_library_side_thread_harness(ctx, sync_each):
    """he_debug_concurrent_mul_relin (bench.py's `concurrent_b1`): pthreads inside the library, every caller on its own handles;
    callers that wait for each result (sync_each) and callers that only enqueue."""
    logN, nq, np_ = 13, 5, 2
    pr, q, p, N, rng, gev, oev, gk, ok = _setup(ctx, logN, nq, np_)
    K, M = 24, 5
    gev.SetCoalescing(64, 50)
    ins = [[np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(2)] for _ in range(K)]
    dev = [[[la.Poly(pr.gQ, nq).upload(c) for c in op] for op in ins[k]] for k in range(K)]
    outs = [[la.Poly(pr.gQ, nq), la.Poly(pr.gQ, nq)] for _ in range(K)]
    wall = ConcurrentMulRelin([(ctx, gev, dev[k][0], dev[k][1], gk, outs[k]) for k in range(K)], nq - 1, M, t=T, sync_each=sync_each)
    assert wall > 0
    st = gev.CoalescingStats()
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
    st = gev.CoalescingStats()
    assert st["calls"] == 4 * K and st["launches"] < st["calls"] and st["largest_batch"] >= 2, st
    # the library-side thread harness on rotations
    outs = [fresh() for _ in range(K)]
    ConcurrentCalls("rotate", [(ctx, gev, ct2[k], None, gk, outs[k]) for k in range(K)], nq - 1, 4, t=gal, sync_each=True)
    for k in range(K):
        assert np.array_equal(np.stack([o.get() for o in outs[k]]), oev.Automorphism(cts[k][:2], gal, ok)), ("harness rotate", k)



def test_mixed_dependent_chains_from_many_threads(ctx):
    """Every caller runs a CHAIN of dependent operations on its own ciphertext -- MulRelin (plain, squaring in place), two
    rotations (one in place: the flagged one-by-one path), GadgetProduct, and direct ring calls (Add) in between, which do not go
    through the queue -- while eleven others do the same with the steps in another order.  What a step reads was written by a
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
    state, res = [fresh(k) for k in range(K)], [None] * K

    def caller(k):
        def f():
            res[k] = run_chain(k, *state[k])
        return f

    _run_threads([caller(k) for k in range(K)])
    ctx.sync()
    for k in range(K):
        assert np.array_equal(np.stack([c.get() for c in res[k]]), ref[k]), k
    st = gev.CoalescingStats()
    assert st["launches"] < st["calls"], st
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

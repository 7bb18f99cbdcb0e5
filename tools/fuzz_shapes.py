#!/usr/bin/env python3
"""Random parameter shapes through the key-switch suite's full-size check (tests/test_gpu_rlwe.py::_full_size_check: GadgetProduct,
Rotate, the hoisted forms, CKKS MulRelin and Rescale at the two top levels, batch of 2, every limb against the oracle):
ring degree, chain length, number and size of special primes and the arithmetic class of every modulus are drawn at random, so
that digit widths that do not divide the chain, single-limb chains, mixed classes inside one digit etc. are all met.

    python tools/fuzz_shapes.py [--seconds 240] [--seed 1]        # on a GPU box; exit status 1 when a shape fails
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattigo_amd as la  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.gpu_common import Pair  # noqa: E402
from tests.helpers import rng_for, uniform_poly  # noqa: E402
from tests.test_gpu_rlwe import _full_size_check  # noqa: E402

QBITS = [36, 40, 45, 45, 45, 46, 50, 55, 55, 58, 60]
PBITS = [45, 55, 55, 60, 61, 61]


def api_case(ctx, rng):
    """One call of the scheme-level entry points with everything drawn at random: operation, level (below the ring's and the key's
    top), a key that stops below the ring's top level, batch size, and which outputs alias which inputs (the in-place forms the
    reference allows: MulRelin(ct0, ct1, ct0 / ct1), squaring, Relinearize and Rotate in place).  Every entry against the oracle."""
    logN = int(rng.choice([11, 12, 13, 13, 14, 15, 16]))
    nq, np_ = int(rng.integers(2, 9)), int(rng.integers(1, 5))
    logq = [int(rng.choice([50, 55, 58, 60]))] + [int(rng.choice(QBITS)) for _ in range(nq - 1)]
    logp = [int(rng.choice(PBITS)) for _ in range(np_)]
    q, p = O.GenModuli(logN + 1, logq, logp)
    pr = Pair(ctx, logN, nq, np_, qmods=q, pmods=p)
    N = pr.N
    nqk = int(rng.integers(1, nq + 1))                       # the key's Q limbs
    level = int(rng.integers(0, nqk))
    B = int(rng.choice([1, 1, 2, 3, 5]))
    op = str(rng.choice(["bgv", "ckks", "relin", "rotate", "gadget", "giant", "giant"]))
    alias = int(rng.integers(0, 4))
    seed = int(rng.integers(1, 1 << 30))
    tag = f"api op={op} logN={logN} logq={logq} logp={logp} nqk={nqk} level={level} B={B} alias={alias} seed={seed}"
    r = rng_for(seed)
    gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
    beta = O.BaseRNSDecompositionVectorSize(nqk - 1, np_ - 1)
    kq = np.stack([np.stack([uniform_poly(r, q[:nqk], N) for _ in range(2)]) for _ in range(beta)])
    kp = np.stack([np.stack([uniform_poly(r, p, N) for _ in range(2)]) for _ in range(beta)])
    gk, ok = gev.NewEvaluationKey(kq, kp), O.EvaluationKey(kq, kp)
    Qm, nl = q[: level + 1], level + 1
    draw = lambda n: np.stack([np.stack([uniform_poly(r, Qm, N) for _ in range(n)]) for _ in range(B)])  # [B][n][limb][N]
    up = lambda h, n: [la.Poly(pr.gQ, nl, B).upload(np.ascontiguousarray(h[:, k])) for k in range(n)]
    fresh = lambda: [la.Poly(pr.gQ, nl, B), la.Poly(pr.gQ, nl, B)]
    T = 65537
    if op in ("bgv", "ckks"):
        ha = draw(2)
        hb = ha if alias == 3 else draw(2)
        da = up(ha, 2)
        db = da if alias == 3 else up(hb, 2)
        out = {0: fresh(), 1: da, 2: db, 3: da}[alias]
        if op == "bgv":
            gev.BGVMulRelin(level, T, da, db, gk, out)
        else:
            gev.CKKSMulRelin(level, da, db, gk, out)
        want = [oev.BGVMulRelin(T, ha[b], hb[b], ok, True) if op == "bgv" else oev.CKKSMulRelin(ha[b], hb[b], ok, True) for b in range(B)]
    elif op == "relin":
        h = draw(3)
        d = up(h, 3)
        out = [d[0] if alias & 1 else la.Poly(pr.gQ, nl, B), d[1] if alias & 2 else la.Poly(pr.gQ, nl, B)]
        gev.Relinearize(level, d, gk, out)
        want = [oev.Relinearize(h[b], ok) for b in range(B)]
    elif op == "rotate":
        h = draw(2)
        d = up(h, 2)
        gal = int(rng.choice([pow(5, int(rng.integers(1, N // 2)), 2 * N), 2 * N - 1]))
        out = d if alias & 1 else fresh()
        gev.Automorphism(level, d, gal, gk, out)
        want = [oev.Automorphism(h[b], gal, ok) for b in range(B)]
    elif op == "giant":
        # he_lintrans_giant_step (round 6): GadgetProductLazy + ringQP.Add + AutomorphismNTTWithIndex[ThenAddLazy] as one call, against
        # the oracle's separate calls; the accumulators start from arbitrary 64-bit words when accumulating
        h = draw(2)                                       # [:, 0] = cx, [:, 1] = the Q part of the addend
        hp = np.stack([uniform_poly(r, p, N) for _ in range(B)])
        acc = bool(alias & 1)
        gal = int(rng.choice([pow(5, int(rng.integers(1, N // 2)), 2 * N), 2 * N - 1]))
        prevQ = [r.integers(0, 1 << 63, size=(B, nl, N), dtype=np.uint64) * np.uint64(2) + np.uint64(alias) for _ in range(2)]
        prevP = [r.integers(0, 1 << 63, size=(B, np_, N), dtype=np.uint64) * np.uint64(2) + np.uint64(1) for _ in range(2)]
        outs = [(la.Poly(pr.gQ, nl, B).upload(prevQ[k]), la.Poly(pr.gP, np_, B).upload(prevP[k])) for k in range(2)]
        gev.LinTransGiantStep(level, la.Poly(pr.gQ, nl, B).upload(np.ascontiguousarray(h[:, 0])), gk, gal,
                              (la.Poly(pr.gQ, nl, B).upload(np.ascontiguousarray(h[:, 1])), la.Poly(pr.gP, np_, B).upload(hp)), outs, acc)
        idx = pr.oQ.AutomorphismNTTIndex(gal)
        for b in range(B):
            wQ, wP = oev.GadgetProductLazy(level, h[b, 0], ok)
            for k in range(2):
                for part, (w, add, mods, prev) in enumerate(((wQ[k], h[b, 1], Qm, prevQ[k][b]), (wP[k], hp[b], p, prevP[k][b]))):
                    v = np.array(w, dtype=np.uint64, copy=True)
                    if k == 0:
                        for i, m in enumerate(mods):
                            sm = v[i] + add[i]
                            v[i] = np.where(sm >= np.uint64(m), sm - np.uint64(m), sm)
                    wantv = v[:, idx]
                    if acc:
                        wantv = prev + wantv
                    gotv = outs[k][part].download().reshape(B, len(mods), N)[b]
                    if not np.array_equal(gotv, wantv):
                        raise AssertionError(f"{tag}: entry {b} component {k} part {part}")
        return tag
    else:
        h = draw(1)
        d = up(h, 1)
        out = fresh()
        gev.GadgetProduct(level, d[0], gk, out)
        want = [oev.GadgetProduct(level, h[b, 0], ok) for b in range(B)]
    got = [o.get().reshape(B, nl, N) for o in out]  # (a batch-1 polynomial comes back without the batch axis)
    for b in range(B):
        for k in range(2):
            if not np.array_equal(got[k][b], np.asarray(want[b])[k]):
                raise AssertionError(f"{tag}: entry {b} component {k}")
    return tag


BIN = ["Add", "AddLazy", "Sub", "SubLazy", "MulCoeffsBarrett", "MulCoeffsBarrettLazy", "MulCoeffsBarrettThenAdd",
       "MulCoeffsBarrettThenAddLazy", "MulCoeffsMontgomery", "MulCoeffsMontgomeryLazy", "MulCoeffsMontgomeryLazyThenNeg",
       "MulCoeffsMontgomeryThenAdd", "MulCoeffsMontgomeryThenAddLazy", "MulCoeffsMontgomeryLazyThenAddLazy",
       "MulCoeffsMontgomeryThenSub", "MulCoeffsMontgomeryThenSubLazy", "MulCoeffsMontgomeryLazyThenSubLazy"]
UN = ["Neg", "Reduce", "ReduceLazy", "MForm", "MFormLazy", "IMForm"]
SCAL = ["AddScalar", "SubScalar", "MulScalar", "MulScalarThenAdd", "MulScalarThenSub"]
DIV = ["DivRoundByLastModulusNTT", "DivRoundByLastModulus", "DivFloorByLastModulusNTT", "DivFloorByLastModulus"]


def ring_case(ctx, rng):
    """One ring-level call with everything drawn at random: degree (logN 4..16), chain, level below the ring's top, batch size, the
    operation (every coefficient-wise formula, the transforms, the rescales, the automorphisms) and whether the output is one of
    the inputs.  Word for word against the oracle (NTTLazy / INTTLazy after Reduce)."""
    logN = int(rng.integers(4, 17))
    nq = int(rng.integers(1, 9 if logN > 13 else 13))
    logq = [int(rng.choice(QBITS + [61])) for _ in range(nq)]
    q, _ = O.GenModuli(logN + 1, logq, [])
    pr = Pair(ctx, logN, nq, qmods=q)
    N = pr.N
    level = int(rng.integers(0, nq))
    B = int(rng.choice([1, 1, 2, 3, 7]))
    kind = str(rng.choice(["bin", "bin", "un", "scal", "ntt", "div", "auto"]))
    inplace = int(rng.integers(0, 3))
    seed = int(rng.integers(1, 1 << 30))
    r = rng_for(seed)
    Qm, nl = q[: level + 1], level + 1
    draw = lambda: np.stack([uniform_poly(r, Qm, N) for _ in range(B)])
    up = lambda h: la.Poly(pr.gQ, nl, B).upload(h)
    g, o = pr.gQ.AtLevel(level), O.Ring(N, Qm)
    a, b, c = draw(), draw(), draw()
    pa, pb, pc = up(a), up(b), up(c)
    each = lambda f: np.stack([f(i) for i in range(B)])
    name = kind
    if kind == "bin":
        name = str(rng.choice(BIN))
        out = {0: pc, 1: pa, 2: pb}[inplace]
        init = {0: c, 1: a, 2: b}[inplace]
        g.binop(name, pa, pb, out)
        want = each(lambda i: o.binop(name, a[i], b[i], init[i]))
    elif kind == "un":
        name = str(rng.choice(UN))
        src = a + (np.uint64(3) * np.array(Qm, dtype=np.uint64)[None, :, None] if "Reduce" in name else np.uint64(0))
        pa = up(src)
        out = pa if inplace else pc
        g.unop(name, pa, out)
        want = each(lambda i: o.unop(name, src[i]))
    elif kind == "scal":
        name = str(rng.choice(SCAL))
        scalar = int(rng.integers(0, 1 << 62))
        out = pc
        g.scalarop(name, pa, scalar, out)
        want = each(lambda i: o.scalarop(name, a[i], scalar, c[i]))
    elif kind == "ntt":
        name = str(rng.choice(["NTT", "INTT", "NTTLazy", "INTTLazy"]))
        out = pa if inplace else pc
        getattr(g, name)(pa, out)
        want = each(lambda i: getattr(o, name.replace("Lazy", ""))(a[i]))
        got = out.get().reshape(B, nl, N)
        got = each(lambda i: o.unop("Reduce", got[i]))
        if not np.array_equal(got, want):
            raise AssertionError(f"ring op={name} logN={logN} logq={logq} level={level} B={B} inplace={inplace} seed={seed}")
        return f"ring op={name} logN={logN} nq={nq} level={level} B={B}"
    elif kind == "div":
        if level == 0:
            return "ring op=div(level 0: skipped)"
        name = str(rng.choice(DIV))
        out = pa if inplace else la.Poly(pr.gQ, nl, B)
        getattr(g, name)(pa, out)
        want = each(lambda i: getattr(o, name)(a[i]))
        got = out.get().reshape(B, nl, N)[:, :level]
        if not np.array_equal(got, want):
            raise AssertionError(f"ring op={name} logN={logN} logq={logq} level={level} B={B} inplace={inplace} seed={seed}")
        return f"ring op={name} logN={logN} nq={nq} level={level} B={B}"
    else:
        gal = int(rng.choice([pow(5, int(rng.integers(1, max(2, N // 2))), 2 * N), 2 * N - 1]))
        name = str(rng.choice(["AutomorphismNTT", "Automorphism"]))
        out = pc
        getattr(g, name)(pa, gal, out)
        want = each(lambda i: getattr(o, name)(a[i], gal))
    got = out.get().reshape(B, nl, N)
    if not np.array_equal(got, want):
        raise AssertionError(f"ring op={name} logN={logN} logq={logq} level={level} B={B} inplace={inplace} seed={seed}")
    return f"ring op={name} logN={logN} nq={nq} level={level} B={B}"


def be_case(ctx, rng):
    """ring.BasisExtender's five entry points, the evaluator's fused ModDownQPtoQNTT and DecomposeNTT at random (levelQ, levelP),
    batch and modulus classes: the word-exact ModUp representatives, the float64 step, partial last digits."""
    logN = int(rng.integers(8, 16))
    nq, np_ = int(rng.integers(1, 13 if logN < 14 else 9)), int(rng.integers(1, 7))
    logq = [int(rng.choice([50, 55, 58, 60]))] + [int(rng.choice(QBITS)) for _ in range(nq - 1)]
    logp = [int(rng.choice(PBITS)) for _ in range(np_)]
    q, p = O.GenModuli(logN + 1, logq, logp)
    pr = Pair(ctx, logN, nq, np_, qmods=q, pmods=p)
    N = pr.N
    levelQ, levelP = int(rng.integers(0, nq)), int(rng.integers(0, np_))
    B = int(rng.choice([1, 2, 3, 5]))
    op = str(rng.choice(["ModUpQtoP", "ModUpPtoQ", "ModDownQPtoQ", "ModDownQPtoQNTT", "ModDownQPtoP", "EvalModDownQPtoQNTT", "DecomposeNTT"]))
    seed = int(rng.integers(1, 1 << 30))
    tag = f"be op={op} logN={logN} logq={logq} logp={logp} levelQ={levelQ} levelP={levelP} B={B} seed={seed}"
    r = rng_for(seed)
    Qm, Pm = q[: levelQ + 1], p[: levelP + 1]
    xq = np.stack([uniform_poly(r, Qm, N) for _ in range(B)])
    xp = np.stack([uniform_poly(r, Pm, N) for _ in range(B)])
    pq, pp = la.Poly(pr.gQ, levelQ + 1, B).upload(xq), la.Poly(pr.gP, levelP + 1, B).upload(xp)
    obe = O.BasisExtender(pr.oQ, pr.oP)
    each = lambda f: np.stack([f(i) for i in range(B)])
    if op in ("EvalModDownQPtoQNTT", "DecomposeNTT"):
        gev, oev = la.Evaluator(pr.gQ, pr.gP), O.Evaluator(pr.oQ, pr.oP)
        if op == "EvalModDownQPtoQNTT":
            out = la.Poly(pr.gQ, levelQ + 1, B)
            gev.ModDownQPtoQNTT(levelQ, levelP, pq, pp, out)
            want, got = each(lambda i: obe.ModDownQPtoQNTT(levelQ, levelP, xq[i], xp[i])), out.get().reshape(B, levelQ + 1, N)
        else:
            levelP = np_ - 1  # (the hoisting buffer is filled at the evaluator's top P level, as the reference's callers do)
            is_ntt = bool(rng.integers(0, 2))
            dec = la.Decomposition(gev, B)
            gev.DecomposeNTT(levelQ, levelP, levelP + 1, pq, is_ntt, dec)
            beta = O.BaseRNSDecompositionVectorSize(levelQ, levelP)
            for i in range(B):
                dq, dp = oev.DecomposeNTT(levelQ, levelP, levelP + 1, xq[i], is_ntt)
                for d in range(beta):
                    for l in range(levelQ + 1):
                        if not np.array_equal(dec.limb(i, d, False, l), dq[d, l]):
                            raise AssertionError(f"{tag} ntt={is_ntt}: entry {i} digit {d} Q limb {l}")
                    for l in range(levelP + 1):
                        if not np.array_equal(dec.limb(i, d, True, l), dp[d, l]):
                            raise AssertionError(f"{tag} ntt={is_ntt}: entry {i} digit {d} P limb {l}")
            return tag
    else:
        gbe = la.BasisExtender(pr.gQ, pr.gP)
        if op == "ModUpQtoP":
            out = la.Poly(pr.gP, levelP + 1, B)
            gbe.ModUpQtoP(levelQ, levelP, pq, out)
            want, nl = each(lambda i: obe.ModUpQtoP(levelQ, levelP, xq[i])), levelP + 1
        elif op == "ModUpPtoQ":
            out = la.Poly(pr.gQ, levelQ + 1, B)
            gbe.ModUpPtoQ(levelP, levelQ, pp, out)
            want, nl = each(lambda i: obe.ModUpPtoQ(levelP, levelQ, xp[i])), levelQ + 1
        elif op == "ModDownQPtoP":
            out = la.Poly(pr.gP, levelP + 1, B)
            gbe.ModDownQPtoP(levelQ, levelP, pq, pp, out)
            want, nl = each(lambda i: obe.ModDownQPtoP(levelQ, levelP, xq[i], xp[i])), levelP + 1
        else:
            out = la.Poly(pr.gQ, levelQ + 1, B)
            getattr(gbe, op)(levelQ, levelP, pq, pp, out)
            want, nl = each(lambda i: getattr(obe, op)(levelQ, levelP, xq[i], xp[i])), levelQ + 1
        got = out.get().reshape(B, nl, N)
    if not np.array_equal(got, want):
        raise AssertionError(tag)
    return tag


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="logN 13..16, chains of up to 14 limbs, up to 6 special primes")
    ap.add_argument("--api", action="store_true", help="single scheme-level calls with random level / key level / batch / aliasing")
    ap.add_argument("--be", action="store_true", help="basis extension entry points and DecomposeNTT at random levels")
    ap.add_argument("--ring", action="store_true", help="single ring-level calls (coefficient-wise formulas, transforms, rescales, automorphisms)")
    a = ap.parse_args()
    rng = np.random.Generator(np.random.PCG64(a.seed))
    ctx = la.Context(0)
    t0, n, bad = time.time(), 0, []
    while time.time() - t0 < a.seconds:
        if a.api or a.ring or a.be:
            try:
                print("ok  ", (api_case if a.api else ring_case if a.ring else be_case)(ctx, rng), flush=True)
            except Exception as e:  # noqa: BLE001
                bad.append(str(e))
                print("FAIL", e, flush=True)
                traceback.print_exc()
            n += 1
            continue
        if a.big:
            logN = int(rng.choice([13, 14, 15, 15, 16]))
            nq = int(rng.integers(3, 15))
            np_ = int(rng.integers(1, 7))
        else:
            logN = int(rng.choice([10, 11, 12, 13, 13, 14, 15, 16]))
            nq = int(rng.integers(2, 11 if logN < 15 else 8))
            np_ = int(rng.integers(1, 5))
        logq = [int(rng.choice([50, 55, 58, 60]))] + [int(rng.choice(QBITS)) for _ in range(nq - 1)]
        logp = [int(rng.choice(PBITS)) for _ in range(np_)]
        seed = int(rng.integers(1, 1 << 30))
        rot = bool(rng.integers(0, 2))
        tag = f"logN={logN} logq={logq} logp={logp} seed={seed} rotate={rot}"
        try:
            _full_size_check(ctx, logN, logq, logp, seed, rot)
            print("ok  ", tag, flush=True)
        except Exception:  # noqa: BLE001
            bad.append(tag)
            print("FAIL", tag, flush=True)
            traceback.print_exc()
        n += 1
    print(f"{n} shapes, {len(bad)} failed", flush=True)
    for b in bad:
        print("  ", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

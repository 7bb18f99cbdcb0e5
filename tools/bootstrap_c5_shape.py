#!/usr/bin/env python3
"""BASELINE config 5 shaped measurement: the operation trace of one CKKS bootstrap at the reference's default parameters
N16QP1546H192H32 (circuits/ckks/bootstrapping/default_parameters.go:33-42): logN = 16, Q = 10 residual + 3 (SlotsToCoeffs)
+ 8 (EvalMod, 60-bit) + 4 (CoeffsToSlots) = 25 limbs, P = 5 x 61 bits (alpha = 5), sparse-secret encapsulation, K = 16,
degree 30, 3 double angles.  Keys, DFT-matrix diagonals and the input are synthetic uniform polynomials (throughput does
not depend on their values; the functional test of the same driver is tests/test_gpu_circuits.py::
test_toy_bootstrapping_end_to_end).  The DFT factors have the diagonal structure of merged radix-2 layers:
{j * stride : |j| < 2^k}.  Prints one JSON line."""
import json
import os
import sys
import time
from fractions import Fraction

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))  # drivers/: the reference's host drivers restated (test scaffolding)
import lattigo_amd as la  # noqa: E402
from bench import uniform  # noqa: E402
from drivers import bootstrapping as BS  # noqa: E402
from drivers import lintrans as LT  # noqa: E402
from drivers import mod1 as M1  # noqa: E402
from lattigo_amd import rlwe as R  # noqa: E402
from drivers import schemes as S  # noqa: E402

# moduli of the right sizes, = 1 mod 2^17 (generated once with the reference's prime search; values only set the sizes)
LOGQ = [60] + [40] * 9 + [39] * 3 + [60] * 8 + [56] * 4
LOGP = [61] * 5


def gen_primes(logs, nth, taken):
    from lattigo_amd import _lib  # noqa: F401
    out = []
    for b in logs:
        x = (1 << b) + 1
        step = nth if b < 61 else -nth  # 61-bit primes are searched downwards: moduli stay below 2^61 (ring/ntt.go:169)
        while True:
            x += step
            if x not in taken and pow(2, x - 1, x) == 1 and all(pow(a, x - 1, x) == 1 for a in (3, 5, 7, 11)):
                out.append(x)
                taken.add(x)
                break
    return out


def build(ctx, B, seed_offset=0, same_input=False):
    """-> (run, info): run() performs one bootstrap of a batch of B ciphertexts on `ctx` (bench.py --workload c5);
    info carries the shape facts and the objects main() needs for its phase report.
    ctx = None: the SAME trace (same seed, same synthetic keys / diagonals / input) on the CPU oracle backend (oracle/circuits.py),
    batch 1 -- the checker: tests/golden/gen_c5_trace_digest.py commits the digest of its output, bench.py compares the device's.
    same_input: every batch entry carries entry 0's ciphertext, so that one oracle run (about a minute of CPU) checks them all."""
    device = ctx is not None
    logN = 16
    N, n, nth = 1 << logN, 1 << (logN - 1), 2 << logN
    taken = set()
    q, p = gen_primes(LOGQ, nth, taken), gen_primes(LOGP, nth, taken)
    if device:
        rq, rp = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
        ev = la.Evaluator(rq, rp)
    else:
        from oracle import circuits as OC
        from oracle import oracle as O
        assert B == 1
        rq, rp = O.Ring(N, q), O.Ring(N, p)
        ev = O.Evaluator(rq, rp)
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 5 + seed_offset))
    top, LP = len(q) - 1, len(p)
    beta = (top + 1 + LP - 1) // LP
    kq, kp = uniform(rng, q, N, (beta, 2)), uniform(rng, p, N, (beta, 2))  # one synthetic key image, uploaded per key

    def key():
        return ev.NewEvaluationKey(kq, kp) if device else O.EvaluationKey(kq, kp)

    dq, dp = uniform(rng, q, N), uniform(rng, p, N)

    def diag(level):
        if not device:
            return (dq[: level + 1], dp)
        return (la.Poly(rq, level + 1).upload(dq[: level + 1]), la.Poly(rp, LP).upload(dp))

    gks = R.GaloisKeySet() if device else None
    keys = gks.keys if device else {}
    mats = {"cts": [], "stc": []}
    # CoeffsToSlots: 4 factors at the top levels (4+4+4+3 radix-2 layers); SlotsToCoeffs: 3 factors (5+5+5)
    plan = [("cts", top - i, k, 1 << s) for i, (k, s) in enumerate([(4, 11), (4, 7), (4, 3), (3, 0)])]
    stc_top = top - 4 - 8
    plan += [("stc", stc_top - i, k, 1 << s) for i, (k, s) in enumerate([(5, 0), (5, 5), (5, 10)])]
    ndiag = 0
    for which, level, k, stride in plan:
        diags = sorted({(j * stride) % n for j in range(-(1 << k) + 1, 1 << k)})
        N1 = LT.FindBestBSGSRatio(diags, n, 1)
        _, r1, r2 = LT.BSGSIndex(diags, n, N1)
        for r in set(r1) | set(r2):
            g = R.GaloisElement(nth, r)
            if r and g not in keys:
                keys[g] = key()
        LTC = LT.LinearTransformation if device else OC.LinearTransformation
        mats[which].append(LTC({d: diag(level) for d in diags}, level, LP - 1, n, N1))
        ndiag += len(diags)
    keys[nth - 1] = key()
    rlk, d2s, s2d = key(), key(), key()
    if device:
        gce = S.CKKSCiphertextEvaluator(ev, rlk)
        lte = LT.LinTransEvaluator(ev, gks)
        be = BS.DeviceBootstrapBackend(gce, lte, R.InnerSumEvaluator(ev, gks), d2s, s2d)
    else:
        gce = OC.CKKSCtEvaluator(ev, rlk)
        be = OC.OracleBootstrapBackend(gce, OC.LinTransEvaluator(ev, keys), OC.InnerSumEvaluator(ev, keys), d2s, s2d)
    pm = M1.Mod1Parameters(q[0], LevelQ=top - 4, LogScale=60, Mod1Type=M1.CosContinuous, K=16, Mod1Degree=30, DoubleAngle=3)
    scales = lambda ms: [Fraction(q[m.LevelQ]) for m in ms]
    boot = BS.Bootstrapper(be, M1.Mod1Evaluator(gce, pm), mats["cts"], scales(mats["cts"]), mats["stc"], scales(mats["stc"]),
                           modup_scale=256.0)
    host0 = [uniform(rng, q[:1], N, (1 if same_input or not device else B,)) for _ in range(2)]
    if device:
        ct0 = [la.Poly(rq, 1, B).upload(np.repeat(h, B, axis=0) if same_input else h) for h in host0]
        run = lambda: boot.Bootstrap(S.Ciphertext(ct0, 0, 1), Fraction(1 << 60))
    else:
        ct0 = [h[0] for h in host0]
        run = lambda: boot.Bootstrap(OC.Ct(list(ct0), 1), Fraction(1 << 60))

    info = {"logN": logN, "L": len(q), "alpha": LP, "galois_keys": len(keys), "dft_diagonals": ndiag}
    run_ = lambda: run()
    run_._parts = (boot, be, ct0, gks, ndiag)  # for main()'s phase report
    run_._shape = {"N": N, "q": q, "p": p, "kq": kq, "kp": kp, "ev": ev, "rq": rq, "key": rlk, "gal": R.GaloisElement(nth, 1)}  # bench.py's CPU leg
    return run_, info


def trace_digest(res, device: bool) -> dict:
    """SHA-256 of every batch entry's refreshed ciphertext (both polynomials, limbs 0..level) of one run()"""
    import hashlib
    if device:
        words = [v.download()[:, : res.level + 1] for v in res.Value]  # [B][limbs][N] per component
        B = words[0].shape[0]
        per = [hashlib.sha256(np.ascontiguousarray(np.stack([w[b] for w in words])).tobytes()).hexdigest() for b in range(B)]
    else:
        per = [hashlib.sha256(np.ascontiguousarray(np.stack([np.asarray(v)[: res.level + 1] for v in res.Value]), dtype=np.uint64).tobytes()).hexdigest()]
    return {"level": int(res.level), "entries": per}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    logN = 16
    ctx = la.Context(0)
    run, _ = build(ctx, B)
    boot, be, ct0, gks, ndiag = run._parts
    res = run()
    ctx.sync()
    iters = 3
    t0 = time.perf_counter()
    for _ in range(iters):
        res = run()
    ctx.sync()
    dt = (time.perf_counter() - t0) / iters
    ctx.prof_begin()
    run()
    prof = ctx.prof_end()
    # phase timing (host wall clock with a device sync after each phase)
    phases = {}
    up = None

    def timed(name, fn):
        ctx.sync()
        t = time.perf_counter()
        out = fn()
        ctx.sync()
        phases[name] = round((time.perf_counter() - t) * 1e3, 2)
        return out

    logSlots = logN - 1
    up = timed("ModUp", lambda: be.modup(S.Ciphertext(ct0, 0, 1), 256.0, logSlots))
    up.Scale = Fraction(1 << 60)
    re_im = timed("CoeffsToSlots", lambda: boot.CoeffsToSlots(up))
    r1 = timed("EvalMod(real)", lambda: boot.mod1.EvaluateNew(re_im[0]))
    r2 = timed("EvalMod(imag)", lambda: boot.mod1.EvaluateNew(re_im[1]))
    timed("SlotsToCoeffs", lambda: boot.SlotsToCoeffs(r1, r2))
    print(json.dumps({"config": "c5-shape", "what": "CKKS bootstrap op trace, logN=16, 25+5 limbs (N16QP1546H192H32 shape), "
                      "synthetic keys / DFT diagonals", "batch": B, "s_per_batch": dt, "bootstraps_per_s": B / dt,
                      "output_level": res.level, "galois_keys": len(gks.keys), "dft_diagonals": ndiag,
                      "kernel_launches": int(sum(v[0] for v in prof.values())), "phase_ms": phases,
                      "kernel_ms": {k: round(v[1], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}}))


if __name__ == "__main__":
    main()

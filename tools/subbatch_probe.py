#!/usr/bin/env python3
"""Infinity-Cache-sized sub-batches: one step of B = 256 MulRelin run as S sub-batches of b ciphertexts,
round-robin over C contexts (HIP streams), so that a sub-batch's decomposition (12 MiB of digits per ciphertext) is consumed by
the NTT + MAC kernel before it leaves the 256 MiB Infinity Cache.  Prints ops/s per (b, C); `python tools/subbatch_probe.py`
(`python tools/subbatch_probe.py 256 b C`: that configuration alone, 2 + 3 steps -- tools/subbatch_pmc.sh wraps it in the PMC passes)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattigo_amd as la  # noqa: E402
from bench import LOGN, T, gen_moduli, uniform  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = 10
    N = 1 << LOGN
    q, p = gen_moduli()
    L, alpha = len(q), len(p)
    beta = (L + alpha - 1) // alpha
    rng = np.random.Generator(np.random.PCG64(1))
    kq, kp = uniform(rng, q, N, (beta, 2)), uniform(rng, p, N, (beta, 2))
    ctxs = [la.Context(0) for _ in range(3)]
    evs = []
    for c in ctxs:
        rq, rp = la.Ring(c, N, q), la.Ring(c, N, p)
        ev = la.Evaluator(rq, rp)
        evs.append((rq, ev, ev.NewEvaluationKey(kq, kp)))
    only = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None  # one (b, C) configuration: for a rocprofv3 pass
    if only:
        steps = 3
    for b in (256, 128, 64, 32, 16, 8):
        for C in (1, 2, 3):
            if only and (b, C) != only:
                continue
            if b == B and C > 1:
                continue
            S = B // b
            work = []  # (ctx index, step fn)
            for s in range(S):
                j = s % C
                rq, ev, rlk = evs[j]
                a = [la.Poly(rq, L, b, zero=False) for _ in range(2)]
                bb = [la.Poly(rq, L, b, zero=False) for _ in range(2)]
                out = [la.Poly(rq, L, b, zero=False), la.Poly(rq, L, b, zero=False)]
                work.append((lambda ev=ev, a=a, bb=bb, rlk=rlk, out=out: ev.BGVMulRelin(L - 1, T, a, bb, rlk, out)))
            for _ in range(2):
                [f() for f in work]
            [c.sync() for c in ctxs]
            t0 = time.perf_counter()
            for _ in range(steps):
                [f() for f in work]
            [c.sync() for c in ctxs]
            dt = time.perf_counter() - t0
            print(f"sub-batch {b:4d} x {S:3d} on {C} stream(s): {B * steps / dt:9,.0f} ops/s ({dt / steps * 1e3:.3f} ms/step)", flush=True)
            del work


if __name__ == "__main__":
    main()

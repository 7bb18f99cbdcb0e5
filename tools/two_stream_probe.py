#!/usr/bin/env python3
"""Does running two half-batches on two contexts (two HIP streams) overlap the ALU-bound and the HBM-bound kernels of the
MulRelin pipeline?  Prints ops/s for 1 x B and for S x (B/S)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattigo_amd as la  # noqa: E402
from bench import LOGN, T, gen_moduli, uniform  # noqa: E402


def build(ctx, B, rng, q, p):
    N = 1 << LOGN
    L, alpha = len(q), len(p)
    beta = (L + alpha - 1) // alpha
    rq, rp = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
    ev = la.Evaluator(rq, rp)
    rlk = ev.NewEvaluationKey(uniform(rng, q, N, (beta, 2)), uniform(rng, p, N, (beta, 2)))
    a = [la.Poly(rq, L, B).upload(uniform(rng, q, N, (B,))) for _ in range(2)]
    b = [la.Poly(rq, L, B).upload(uniform(rng, q, N, (B,))) for _ in range(2)]
    out = [la.Poly(rq, L, B), la.Poly(rq, L, B)]
    return lambda: ev.BGVMulRelin(L - 1, T, a, b, rlk, out)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = 20
    q, p = gen_moduli()
    rng = np.random.Generator(np.random.PCG64(1))
    for S in (1, 2, 4):
        ctxs = [la.Context(0) for _ in range(S)]
        fns = [build(c, B // S, rng, q, p) for c in ctxs]
        for _ in range(3):
            [f() for f in fns]
        [c.sync() for c in ctxs]
        t0 = time.perf_counter()
        for _ in range(steps):
            [f() for f in fns]
        [c.sync() for c in ctxs]
        dt = time.perf_counter() - t0
        print(f"streams={S} batch/stream={B // S}: {B * steps / dt:,.0f} ops/s ({dt / steps * 1e3:.3f} ms/step)", flush=True)
        del fns, ctxs


if __name__ == "__main__":
    main()

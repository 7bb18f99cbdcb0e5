#!/usr/bin/env python3
"""Per-kernel time of the stand-alone Ring.NTT / INTT (bench.py's NTT/s leg) at logN = 15 / 16: tools/ntt_prof.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattigo_amd as la
import bench

ctx = la.Context(0)
rng = np.random.Generator(np.random.PCG64(1))
for logN, mods, B in ((15, bench.gen_moduli()[0], 64), (16, bench.C4_Q, 32), (14, bench.gen_moduli()[0][:8], 128)):
    N = 1 << logN
    r = la.Ring(ctx, N, mods)
    x = la.Poly(r, len(mods), B).upload(bench.uniform(rng, mods, N, (B,)))
    for inv in (False, True):
        f = (lambda: r.INTT(x, x)) if inv else (lambda: r.NTT(x, x))
        for _ in range(3):
            f()
        ctx.prof_begin()
        for _ in range(10):
            f()
        prof = ctx.prof_end()
        tot = sum(v[1] for v in prof.values()) / 10
        gb = 2 * len(mods) * B * N * 8 / 1e9
        print(json.dumps({"logN": logN, "inverse": inv, "limbs": len(mods), "batch": B, "ms": round(tot, 4), "alg_TBs": round(gb / tot, 3),
                          "kernels_ms": {k: round(v[1] / 10, 4) for k, v in prof.items()},
                          "kernel_TBs": {k: round(gb * (sum(1 for m in mods if (m < (1 << 47)) == ('f64' in k)) / len(mods) if 'rows' in k else 1.0) / (v[1] / 10), 2) for k, v in prof.items()}}))

#!/usr/bin/env python3
"""Stand-alone Ring.NTT rate against the batch per call: does a working set that fits the 256 MiB Infinity Cache (the two
passes of a transform touch every limb twice) change the rate?  Prints limb-NTT/s per (logN, batch)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattigo_amd as la  # noqa: E402
from bench import C4_Q, gen_moduli, uniform  # noqa: E402

ctx = la.Context(0)
for logN, mods in ((15, gen_moduli()[0]), (16, C4_Q)):
    N = 1 << logN
    r = la.Ring(ctx, N, mods)
    for B in (4, 8, 16, 32, 64, 128, 256):
        total = 256 if logN == 15 else 128      # entries transformed per timed pass, as total // B calls
        # uniform random words: an all-zero (or recycled) buffer draws less power and clocks higher -- a first version of this sweep
        # on unwritten scratch read 4.75 M limb-NTT/s at batch 256 where real data gives 4.1-4.4
        rng = np.random.Generator(np.random.PCG64(B))
        xs = [la.Poly(r, len(mods), B, zero=False).upload(uniform(rng, mods, N, (B,))) for _ in range(max(1, total // B))]
        for inverse in (False, True):
            f = r.INTT if inverse else r.NTT
            for x in xs:
                f(x, x)
            ctx.timer_start()
            for _ in range(10):
                for x in xs:
                    f(x, x)
            ms = ctx.timer_stop() / 10
            n = len(xs) * B * len(mods)
            print(f"logN={logN} batch={B:3d} x {len(xs):2d} calls ({B * len(mods) * N * 8 / 2**20:6.0f} MiB per call) {'INTT' if inverse else ' NTT'}: "
                  f"{n / (ms * 1e-3) / 1e6:6.2f} M limb-NTT/s, {2 * n * N * 8 / (ms * 1e-3) / 1e12:5.2f} TB/s algorithmic", flush=True)
        del xs

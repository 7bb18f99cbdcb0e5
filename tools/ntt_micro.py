"""NTT-only microbenchmark (for rocprofv3 PMC passes): forward NTT of B polys with L limbs at logN."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lattigo_amd as la
import bench

logN = int(sys.argv[1]) if len(sys.argv) > 1 else 15
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
q, p = bench.gen_moduli()
ctx = la.Context(0)
ring = la.Ring(ctx, 1 << logN, q)
rng = np.random.default_rng(0)
x = la.Poly(ring, len(q), B).upload(bench.uniform(rng, q, 1 << logN, (B,)))
y = la.Poly(ring, len(q), B)
for _ in range(3):
    ring.NTT(x, y)
ctx.timer_start()
for _ in range(iters):
    ring.NTT(x, y)
ms = ctx.timer_stop()
print(f"logN={logN} B={B} L={len(q)}: {iters * len(q) * B / (ms * 1e-3):.3e} limb-NTT/s, {ms / iters * 1e3:.1f} us per call")

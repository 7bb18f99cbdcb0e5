"""Why BASELINE's second metric (stand-alone Ring.NTT) read 4.39 M limb-NTT/s in round 3's line and 4.1 M since:
the round-3 line measured 20 transforms of a batch of 64 (192 MiB working set, 3.5 ms of GPU time), rounds 4-5 measure 20 transforms
of a batch of 256 (768 MiB, 15 ms).  This probe runs the same kernels both ways -- batch 64 / 256, 20 / 200 / 1000 back-to-back calls,
cold (after 2 s of idle) and hot (right after 0.5 s of MulRelin) -- and reads the shader clock while each run is in flight.
usage (GPU box): python tools/ntt_drift_probe.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import lattigo_amd as la
from lattigo_amd.dist import ControlPlane
import argparse

ctx = la.Context(0)
rng = np.random.Generator(np.random.PCG64(7))
q = bench.gen_moduli()[0]
N = 1 << 15
r = la.Ring(ctx, N, q)
W = bench.setup_c3(la, ctx, 0, 128, ControlPlane(), argparse.Namespace(replicate_keys="none"))
out = []
for B in (64, 256):
    x = la.Poly(r, len(q), B).upload(bench.uniform(rng, q, N, (B,)))
    for state in ("cold", "hot"):
        for iters in (20, 200, 1000):
            if state == "cold":
                ctx.sync(); time.sleep(2.0)
            else:
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.5:
                    W["step"]()
                ctx.sync()
            ctx.timer_start()
            for _ in range(iters):
                r.NTT(x, x)
            clk = bench.gpu_clock(0)
            ms = ctx.timer_stop() / iters
            rec = {"batch": B, "state": state, "iters": iters, "ms": round(ms, 4), "M_limb_ntt_per_s": round(len(q) * B / ms / 1e3, 3),
                   "sclk_under_load": clk and clk.get("sclk_mhz"), "power_w": clk and clk.get("power_w")}
            out.append(rec)
            print(json.dumps(rec), flush=True)
    del x

#!/usr/bin/env python3
"""One key switch split over the ranks by digit (lattigo_amd/dist.py SplitGadgetProductHoisted) against the unsplit call, at the
BASELINE config-4 shape (CKKS logN=16, 20+4 limbs, beta = 5 digits).  Run under torch.distributed.run, one rank per GPU:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=N --master-addr 127.0.0.1 tools/split_probe.py [--transport rccl|host] [--batch B]
(HERING_FORCE_DEVICE=0 lets several ranks share one GPU for a functional run).  Rank 0 prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--transport", default="rccl")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--logn", type=int, default=16)
args = ap.parse_args()
if args.transport == "rccl":
    import torch  # noqa: F401  (before libhering: one HIP runtime in the process)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattigo_amd as la  # noqa: E402
from bench import uniform  # noqa: E402
from lattigo_amd import rlwe as R  # noqa: E402
from lattigo_amd.dist import ControlPlane  # noqa: E402
from tools.bench_configs import C4_P, C4_Q  # noqa: E402

cp = ControlPlane()
ctx = la.Context(int(os.environ.get("HERING_FORCE_DEVICE", cp.local_rank)))
N = 1 << args.logn
if args.logn != 16:  # functional runs at a smaller ring: same chain shape from the oracle's prime search is not needed, reuse sizes
    from oracle import oracle as O
    q, p = O.GenModuli(args.logn + 1, [60] + [45] * 19, [61] * 4)
else:
    q, p = C4_Q, C4_P
rq, rp = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
ev = la.Evaluator(rq, rp)
L, B, beta = len(q), args.batch, 5
rng = np.random.default_rng(7)
key = ev.NewEvaluationKey(uniform(rng, q, N, (beta, 2)), uniform(rng, p, N, (beta, 2)))
cx = la.Poly(rq, L, B).upload(uniform(rng, q, N, (B,)))
dec = R.Decomposition(ev, B)
ct = [la.Poly(rq, L, B), la.Poly(rq, L, B)]
ref = [la.Poly(rq, L, B), la.Poly(rq, L, B)]


def timed(fn):
    for _ in range(2):
        fn()
    ctx.sync(); cp.barrier()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        fn()
    ctx.sync(); cp.barrier()
    return cp.max_over_ranks(time.perf_counter() - t0) / args.iters


def unsplit():
    ev.DecomposeNTT(L - 1, len(p) - 1, len(p), cx, True, dec)
    ev.GadgetProductHoisted(L - 1, dec, key, ref)


def split():
    ev.DecomposeNTT(L - 1, len(p) - 1, len(p), cx, True, dec)
    cp.SplitGadgetProductHoisted(ev, L - 1, dec, key, ct, transport=args.transport)


t_un, t_sp = timed(unsplit), timed(split)
same = all(np.array_equal(a.get(), b.get()) for a, b in zip(ct, ref))
if cp.rank == 0:
    print(json.dumps({"what": "one hoisted key switch, digits split over ranks vs every rank computing it whole", "logN": args.logn,
                      "world": cp.world, "batch": B, "transport": args.transport, "unsplit_ms": round(t_un * 1e3, 3),
                      "split_ms": round(t_sp * 1e3, 3), "identical_words": bool(same),
                      "allreduce_MiB_per_ciphertext": round(2 * (L + len(p)) * N * 8 / 2 ** 20, 1)}))
cp.close()
sys.exit(0 if same else 1)

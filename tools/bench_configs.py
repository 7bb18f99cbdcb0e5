#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE configs (bench.py stays on the headline config 3):
  c1  ring.NTT forward+inverse, N=2^12, one 61-bit prime          (NTT pairs/s, device-resident and host-slice)
  ntt limb-NTT/s at logN = 15 and 16 on the config-3 / config-4 chains
  c2  CKKS logN=14, LogQ=[50,40x7], LogP=[60]: Mul+Rescale and MulRelin+Rescale
  c4  CKKS logN=16, LogQ=[60,45x19], LogP=[61x4]: Rotate (automorphism + Galois key-switch)
  int61  CKKS logN=15, 8 + 4 limbs of 61 bits (the reference's ring test primes): MulRelin on the all-integer kernels
Synthetic uniform inputs, HIP-event timing on the context stream, one JSON line per measurement."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # drivers/ (test scaffolding)
import lattigo_amd as la  # noqa: E402
from bench import uniform  # noqa: E402

# GenModuli outputs (core/rlwe/params.go:811) for the configs, pinned
C2_Q = [1125899908022273, 1099511922689, 1099512938497, 1099510054913, 1099514314753, 1099514478593, 1099508121601,
        1099507695617]
C2_P = [1152921504606748673]
C4_Q = [1152921504606584833, 35184372744193, 35184373006337, 35184368025601, 35184376545281, 35184377331713, 35184378511361,
        35184379035649, 35184365273089, 35184380870657, 35184363569153, 35184382967809, 35184383229953, 35184383754241,
        35184385196033, 35184358850561, 35184386899969, 35184388734977, 35184355704833, 35184353083393]
C4_P = [2305843009211596801, 2305843009210023937, 2305843009208713217, 2305843009202159617]


def timed(ctx, fn, iters, warm=3):
    for _ in range(warm):
        fn()
    ctx.timer_start()
    for _ in range(iters):
        fn()
    return ctx.timer_stop() / iters


def main():
    ctx = la.Context(0)
    rng = np.random.Generator(np.random.PCG64(0x1A77160))
    out = []

    # ---- c1
    N = 1 << 12
    q1 = [0x1fffffffffe00001]
    r1 = la.Ring(ctx, N, q1)
    for B in (1, 4096):
        x = la.Poly(r1, 1, B).upload(uniform(rng, q1, N, (B,)))
        ms = timed(ctx, lambda: (r1.NTT(x, x), r1.INTT(x, x)), 50)
        out.append({"config": "c1", "what": "ring.NTT+INTT N=2^12 1 prime, device resident", "batch": B, "ms": ms,
                    "ntt_pairs_per_s": B / (ms * 1e-3), "alg_GBs": B * 131072 / (ms * 1e-3) / 1e9})
    xh = uniform(rng, q1, N)[0]
    import time
    t0 = time.perf_counter()
    for _ in range(200):
        xh = r1.Backward(0, r1.Forward(0, xh))
    out.append({"config": "c1", "what": "host-slice NumberTheoreticTransformer (H2D+D2H per call)",
                "ntt_pairs_per_s": 200 / (time.perf_counter() - t0)})

    # ---- NTT/s at logN 15 / 16
    from bench import gen_moduli
    q3, p3 = gen_moduli()
    for logN, mods, B in ((15, q3, 64), (16, C4_Q, 32)):
        r = la.Ring(ctx, 1 << logN, mods)
        x = la.Poly(r, len(mods), B).upload(uniform(rng, mods, 1 << logN, (B,)))
        ms = timed(ctx, lambda: r.NTT(x, x), 20)
        out.append({"config": "ntt", "what": f"Ring.NTT logN={logN}, {len(mods)} limbs, batch {B}", "ms": ms,
                    "limb_ntt_per_s": len(mods) * B / (ms * 1e-3), "alg_GBs": 2 * len(mods) * B * (8 << logN) / (ms * 1e-3) / 1e9})

    # ---- c2
    N = 1 << 14
    rq, rp = la.Ring(ctx, N, C2_Q), la.Ring(ctx, N, C2_P)
    ev = la.Evaluator(rq, rp)
    L, B = len(C2_Q), 128
    rlk = ev.NewEvaluationKey(uniform(rng, C2_Q, N, (L, 2)), uniform(rng, C2_P, N, (L, 2)))
    a = [la.Poly(rq, L, B).upload(uniform(rng, C2_Q, N, (B,))) for _ in range(2)]
    b = [la.Poly(rq, L, B).upload(uniform(rng, C2_Q, N, (B,))) for _ in range(2)]
    o3 = [la.Poly(rq, L, B) for _ in range(3)]
    r3 = [la.Poly(rq, L - 1, B) for _ in range(3)]
    ms = timed(ctx, lambda: (ev.CKKSMulRelin(L - 1, a, b, None, o3), ev.Rescale(L - 1, 1, o3, r3)), 10)
    out.append({"config": "c2", "what": "CKKS logN=14 L=8: Mul + Rescale", "batch": B, "ms": ms, "ops_per_s": B / (ms * 1e-3),
                "alg_GBs": B * 12.625 * 2**20 / (ms * 1e-3) / 1e9})
    ms = timed(ctx, lambda: (ev.CKKSMulRelin(L - 1, a, b, rlk, o3[:2]), ev.Rescale(L - 1, 1, o3[:2], r3[:2])), 10)
    out.append({"config": "c2", "what": "CKKS logN=14 L=8 alpha=1: MulRelin + Rescale", "batch": B, "ms": ms,
                "ops_per_s": B / (ms * 1e-3), "alg_GBs": B * 27.75 * 2**20 / (ms * 1e-3) / 1e9})

    # ---- c4
    N = 1 << 16
    rq, rp = la.Ring(ctx, N, C4_Q), la.Ring(ctx, N, C4_P)
    ev = la.Evaluator(rq, rp)
    L, B, beta = len(C4_Q), 16, 5
    gk = ev.NewEvaluationKey(uniform(rng, C4_Q, N, (beta, 2)), uniform(rng, C4_P, N, (beta, 2)))
    ct = [la.Poly(rq, L, B).upload(uniform(rng, C4_Q, N, (B,))) for _ in range(2)]
    o2 = [la.Poly(rq, L, B) for _ in range(2)]
    gal = pow(5, 1, 2 * N)
    ms = timed(ctx, lambda: ev.Automorphism(L - 1, ct, gal, gk, o2), 10)
    out.append({"config": "c4", "what": "CKKS logN=16 L=20 alpha=4: Rotate", "batch": B, "ms": ms, "ops_per_s": B / (ms * 1e-3),
                "alg_GBs": B * 160 * 2**20 / (ms * 1e-3) / 1e9})
    dec = la.Decomposition(ev, B)
    ev.DecomposeNTT(L - 1, 3, 4, ct[1], True, dec)
    ms = timed(ctx, lambda: ev.AutomorphismHoisted(L - 1, ct, dec, gal, gk, o2), 10)
    out.append({"config": "c4", "what": "CKKS logN=16: hoisted Rotate (decomposition shared)", "batch": B, "ms": ms,
                "ops_per_s": B / (ms * 1e-3)})
    # ---- lintrans / inner sum on the config-4 chain: host drivers over the device operators
    from drivers import lintrans as LT
    from lattigo_amd import rlwe as R
    B = 4
    nth, slots = 2 * N, N // 2
    ct = [la.Poly(rq, L, B).upload(uniform(rng, C4_Q, N, (B,))) for _ in range(2)]
    gks = R.GaloisKeySet()

    def need(galels):
        for g in galels:
            if g not in gks.keys:
                gks.keys[g] = ev.NewEvaluationKey(uniform(rng, C4_Q, N, (beta, 2)), uniform(rng, C4_P, N, (beta, 2)))

    def diag():
        return (la.Poly(rq, L).upload(uniform(rng, C4_Q, N)), la.Poly(rp, len(C4_P)).upload(uniform(rng, C4_P, N)))

    lte = LT.LinTransEvaluator(ev, gks)
    for name, diags, N1 in (("BSGS 32 diagonals (N1=8: 7 baby + 3 giant rotations)", list(range(32)), 8),
                            ("naive 8 diagonals (single hoisting)", list(range(8)), 0)):
        need(LT.GaloisElements(nth, diags, slots, -1) if N1 == 0 else
             [R.GaloisElement(nth, r) for r in sum(LT.BSGSIndex(diags, slots, N1)[1:], []) if r])
        lt = LT.LinearTransformation({d: diag() for d in diags}, L - 1, len(C4_P) - 1, slots, N1)
        o2 = [la.Poly(rq, L, B) for _ in range(2)]
        ms = timed(ctx, lambda: lte.EvaluateMany(L - 1, ct, [lt], [o2]), 3, warm=1)
        ctx.prof_begin()
        lte.EvaluateMany(L - 1, ct, [lt], [o2])
        prof = ctx.prof_end()
        out.append({"config": "lintrans", "what": f"CKKS logN=16 L=20 alpha=4: lintrans.EvaluateMany, {name}", "batch": B,
                    "ms": ms, "ops_per_s": B / (ms * 1e-3),
                    "kernel_ms": {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}})
        del lt
    ise = R.InnerSumEvaluator(ev, gks)
    need(R.GaloisElementsForInnerSum(nth, 1, 64))
    o2 = [la.Poly(rq, L, B) for _ in range(2)]
    ms = timed(ctx, lambda: ise.InnerSum(L - 1, ct, 1, 64, o2), 3, warm=1)
    out.append({"config": "innersum", "what": "CKKS logN=16 L=20 alpha=4: InnerSum(batch=1, n=64) (6 hoisted rotations)",
                "batch": B, "ms": ms, "ops_per_s": B / (ms * 1e-3)})
    # ---- an all-integer chain: the reference's own 61-bit ring test primes (ring/test_params.go:15-32), logN = 15, 8 + 4 limbs.
    # No modulus below 2^47: every kernel takes its integer path (Harvey-form row transforms, the LDS-parking basis extension of
    # a 4-limb digit with three column stages, 128-bit inner product)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Pi60, Qi60
    N, q61, p61, B = 1 << 15, Qi60[:8], Pi60[:4], 64
    rq, rp = la.Ring(ctx, N, q61), la.Ring(ctx, N, p61)
    ev = la.Evaluator(rq, rp)
    L, beta = len(q61), 2
    rlk = ev.NewEvaluationKey(uniform(rng, q61, N, (beta, 2)), uniform(rng, p61, N, (beta, 2)))
    a = [la.Poly(rq, L, B).upload(uniform(rng, q61, N, (B,))) for _ in range(2)]
    b = [la.Poly(rq, L, B).upload(uniform(rng, q61, N, (B,))) for _ in range(2)]
    o2 = [la.Poly(rq, L, B) for _ in range(2)]
    ms = timed(ctx, lambda: ev.CKKSMulRelin(L - 1, a, b, rlk, o2), 10)
    ctx.prof_begin()
    ev.CKKSMulRelin(L - 1, a, b, rlk, o2)
    prof = ctx.prof_end_bytes()
    limb = N * 8
    out.append({"config": "int61", "what": "CKKS logN=15, 8 Q + 4 P limbs of 61 bits (ring/test_params.go primes): MulRelin, all-integer kernels",
                "batch": B, "ms": ms, "ops_per_s": B / (ms * 1e-3),
                "alg_GBs": B * (6 * L + 2 * beta * (L + 4)) * limb / (ms * 1e-3) / 1e9,
                "kernel_ms": {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
                "kernel_GBs": {k: round(v[2] / (v[1] * 1e-3) / 1e9) for k, v in prof.items() if v[1] > 0}})
    for line in out:
        print(json.dumps(line))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ring-arithmetic backend.

Workload (BASELINE.json configs[2], the one its metric "ciphertext-mul+relin ops/s at logN=15"
is quoted on): BGV, logN=15, 12 Q-limbs (LogQ=[55,45x11]), 3 P-limbs (LogP=[55x3]), T=65537,
ct x ct Mul + Relinearize (schemes/bgv/evaluator.go:592 tensorStandard + GadgetProduct).
A "step" is one MulRelin over a batch of B independent ciphertext pairs already resident in HBM.
Synthetic inputs: coefficients uniform in [0, q_i), PCG64 seed 0x1A77160 + 2.

    python bench.py --gpus N --steps K --warmup W [--batch B]

N > 1: launched by torch.distributed.run, one rank per GPU; independent ciphertexts are sharded
across ranks (weak scaling, no data-path collective -- SURVEY.md section 8e); ranks synchronise only
for the barrier around the timed region and the MAX over ranks of the elapsed time.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOGN = 15
LOGQ = [55] + [45] * 11
LOGP = [55] * 3
T = 65537
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def gen_moduli():
    """NTT-friendly primes for the workload.  The product path needs only the primes; they come
    from the oracle's restated GenModuli (core/rlwe/params.go:811) when it is available and are
    pinned here so the bench does not depend on it."""
    # = GenModuli(LogNthRoot=16, LogQ=[55,45x11], LogP=[55x3])
    q = [36028797019488257, 35184372744193, 35184373006337, 35184373989377, 35184368877569, 35184368025601,
         35184367828993, 35184376545281, 35184377331713, 35184366911489, 35184378511361, 35184378707969]
    p = [36028797020209153, 36028797017456641, 36028797020602369]
    return q, p


def uniform(rng, moduli, N, lead=()):
    out = np.empty(tuple(lead) + (len(moduli), N), dtype=np.uint64)
    for i, m in enumerate(moduli):
        out[..., i, :] = rng.integers(0, int(m), size=tuple(lead) + (N,), dtype=np.uint64)
    return out


def cpu_baseline(q, p, kq, kp, seconds=12.0):
    """The restated reference (oracle/, a scalar C port) timed on the host cores: one MulRelin per
    call, independent ciphertexts on one thread per core (the reference's own parallel mode is
    b.RunParallel over independent outputs, schemes/ckks/ckks_benchmarks_test.go:116)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O
    N = 1 << LOGN
    cores = os.cpu_count() or 1
    ringQ, ringP = O.Ring(N, q), O.Ring(N, p)
    ev = O.Evaluator(ringQ, ringP)
    rlk = O.EvaluationKey(kq, kp)
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 2))
    ct0, ct1 = uniform(rng, q, N, (2,)), uniform(rng, q, N, (2,))
    t0 = time.perf_counter()
    ev.BGVMulRelin(T, ct0, ct1, rlk, True)
    one = time.perf_counter() - t0
    # bounded sample: every thread repeats the op until a shared deadline (~`seconds` of wall time)
    deadline = time.perf_counter() + seconds
    counts = [0] * cores

    def work(i):
        while True:
            ev.BGVMulRelin(T, ct0, ct1, rlk, True)  # ctypes releases the GIL
            counts[i] += 1
            if time.perf_counter() >= deadline:
                return

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    n = sum(counts)
    return {"value": n / dt, "unit": "ctxt-mul+relin ops/s", "cores": cores, "kind": "port",
            "single_thread_ops_s": 1.0 / one,
            "sample": f"{n} BGV MulRelin (logN=15, 12+3 limbs) over {cores} threads in {dt:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="independent ciphertext pairs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--replicate-keys", choices=["none", "rccl", "host"], default="none",
                    help="N > 1: rank 0's relinearisation key is replicated to every rank before the timed region (RCCL broadcast "
                         "into the key's device storage, or gloo through host memory) instead of each rank drawing its own")
    ap.add_argument("--microbench", action="store_true", help="also print NTT/s and the modmul probe to stderr")
    args = ap.parse_args()

    from lattigo_amd.dist import ControlPlane
    cp = ControlPlane()  # gloo control plane only (barrier + MAX of the elapsed time); no data-path collective
    rank, local_rank, world = cp.rank, cp.local_rank, cp.world
    if world > 1:
        import torch
        torch.cuda.set_device(int(os.environ.get("HERING_FORCE_DEVICE", local_rank)))

    import lattigo_amd as la
    # HERING_FORCE_DEVICE: test hook to exercise the multi-rank path on a box with fewer GPUs than ranks
    dev = int(os.environ.get("HERING_FORCE_DEVICE", local_rank if world > 1 else 0))
    ctx = la.Context(dev)
    N, B = 1 << LOGN, args.batch
    q, p = gen_moduli()
    L, alpha = len(q), len(p)
    beta = (L + alpha - 1) // alpha
    ringQ, ringP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
    ev = la.Evaluator(ringQ, ringP)
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 2 + 1000 * rank))
    kq, kp = uniform(rng, q, N, (beta, 2)), uniform(rng, p, N, (beta, 2))
    rlk = ev.NewEvaluationKey(kq, kp)
    if world > 1 and args.replicate_keys != "none":
        rlk = cp.ReplicateEvaluationKey(ev, rlk if rank == 0 else None, src=0, transport=args.replicate_keys)
    a = [la.Poly(ringQ, L, B).upload(uniform(rng, q, N, (B,))) for _ in range(2)]
    b = [la.Poly(ringQ, L, B).upload(uniform(rng, q, N, (B,))) for _ in range(2)]
    out = [la.Poly(ringQ, L, B), la.Poly(ringQ, L, B)]

    def step():
        ev.BGVMulRelin(L - 1, T, a, b, rlk, out)

    def barrier():
        ctx.sync()
        if world > 1:
            import torch
            torch.cuda.synchronize()
        cp.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = cp.max_over_ranks(elapsed)

    if rank != 0:
        cp.close()
        return

    ops = world * B * args.steps
    value = ops / elapsed
    limb = N * 8
    alg_bytes_op = (6 * L + 2 * beta * (L + alpha)) * limb  # 48 MiB

    # ---- roofline leg: per-kernel HIP-event timing over an identical region --------------------
    ctx.prof_begin()
    for _ in range(args.steps):
        step()
    prof = ctx.prof_end()
    total_ms = sum(v[1] for v in prof.values())
    dom = max(prof.items(), key=lambda kv: kv[1][1])
    dom_name, (dom_launches, dom_ms) = dom
    # algorithmic bytes per step of each kernel family (what the kernel must read + write once for B
    # ciphertexts; twiddles / constants excluded, resident).  Limbs below 2^47 run the double-precision row
    # kernels ("*_f64"), the others the 64-bit integer ones.
    nonown = beta * (L + alpha) - L                # limbs written by the decomposition
    small_q = [m < (1 << 47) for m in q]
    small_p = [m < (1 << 47) for m in p]
    dec_small = dec_big = 0                        # forward row passes of DecomposeNTT (non-own limbs)
    for d in range(beta):
        for l in range(L):
            if not (d * alpha <= l < (d + 1) * alpha):
                dec_small, dec_big = dec_small + small_q[l], dec_big + (not small_q[l])
        dec_small, dec_big = dec_small + sum(small_p), dec_big + (alpha - sum(small_p))
    nsq, nsp = sum(small_q), sum(small_p)
    n_small = nsq + nsp                            # limbs of QP on the double-precision path
    n_big = L + alpha - n_small
    per_step_bytes = {
        # fused forward row NTT + key MAC on the small limbs: decomposed non-own limbs + own limbs of c2 in, the key
        # once, both accumulators out
        "ntt_mac_f64": ((dec_small + nsq) * B + 2 * beta * n_small + 2 * n_small * B) * limb,
        # ModDown NTTs of both components with the fused epilogue (in, acc, add, out)
        "ntt_rows_fwd_f64": 4 * 2 * nsq * limb * B,
        # decomposition NTTs of the large limbs (in + out) + their share of the ModDown NTTs
        "ntt_rows_fwd": (2 * dec_big + 4 * 2 * (L - nsq)) * limb * B,
        # INTT(c2) + INTT of the P part of both accumulators
        "ntt_rows_inv_f64": 2 * (nsq + 2 * nsp) * limb * B,
        "ntt_rows_inv": 2 * ((L - nsq) + 2 * (alpha - nsp)) * limb * B,
        # key MAC on the large limbs: beta digits in, the key once, both accumulators out
        "ks_inner": (beta * n_big * B + 2 * beta * n_big + 2 * n_big * B) * limb,
        "tensor": 7 * L * limb * B,
        # fused basis extension: decomposition (L in, beta*(L+alpha)-L out) + ModDown (2*alpha in, 2*L out)
        "modup": (L + nonown + 2 * alpha + 2 * L) * limb * B,
    }
    dom_bytes_launch = per_step_bytes.get(dom_name, 0) * args.steps / max(dom_launches, 1)
    dom_avg_ms = dom_ms / max(dom_launches, 1)
    achieved = dom_bytes_launch / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
    traffic = None  # HBM bytes per launch from the committed rocprofv3 PMC pass (FETCH_SIZE x2 + WRITE_SIZE)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if pmc.get("batch") == B and dom_name in pmc.get("kernels", {}):
            traffic = pmc["kernels"][dom_name]["hbm_bytes_per_launch"]
    except Exception:
        pass
    valu = None  # VALU utilisation of the dominant kernel from the committed SQ-counter pass (profiles/r01_sq_counters.json)
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "r01_sq_counters.json")))
        valu = sq["kernels"][dom_name]["valu_util"]
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "valu_util": valu,
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_ms": dom_avg_ms, "launches": dom_launches,
                "alg_bytes_per_launch": dom_bytes_launch,
                "whole_op": {"alg_bytes_per_op": alg_bytes_op, "achieved_GBs": alg_bytes_op * value / world / 1e9,
                             "frac": alg_bytes_op * value / world / 1e9 / HBM_PEAK_GBS},
                "kernel_ms_per_step": {k: v[1] / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
                "kernel_time_sum_ms_per_step": total_ms / args.steps}

    line = {
        "metric": "ciphertext-mul+relin ops/s", "value": value, "unit": "ctxt-mul+relin ops/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "BGV logN=15, 12 Q-limbs [55,45x11] + 3 P-limbs [55x3], T=65537: ct x ct MulRelin "
                               "(tensor + gadget product + ModDown), inputs resident in HBM",
                   "batch_per_gpu": B, "logN": LOGN, "L": L, "alpha": alpha, "beta": beta,
                   "parallelism": f"{world} independent replicas, ciphertext-sharded"},
        "hip_event_ms_per_step": ev_ms / args.steps,
        "roofline": roofline,
    }
    if args.microbench:
        x = la.Poly(ringQ, L, B).upload(uniform(rng, q, N, (B,)))
        for _ in range(3):
            ringQ.NTT(x, x)
        ctx.timer_start()
        for _ in range(20):
            ringQ.NTT(x, x)
        ms = ctx.timer_stop()
        line["ntt_limb_per_s"] = 20 * L * B / (ms * 1e-3)
        line["modmul_per_s"] = ctx.probe_modmul(256)
    if not args.no_cpu_baseline and world == 1:
        try:
            line["cpu_baseline"] = cpu_baseline(q, p, kq, kp)
        except Exception as e:  # the oracle is optional test infrastructure
            line["cpu_baseline"] = {"error": str(e)}
    print(json.dumps(line), flush=True)
    cp.close()


if __name__ == "__main__":
    main()

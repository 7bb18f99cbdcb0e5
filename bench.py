#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ring-arithmetic backend.

Default workload (BASELINE.json configs[2], the one its metric "ciphertext-mul+relin ops/s at logN=15" is quoted on):
BGV, logN=15, 12 Q-limbs (LogQ=[55,45x11]), 3 P-limbs (LogP=[55x3]), T=65537, ct x ct Mul + Relinearize
(schemes/bgv/evaluator.go:592 tensorStandard + GadgetProduct).  A "step" is one MulRelin over a batch of B independent
ciphertext pairs already resident in HBM.  Synthetic inputs: coefficients uniform in [0, q_i), PCG64 seed
0x1A77160 + 2.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--workload c3|c4|c5]

--workload c4: BASELINE configs[3], CKKS logN=16, 20+4 limbs, Rotate (automorphism + Galois key-switch);
--workload c5: BASELINE configs[4], the operation trace of one CKKS bootstrap at the N16QP1546H192H32 shape.
N > 1: launched by torch.distributed.run, one rank per GPU; independent ciphertexts are sharded across ranks (weak
scaling, no data-path collective -- SURVEY.md section 8e); ranks synchronise only for the barrier around the timed region
and the MAX over ranks of the elapsed time.  After the timed region the output of the LAST step is checked against the CPU
oracle on three batch entries per rank ("verified"; a mismatch makes the run fail).  Prints ONE JSON line on rank 0.
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

# GenModuli outputs (core/rlwe/params.go:811) for the configs, pinned (tests/test_gpu_headline.py checks c3 against the
# oracle's restated GenModuli)
C4_Q = [1152921504606584833, 35184372744193, 35184373006337, 35184368025601, 35184376545281, 35184377331713, 35184378511361,
        35184379035649, 35184365273089, 35184380870657, 35184363569153, 35184382967809, 35184383229953, 35184383754241,
        35184385196033, 35184358850561, 35184386899969, 35184388734977, 35184355704833, 35184353083393]
C4_P = [2305843009211596801, 2305843009210023937, 2305843009208713217, 2305843009202159617]


def gen_moduli():
    """NTT-friendly primes of the headline workload = GenModuli(LogNthRoot=16, LogQ=[55,45x11], LogP=[55x3])."""
    q = [36028797019488257, 35184372744193, 35184373006337, 35184373989377, 35184368877569, 35184368025601,
         35184367828993, 35184376545281, 35184377331713, 35184366911489, 35184378511361, 35184378707969]
    p = [36028797020209153, 36028797017456641, 36028797020602369]
    return q, p


def uniform(rng, moduli, N, lead=()):
    out = np.empty(tuple(lead) + (len(moduli), N), dtype=np.uint64)
    for i, m in enumerate(moduli):
        out[..., i, :] = rng.integers(0, int(m), size=tuple(lead) + (N,), dtype=np.uint64)
    return out


def physical_cores():
    try:
        seen, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline(q, p, kq, kp, seconds=12.0):
    """The restated reference (oracle/, a scalar C port) timed on the host cores: independent MulRelin calls on one shared
    evaluator from n OS threads -- a C-level pthread loop with pooled scratch, the shape of the reference's own parallel
    benchmarks (b.RunParallel, schemes/ckks/ckks_benchmarks_test.go:95-325; scratch from core/rlwe/pool.go).  Thread counts
    1 / 16 / 64 / all logical CPUs are reported so that the scaling is visible; `value` is the best of them."""
    from oracle import oracle as O
    N = 1 << LOGN
    logical = os.cpu_count() or 1
    ringQ, ringP = O.Ring(N, q), O.Ring(N, p)
    ev = O.Evaluator(ringQ, ringP)
    rlk = O.EvaluationKey(kq, kp)
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 2))
    ct0, ct1 = uniform(rng, q, N, (2,)), uniform(rng, q, N, (2,))
    counts = sorted({1, min(16, logical), min(64, logical), logical})
    per = seconds / len(counts)
    sweep, best = {}, (0.0, 1, "")
    for n in counts:
        done, dt = ev.BenchBGVMulRelin(T, ct0, ct1, rlk, n, per)
        rate = done / dt
        sweep[str(n)] = rate
        if rate > best[0]:
            best = (rate, n, f"{done} BGV MulRelin (logN=15, 12+3 limbs) on {n} threads in {dt:.1f}s")
    return {"value": best[0], "unit": "ctxt-mul+relin ops/s", "cores": best[1], "kind": "port", "sample": best[2],
            "threads_sweep_ops_s": sweep, "logical_cpus": logical, "physical_cores": physical_cores(),
            "single_thread_ops_s": sweep["1"]}


# -------------------------------------------------------------------------------------------------------------------
# workloads: each returns step(), the units one step processes, a verifier of the last step's output and report fields
# -------------------------------------------------------------------------------------------------------------------
def pick_entries(B):
    return sorted({0, B // 2, B - 1})


def setup_c3(la, ctx, rank, B, cp, args):
    N = 1 << LOGN
    q, p = gen_moduli()
    L, alpha = len(q), len(p)
    beta = (L + alpha - 1) // alpha
    ringQ, ringP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
    ev = la.Evaluator(ringQ, ringP)
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 2 + 1000 * rank))
    kq, kp = uniform(rng, q, N, (beta, 2)), uniform(rng, p, N, (beta, 2))
    rlk = ev.NewEvaluationKey(kq, kp)
    if cp.world > 1 and args.replicate_keys != "none":
        rlk = cp.ReplicateEvaluationKey(ev, rlk if rank == 0 else None, src=0, transport=args.replicate_keys)
        kw = rlk.download()
        kq, kp = kw[:, :, :L], kw[:, :, L:]
    keep = pick_entries(B)
    host_in = []
    a, b = [], []
    for dst in (a, a, b, b):
        h = uniform(rng, q, N, (B,))
        dst.append(la.Poly(ringQ, L, B).upload(h))
        host_in.append(h[keep].copy())
        del h
    out = [la.Poly(ringQ, L, B), la.Poly(ringQ, L, B)]

    def step():
        ev.BGVMulRelin(L - 1, T, a, b, rlk, out)

    def verify():
        """every limb of three batch entries of the timed configuration's output against the oracle (outside the timed region)"""
        from oracle import oracle as O
        oev = O.Evaluator(O.Ring(N, q), O.Ring(N, p))
        orlk = O.EvaluationKey(np.ascontiguousarray(kq), np.ascontiguousarray(kp))
        g0, g1 = out[0], out[1]
        for i, e in enumerate(keep):
            want = oev.BGVMulRelin(T, np.stack([host_in[0][i], host_in[1][i]]), np.stack([host_in[2][i], host_in[3][i]]), orlk, True)
            for limb in range(L):
                if not (np.array_equal(g0.download_limb(e, limb), want[0][limb]) and np.array_equal(g1.download_limb(e, limb), want[1][limb])):
                    return False, f"batch entry {e}, limb {limb} differs from the oracle"
        return True, f"entries {keep} x {L} limbs x 2 components equal oracle.BGVMulRelin"

    limb = N * 8
    nonown = beta * (L + alpha) - L
    small_q = [m < (1 << 47) for m in q]
    small_p = [m < (1 << 47) for m in p]
    dec_small = dec_big = 0
    for d in range(beta):
        for l in range(L):
            if not (d * alpha <= l < (d + 1) * alpha):
                dec_small, dec_big = dec_small + small_q[l], dec_big + (not small_q[l])
        dec_small, dec_big = dec_small + sum(small_p), dec_big + (alpha - sum(small_p))
    nsq, nsp = sum(small_q), sum(small_p)
    n_small = nsq + nsp
    n_big = L + alpha - n_small
    # algorithmic bytes per step of each kernel family (what the kernel must read + write once for B ciphertexts; twiddles /
    # constants excluded, resident); see DESIGN.md section 4 for the pipeline these follow
    kernel_bytes = {
        "ntt_mac_f64": ((dec_small + nsq) * B + 2 * beta * n_small + 2 * n_small * B) * limb,
        "ntt_rows_fwd_f64": 4 * 2 * nsq * limb * B,
        "ntt_rows_fwd": (2 * dec_big + 4 * 2 * (L - nsq)) * limb * B,
        "ntt_rows_inv_f64": 2 * (nsq + 2 * nsp) * limb * B,
        "ntt_rows_inv": 2 * ((L - nsq) + 2 * (alpha - nsp)) * limb * B,
        "ks_inner": (beta * n_big * B + 2 * beta * n_big + 2 * n_big * B) * limb,
        "tensor": 7 * L * limb * B,
        "modup": (L + nonown + 2 * alpha + 2 * L) * limb * B,
    }
    return {
        "metric": "ciphertext-mul+relin ops/s", "unit": "ctxt-mul+relin ops/s", "step": step, "units": B, "verify": verify,
        "kernel_bytes": kernel_bytes,
        # (6L + 2 beta (L+alpha)) limbs = 48 MiB with the key charged to every op; one key read
        # serves the B ciphertexts of a step, so the batch-amortised figure is 6L limbs + key / B
        "alg_bytes_per_op": (6 * L + 2 * beta * (L + alpha)) * limb,
        "alg_bytes_per_op_amortised": 6 * L * limb + 2 * beta * (L + alpha) * limb / B,
        "cpu": lambda: cpu_baseline(q, p, np.ascontiguousarray(kq), np.ascontiguousarray(kp)),
        "config": {"workload": "BGV logN=15, 12 Q-limbs [55,45x11] + 3 P-limbs [55x3], T=65537: ct x ct MulRelin "
                               "(tensor + gadget product + ModDown), inputs resident in HBM",
                   "batch_per_gpu": B, "logN": LOGN, "L": L, "alpha": alpha, "beta": beta},
    }


def setup_c4(la, ctx, rank, B, cp, args):
    logN, q, p = 16, C4_Q, C4_P
    N, L, alpha = 1 << logN, len(q), len(p)
    beta = (L + alpha - 1) // alpha
    ringQ, ringP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
    ev = la.Evaluator(ringQ, ringP)
    # the Galois key is the same on every rank (a replicated key: same seed; `--replicate-keys` sends rank 0's instead)
    krng = np.random.Generator(np.random.PCG64(0x1A77160 + 3))
    kq, kp = uniform(krng, q, N, (beta, 2)), uniform(krng, p, N, (beta, 2))
    gk = ev.NewEvaluationKey(kq, kp)
    if cp.world > 1 and args.replicate_keys != "none":
        gk = cp.ReplicateEvaluationKey(ev, gk if rank == 0 else None, src=0, transport=args.replicate_keys)
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 3 + 1000 * (rank + 1)))
    keep = pick_entries(B)
    ct, host_in = [], []
    for _ in range(2):
        h = uniform(rng, q, N, (B,))
        ct.append(la.Poly(ringQ, L, B).upload(h))
        host_in.append(h[keep].copy())
        del h
    out = [la.Poly(ringQ, L, B), la.Poly(ringQ, L, B)]
    gal = pow(5, 1, 2 * N)

    def step():
        ev.Automorphism(L - 1, ct, gal, gk, out)

    def verify():
        from oracle import oracle as O
        oev = O.Evaluator(O.Ring(N, q), O.Ring(N, p))
        ogk = O.EvaluationKey(kq, kp)
        for i, e in enumerate(keep[:2]):
            want = oev.Automorphism(np.stack([host_in[0][i], host_in[1][i]]), gal, ogk)
            for limb in range(L):
                if not (np.array_equal(out[0].download_limb(e, limb), want[0][limb]) and np.array_equal(out[1].download_limb(e, limb), want[1][limb])):
                    return False, f"batch entry {e}, limb {limb} differs from the oracle"
        return True, f"entries {keep[:2]} x {L} limbs x 2 components equal oracle.Automorphism"

    limb = N * 8
    return {
        "metric": "ciphertext rotate ops/s", "unit": "ctxt-rotate ops/s", "step": step, "units": B, "verify": verify,
        "kernel_bytes": {},
        "alg_bytes_per_op": (4 * L + 2 * beta * (L + alpha)) * limb,  # 160 MiB
        "alg_bytes_per_op_amortised": 4 * L * limb + 2 * beta * (L + alpha) * limb / B,
        "cpu": None,
        "config": {"workload": "CKKS logN=16, 20 Q-limbs [60,45x19] + 4 P-limbs [61x4]: Rotate (automorphism + Galois "
                               "key-switch), inputs resident in HBM, Galois key replicated on every GPU",
                   "batch_per_gpu": B, "logN": logN, "L": L, "alpha": alpha, "beta": beta},
    }


def setup_c5(la, ctx, rank, B, cp, args):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bootstrap_c5_shape as C5
    run, info = C5.build(ctx, B, seed_offset=1000 * rank)
    return {
        "metric": "bootstraps/s", "unit": "ctxt-bootstraps/s", "step": run, "units": B, "verify": None, "kernel_bytes": {},
        "alg_bytes_per_op": None, "alg_bytes_per_op_amortised": None, "cpu": None,
        "config": dict({"workload": "CKKS bootstrap operation trace at the N16QP1546H192H32 shape (logN=16, 25+5 limbs; "
                                    "ModUp, CoeffsToSlots, EvalMod x2, SlotsToCoeffs), synthetic keys and DFT diagonals, "
                                    "batch split b mod G over the GPUs", "batch_per_gpu": B}, **info),
    }


WORKLOADS = {"c3": (setup_c3, 128), "c4": (setup_c4, 16), "c5": (setup_c5, 4)}


def ntt_rates(la, ctx):
    """BASELINE.json's "and NTT/s": stand-alone Ring.NTT (forward, in place) on the config-3 and config-4 Q chains."""
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 9))
    out = {}
    for name, logN, mods, B in (("logN15_L12", 15, gen_moduli()[0], 64), ("logN16_L20", 16, C4_Q, 32)):
        N = 1 << logN
        r = la.Ring(ctx, N, mods)
        x = la.Poly(r, len(mods), B).upload(uniform(rng, mods, N, (B,)))
        for _ in range(3):
            r.NTT(x, x)
        ctx.timer_start()
        for _ in range(20):
            r.NTT(x, x)
        ms = ctx.timer_stop() / 20
        out[name] = {"limb_ntt_per_s": len(mods) * B / (ms * 1e-3), "batch": B, "ms": ms,
                     "alg_GBs": 2 * len(mods) * B * N * 8 / (ms * 1e-3) / 1e9,
                     "frac_of_hbm_peak": 2 * len(mods) * B * N * 8 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del x, r
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--batch", type=int, default=0, help="independent ciphertexts per GPU per step (default: per workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed configuration's output")
    ap.add_argument("--no-ntt", action="store_true", help="skip the stand-alone NTT/s measurement")
    ap.add_argument("--replicate-keys", choices=["none", "rccl", "host"], default="none",
                    help="N > 1: rank 0's evaluation key is replicated to every rank before the timed region (RCCL broadcast "
                         "into the key's device storage, or gloo through host memory) instead of each rank drawing its own")
    ap.add_argument("--microbench", action="store_true", help="also report the modular-multiply probe")
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
    setup, default_B = WORKLOADS[args.workload]
    B = args.batch or default_B
    W = setup(la, ctx, rank, B, cp, args)
    step = W["step"]

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

    # ---- parity of what was timed: the output of the last timed step against the oracle, on every rank ------------
    verified, vmsg = None, "skipped"
    if W["verify"] is not None and not args.no_verify:
        ok, vmsg = W["verify"]()
        bad = cp.sum_over_ranks(0.0 if ok else 1.0)
        verified = bad == 0.0

    if rank != 0:
        cp.close()
        if verified is False:
            sys.exit(3)
        return

    ops = world * W["units"] * args.steps
    value = ops / elapsed

    # ---- roofline leg: per-kernel HIP-event timing over an identical region --------------------------------------
    ctx.prof_begin()
    for _ in range(args.steps):
        step()
    prof = ctx.prof_end()
    total_ms = sum(v[1] for v in prof.values())
    dom_name, (dom_launches, dom_ms) = max(prof.items(), key=lambda kv: kv[1][1])
    kb = W["kernel_bytes"]
    dom_bytes_launch = kb.get(dom_name, 0) * args.steps / max(dom_launches, 1)
    dom_avg_ms = dom_ms / max(dom_launches, 1)
    achieved = dom_bytes_launch / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 and dom_bytes_launch else None
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass of this command (FETCH_SIZE x 2 +
    # WRITE_SIZE, the gfx950 correction of the microarch guide; tools/round_artifacts.sh): NOT measured by this run
    traffic, traffic_src = None, None
    for name in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if pmc.get("batch") == B and pmc.get("workload", "c3") == args.workload and dom_name in pmc.get("kernels", {}):
                traffic, traffic_src = pmc["kernels"][dom_name]["hbm_bytes_per_launch"], f"profiles/{name} (committed rocprofv3 --pmc pass, not this run)"
                break
        except Exception:
            pass
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": dom_avg_ms, "launches": dom_launches, "alg_bytes_per_launch": dom_bytes_launch or None,
                "kernel_ms_per_step": {k: v[1] / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
                "kernel_GBs": {k: kb[k] / (v[1] / args.steps * 1e-3) / 1e9 for k, v in prof.items() if k in kb and v[1] > 0},
                "kernel_time_sum_ms_per_step": total_ms / args.steps}
    if W["alg_bytes_per_op"]:
        per_gpu = value / world
        roofline["whole_op"] = {
            "alg_bytes_per_op": W["alg_bytes_per_op"], "achieved_GBs": W["alg_bytes_per_op"] * per_gpu / 1e9,
            "frac": W["alg_bytes_per_op"] * per_gpu / 1e9 / HBM_PEAK_GBS,
            "alg_bytes_per_op_batch_amortised": W["alg_bytes_per_op_amortised"],
            "achieved_GBs_batch_amortised": W["alg_bytes_per_op_amortised"] * per_gpu / 1e9,
            "frac_batch_amortised": W["alg_bytes_per_op_amortised"] * per_gpu / 1e9 / HBM_PEAK_GBS}

    cfg = dict(W["config"])
    cfg["parallelism"] = f"{world} independent replicas, ciphertext-sharded"
    line = {
        "metric": W["metric"], "value": value, "unit": W["unit"], "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": cfg, "verified": verified, "verified_detail": vmsg,
        "hip_event_ms_per_step": ev_ms / args.steps,
        "roofline": roofline,
    }
    if not args.no_ntt:
        line["ntt"] = ntt_rates(la, ctx)
        line["ntt_limb_per_s"] = line["ntt"]["logN15_L12"]["limb_ntt_per_s"]
    if args.microbench:
        line["modmul_per_s"] = ctx.probe_modmul(256)
    if not args.no_cpu_baseline and world == 1 and W["cpu"] is not None:
        try:
            line["cpu_baseline"] = W["cpu"]()
        except Exception as e:  # the oracle is optional test infrastructure
            line["cpu_baseline"] = {"error": str(e)}
    print(json.dumps(line), flush=True)
    cp.close()
    if verified is False:
        sys.exit(3)


if __name__ == "__main__":
    main()

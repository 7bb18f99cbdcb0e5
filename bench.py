#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ring-arithmetic backend.

Default workload (BASELINE.json configs[2], the one its metric "ciphertext-mul+relin ops/s at logN=15" is quoted on):
BGV, logN=15, 12 Q-limbs (LogQ=[55,45x11]), 3 P-limbs (LogP=[55x3]), T=65537, ct x ct Mul + Relinearize
(schemes/bgv/evaluator.go:592 tensorStandard + GadgetProduct).  A "step" is one MulRelin over a batch of B (default 256) independent
ciphertext pairs already resident in HBM.  Synthetic inputs: coefficients uniform in [0, q_i), PCG64 seed
0x1A77160 + 2.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--workload c2|c3|c4|c5]

--workload c2: BASELINE configs[1], CKKS logN=14, 8-level chain: Evaluator.Mul (degree 2) + Rescale;
--workload c4: BASELINE configs[3], CKKS logN=16, 20+4 limbs, Rotate (automorphism + Galois key-switch);
--workload c5: BASELINE configs[4], the operation trace of one CKKS bootstrap at the N16QP1546H192H32 shape.
N > 1: launched by torch.distributed.run, one rank per GPU; independent ciphertexts are sharded across ranks (weak
scaling, no data-path collective -- SURVEY.md section 8e); ranks synchronise only for the barrier around the timed region
and the MAX over ranks of the elapsed time.  After the timed region the output of the LAST step is checked against the CPU
oracle on EVERY batch entry of every rank ("verified"; a mismatch makes the run fail).  Prints ONE JSON line on rank 0.

Byte accounting (DESIGN.md sections 4 and 6): per-kernel algorithmic bytes come from the launchers themselves (he_prof_end_bytes:
every polynomial stream a launch reads or writes, once), per-op algorithmic bytes from the per-primitive formulas of SURVEY.md
section 8(d) accumulated at the C ABI over the timed operation trace (he_alg_bytes) -- for c3 both are checked against the
closed forms below, so the model cannot drift from the pipeline again.
"""
from __future__ import annotations

import argparse
import json
import os
import re
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
C2_Q = [1125899908022273, 1099511922689, 1099512938497, 1099510054913, 1099514314753, 1099514478593, 1099508121601,
        1099507695617]                      # CKKS logN=14, LogQ=[50,40x7] (schemes/ckks/ckks_benchmarks_test.go:24-32)
C2_P = [1152921504606748673]                # LogP=[60]
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


def graph_replay(la, ctx, step, n):
    """The same step recorded once as a hipGraph (he_graph_begin / he_graph_end) and replayed n times: one enqueue per step
    instead of one per kernel -- what a latency-bound single-ciphertext chain costs without the launch queue."""
    try:
        with ctx.capture() as g:
            step()
        g.launch()
        ctx.sync()
        t = time.perf_counter()
        for _ in range(n):
            g.launch()
        ctx.sync()
        dt = (time.perf_counter() - t) / n
        out = {"latency_ms": dt * 1e3, "ops_per_s": 1.0 / dt, "nodes": g.nodes(),
               "note": "the step captured once (he_graph_*) and replayed back to back"}
        g.close()
        return out
    except la.HeringError as e:
        return {"error": str(e)}


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


def cpu_quota_cores():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited.  The GPU boxes of
    this pool expose 256 logical CPUs but grant 16: beyond that many busy threads the kernel throttles the whole group, which is
    what made the thread sweeps of rounds 1-2 peak at 16 threads and FALL beyond."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def cpu_baseline(kind, N, q, p, kq, kp, unit, what, seconds=12.0, t=0, gal=0):
    """The restated reference (oracle/, a scalar C port -- NOT Lattigo's Go code, which cannot be built here) timed on the host
    cores: independent calls on one shared evaluator from n OS threads -- a C-level pthread loop with pooled scratch, the shape
    of the reference's own parallel benchmarks (b.RunParallel, schemes/ckks/ckks_benchmarks_test.go:95-325; scratch from
    core/rlwe/pool.go).  Every thread is pinned to one CPU and works on its own first-touched copy of the inputs and the key
    (NUMA placement).  Thread counts 1 / half the usable cores / the usable cores / twice that are reported so that the scaling is
    visible; "usable" = min(logical CPUs in the affinity set, the cgroup's CPU quota) -- the quota, not the socket, is what the
    container gets; `value` is the best of the sweep and `cores` the thread count that gave it."""
    from oracle import oracle as O
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    phys = physical_cores()
    quota = cpu_quota_cores()
    usable = max(1, min(logical, int(quota))) if quota else min(logical, phys or logical)
    ev = O.Evaluator(O.Ring(N, q), O.Ring(N, p))
    key = O.EvaluationKey(kq, kp) if kq is not None else None
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 2))
    ct0 = uniform(rng, q, N, (2,))
    ct1 = uniform(rng, q, N, (2,)) if kind != "rotate" else None
    counts = sorted({1, max(1, usable // 2), usable, min(logical, 2 * usable)})
    per = seconds / len(counts)
    sweep, best = {}, (0.0, 1, "")
    for n in counts:
        done, dt = ev.BenchOp(kind, ct0, ct1, key, t=t, gal=gal, nthreads=n, seconds=per)
        rate = done / dt
        sweep[str(n)] = rate
        if rate > best[0]:
            best = (rate, n, f"{done} {what} on {n} pinned threads in {dt:.1f}s")
    return {"value": best[0], "unit": unit, "cores": best[1], "kind": "port", "sample": best[2],
            "threads_sweep_ops_s": sweep, "logical_cpus": logical, "physical_cores": phys, "cgroup_cpu_quota_cores": quota,
            "usable_cores": usable,
            "single_thread_ops_s": sweep["1"],
            "note": "scalar C restatement of the reference (oracle/), one pinned thread per CPU with private copies of inputs and key; not the Go code"}


def key_digest(key) -> int:
    """48 bits of the SHA-256 of an evaluation key's device words (exact in a float64: compared across ranks over gloo)"""
    import hashlib
    return int(hashlib.sha256(key.download().tobytes()).hexdigest()[:12], 16)


def replicate_key(cp, ev, key, rank, args):
    """N > 1: rank 0's key on every rank (one-time setup, before the timed region).  A transport that fails on any rank is
    given up by ALL ranks together (the outcome is agreed over the gloo control plane): rccl -> host -> each rank keeps the
    key it drew itself, so that a broken transport costs the replication, not the measurement.  -> (key, replicated?)"""
    if cp.world == 1 or args.replicate_keys.startswith("none"):
        return key, False
    for transport in ([args.replicate_keys, "host"] if args.replicate_keys == "rccl" else [args.replicate_keys]):
        ok, out = 1.0, None
        try:
            out = cp.ReplicateEvaluationKey(ev, key if rank == 0 else None, src=0, transport=transport)
        except Exception as e:  # noqa: BLE001 -- any failure of the transport
            sys.stderr.write(f"[rank {rank}] key replication over {transport} failed: {e}\n")
            ok = 0.0
        if cp.sum_over_ranks(ok) == cp.world:
            args.replicate_keys = transport
            return out, True
    args.replicate_keys = "none (transports failed)"
    return key, False


# -------------------------------------------------------------------------------------------------------------------
# workloads: each returns step(), the units one step processes, a verifier of the last step's output and report fields
# -------------------------------------------------------------------------------------------------------------------
def valu_model_mulrelin(q, p, N):
    """Modular-multiply equivalents of ONE MulRelin in closed form (NTT (N/2) logN + N per limb, basis
    extension L_src x L_dst x N, key inner product 2 beta (L + alpha) N, tensor 6 L N, ModDown's last op 2 L N), split by the
    arithmetic class of the limb they run on: moduli below 2^47 are computed with the exact double-precision product
    (csrc/kernels.hip modmul_f64), the others with 64-bit Montgomery products.  The two classes have different measured
    ceilings (he_probe_modmul_f64 / he_probe_modmul), so the ALU-bound time of the operation is
    n_f64 / rate_f64 + n_int / rate_int -- the binding roofline of this path: its kernels are VALU-issue-bound well before HBM."""
    L, alpha = len(q), len(p)
    beta = (L + alpha - 1) // alpha
    logN = N.bit_length() - 1
    f64 = lambda m: m < (1 << 47)
    ntt = (N // 2) * logN + N
    cnt = {"f64": 0.0, "int": 0.0}
    parts = {}

    def add(name, modulus, n):
        k = "f64" if f64(modulus) else "int"
        cnt[k] += n
        parts.setdefault(name, {"f64": 0.0, "int": 0.0})[k] += n

    for m in q:                                    # inverse NTT of c2; tensor (6 products); last op of ModDown, both components
        add("ntt", m, ntt)
        add("tensor", m, 6 * N)
        add("moddown_last", m, 2 * N)
        add("ntt", m, 2 * ntt)                     # forward NTT of the two basis-extended P parts
    for m in p:
        add("ntt", m, 2 * ntt)                     # inverse NTT of the P parts of both accumulators
        add("basis_extension", m, 2 * N)           # y_i of ModDown's sources
    for m in q:
        add("basis_extension", m, 2 * alpha * N)   # ModUpPtoQ: alpha sources into every Q limb, both components
    for d in range(beta):
        own = range(d * alpha, min((d + 1) * alpha, L))
        for i in own:
            add("basis_extension", q[i], N)        # y_i of the digit's sources
        dsts = [m for i, m in enumerate(q) if i not in own] + list(p)
        for m in dsts:
            add("basis_extension", m, len(own) * N)
            add("ntt", m, ntt)                     # forward NTT of the decomposed digit
    for m in list(q) + list(p):
        add("key_mac", m, 2 * beta * N)
    return cnt, parts


def verify_batch(kind, N, q, p, kq, kp, host_in, outs, out_limbs, t=0, gal=0, chunk=64):
    """EVERY batch entry and limb of the timed configuration's last output against the oracle (outside the timed region): the
    oracle runs the entries on the host's cores (oracle.Evaluator.BatchOp), in chunks so that the host copies stay bounded.
    host_in: the uploaded inputs [B][L][N] (two polynomials for a rotation, four for a product); outs: the result Polys."""
    from oracle import oracle as O
    oev = O.Evaluator(O.Ring(N, q), O.Ring(N, p))
    key = O.EvaluationKey(np.ascontiguousarray(kq), np.ascontiguousarray(kp)) if kq is not None else None
    B = host_in[0].shape[0]
    got = [o.download() for o in outs]
    for b0 in range(0, B, chunk):
        b1 = min(B, b0 + chunk)
        op0 = np.stack([host_in[0][b0:b1], host_in[1][b0:b1]], axis=1)
        op1 = np.stack([host_in[2][b0:b1], host_in[3][b0:b1]], axis=1) if len(host_in) == 4 else None
        want = oev.BatchOp(kind, op0, op1, key, t=t, gal=gal)
        for k in range(len(outs)):
            if not np.array_equal(got[k][b0:b1, :out_limbs], want[:, k]):
                bad = np.argwhere(got[k][b0:b1, :out_limbs] != want[:, k])[0]
                return False, f"batch entry {b0 + int(bad[0])}, component {k}, limb {int(bad[1])} differs from the oracle"
    return True, f"{B}/{B} entries x {out_limbs} limbs x {len(outs)} components equal the oracle's {kind}"


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
    rlk, replicated = replicate_key(cp, ev, rlk, rank, args)
    if replicated:
        kw = rlk.download()
        kq, kp = kw[:, :, :L], kw[:, :, L:]
    host_in = []
    a, b = [], []
    for dst in (a, a, b, b):
        h = uniform(rng, q, N, (B,))
        dst.append(la.Poly(ringQ, L, B).upload(h))
        host_in.append(h)
    out = [la.Poly(ringQ, L, B), la.Poly(ringQ, L, B)]

    def step():
        ev.BGVMulRelin(L - 1, T, a, b, rlk, out)

    def verify():
        return verify_batch("bgv_mulrelin", N, q, p, kq, kp, host_in, out, L, t=T)

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
    # algorithmic bytes per step of each kernel family in closed form (what the kernel must read + write once for B ciphertexts;
    # twiddles / constants excluded, resident), for the pipeline of DESIGN.md section 4 WITH the tensor product formed in the ModDown
    # epilogue: the tensor kernel reads a1, b1 and writes c2; the forward rows of the epilogue read the extension, the
    # accumulator and the four inputs of the product (shared by the two components of an entry) and write the result.  main()
    # checks the library's own per-launch accounting against these.
    # ... and, when no special prime is of the double-precision class (the headline chain), WITH that epilogue inside the NTT + MAC
    # kernel (NttMacEpilogue): its launch then also reads the two extension rows and the four inputs and writes the final
    # outputs in place of the accumulators, and the double-precision forward-row launch does not exist.
    mac_epilogue = nsp == 0 and nsq > 0 and os.environ.get("HERING_NO_MAC_EPILOGUE", "0") in ("", "0")
    prod_in = nsq > 0 and os.environ.get("HERING_NO_PROD_PROLOGUE", "0") in ("", "0")
    kernel_bytes = {
        "ntt_mac_f64": ((dec_small + nsq) * B + 2 * beta * n_small + 2 * n_small * B + (6 * nsq * B if mac_epilogue else 0)) * limb,
        "ntt_rows_fwd_f64": (2 * (1 + 1 + 1) + 4) * nsq * limb * B,
        "ntt_rows_fwd": (2 * dec_big + (2 * (1 + 1 + 1) + 4) * (L - nsq)) * limb * B,
        # (the inverse rows of the double-precision limbs form c2 = T(a1, b1) themselves: two inputs in, c2 and the transform out;
        # the tensor kernel covers the integer-class limbs only)
        "ntt_rows_inv_f64": ((4 if prod_in else 2) * nsq + 2 * 2 * nsp) * limb * B,
        "ntt_rows_inv": 2 * ((L - nsq) + 2 * (alpha - nsp)) * limb * B,
        "ks_inner": (beta * n_big * B + 2 * beta * n_big + 2 * n_big * B) * limb,
        "tensor": 3 * (L - nsq if prod_in else L) * limb * B,
        "modup": (L + nonown + 2 * alpha + 2 * L) * limb * B,
    }
    if mac_epilogue:
        del kernel_bytes["ntt_rows_fwd_f64"]
    return {
        "metric": "ciphertext-mul+relin ops/s", "unit": "ctxt-mul+relin ops/s", "step": step, "units": B, "verify": verify,
        "kernel_bytes": kernel_bytes,
        # (6L + 2 beta (L+alpha)) limbs = 48 MiB with the key charged to every op; one key read
        # serves the B ciphertexts of a step, so the batch-amortised figure is 6L limbs + key / B
        "alg_bytes_per_op": (6 * L + 2 * beta * (L + alpha)) * limb,
        "alg_bytes_per_op_amortised": 6 * L * limb + 2 * beta * (L + alpha) * limb / B,
        "valu_model": valu_model_mulrelin(q, p, N),
        "key_digest": lambda: key_digest(rlk),
        "cpu": lambda: cpu_baseline("bgv_mulrelin", N, q, p, np.ascontiguousarray(kq), np.ascontiguousarray(kp),
                                    "ctxt-mul+relin ops/s", "BGV MulRelin (logN=15, 12+3 limbs)", t=T),
        "config": {"workload": "BGV logN=15, 12 Q-limbs [55,45x11] + 3 P-limbs [55x3], T=65537: ct x ct MulRelin "
                               "(tensor + gadget product + ModDown), inputs resident in HBM",
                   "batch_per_gpu": B, "logN": LOGN, "L": L, "alpha": alpha, "beta": beta},
    }


def setup_c2(la, ctx, rank, B, cp, args):
    """BASELINE configs[1]: CKKS logN=14, 8-level chain, Evaluator.Mul (degree-2 result, no relinearisation) + Rescale of the
    three polynomials (schemes/ckks/evaluator.go:764-872, :477-515; the reference's BenchmarkCKKS Mul / Rescale pair)."""
    logN, q, p = 14, C2_Q, C2_P
    N, L = 1 << logN, len(q)
    ringQ, ringP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
    ev = la.Evaluator(ringQ, ringP)
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 1 + 1000 * rank))
    host_in, a, b = [], [], []
    for dst in (a, a, b, b):
        h = uniform(rng, q, N, (B,))
        dst.append(la.Poly(ringQ, L, B).upload(h))
        host_in.append(h)
    o3 = [la.Poly(ringQ, L, B) for _ in range(3)]
    r3 = [la.Poly(ringQ, L - 1, B) for _ in range(3)]

    def step():
        ev.CKKSMulRelin(L - 1, a, b, None, o3)
        ev.Rescale(L - 1, 1, o3, r3)

    def verify():
        return verify_batch("ckks_mul_rescale", N, q, p, None, None, host_in, r3, L - 1, chunk=128)

    limb = N * 8
    return {
        "metric": "ciphertext mul+rescale ops/s", "unit": "ctxt-mul+rescale ops/s", "step": step, "units": B, "verify": verify,
        "kernel_bytes": None,
        "alg_bytes_per_op": (7 * L + 3 * (2 * L - 1)) * limb,  # Mul 7 L + Rescale 3 (2L - 1) = 12.625 MiB
        "alg_bytes_per_op_amortised": (7 * L + 3 * (2 * L - 1)) * limb,
        "cpu": lambda: cpu_baseline("ckks_mul_rescale", N, q, p, None, None, "ctxt-mul+rescale ops/s", "CKKS Mul + Rescale (logN=14, 8 limbs)"),
        "config": {"workload": "CKKS logN=14, 8 Q-limbs [50,40x7] (+ 1 P-limb, unused): ct x ct Mul (degree 2) + Rescale of the three "
                               "polynomials, inputs resident in HBM", "batch_per_gpu": B, "logN": logN, "L": L},
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
    gk, _ = replicate_key(cp, ev, gk, rank, args)
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 3 + 1000 * (rank + 1)))
    ct, host_in = [], []
    for _ in range(2):
        h = uniform(rng, q, N, (B,))
        ct.append(la.Poly(ringQ, L, B).upload(h))
        host_in.append(h)
    out = [la.Poly(ringQ, L, B), la.Poly(ringQ, L, B)]
    gal = pow(5, 1, 2 * N)

    def step():
        ev.Automorphism(L - 1, ct, gal, gk, out)

    def verify():
        return verify_batch("rotate", N, q, p, kq, kp, host_in, out, L, gal=gal, chunk=32)

    limb = N * 8
    return {
        "metric": "ciphertext rotate ops/s", "unit": "ctxt-rotate ops/s", "step": step, "units": B, "verify": verify,
        "kernel_bytes": None,
        "alg_bytes_per_op": (4 * L + 2 * beta * (L + alpha)) * limb,  # 160 MiB
        "alg_bytes_per_op_amortised": 4 * L * limb + 2 * beta * (L + alpha) * limb / B,
        "key_digest": lambda: key_digest(gk),
        "cpu": lambda: cpu_baseline("rotate", N, q, p, kq, kp, "ctxt-rotate ops/s", "CKKS Rotate (logN=16, 20+4 limbs)", gal=gal),
        "config": {"workload": "CKKS logN=16, 20 Q-limbs [60,45x19] + 4 P-limbs [61x4]: Rotate (automorphism + Galois "
                               "key-switch), inputs resident in HBM, Galois key replicated on every GPU",
                   "batch_per_gpu": B, "logN": logN, "L": L, "alpha": alpha, "beta": beta},
    }


def c5_cpu_baseline(la, ctx, shape, trace_bytes_per_bootstrap):
    """Bounded CPU sample for the bootstrap trace: the trace is a few thousand ring / key-switch calls, four fifths of its
    algorithmic bytes key switches; a whole bootstrap on the scalar oracle takes about a minute per core (tests/golden/
    gen_c5_digest.py).  The sample is its dominant primitive at its top level -- Rotate at logN = 16, 25 + 5 limbs -- timed like
    the c4 baseline, and converted to bootstraps/s through SURVEY 8(d) bytes: rate x (bytes of one Rotate, as the library
    accounts it) / (bytes of one bootstrap trace, same accounting)."""
    N, q, p = shape["N"], shape["q"], shape["p"]
    ev, rq = shape["ev"], shape["rq"]
    ct = [la.Poly(rq, len(q)), la.Poly(rq, len(q))]
    out = [la.Poly(rq, len(q)), la.Poly(rq, len(q))]
    ctx.alg_bytes(reset=True)
    ctx.alg_valu(reset=True)
    ev.Automorphism(len(q) - 1, ct, shape["gal"], shape["key"], out)
    ctx.sync()
    rot_bytes = ctx.alg_bytes(reset=True)[0]
    r = cpu_baseline("rotate", N, q, p, shape["kq"], shape["kp"], "ctxt-rotate ops/s", "CKKS Rotate (logN=16, 25+5 limbs)",
                     gal=shape["gal"])
    scale = rot_bytes / trace_bytes_per_bootstrap
    r["sample"] += (f"; one Rotate = {rot_bytes / 2**20:.0f} MiB of the {trace_bytes_per_bootstrap / 2**30:.1f} GiB one bootstrap trace "
                    "moves (SURVEY 8(d) accounting by the library): rates scaled by that ratio")
    r["rotate_ops_s"] = r["value"]
    r["value"] *= scale
    r["unit"] = "ctxt-bootstraps/s"
    r["single_thread_ops_s"] *= scale
    r["threads_sweep_ops_s"] = {k: v * scale for k, v in r["threads_sweep_ops_s"].items()}
    r["note"] += "; an extrapolation from the trace's dominant primitive, not a timed bootstrap"
    return r


def setup_c5(la, ctx, rank, B, cp, args):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bootstrap_c5_shape as C5
    # every batch entry (on every rank) carries the same synthetic input: one oracle-backed run of the trace costs a minute of CPU,
    # and its committed digest (tests/golden/c5_trace_digest.json, made by tests/golden/gen_c5_trace_digest.py) then checks every
    # entry of the device's output.  Timing does not depend on the words: the entries are distinct polynomials in HBM.
    run, info = C5.build(ctx, B, seed_offset=0, same_input=True)
    last = {}

    def step():
        last["res"] = run()
        return last["res"]

    def verify():
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_trace_digest.json")))
        got = C5.trace_digest(last["res"], device=True)
        if got["level"] != golden["level"]:
            return False, f"output level {got['level']} != {golden['level']}"
        bad = [b for b, d in enumerate(got["entries"]) if d != golden["entries"][0]]
        if bad:
            return False, f"batch entries {bad[:8]} differ from the oracle-backed trace (tests/golden/c5_trace_digest.json)"
        return True, f"{len(got['entries'])}/{len(got['entries'])} entries: SHA-256 of the refreshed ciphertext equals the oracle-backed trace's"

    step._shape = run._shape
    return {
        "cpu_c5": lambda trace_bytes: c5_cpu_baseline(la, ctx, run._shape, trace_bytes),
        "metric": "bootstraps/s", "unit": "ctxt-bootstraps/s", "step": step, "units": B, "verify": verify, "kernel_bytes": None,
        # no closed form: summed over the operation trace by the library (he_alg_bytes), see main()
        "alg_bytes_per_op": None, "alg_bytes_per_op_amortised": None, "cpu": None,
        "config": dict({"workload": "CKKS bootstrap operation trace at the N16QP1546H192H32 shape (logN=16, 25+5 limbs; "
                                    "ModUp, CoeffsToSlots, EvalMod x2, SlotsToCoeffs), synthetic keys and DFT diagonals, every "
                                    "batch entry the same synthetic ciphertext (checked against the oracle-backed run of the "
                                    "trace), batch split b mod G over the GPUs", "batch_per_gpu": B}, **info),
    }


def concurrent_b1(la, ctx, workload="c3", ks=(1, 4, 16, 64), ops_per_run=6000, max_batch=64, window_us=30):
    """What the reference's own interface delivers: its operator API takes ONE ciphertext per call (schemes/schemes.go:14-28) and
    scales by concurrent callers (b.RunParallel over evaluators sharing tables and keys, schemes/ckks/ckks_benchmarks_test.go:
    116-207).  K OS threads (pthreads inside the library, he_debug_concurrent_mul_relin), each repeating MulRelin on its own
    batch-1 ciphertexts at the headline shape:
      * separate_contexts: one context (= HIP stream), ring, evaluator and key copy per caller -- K independent streams of
        single-ciphertext launches sharing the GPU (what existed before round 4);
      * coalesced: all callers on ONE evaluator with its submission queue on (he_evaluator_set_coalescing): calls waiting at
        the same time become one batched launch over the callers' own polynomials;
      * coalesced_sync_each: the same with every caller waiting for its result (he_ctx_sync) before its next call.
    Every figure is the median of three runs of about ops_per_run calls (the callers share the host's CPU quota with each other:
    single runs of a few milliseconds scatter by 20 %); every caller's output is compared with the oracle after every run.
    workload "c3": BGV MulRelin at the headline shape; "c4": CKKS Rotate (Automorphism + Galois key switch) at logN = 16, 20 + 4 limbs."""
    from lattigo_amd.rlwe import ConcurrentCalls
    from oracle import oracle as O
    rotate = workload == "c4"
    N, (q, p) = (1 << 16, (C4_Q, C4_P)) if rotate else (1 << LOGN, gen_moduli())
    op, tt = ("rotate", 5) if rotate else ("bgv_mulrelin", T)
    L, alpha = len(q), len(p)
    beta = (L + alpha - 1) // alpha
    kmax = max(ks)
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 77))
    kq, kp = uniform(rng, q, N, (beta, 2)), uniform(rng, p, N, (beta, 2))
    host = [uniform(rng, q, N, (kmax,)) for _ in range(2 if rotate else 4)]  # a0, a1 [, b0, b1]: [K][L][N]
    want = O.Evaluator(O.Ring(N, q), O.Ring(N, p)).BatchOp(
        op, np.stack([host[0], host[1]], axis=1), None if rotate else np.stack([host[2], host[3]], axis=1), O.EvaluationKey(kq, kp),
        t=0 if rotate else T, gal=tt if rotate else 0)
    ConcurrentMulRelin = lambda callers, level, iters, t=0, sync_each=False: ConcurrentCalls(op, callers, level, iters, tt, sync_each)

    def callers_on(c, n):
        ringQ, ringP = la.Ring(c, N, q), la.Ring(c, N, p)
        ev = la.Evaluator(ringQ, ringP)
        rlk = ev.NewEvaluationKey(kq, kp)
        mk = lambda k: ([la.Poly(ringQ, L).upload(host[0][k]), la.Poly(ringQ, L).upload(host[1][k])],
                        None if rotate else [la.Poly(ringQ, L).upload(host[2][k]), la.Poly(ringQ, L).upload(host[3][k])],
                        [la.Poly(ringQ, L, zero=False), la.Poly(ringQ, L, zero=False)])
        return ev, rlk, [mk(k) for k in n]

    def check(callers, base=0):
        for i, cl in enumerate(callers):
            got = np.stack([o.download()[0] for o in cl[5]])
            if not np.array_equal(got, want[base + i]):
                return False
        return True

    iters_of = lambda K: max(60, min(600, ops_per_run // K))

    def rate(callers, K, sync_each=False):
        runs = []
        for _ in range(3):
            runs.append(K * iters_of(K) / ConcurrentMulRelin(callers[:K], L - 1, iters_of(K), t=T, sync_each=sync_each))
        return sorted(runs)[1]

    out = {"K": list(ks), "iters_per_caller": [iters_of(K) for K in ks], "runs": "median of 3", "max_batch": max_batch,
           "window_us": window_us, "unit": "ctxt-rotate ops/s" if rotate else "ctxt-mul+relin ops/s", "operation": op,
           "note": "K OS threads, one batch-1 operation per call (he_debug_concurrent_mul_relin); host wall clock from the common "
                   "start to the last caller's final sync"}
    ok = True
    # one shared evaluator, submission queue on
    ev, rlk, cs = callers_on(ctx, range(kmax))
    shared = [(ctx, ev, a, b, rlk, o) for a, b, o in cs]
    ev.SetCoalescing(max_batch, window_us)
    # deferred: he_ctx_set_deferred -- the calls return once filed, the context's dispatcher thread launches them
    for name, sync_each, depth in (("coalesced", False, 0), ("coalesced_sync_each", True, 0), ("deferred", False, 8), ("deferred_sync_each", True, 8)):
        ctx.SetDeferred(depth)
        rates = []
        for K in ks:
            ConcurrentMulRelin(shared[:K], L - 1, 3, t=T, sync_each=sync_each)
            for cl in shared[:K]:
                cl[5][0].Zero(); cl[5][1].Zero()
            s0 = ev.CoalescingStats()
            rates.append(rate(shared, K, sync_each))
            s1 = ev.CoalescingStats()
            out.setdefault(name + "_mean_batch", []).append((s1["calls"] - s0["calls"]) / max(1, s1["launches"] - s0["launches"]))
            ok = ok and check(shared[:K])
        out[name] = rates
    ev.SetCoalescing(0, 0)
    # the same handles with the queue off: K threads serialising on one context's stream
    rates = []
    for K in ks:
        ConcurrentMulRelin(shared[:K], L - 1, 3, t=T)
        rates.append(rate(shared, K))
        ok = ok and check(shared[:K])
    out["one_context_uncoalesced"] = rates
    del shared, cs, ev, rlk
    # one context per caller
    sep = []
    for k in range(kmax):
        c = la.Context(ctx.device_id)
        e, r, cl = callers_on(c, [k])
        sep.append((c, e, cl[0][0], cl[0][1], r, cl[0][2]))
    rates = []
    for K in ks:
        ConcurrentMulRelin(sep[:K], L - 1, 3, t=T)
        rates.append(rate(sep, K))
        ok = ok and check(sep[:K])
    out["separate_contexts"] = rates
    out["verified"] = bool(ok)
    return out


def concurrent_c5(la, ctx, ks=(4, 16, 32), window_us=100, max_batch=64, rounds=3, depth=8):
    """BASELINE config 5 in the shape of the reference's own benchmark: BenchmarkConcurrentBootstrap runs b.RunParallel over
    bootstrappers (circuits/ckks/bootstrapping/evaluator_benchmarks_test.go:14-42) -- K callers, ONE ciphertext each, every call of
    the circuit a single-ciphertext call on shared keys and matrices.  The drivers above the operator interface exist here only as
    Python restatements (tests/drivers), and K interpreter threads would serialise on the interpreter's lock (measured: 19
    bootstraps/s at K = 16, below one lone caller's 33).  So ONE run of the driver on a batch-1 ciphertext is RECORDED (every he_*
    call with its arguments: ModUp, CoeffsToSlots, EvalMod x2, SlotsToCoeffs -- about 1 900 calls of the public entry points,
    allocations included) and replayed by K pthreads inside the library (he_debug_replay, include/hering_debug.h): the same calls
    through the same entry points, each thread on its own ciphertext, temporaries and hoisting buffers, all threads on ONE evaluator
    whose context's submission queue is on -- what K goroutines over the cgo package would do.  Every caller's refreshed ciphertext
    must hash to the committed digest of the oracle-backed trace."""
    import gc
    import hashlib
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bootstrap_c5_shape as C5
    from fractions import Fraction
    from drivers import schemes as S
    from lattigo_amd import _lib
    L = _lib.load()
    kmax = max(ks)
    run, _ = C5.build(ctx, 1, seed_offset=0, same_input=True)
    boot, _, ct0, _, _ = run._parts
    run()  # warm-up with the queue off: index tables, plans, lazily built helpers -- shared objects exist before the recording
    ctx.sync()
    gc.collect()
    _lib.trace_begin()
    res = boot.Bootstrap(S.Ciphertext(ct0, 0, 1), Fraction(1 << 60))
    program = _lib.trace_end()
    ctx.sync()
    rq = run._shape["rq"]
    host = ct0[0].download(), ct0[1].download()
    ins = [[la.Poly(rq, 1, 1).upload(h) for h in host] for _ in range(kmax)]
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_trace_digest.json")))
    level = int(res.level)
    ok = level == golden["level"] and C5.trace_digest(res, device=True)["entries"][0] == golden["entries"][0]
    out = {"K": list(ks), "unit": "ctxt-bootstraps/s", "max_batch": max_batch, "window_us": window_us, "rounds_per_caller": rounds,
           "calls_per_bootstrap": int(sum(1 for _ in _program_calls(program))),
           "host": "K pthreads replaying the recorded call sequence of one driver run (he_debug_replay): public entry points only",
           "coalesced": [], "mean_batch": [], "served_one_by_one": [],
           "deferred": {"depth": depth, "coalesced": [], "mean_batch": [],
                        "note": "he_ctx_set_deferred: calls return once filed, the context's dispatcher thread launches them"}}

    def fetch(h):
        nl, b, n = C.c_int(), C.c_int(), C.c_int()
        _lib.check(L.he_poly_shape(int(h), C.byref(nl), C.byref(b), C.byref(n)))
        a = np.empty((b.value, nl.value, n.value), dtype=np.uint64)
        _lib.check(L.he_poly_download(int(h), a.ctypes.data_as(_lib.u64p), a.size))
        _lib.check(L.he_poly_free(int(h)))
        return a

    def go(K, coalesce, n_rounds, deferred=0):
        nonlocal ok
        ctx.SetCoalescing(max_batch if coalesce else 0, window_us)
        if coalesce:
            ctx.SetDeferred(deferred)
        s0 = ctx.CoalescingStats()
        wall, outs = _lib.replay(ctx.h, program, K, n_rounds, [ct0[0].h, ct0[1].h], [[p.h for p in ins[k]] for k in range(K)],
                                 [v.h for v in res.Value])
        s1 = ctx.CoalescingStats()
        ctx.SetCoalescing(0, 0)
        if os.environ.get("HERING_REPLAY_PROFILE"):
            dbg = (C.c_uint64 * 16)()
            L.he_debug_queue_counters(ctx.h, dbg)
            print("queue counters (cumulative):", [int(x) for x in dbg], file=sys.stderr)
            ops = (C.c_uint64 * 64)()
            L.he_debug_queue_op_stats(ctx.h, ops)
            cur = [int(x) for x in ops]
            prev = getattr(go, "_ops", [0] * 64)
            go._ops = cur
            print("queue batches by operation (op: launches, mean batch):",
                  {i: (cur[2 * i] - prev[2 * i], round((cur[2 * i + 1] - prev[2 * i + 1]) / (cur[2 * i] - prev[2 * i]), 1))
                   for i in range(32) if cur[2 * i] > prev[2 * i]}, file=sys.stderr)
            prof = (C.c_uint64 * 192)()
            L.he_debug_replay_profile(prof, 64, 1)
            names = {v[0]: k for k, v in _lib._TRACE_FNS.items()}
            rows = sorted(((prof[3 * f], prof[3 * f + 1], prof[3 * f + 2], names.get(f, str(f))) for f in range(64) if prof[3 * f + 1]), reverse=True)
            print(f"replay profile K={K} coalesce={coalesce} wall={wall:.3f}s:", [(n, int(us), int(c), int(mx)) for us, c, mx, n in rows[:12]], file=sys.stderr)
        for k in range(K):
            words = [fetch(h)[:, : level + 1] for h in outs[k]]
            d = hashlib.sha256(np.ascontiguousarray(np.stack([w[0] for w in words])).tobytes()).hexdigest()
            ok = ok and d == golden["entries"][0]
        return K * n_rounds / wall, (s1["calls"] - s0["calls"]) / max(1, s1["launches"] - s0["launches"]), s1["one_by_one"] - s0["one_by_one"]

    go(min(4, kmax), True, 1)  # arena, pools, the queue's history
    if os.environ.get("HERING_C5_ONLY"):  # diagnosis (tools/c5_gap_analysis.py): one mode, the timed run last in the process
        dd = depth if os.environ["HERING_C5_ONLY"] == "deferred" else 0
        go(kmax, True, 1, dd)
        r, mb, _ = go(kmax, True, rounds, dd)
        return {"only": os.environ["HERING_C5_ONLY"], "K": kmax, "rate": r, "mean_batch": mb, "wall_s": kmax * rounds / r, "verified": bool(ok)}
    for K in ks:
        go(K, True, 1)  # the buffer pool grows to K callers' temporaries (hipMalloc synchronises the device)
        r, mb, obo = go(K, True, rounds)
        out["coalesced"].append(r); out["mean_batch"].append(mb); out["served_one_by_one"].append(obo)
        if depth > 0:
            go(K, True, 1, depth)
            r, mb, obo = go(K, True, rounds, depth)
            out["deferred"]["coalesced"].append(r); out["deferred"]["mean_batch"].append(mb)
    out["uncoalesced_K%d" % ks[0]] = go(ks[0], False, rounds)[0]
    out["lone_caller"] = go(1, False, rounds)[0]
    if depth > 0:
        out["deferred"]["lone_caller"] = go(1, True, rounds, depth)[0]
    out["verified"] = bool(ok)
    return out


def _program_calls(program):
    """the calls of a recorded program (lattigo_amd/_lib.py trace_end; encoding: csrc/replay.cpp)"""
    i, n = 0, len(program)
    while i < n:
        fn, na = int(program[i]), int(program[i + 1])
        i += 2
        for _ in range(na):
            kind = int(program[i]); i += 1
            if kind in (3, 4):
                i += 1 + int(program[i])
            elif kind != 5:
                i += 1
        yield fn


def gpu_clock(device_id=0):
    """Current shader / memory clock of the GPU (MHz) and its power draw (W) from the driver's sysfs tables of THE device this rank
    runs on (found through its PCI bus id: the node's other GPUs are listed in sysfs too): pp_dpm_sclk / pp_dpm_mclk, the line
    marked '*' is the level in use.  Read before, during and after the timed region so that box-to-box and power-management
    effects (quiet data clocks ~15 % higher) can be told from kernel changes.  None where the files are absent."""
    import glob
    import lattigo_amd as la
    if device_id not in _CLOCK_DIR:
        d = None
        bus = la.device_pci_bus_id(device_id)
        if bus and os.path.exists(f"/sys/bus/pci/devices/{bus}/pp_dpm_sclk"):
            d = f"/sys/bus/pci/devices/{bus}"
        _CLOCK_DIR[device_id] = (d, bus)
    d, bus = _CLOCK_DIR[device_id]
    if d is None:
        return None
    out = {}
    for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
        try:
            levels = [ln for ln in open(os.path.join(d, name)).read().splitlines() if ln.strip()]
            cur = [ln for ln in levels if ln.rstrip().endswith("*")] or levels[-1:]
            out[key] = float(re.search(r"(\d+(?:\.\d+)?)\s*[Mm][Hh]z", cur[0]).group(1))
        except (OSError, AttributeError, IndexError, ValueError):
            out[key] = None
    for key, pat, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6), ("temp_c", "temp1_input", 1e-3)):
        try:
            if key not in out:
                out[key] = float(open(glob.glob(os.path.join(d, "hwmon", "hwmon*", pat))[0]).read()) * scale
        except (OSError, IndexError, ValueError):
            pass
    out["pci_bus_id"] = bus
    return out


_CLOCK_DIR = {}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (torch.distributed.run, one
    rank per GPU, rendezvous on 127.0.0.1) and hand over -- so that --gpus means N however the script is invoked.  A node with
    fewer than N GPUs is refused (exit 2) unless HERING_FORCE_DEVICE (test hook: ranks share one GPU) is set."""
    import socket
    import lattigo_amd as la
    have = la.device_count()
    if have < args.gpus and "HERING_FORCE_DEVICE" not in os.environ:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} asked, this node has {have} GPU(s); no line printed\n")
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write(f"bench.py: --gpus {args.gpus} without a launcher: starting {args.gpus} ranks ({' '.join(cmd[1:8])} ...)\n")
    sys.stderr.flush()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


# the other BASELINE configurations in the default line (`other_configs`): (workload, steps, warmup); c5's step is a whole
# bootstrap trace of 16 ciphertexts (~0.18 s), so three timed steps keep the default run within minutes
OTHER_CONFIGS = (("c2", 10, 2), ("c4", 10, 2), ("c5", 3, 1))


def other_configs(args):
    """BASELINE configs 2, 4 and 5 measured by the SAME script in child processes (one after the other, after the headline's own
    context is gone): every entry of their timed batch verified against the oracle / the committed digest, their dominant kernel
    timed by HIP events in that run.  Returns {workload: summary}; a child that fails is reported as {"error": ...} and makes the
    run exit non-zero."""
    import subprocess
    out = {}
    for wl, steps, warmup in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", wl, "--steps", str(steps), "--warmup", str(warmup),
               "--no-b1", "--no-concurrent", "--no-ntt", "--no-cpu-baseline", "--no-other-configs"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.other_timeout)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not lines:
                out[wl] = {"error": f"exit {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
                continue
            j = json.loads(lines[-1])
        except Exception as e:  # noqa: BLE001 -- timeout, unparsable line
            out[wl] = {"error": str(e)[-400:]}
            continue
        rf = j.get("roofline") or {}
        valu = rf.get("valu") or {}
        out[wl] = {
            "metric": j["metric"], "value": j["value"], "unit": j["unit"], "steps": j["steps"], "warmup": j["warmup"],
            "ms_per_step": j["ms_per_step"], "batch": j["config"].get("batch_per_gpu"), "workload": j["config"]["workload"],
            "verified": j.get("verified"), "verified_detail": j.get("verified_detail"), "exit_code": r.returncode,
            "roofline": {"kernel": rf.get("kernel"), "bound": rf.get("bound"), "achieved": rf.get("achieved"), "peak": rf.get("peak"),
                         "unit": rf.get("unit"), "frac": rf.get("frac"), "avg_launch_ms": rf.get("avg_launch_ms"),
                         "launches": rf.get("launches"), "alg_bytes_per_launch": rf.get("alg_bytes_per_launch"),
                         "traffic": rf.get("traffic"), "traffic_source": rf.get("traffic_source"),
                         "whole_op_frac": (rf.get("whole_op") or {}).get("frac"),
                         "kernel_ms_per_step": dict(list((rf.get("kernel_ms_per_step") or {}).items())[:8])},
            "valu": {"frac": valu.get("frac"), "instr_frac": (valu.get("instr") or {}).get("frac"),
                     "modmul_equiv_per_op": valu.get("modmul_equiv_per_op"), "ideal_us_per_op": valu.get("ideal_us_per_op"),
                     "achieved_us_per_op": valu.get("achieved_us_per_op")},
            "gpu_clock": j.get("gpu_clock"), "accounting_problems": j.get("accounting_problems"),
            "wall_s": time.perf_counter() - t0,
        }
    return out


# default batches from sweeps on MI355X (round 3): c2 128 / 256 / 512 / 1024 / 2048: 204k / 223k / 243k / 257k / 263k (a logN = 14, 8-limb
# ciphertext is small: the launches need the larger batch to fill the chip); c3 64 / 128 / 192 / 256 / 512: 33.9k / 36.6k / 37.5k /
# 37.8k / 38.0k on one box (the persistent NTT+MAC kernel's tail shrinks with more items per workgroup; flat beyond 256);
# c4 16 / 32 / 64 / 128: 8.6k / 9.2k / 9.6k / 9.8k (round 6: 64 / 128: 11.8k / 12.0k); c5 8 / 16 / 32: 73 / 84 / 85 (round 6, one box:
# 16 / 24 / 32 / 48: 94.3 / 99.6 / 100.2 / 102.1 bootstraps/s -- a bootstrap's launches at its low levels are small: the default is 32 since
# round 6, rounds 1-5 reported batch 16)
WORKLOADS = {"c2": (setup_c2, 1024), "c3": (setup_c3, 256), "c4": (setup_c4, 128), "c5": (setup_c5, 32)}


def ntt_rates(la, ctx):
    """BASELINE.json's "and NTT/s": stand-alone Ring.NTT and Ring.INTT (in place; the reference benches both,
    ring/ntt_benchmark_test.go:10-25) on the config-3 and config-4 Q chains.  Batches from tools/ntt_batch_sweep.py: the rate
    saturates from 64 entries per call (logN = 15: 4.6 / 4.6 / 4.75 M limb-NTT/s at 64 / 128 / 256)."""
    rng = np.random.Generator(np.random.PCG64(0x1A77160 + 9))
    out = {}
    for name, logN, mods, B in (("logN15_L12", 15, gen_moduli()[0], 256), ("logN16_L20", 16, C4_Q, 64)):
        N = 1 << logN
        r = la.Ring(ctx, N, mods)
        x = la.Poly(r, len(mods), B).upload(uniform(rng, mods, N, (B,)))
        res = {"batch": B}
        for what, f in (("ntt", r.NTT), ("intt", r.INTT)):
            # steady state: the device needs tens of milliseconds of work after an idle spell before it runs at its sustained rate
            # (tools/ntt_drift_probe.py, profiles/r06_ntt_drift_probe.txt: 20 calls right after idle read 4.09 M limb-NTT/s at batch
            # 256 and 4.31 M at batch 64 -- the figures of rounds 4-5 and of round 3 -- where 200 calls, or 20 calls on a busy
            # device, read 4.6 M and 4.8 M); rounds 3-5 timed 20 calls after 3 warm-up calls, i.e. the ramp
            for _ in range(80):
                f(x, x)
            ctx.timer_start()
            for _ in range(100):
                f(x, x)
            ms = ctx.timer_stop() / 100
            gbs = 2 * len(mods) * B * N * 8 / (ms * 1e-3) / 1e9
            res[what] = {"limb_ntt_per_s": len(mods) * B / (ms * 1e-3), "ms": ms, "alg_GBs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}
        res["protocol"] = "80 untimed + 100 timed in-place calls back to back (steady state; rounds 3-5: 3 + 20 calls after an idle spell)"
        res.update({"limb_ntt_per_s": res["ntt"]["limb_ntt_per_s"], "ms": res["ntt"]["ms"], "alg_GBs": res["ntt"]["alg_GBs"],
                    "frac_of_hbm_peak": res["ntt"]["frac_of_hbm_peak"], "limb_intt_per_s": res["intt"]["limb_ntt_per_s"]})
        out[name] = res
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
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the per-kernel HIP-event leg (and with it the roofline object): for runs under rocprofv3 --pmc, whose "
                         "counter collection does not survive the tens of thousands of events a bootstrap trace records")
    ap.add_argument("--no-b1", action="store_true", help="skip the single-ciphertext (batch 1) rate / latency measurement")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the concurrent single-ciphertext callers measurement (c3)")
    ap.add_argument("--co-window", type=int, default=30, help="concurrent_b1: gathering window of the submission queue, microseconds")
    ap.add_argument("--co-batch", type=int, default=64, help="concurrent_b1: largest coalesced batch")
    ap.add_argument("--only-concurrent", action="store_true", help="print only the concurrent_b1 object (window sweeps)")
    ap.add_argument("--replicate-keys", choices=["auto", "none", "rccl", "host"], default="auto",
                    help="N > 1: rank 0's evaluation key is replicated to every rank before the timed region (RCCL broadcast "
                         "into the key's device storage, or gloo through host memory) instead of each rank drawing its own; "
                         "auto = rccl when the node has a GPU per rank, none otherwise")
    ap.add_argument("--selftest", action="store_true",
                    help="N > 1: fail (exit 5) unless every rank runs on its own GPU, the RCCL communicator reaches all N ranks and every "
                         "rank's evaluation key equals rank 0's after the replication")
    ap.add_argument("--microbench", action="store_true", help="also report the modular-multiply probe")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default workload, one GPU: do not measure BASELINE configs 2, 4 and 5 into the line's `other_configs`")
    ap.add_argument("--other-timeout", type=int, default=420, help="seconds a child run of `other_configs` may take")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus > 1 and env_world == 0:
        self_launch(args)  # does not return
    if env_world not in (0, args.gpus):
        # a line whose n_gpus differs from what was asked is never printed
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; no line printed\n")
        sys.exit(2)

    from lattigo_amd.dist import ControlPlane
    cp = ControlPlane()  # gloo control plane only (barrier + MAX of the elapsed time); no data-path collective
    rank, local_rank, world = cp.rank, cp.local_rank, cp.world
    import lattigo_amd as la
    asked = args.replicate_keys
    if args.replicate_keys == "auto":
        # one GPU per rank: the RCCL broadcast (driven by libhering on the context's stream -- torch only carries the gloo control
        # plane, so neither an import order nor torch.cuda is involved); ranks sharing a GPU (test hook) cannot form a communicator
        args.replicate_keys = "none"
        if world > 1 and "HERING_FORCE_DEVICE" not in os.environ and la.device_count() >= world:
            args.replicate_keys = "rccl"

    # HERING_FORCE_DEVICE: test hook to exercise the multi-rank path on a box with fewer GPUs than ranks
    dev = int(os.environ.get("HERING_FORCE_DEVICE", local_rank if world > 1 else 0))
    ctx = la.Context(dev)
    if args.only_concurrent:
        print(json.dumps(concurrent_b1(la, ctx, args.workload, window_us=args.co_window, max_batch=args.co_batch)), flush=True)
        cp.close()
        return
    setup, default_B = WORKLOADS[args.workload]
    B = args.batch or default_B
    W = setup(la, ctx, rank, B, cp, args)
    step = W["step"]

    def barrier():
        ctx.sync()
        cp.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.alg_bytes(reset=True)
    ctx.alg_valu(reset=True)
    clock = {"before": gpu_clock(dev)}
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    clock["under_load"] = gpu_clock(dev)  # the steps are enqueued, the device is still running them (a sysfs read: microseconds)
    ev_ms = ctx.timer_stop()
    barrier()
    elapsed_rank = time.perf_counter() - t0
    clock["after"] = gpu_clock(dev)
    clock["source"] = "amdgpu sysfs of this rank's device: pp_dpm_sclk / pp_dpm_mclk (level in use), hwmon power / temperature"
    elapsed = cp.max_over_ranks(elapsed_rank)
    elapsed_min = -cp.max_over_ranks(-elapsed_rank)
    alg_trace = ctx.alg_bytes(reset=True)  # SURVEY 8(d) per-primitive bytes of the timed steps (key per entry, key per call)
    valu_trace = ctx.alg_valu(reset=True)  # SURVEY 8(d) multiply counts of the timed steps, by arithmetic class
    # what the control plane and (when keys were replicated over RCCL) the RCCL communicator saw
    ranks_seen = {"control_plane_gloo": int(cp.sum_over_ranks(1.0)), "rccl": cp.rccl_world() if args.replicate_keys == "rccl" else None}
    # a replication that was asked for (or chosen by `auto` because the node has a GPU per rank) and did not happen is a failure of
    # the run, not a footnote: the multi-GPU record must not silently be N independent key sets
    multi_gpu_problems = []
    if world > 1 and asked != "none" and args.replicate_keys.startswith("none") and (asked != "auto" or la.device_count() >= world):
        multi_gpu_problems.append(f"key replication failed: asked for {asked!r}, ended with {args.replicate_keys!r}")
    if world > 1 and args.replicate_keys == "rccl" and ranks_seen["rccl"] != world:
        multi_gpu_problems.append(f"the RCCL communicator reaches {ranks_seen['rccl']} of {world} ranks")
    key_digest = W.get("key_digest")
    if world > 1 and key_digest is not None and not args.replicate_keys.startswith("none"):
        d = float(key_digest())
        if cp.max_over_ranks(d) != d or -cp.max_over_ranks(-d) != d:
            multi_gpu_problems.append("evaluation keys differ between the ranks after the replication")
    if args.selftest and world > 1:
        devs = cp.sum_over_ranks(float(1 << (ctx.device_id % 48)))  # one bit per device in use: all distinct iff the sum has `world` bits
        if bin(int(devs)).count("1") != world:
            multi_gpu_problems.append("ranks share a GPU")
        if args.replicate_keys != "rccl":
            multi_gpu_problems.append(f"selftest wants the RCCL leg, the run used {args.replicate_keys!r}")
    n_bad = cp.sum_over_ranks(1.0 if multi_gpu_problems else 0.0)

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
        if n_bad:
            sys.stderr.write(f"[rank {rank}] multi-GPU problems: {multi_gpu_problems}\n")
            sys.exit(5)
        return

    ops = world * W["units"] * args.steps
    value = ops / elapsed

    if args.no_kernel_timing:
        print(json.dumps({"metric": W["metric"], "value": value, "unit": W["unit"], "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "verified": verified, "roofline": None,
                          "gpu_clock": clock,
                          "config": dict(W["config"])}), flush=True)
        cp.close()
        return
    # ---- roofline leg: per-kernel HIP-event timing over an identical region --------------------------------------
    ctx.prof_begin()
    for _ in range(args.steps):
        step()
    prof = ctx.prof_end_bytes()  # {kernel: (launches, ms, algorithmic bytes of those launches -- counted by the launchers)}
    total_ms = sum(v[1] for v in prof.values())
    dom_name, (dom_launches, dom_ms, dom_bytes) = max(prof.items(), key=lambda kv: kv[1][1])
    kb = {k: v[2] / args.steps for k, v in prof.items()}  # per step
    kb_check = None
    if W["kernel_bytes"]:  # closed forms of this workload's pipeline: the library's accounting must agree
        bad = {k: (kb.get(k), v) for k, v in W["kernel_bytes"].items() if abs(kb.get(k, 0.0) - v) > 0.005 * v}
        kb_check = "library launch accounting == closed forms of DESIGN.md section 4" if not bad else f"MISMATCH {bad}"
    dom_bytes_launch = dom_bytes / max(dom_launches, 1)
    dom_avg_ms = dom_ms / max(dom_launches, 1)
    achieved = dom_bytes_launch / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 and dom_bytes_launch else None
    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass of this command (FETCH_SIZE x 2 +
    # WRITE_SIZE, the gfx950 correction of the microarch guide; tools/round_artifacts.sh): NOT measured by this run
    traffic, traffic_src = None, None
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if "pmc_traffic" in f and f.endswith(".json")), reverse=True):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if pmc.get("batch") == B and pmc.get("workload", "c3") == args.workload and dom_name in pmc.get("kernels", {}):
                traffic, traffic_src = pmc["kernels"][dom_name]["hbm_bytes_per_launch"], f"profiles/{name} (committed rocprofv3 --pmc pass, not this run)"
                break
        except Exception:
            pass
    kernel_GBs = {k: kb[k] / (v[1] / args.steps * 1e-3) / 1e9 for k, v in prof.items() if kb.get(k) and v[1] > 0}
    # (a launch whose whole working set fits the 256 MB Infinity Cache can legitimately exceed the HBM rate -- the bootstrap trace
    # has such launches at its lowest levels: only launches of more than 1 GiB are held against the peak)
    over_peak = {k: g for k, g in kernel_GBs.items() if g > HBM_PEAK_GBS and prof[k][2] / max(prof[k][0], 1) > (1 << 30)}
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": dom_avg_ms, "launches": dom_launches, "alg_bytes_per_launch": dom_bytes_launch or None,
                "kernel_ms_per_step": {k: v[1] / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
                "kernel_launches_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
                "kernel_alg_bytes_per_step": {k: kb[k] for k in sorted(kb, key=lambda k: -prof[k][1])},
                "kernel_GBs": kernel_GBs, "kernel_bytes_check": kb_check,
                "kernel_bytes_source": "he_prof_end_bytes: every polynomial stream a launch reads or writes, once (csrc/kernels.hip)",
                "kernel_time_sum_ms_per_step": total_ms / args.steps,
                "kernel_time_note": "per-launch HIP events inflate kernel times by ~3 %: the sum may exceed ms_per_step"}
    # per-op algorithmic bytes: SURVEY 8(d)'s per-primitive formulas summed over the timed operation trace by the library
    per_op_trace = alg_trace[0] / (W["units"] * args.steps)
    per_op_trace_am = alg_trace[1] / (W["units"] * args.steps)
    alg_op = W["alg_bytes_per_op"] or per_op_trace
    alg_op_am = W["alg_bytes_per_op_amortised"] or per_op_trace_am
    per_gpu = value / world
    roofline["whole_op"] = {
        "alg_bytes_per_op": alg_op, "achieved_GBs": alg_op * per_gpu / 1e9, "frac": alg_op * per_gpu / 1e9 / HBM_PEAK_GBS,
        "alg_bytes_per_op_batch_amortised": alg_op_am, "achieved_GBs_batch_amortised": alg_op_am * per_gpu / 1e9,
        "frac_batch_amortised": alg_op_am * per_gpu / 1e9 / HBM_PEAK_GBS,
        "alg_bytes_per_op_from_trace": per_op_trace,
        "source": ("closed form of SURVEY.md section 8(d); the library's per-primitive accounting of the timed trace gives "
                   f"{per_op_trace / 2**20:.2f} MiB") if W["alg_bytes_per_op"] else
                  "SURVEY.md section 8(d) per-primitive formulas summed over the timed operation trace (he_alg_bytes)"}
    valu_problem = None
    if world == 1:
        # The VALU roofline, for every workload: modular multiplies per operation -- counted by the library per primitive call of the
        # timed steps (he_alg_valu: the closed forms of SURVEY.md section 8(d), by the arithmetic class of each limb) -- against the
        # chip's two measured multiply ceilings (he_probe_modmul: the production 16-instruction Montgomery sequence;
        # he_probe_modmul_f64: the 6-instruction exact double-precision product).  `frac` prices multiplies only; `instr.frac` prices
        # the instructions an implementation cannot avoid: a butterfly is a product PLUS an add, a subtract and the range handling
        # (22 instructions on the integer path, 16 of them the product; 10 double-precision operations, 6 of them the product -- the
        # kernels' own sequences, csrc/kernels.hip bfly_fwd / rows_round16_f64), issued at the rate the probes measure for the
        # product's instructions.  What is left between instr.frac and 1 is stall (dependent chains, LDS exchanges, waits).
        ops_timed = W["units"] * args.steps
        mul_int, mul_f64, bf_int, bf_f64 = (x / ops_timed for x in valu_trace)
        rate_int, rate_f64 = ctx.probe_modmul(256), ctx.probe_modmul_f64(256)
        ideal_s = mul_int / rate_int + mul_f64 / rate_f64
        instr_int = bf_int * 22 + (mul_int - bf_int) * 16
        instr_f64 = bf_f64 * 10 + (mul_f64 - bf_f64) * 6
        ideal_instr_s = instr_int / (16 * rate_int) + instr_f64 / (6 * rate_f64)
        sq, sq_src = None, None
        for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if "sq_counters" in f and f.endswith(".json")), reverse=True):
            try:
                sqj = json.load(open(os.path.join(ROOT, "profiles", name)))
                if sqj.get("workload", "c3") != args.workload:
                    continue
                sq = {k: {kk: v[kk] for kk in ("valu_issue_share_of_wave_time", "valu_insts_per_wave", "wait_any_share",
                                               "wait_inst_any_share", "lds_bank_conflict_share") if kk in v}
                      for k, v in (sqj.get("launch_groups") or sqj.get("kernels") or {}).items()}
                sq_src = f"profiles/{name} (committed rocprofv3 --pmc SQ pass of this command, not this run)"
                break
            except Exception:
                pass
        roofline["valu"] = {
            "bound": "valu (integer / double-precision multiply issue)",
            "modmul_equiv_per_op": mul_int + mul_f64, "modmul_equiv_int": mul_int, "modmul_equiv_f64": mul_f64,
            "butterflies_int": bf_int, "butterflies_f64": bf_f64,
            "count_source": "he_alg_valu: closed forms per primitive call of the timed steps (csrc/api.cpp struct Valu)",
            "peak_int_modmul_per_s": rate_int, "peak_f64_modmul_per_s": rate_f64,
            "peak_source": "he_probe_modmul / he_probe_modmul_f64 run in this process: dependent MRedLazy / modmul_f64 chains, 4 per thread, 4 M threads",
            "ideal_us_per_op": ideal_s * 1e6, "achieved_us_per_op": 1e6 / per_gpu, "frac": ideal_s * per_gpu,
            "instr": {"int_instr_per_op": instr_int, "f64_instr_per_op": instr_f64,
                      "model": "butterfly = 22 integer instructions (16 the Montgomery product) / 10 double-precision operations (6 the "
                               "exact product); other products 16 / 6; issue rate = 16 x / 6 x the measured product rates",
                      "ideal_us_per_op": ideal_instr_s * 1e6, "frac": ideal_instr_s * per_gpu},
            "per_kernel_sq": sq, "per_kernel_sq_source": sq_src,
            "note": "frac = ALU-bound time / measured time, multiplies only; instr.frac counts a butterfly's unavoidable companions too"}
        if W.get("valu_model"):
            cnt, parts = W["valu_model"]
            roofline["valu"]["breakdown_per_op"] = parts
            roofline["valu"]["closed_form_check"] = "library's per-primitive counts == valu_model_mulrelin (bench.py)"
            if abs(cnt["int"] - mul_int) > 1e-6 * cnt["int"] or abs(cnt["f64"] - mul_f64) > 1e-6 * max(cnt["f64"], 1.0):
                valu_problem = f"multiply counts: library {mul_int, mul_f64} != closed form {cnt['int'], cnt['f64']}"
                roofline["valu"]["closed_form_check"] = "MISMATCH: " + valu_problem
        # which roofline binds: the larger of the two ideal times per operation
        hbm_ideal_s = alg_op_am / (HBM_PEAK_GBS * 1e9)
        roofline["bound"] = "valu" if ideal_instr_s > hbm_ideal_s else "hbm"
        roofline["bound_detail"] = {"hbm_ideal_us_per_op": hbm_ideal_s * 1e6, "valu_ideal_us_per_op": ideal_instr_s * 1e6,
                                    "note": "`bound` = the larger ideal time per operation (algorithmic bytes at 8 TB/s vs unavoidable "
                                            "instructions at the measured issue rates); achieved / peak / frac above stay the dominant "
                                            "kernel's HBM figures, as the bench contract defines them"}
    problems = []
    if valu_problem:
        problems.append(valu_problem)
    if over_peak:
        problems.append(f"kernel_GBs above the HBM peak (stale byte model?): {over_peak}")
    if kb_check and kb_check.startswith("MISMATCH"):
        problems.append(kb_check)
    if W["alg_bytes_per_op"] and abs(per_op_trace - W["alg_bytes_per_op"]) > 0.005 * W["alg_bytes_per_op"]:
        problems.append(f"per-op trace accounting {per_op_trace} != closed form {W['alg_bytes_per_op']}")

    cfg = dict(W["config"])
    cfg["parallelism"] = f"{world} independent replicas, ciphertext-sharded"
    line = {
        "metric": W["metric"], "value": value, "unit": W["unit"], "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": cfg, "verified": verified, "verified_detail": vmsg,
        "hip_event_ms_per_step": ev_ms / args.steps,
        "rank_ms_per_step": {"min": elapsed_min / args.steps * 1e3, "max": elapsed / args.steps * 1e3},
        "gpu_clock": clock,
        "ranks_seen": ranks_seen, "replicate_keys": args.replicate_keys,
        "multi_gpu_selftest": (None if world == 1 else ("ok" if not n_bad else f"FAILED on {int(n_bad)} rank(s): {multi_gpu_problems}")),
        "roofline": roofline,
    }
    if not args.no_b1 and world == 1 and args.workload != "c5" and B != 1:
        # single-ciphertext figures ("report best and B=1"): the same operation on ONE ciphertext
        W1 = setup(la, ctx, rank, 1, cp, args)
        for _ in range(5):
            W1["step"]()
        ctx.sync()
        n1 = 200
        t1 = time.perf_counter()
        for _ in range(n1):
            W1["step"]()
        ctx.sync()
        dt1 = (time.perf_counter() - t1) / n1
        line["b1"] = {"batch": 1, "ops_per_s": 1.0 / dt1, "latency_ms": dt1 * 1e3,
                      "note": "one ciphertext per call, back-to-back calls on one stream, host wall clock incl. launch overhead"}
        line["b1"]["graph"] = graph_replay(la, ctx, W1["step"], n1)
        del W1
    elif args.workload == "c5" and world == 1 and not args.no_b1 and B != 1:
        W1 = setup(la, ctx, rank, 1, cp, args)
        W1["step"]()
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(3):
            W1["step"]()
        ctx.sync()
        dt1 = (time.perf_counter() - t1) / 3
        line["b1"] = {"batch": 1, "ops_per_s": 1.0 / dt1, "latency_ms": dt1 * 1e3, "note": "one ciphertext per bootstrap"}
        line["b1"]["graph"] = graph_replay(la, ctx, W1["step"], 5)
        del W1
    if not args.no_concurrent and world == 1 and args.workload in ("c3", "c4"):
        try:
            line["concurrent_b1"] = concurrent_b1(la, ctx, args.workload, window_us=args.co_window, max_batch=args.co_batch)
            if line["concurrent_b1"]["verified"] is False:
                problems.append("concurrent_b1: a caller's output differs from the oracle")
        except la.HeringError as e:
            line["concurrent_b1"] = {"error": str(e)}
            problems.append(f"concurrent_b1 failed: {e}")
        # the same shape from a COMPILED host through the public interface only (include/hering.hpp; tests/cpp/run_parallel.cpp:
        # std::thread per caller, its own process and context): what a Go caller of the cgo package would see
        exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "cpp", "run_parallel")
        if args.workload == "c3" and os.path.exists(exe) and "error" not in line["concurrent_b1"]:
            import subprocess
            runs = []
            for sync_each, deferred in ((0, 0), (1, 0), (0, 8), (1, 8)):  # deferred: he_ctx_set_deferred, the calls return once filed
                try:
                    r = subprocess.run([exe, "64", "96", str(sync_each), "1", "c3", str(args.co_batch), str(args.co_window), str(deferred)],
                                       capture_output=True, text=True, timeout=120)
                    runs.append(json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-300:]})
                except Exception as e:  # noqa: BLE001 -- a missing / stale binary must not cost the bench line
                    runs.append({"error": str(e)})
            line["concurrent_b1"]["compiled_host"] = runs
            if any(x.get("verified") is False for x in runs):
                problems.append("concurrent_b1.compiled_host: a caller's output differs from the oracle")
    if not args.no_concurrent and world == 1 and args.workload == "c2":
        # config 2 through the one-ciphertext interface (round 5: the queue serves Mul without a key and Rescale as well): a COMPILED
        # host, std::thread per ciphertext, four interface calls per operation, every caller's result checked against the oracle
        exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "cpp", "run_parallel")
        if os.path.exists(exe):
            import subprocess
            runs = []
            # (K, max_batch in entries: a Rescale files one request per polynomial, so 4 K lets every caller's three share a launch,
            #  wait for each result, queue on, deferred depth in requests per caller, calls per caller)
            for K, mb, sync_each, co, deferred, iters in ((16, 64, 0, 1, 0, 1000), (32, 128, 0, 1, 0, 500), (64, 256, 0, 1, 0, 400), (64, 256, 1, 1, 0, 200),
                                                          (16, 64, 0, 1, 128, 2000), (32, 128, 0, 1, 128, 2000), (64, 256, 0, 1, 128, 1000),
                                                          (64, 256, 1, 1, 128, 200), (64, 256, 0, 0, 0, 100)):
                try:
                    r = subprocess.run([exe, str(K), str(iters), str(sync_each), str(co), "c2", str(mb), str(max(args.co_window, 100)),
                                        str(deferred)], capture_output=True, text=True, timeout=180)
                    runs.append(json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-300:]})
                except Exception as e:  # noqa: BLE001
                    runs.append({"error": str(e)})
            line["concurrent_b1"] = {"unit": "ctxt-mul+rescale ops/s", "compiled_host": runs,
                                     "note": "tests/cpp/run_parallel.cpp c2: K threads, one ciphertext per call (Mul, then Rescale: "
                                             "he_rescale_polys, the three polynomials in one call) on one evaluator; deferred_depth > 0: "
                                             "he_ctx_set_deferred (the calls return once filed, the context's dispatcher launches); the last run: the "
                                             "same callers with the queue off.  More callers than the host's CPU quota (16 here) lose "
                                             "to the scheduler, not to the GPU"}
            if any(x.get("verified") is False for x in runs):
                problems.append("concurrent_b1.compiled_host: a caller's output differs from the oracle")
    if not args.no_concurrent and world == 1 and args.workload == "c5":
        try:
            line["concurrent_b1"] = concurrent_c5(la, ctx, window_us=max(args.co_window, 100))
            if line["concurrent_b1"].get("verified") is False:
                problems.append("concurrent_b1: a caller's refreshed ciphertext differs from the committed digest")
        except la.HeringError as e:
            line["concurrent_b1"] = {"error": str(e)}
            problems.append(f"concurrent_b1 failed: {e}")
    if not args.no_ntt:
        line["ntt"] = ntt_rates(la, ctx)
        line["ntt_limb_per_s"] = line["ntt"]["logN15_L12"]["limb_ntt_per_s"]
    if args.microbench:
        line["modmul_per_s"] = ctx.probe_modmul(256)
    if not args.no_cpu_baseline and world == 1 and W.get("cpu_c5") is not None:
        try:
            line["cpu_baseline"] = W["cpu_c5"](per_op_trace)
        except Exception as e:
            line["cpu_baseline"] = {"error": str(e)}
    if not args.no_cpu_baseline and world == 1 and W["cpu"] is not None:
        try:
            line["cpu_baseline"] = W["cpu"]()
        except Exception as e:  # the oracle is optional test infrastructure
            line["cpu_baseline"] = {"error": str(e)}
    if world == 1 and args.workload == "c3" and not args.batch and not args.no_other_configs:
        # the other BASELINE configurations, driver-visible: child runs of this script (their own context; this one's polynomials
        # and keys are released first)
        del W, step
        import gc
        gc.collect()
        line["other_configs"] = other_configs(args)
        if line.get("ntt"):
            line["other_configs"]["ntt"] = {k: {kk: v[kk] for kk in ("batch", "limb_ntt_per_s", "limb_intt_per_s", "ms", "alg_GBs", "frac_of_hbm_peak")}
                                            for k, v in line["ntt"].items()}
        for wl, r in line["other_configs"].items():
            if wl == "ntt":
                continue
            if "error" in r:
                problems.append(f"other_configs.{wl}: {r['error'][-200:]}")
            elif r.get("verified") is not True or r.get("exit_code"):
                problems.append(f"other_configs.{wl}: verified={r.get('verified')} exit={r.get('exit_code')} {r.get('accounting_problems')}")
    if problems:
        line["accounting_problems"] = problems
    print(json.dumps(line), flush=True)
    cp.close()
    if verified is False:
        sys.exit(3)
    if n_bad:
        sys.exit(5)
    if problems:
        sys.exit(4)


if __name__ == "__main__":
    main()

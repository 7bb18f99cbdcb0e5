"""Multi-GPU readiness (`-m gpu`; SURVEY.md section 8e: one process per GPU, ciphertexts sharded, keys replicated).

What one GPU can check runs always: every bench.py workload under torch.distributed.run with two ranks sharing the box's GPU
(HERING_FORCE_DEVICE), each rank's output verified against the oracle.  What needs two or more GPUs enables itself when the
box has them (the driver's 8-GPU node): the multi-rank RCCL key broadcast over xGMI, and contexts on different devices driven
alternately from one thread and concurrently from several (every he_* call selects its context's device)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _device_count() -> int:
    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                         timeout=300)
    return int(out.stdout.strip() or 0)


@pytest.mark.parametrize("workload,batch", [("c3", 4), ("c4", 2), ("c5", 1)])
def test_bench_two_ranks_every_workload(workload, batch):
    """bench.py's multi-rank entry point for BASELINE configs 3, 4 and 5: two ranks (here on one GPU), ciphertexts sharded,
    keys replicated from rank 0 through the host transport, every rank's last step verified against the oracle."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HERING_FORCE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload",
           workload, "--batch", str(batch), "--no-cpu-baseline", "--no-ntt"] + (["--replicate-keys", "host"] if workload != "c5" else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["batch_per_gpu"] == batch
    assert line["verified"] is True, line["verified_detail"]  # (c5: every entry against the oracle-backed trace's committed digest)


def test_bench_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` invoked plainly (no torch.distributed.run around it, the way the driver calls `--gpus 1`):
    the script starts its two ranks itself and the line says n_gpus 2 with both ranks on the control plane (VERDICT r5 weak #7:
    --gpus used to be parsed and ignored)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HERING_FORCE_DEVICE="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-ntt", "--replicate-keys", "host"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"]["control_plane_gloo"] == 2
    assert line["verified"] is True and line["config"]["batch_per_gpu"] == 4


def test_bench_refuses_more_gpus_than_the_node_has():
    """without the test hook, asking for more GPUs than the node has prints no line (exit 2) instead of a mislabelled one"""
    n = _device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HERING_FORCE_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 2 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], (out.stdout[-500:], out.stderr[-500:])


_RCCL_WORKER = """
import hashlib, os, sys
sys.path.insert(0, %r)
import numpy as np
import lattigo_amd as la
from lattigo_amd.dist import ControlPlane
from oracle import oracle as O
from tests.helpers import rng_for, uniform_poly
cp = ControlPlane()
dev = cp.local_rank
ctx = la.Context(dev)                     # one rank per GPU; the RCCL leg runs inside libhering on this context's stream
q, p = O.GenModuli(13, [55, 45, 45, 45], [55, 46])
N = 1 << 12
gQ, gP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
gev = la.Evaluator(gQ, gP)
rng = rng_for(2970)                       # same stream on every rank: the same cx; only rank 0 draws a key
cx = uniform_poly(rng, q, N)
key = None
if cp.rank == 0:
    kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(2)])
    kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(2)])
    key = gev.NewEvaluationKey(kq, kp)
key = cp.ReplicateEvaluationKey(gev, key, src=0, transport="rccl")   # GPU-to-GPU broadcast into the key's device storage
ct = [la.Poly(gQ, 4), la.Poly(gQ, 4)]
gev.GadgetProduct(3, la.Poly(gQ, 4).upload(cx), key, ct)
got = np.stack([c.get() for c in ct])
if cp.rank == 0:
    want = O.Evaluator(O.Ring(N, q), O.Ring(N, p)).GadgetProduct(3, cx, O.EvaluationKey(kq, kp))
    assert np.array_equal(got, want)
digest = int(hashlib.sha256(got.tobytes()).hexdigest()[:12], 16)
assert cp.max_over_ranks(digest) == digest == -cp.max_over_ranks(-digest)   # every rank computed rank 0's (oracle-checked) words
if cp.rank == 0:
    print("RCCL_REPLICATED", cp.world)
cp.close()
"""


def test_multi_rank_rccl_key_replication(tmp_path):
    """The one RCCL collective of the design (key distribution) with one rank per GPU."""
    n = min(_device_count(), 8)
    if n < 2:
        pytest.skip("needs at least two GPUs (runs on the driver's multi-GPU node)")
    script = tmp_path / "worker.py"
    script.write_text(_RCCL_WORKER % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", "29563", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stderr[-3000:]
    assert f"RCCL_REPLICATED {n}" in out.stdout


def test_contexts_on_different_devices_select_their_device():
    """Handles may be used from any OS thread and every call selects its context's device (hering.h conventions): two
    contexts on two GPUs driven alternately from one thread, then concurrently from two threads."""
    import lattigo_amd as la
    from oracle import oracle as O
    from tests.helpers import rng_for, uniform_poly
    if _device_count() < 2:
        pytest.skip("needs at least two GPUs (runs on the driver's multi-GPU node)")
    N = 1 << 12
    q, p = O.GenModuli(13, [55, 45, 45], [55])
    oev = O.Evaluator(O.Ring(N, q), O.Ring(N, p))
    rng = rng_for(2980)
    kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(3)])
    kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(3)])
    cxs = [uniform_poly(rng, q, N) for _ in range(2)]
    wants = [oev.GadgetProduct(2, cx, O.EvaluationKey(kq, kp)) for cx in cxs]
    envs = []
    for dev in (0, 1):
        ctx = la.Context(dev)
        gQ, gP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
        gev = la.Evaluator(gQ, gP)
        envs.append((ctx, gQ, gev, gev.NewEvaluationKey(kq, kp)))

    def run(dev, i):
        ctx, gQ, gev, key = envs[dev]
        ct = [la.Poly(gQ, 3), la.Poly(gQ, 3)]
        gev.GadgetProduct(2, la.Poly(gQ, 3).upload(cxs[i]), key, ct)
        assert np.array_equal(np.stack([c.get() for c in ct]), wants[i]), (dev, i)

    for rep in range(3):  # interleaved on one thread: no call may inherit the other context's device
        run(0, 0), run(1, 1), run(1, 0), run(0, 1)
    errs = []

    def worker(dev):
        try:
            for rep in range(10):
                run(dev, rep & 1)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(d,)) for d in (0, 1)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


_SPLIT_WORKER = """
import os, sys
sys.path.insert(0, %r)
import numpy as np
import lattigo_amd as la
from lattigo_amd.dist import ControlPlane
from lattigo_amd import rlwe as R
from oracle import oracle as O
from tests.helpers import rng_for, uniform_poly
cp = ControlPlane()
ctx = la.Context(int(os.environ.get("HERING_FORCE_DEVICE", cp.local_rank)))
q, p = O.GenModuli(14, [55, 45, 45, 45, 45, 45, 45], [55, 46])      # beta = 4 digits of alpha = 2 limbs
N, L, B = 1 << 13, 7, 2
gQ, gP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
gev = la.Evaluator(gQ, gP)
rng = rng_for(2990)                       # the same stream on every rank: same key, same input
beta = 4
kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(beta)])
kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(beta)])
cx = np.stack([uniform_poly(rng, q, N) for _ in range(B)])
share = cp.digit_range(beta, cp.rank, cp.world)
kq_r, kp_r = kq.copy(), kp.copy()         # a rank only needs its digits: poison the others
for d in range(beta):
    if d not in share:
        kq_r[d] = 0x5A5A5A5A; kp_r[d] = 0x5A5A5A5A
key = gev.NewEvaluationKey(kq_r, kp_r)
pcx = la.Poly(gQ, L, B).upload(cx)
dec = R.Decomposition(gev, B)
gev.DecomposeNTT(L - 1, len(p) - 1, len(p), pcx, True, dec)
ct = [la.Poly(gQ, L, B), la.Poly(gQ, L, B)]
cp.SplitGadgetProductHoisted(gev, L - 1, dec, key, ct, transport=%r)
got = np.stack([c.get() for c in ct], axis=1)      # [B][2][L][N]
oev = O.Evaluator(O.Ring(N, q), O.Ring(N, p))
for b in range(B):
    want = oev.GadgetProduct(L - 1, cx[b], O.EvaluationKey(kq, kp))
    assert np.array_equal(got[b], want), (cp.rank, b)
if cp.rank == 0:
    print("SPLIT_OK", cp.world)
cp.close()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_one_key_switch_split_over_ranks_by_digit(tmp_path, world):
    """SURVEY.md section 8e's single-op split: every rank holds only its digits of the key (the others poisoned), accumulates
    them with he_gadget_product_hoisted_lazy_digits, the partial accumulators are all-reduced (here: gloo through the host,
    ranks sharing the box's GPU) and ModDown finishes: every rank's result equals the oracle's GadgetProduct bit for bit."""
    script = tmp_path / "worker.py"
    script.write_text(_SPLIT_WORKER % (ROOT, "host"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29565 + world), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1", HERING_FORCE_DEVICE="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    assert f"SPLIT_OK {world}" in out.stdout


def test_one_key_switch_split_over_gpus_rccl(tmp_path):
    """the same with one rank per GPU and the all-reduce over RCCL / xGMI, in place on the accumulators' device storage"""
    n = min(_device_count(), 4)
    if n < 2:
        pytest.skip("needs at least two GPUs (runs on the driver's multi-GPU node)")
    script = tmp_path / "worker.py"
    script.write_text((_SPLIT_WORKER % (ROOT, "rccl")).replace('os.environ.get("HERING_FORCE_DEVICE", cp.local_rank)', "cp.local_rank"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", "29569", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stderr[-3000:]
    assert f"SPLIT_OK {n}" in out.stdout


_RCCL_SOLO_WORKER = """
import ctypes as C, sys
sys.path.insert(0, %r)
import numpy as np
import lattigo_amd as la                  # no torch in this process: libhering loads RCCL itself (he_rccl_*)
from lattigo_amd import rlwe as R
from lattigo_amd._lib import H, check, load
from oracle import oracle as O
from tests.helpers import rng_for, uniform_poly
assert "torch" not in sys.modules
ctx = la.Context(0)
N = 1 << 11
q, p = O.GenModuli(12, [55, 45, 45, 45], [55, 46])
gQ, gP = la.Ring(ctx, N, q), la.Ring(ctx, N, p)
gev = la.Evaluator(gQ, gP)
rng = rng_for(2990)
kq = np.stack([np.stack([uniform_poly(rng, q, N) for _ in range(2)]) for _ in range(2)])
kp = np.stack([np.stack([uniform_poly(rng, p, N) for _ in range(2)]) for _ in range(2)])
key = gev.NewEvaluationKey(kq, kp)
ident = (C.c_uint8 * 128)()
check(load().he_rccl_unique_id(ident))
comm = H()
check(load().he_rccl_comm_create(ctx.h, ident, 0, 1, C.byref(comm)))
n = C.c_int()
check(load().he_rccl_comm_ranks(comm.value, C.byref(n)))
assert n.value == 1
check(load().he_evk_broadcast(comm.value, key.h, 0))          # a one-rank broadcast leaves the root's words as they are
cx = uniform_poly(rng, q, N)
ct = [la.Poly(gQ, 4), la.Poly(gQ, 4)]
gev.GadgetProduct(3, la.Poly(gQ, 4).upload(cx), key, ct)
want = O.Evaluator(O.Ring(N, q), O.Ring(N, p)).GadgetProduct(3, cx, O.EvaluationKey(kq, kp))
assert np.array_equal(np.stack([c.get() for c in ct]), want)
pcx = la.Poly(gQ, 4).upload(cx)
check(load().he_poly_all_reduce_sum(comm.value, pcx.h))       # the sum over one rank
assert np.array_equal(pcx.get(), cx)
assert load().he_evk_broadcast(comm.value, key.h, 3) != 0     # root outside the communicator: refused
check(load().he_rccl_comm_destroy(comm.value))
print("RCCL_SOLO_OK")
"""


def test_library_side_rccl_leg_one_rank(tmp_path):
    """The RCCL leg libhering drives itself (he_rccl_unique_id / he_rccl_comm_create / he_evk_broadcast / he_poly_all_reduce_sum /
    he_rccl_comm_ranks) on a one-rank communicator, in a process that never imports torch: librccl is loaded at run time next to
    the HIP runtime in use.  (The multi-rank leg needs a GPU per rank: test_multi_rank_rccl_key_replication.)"""
    script = tmp_path / "worker.py"
    script.write_text(_RCCL_SOLO_WORKER % ROOT)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "RCCL_SOLO_OK" in out.stdout


"""world_size-2 and world_size-8 gloo tests of the multi-GPU control plane (lattigo_amd/dist.py): sharding of
independent ciphertexts, barrier, MAX/SUM over ranks.  Runs on CPU; the per-rank compute is the
GPU path covered by the `-m gpu` tests."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %r)
    from lattigo_amd.dist import ControlPlane
    cp = ControlPlane()
    W = cp.world
    assert W in (2, 8)
    items = 7 if W == 2 else 29               # ciphertext b goes to rank b mod W (SURVEY.md section 8e)
    mine = list(cp.shard(items))
    assert mine == list(range(cp.rank, items, W))
    cp.barrier()
    t = cp.max_over_ranks(1.0 + cp.rank)      # the last rank is the slow one
    n = cp.sum_over_ranks(len(mine))
    assert t == float(W) and n == float(items), (t, n)
    # key replication, host leg: a Galois key in the reference's wire format travels from rank 0 (core/rlwe/keys.go:628)
    import hashlib
    import numpy as np
    from lattigo_amd import wire
    blob = None
    if cp.rank == 0:
        rng = np.random.default_rng(5)
        kq = rng.integers(0, 1 << 50, size=(2, 2, 3, 64), dtype=np.uint64)
        kp = rng.integers(0, 1 << 50, size=(2, 2, 1, 64), dtype=np.uint64)
        blob = wire.galois_key_marshal(5, 128, kq, kp, 0, None)
    shape = cp.broadcast_object((2, 3, 1, 0, None) if cp.rank == 0 else None)
    assert shape == (2, 3, 1, 0, None)
    got = cp.broadcast_bytes(blob, src=0)
    g, nth, q, p_, base_two, nj = wire.galois_key_unmarshal(got)
    assert (g, nth) == (5, 128) and q.shape == (2, 2, 3, 64) and p_.shape == (2, 2, 1, 64)
    digest = int(hashlib.sha256(got).hexdigest()[:12], 16)
    assert cp.max_over_ranks(digest) == digest == -cp.max_over_ranks(-digest)  # identical bytes on both ranks
    if cp.rank == 0:
        print("AGG", W, n / t)
    cp.close()
""") % ROOT


import pytest


@pytest.mark.parametrize("world,agg", [(2, "AGG 2 3.5"), (8, "AGG 8 3.625")])
def test_control_plane(tmp_path, world, agg):
    """world size 2, and 8 = one rank per GPU of the node the driver scales to"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
         "--master-port", str(29533 + world), str(script)],
        capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert agg in out.stdout


def test_digit_shares_partition_the_decomposition():
    """ControlPlane.digit_range (single key switch split by digit): contiguous, disjoint, covering, balanced to one digit; ranks
    beyond beta get an empty share"""
    from lattigo_amd.dist import ControlPlane
    for beta in range(1, 13):
        for world in range(1, 10):
            shares = [ControlPlane.digit_range(beta, r, world) for r in range(world)]
            assert [d for s in shares for d in s] == list(range(beta))
            sizes = [len(s) for s in shares]
            assert max(sizes) - min(sizes) <= 1


def test_rccl_rendezvous_has_a_deadline(monkeypatch):
    """The RCCL communicator is created on a helper thread with a deadline: a rendezvous that never completes (a peer that does not
    join) must end as an agreed failure -- the callers then fall back to the host transport -- instead of hanging the job, a fast
    failure must not be mistaken for a stuck thread, and a normal return hands the communicator over."""
    import threading
    import time

    import lattigo_amd._lib as L
    from lattigo_amd.dist import ControlPlane

    class FakeLib:
        def __init__(self, mode):
            self.mode, self.gate = mode, threading.Event()

        def he_rccl_available(self, yes):
            yes._obj.value = 1
            return 0

        def he_rccl_unique_id(self, ident):
            return 0

        def he_rccl_comm_create(self, ctx, ident, rank, world, h):
            if self.mode == "hang":
                self.gate.wait(20)
                return -3
            if self.mode == "fail":
                return -3
            h._obj.value = 4242
            return 0

        def he_rccl_comm_destroy(self, h):
            return 0

        def he_last_error(self):
            return b"fake"

    class Ctx:
        h = 1

    monkeypatch.setenv("HERING_RCCL_TIMEOUT", "0.3")
    for mode in ("hang", "fail", "ok"):
        fake = FakeLib(mode)
        monkeypatch.setattr(L, "load", lambda fake=fake: fake)
        cp = ControlPlane()
        assert cp.world == 1
        t0 = time.time()
        if mode == "ok":
            assert cp._rccl_comm(Ctx) == 4242 and not cp._rccl_abandoned
        else:
            with pytest.raises(L.HeringError, match="did not complete on every rank"):
                cp._rccl_comm(Ctx)
            assert cp._rccl_abandoned == (mode == "hang")
            assert cp._rccl is None
        assert time.time() - t0 < 5
        fake.gate.set()


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """bench.py never prints a line whose n_gpus differs from --gpus: a launcher that started another number of ranks is refused
    before anything touches a device (runs without a GPU)."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2 and "no line printed" in out.stderr and not out.stdout.strip(), (out.stdout, out.stderr)

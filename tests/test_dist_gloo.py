"""world_size-2 gloo test of the multi-GPU control plane (lattigo_amd/dist.py): sharding of
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
    assert cp.world == 2
    mine = list(cp.shard(7))
    assert mine == ([0, 2, 4, 6] if cp.rank == 0 else [1, 3, 5])
    cp.barrier()
    t = cp.max_over_ranks(1.0 + cp.rank)      # rank 1 is the slow one
    n = cp.sum_over_ranks(len(mine))
    assert t == 2.0 and n == 7.0, (t, n)
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
        print("AGG", n / t)
    cp.close()
""") % ROOT


def test_two_rank_control_plane(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29533", str(script)],
        capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "AGG 3.5" in out.stdout

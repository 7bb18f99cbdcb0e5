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

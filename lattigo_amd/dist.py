"""Control plane for the multi-GPU runs: one process per GPU, independent
ciphertexts sharded across ranks, evaluation keys replicated, NO data-path collective.  The only
cross-rank traffic is a barrier around the timed region and a MAX-reduce of the elapsed time; both
run over gloo on host tensors, so the same code is testable with world_size 2 on CPU."""
from __future__ import annotations

import os


class ControlPlane:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self._dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max_over_ranks(self, x: float) -> float:
        if self._dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: float) -> float:
        if self._dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def shard(self, n_items: int) -> range:
        """Ciphertext b goes to rank b mod world (independent units, embarrassingly parallel)."""
        return range(self.rank, n_items, self.world)

    def close(self):
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None

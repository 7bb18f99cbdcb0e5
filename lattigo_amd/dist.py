"""Control plane for the multi-GPU runs: one process per GPU, independent
ciphertexts sharded across ranks, evaluation keys replicated, NO data-path collective.  The steady-state
cross-rank traffic is a barrier around the timed region and a MAX-reduce of the elapsed time; both
run over gloo on host tensors, so the same code is testable with world_size 2 on CPU.  The one-time
replication of evaluation keys (ReplicateEvaluationKey) goes GPU-to-GPU as an RCCL broadcast over xGMI
straight into the key's device storage, or over gloo through host memory."""
from __future__ import annotations

import os


class ControlPlane:
    def __init__(self, init_single: bool = False):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self._dist = None
        self._rccl = None
        if self.world > 1 or init_single:  # init_single: a one-rank group (tests of the collective plumbing)
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

    def broadcast_object(self, obj, src: int = 0):
        if self._dist is None:
            return obj
        box = [obj if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def broadcast_bytes(self, blob, src: int = 0) -> bytes:
        """A byte string (e.g. a key in the reference's wire format, lattigo_amd/wire.py) from rank src to every rank."""
        if self._dist is None:
            return bytes(blob)
        import numpy as np
        import torch
        n = self.broadcast_object(len(blob) if self.rank == src else None, src)
        t = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()) if self.rank == src else torch.empty(n, dtype=torch.uint8)
        self._dist.broadcast(t, src=src)
        return t.numpy().tobytes()

    def _rccl_group(self, device: int):
        if self._rccl is None:
            import torch
            # torch ships its own libamdhip64; libhering binds to whichever copy the process loaded first, and two HIP
            # runtimes in one process cannot both own the GPU.  `import torch` before lattigo_amd makes them share one.
            with open("/proc/self/maps") as f:
                copies = {ln.split()[-1] for ln in f if "libamdhip64" in ln}
            if len(copies) > 1:
                raise RuntimeError("two HIP runtimes are loaded (%s): import torch before lattigo_amd to use the RCCL transport"
                                   % ", ".join(sorted(copies)))
            torch.cuda.set_device(device)
            self._rccl = self._dist.new_group(backend="nccl")  # "nccl" is RCCL on ROCm
        return self._rccl

    def ReplicateEvaluationKey(self, evaluator, key=None, src: int = 0, transport: str = "rccl"):
        """Rank src holds `key` (rlwe.EvaluationKey); every rank returns a device-resident copy of it.  transport "rccl":
        the peers allocate an empty key of the same shape and the key words are broadcast GPU-to-GPU into it
        (he_evk_device_buffer / he_evk_commit); "host": downloaded on src, broadcast over gloo, uploaded by the peers."""
        from .rlwe import EvaluationKey
        if self._dist is None:
            return key
        shape = self.broadcast_object(key.Shape() if self.rank == src else None, src)
        beta, nQk, nPk, base_two, nj = shape
        if transport == "rccl":
            import torch
            dev = evaluator.ringQ.ctx.device_id
            if self.rank != src:
                key = EvaluationKey(evaluator, None, None, base_two, nj, shape=(beta, nQk, nPk))
            ptr, nbytes = key.DeviceBuffer()  # drains the context's stream

            class _DeviceWords:
                __cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<i8", "data": (ptr, False), "version": 2}

            t = torch.as_tensor(_DeviceWords(), device=f"cuda:{dev}")
            assert t.data_ptr() == ptr
            self._dist.broadcast(t, src=src, group=self._rccl_group(dev))
            torch.cuda.synchronize(dev)
            if self.rank != src:
                key.Commit()
            return key
        if transport != "host":
            raise ValueError("transport is 'rccl' or 'host'")
        import numpy as np
        import torch
        words = key.download() if self.rank == src else np.empty((beta, 2, nQk + nPk, evaluator.ringQ.N), dtype=np.uint64)
        t = torch.from_numpy(words.view(np.int64))
        self._dist.broadcast(t, src=src)
        if self.rank == src:
            return key
        return EvaluationKey(evaluator, words[:, :, :nQk], words[:, :, nQk:], base_two, nj)

    def shard(self, n_items: int) -> range:
        """Ciphertext b goes to rank b mod world (independent units, embarrassingly parallel)."""
        return range(self.rank, n_items, self.world)

    def close(self):
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None

"""Control plane for the multi-GPU runs: one process per GPU, independent
ciphertexts sharded across ranks, evaluation keys replicated, NO data-path collective.  The steady-state
cross-rank traffic is a barrier around the timed region and a MAX-reduce of the elapsed time; both
run over gloo on host tensors, so the same code is testable with world_size 2 on CPU.  The one-time
replication of evaluation keys (ReplicateEvaluationKey) goes GPU-to-GPU as an RCCL broadcast over xGMI
straight into the key's device storage, or over gloo through host memory.  The RCCL leg is driven by libhering
itself (he_rccl_* / he_evk_broadcast, include/hering.h) on the context's stream: torch is used for the gloo
control plane only -- CPU tensors -- so no import order has to be respected and no second HIP runtime is touched."""
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

    def rccl_world(self):
        """Ranks of the RCCL communicator as RCCL itself counts them (an all-reduce of ones on the device, he_rccl_comm_ranks), or
        None when no communicator was created in this process (no key replication / split key switch over RCCL took place)."""
        if self._rccl is None:
            return None
        import ctypes as C
        from ._lib import check, load
        n = C.c_int()
        check(load().he_rccl_comm_ranks(self._rccl[1], C.byref(n)))
        return int(n.value)

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

    def _rccl_comm(self, ctx):
        """The RCCL communicator of this process's context (created on first use; collective: every rank must call it): rank 0
        draws the id, the gloo control plane hands it round, he_rccl_comm_create joins."""
        if self._rccl is None:
            import ctypes as C
            from ._lib import H, HeringError, check, load
            # every step that can fail on ONE rank is agreed on over the control plane before anybody enters a collective: a rank
            # that raised while the others wait inside ncclCommInitRank (or inside the id's broadcast) would hang the job
            yes = C.c_int()
            check(load().he_rccl_available(C.byref(yes)))
            if self.sum_over_ranks(float(yes.value)) != self.world:
                raise HeringError(-3, "RCCL is not available on every rank")
            ident, ok = (C.c_uint8 * 128)(), 1
            if self.rank == 0 and load().he_rccl_unique_id(ident) != 0:
                ok = 0
            blob = self.broadcast_bytes(bytes([ok]) + bytes(ident), src=0) if self._dist is not None else bytes([ok]) + bytes(ident)
            if blob[0] == 0:
                raise HeringError(-3, "rank 0 could not draw an RCCL id")
            ident = (C.c_uint8 * 128)(*blob[1:])
            # the rendezvous itself, with a deadline (HERING_RCCL_TIMEOUT seconds, default 120): a peer that never joins -- a rank
            # on a broken link, an interface RCCL cannot bootstrap over -- would otherwise hang the whole job inside
            # ncclCommInitRank.  It runs on a helper thread that is abandoned on timeout (the library does not hold the context's
            # lock across the rendezvous); the outcome is agreed on, so that either every rank has a communicator or none uses one.
            joined = self._rccl_join(ctx, ident, float(os.environ.get("HERING_RCCL_TIMEOUT", "120")))
            if self.sum_over_ranks(1.0 if joined is not None else 0.0) != self.world:
                self._rccl_abandoned = self._rccl_abandoned or self._rccl_thread_stuck
                if joined is not None:
                    load().he_rccl_comm_destroy(joined)
                raise HeringError(-3, "the RCCL rendezvous did not complete on every rank within the deadline")
            self._rccl = (ctx, joined)
        if self._rccl[0] is not ctx:
            raise RuntimeError("the RCCL communicator of this process belongs to another context")
        return self._rccl[1]

    _rccl_abandoned = False     # a helper thread of this process is still inside ncclCommInitRank: leave with os._exit (close())
    _rccl_thread_stuck = False  # ... as of the last _rccl_join

    def _rccl_join(self, ctx, ident, timeout_s: float):
        """he_rccl_comm_create on a helper thread; the communicator handle, or None when the call failed or did not return in time"""
        import ctypes as C
        import threading
        from ._lib import H, load
        box = {}

        def work():
            h = H()
            box["rc"] = load().he_rccl_comm_create(ctx.h, ident, self.rank, self.world, C.byref(h))
            box["h"] = h.value

        t = threading.Thread(target=work, daemon=True, name="hering-rccl-rendezvous")
        t.start()
        t.join(timeout_s)
        self._rccl_thread_stuck = t.is_alive()
        if t.is_alive() or box.get("rc") != 0:
            return None
        return box["h"]

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
            from ._lib import check, load
            if self.rank != src:
                key = EvaluationKey(evaluator, None, None, base_two, nj, shape=(beta, nQk, nPk))
            # on the context's stream: the words land in the key's device storage, the derived copy is refreshed behind them
            check(load().he_evk_broadcast(self._rccl_comm(evaluator.ringQ.ctx), key.h, src))
            evaluator.ringQ.ctx.sync()
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

    # ---- one key switch split over the ranks by digit ("RCCL ... when a single op is split") ----------
    @staticmethod
    def digit_range(beta: int, rank: int, world: int) -> range:
        """contiguous share of the beta digits of the RNS decomposition (ranks beyond beta get an empty share)"""
        return range(beta * rank // world, beta * (rank + 1) // world)

    def AllReduceSumPolys(self, polys, rings, transport: str = "rccl"):
        """In place: every polynomial becomes the limb-wise sum over the ranks, reduced to [0, q).  Inputs canonical; the word
        sum of `world` canonical residues must not wrap (world * q < 2^64, checked)."""
        if self._dist is None:
            return
        for p, r in zip(polys, rings):
            if self.world * max(int(q) for q in r.moduli) >= 1 << 64:
                raise ValueError("AllReduceSumPolys: world * q exceeds 64 bits")
        if transport == "rccl":
            from ._lib import check, load
            comm = self._rccl_comm(rings[0].ctx)
            for p in polys:  # addition of the 64-bit words mod 2^64, in place, on the context's stream (he_poly_all_reduce_sum)
                check(load().he_poly_all_reduce_sum(comm, p.h))
        elif transport == "host":
            import numpy as np
            import torch
            for p in polys:
                w = p.download()
                t = torch.from_numpy(w.view(np.int64))
                self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
                p.upload(w)
        else:
            raise ValueError("transport is 'rccl' or 'host'")
        for p, r in zip(polys, rings):
            r.AtLevel(p.n_limbs - 1).Reduce(p, p)

    def SplitGadgetProductHoisted(self, evaluator, levelQ: int, decomp, evk, ct, transport: str = "rccl"):
        """Evaluator.GadgetProductHoisted with the digits of the inner product shared out over the ranks: every rank holds
        the same decomposition (decomp, from DecomposeNTT) and needs only ITS digits of the key to be meaningful; it
        accumulates digits digit_range(beta, rank, world), the four partial (Q, P) accumulators are summed across the ranks
        (RCCL all-reduce over xGMI, or gloo through the host) and every rank finishes with ModDown.  Bit-identical to the
        unsplit call.  This trades a 2 (L + alpha)-limb all-reduce per ciphertext for 1/world of the key memory and of the
        inner-product work: the decomposition and the ModDown are not split."""
        from .ring import Poly
        rQ, rP = evaluator.ringQ, evaluator.ringP
        levelP = evk.LevelP()
        levelQ = min(levelQ, evk.LevelQ())  # utils.Min(levelQ, gadgetCt.LevelQ()), as the unsplit call (and the C side) does
        from .rlwe import BaseRNSDecompositionVectorSize
        beta = BaseRNSDecompositionVectorSize(levelQ, levelP)
        share = self.digit_range(beta, self.rank, self.world)
        B = decomp.batch
        ctQP = [(Poly(rQ, levelQ + 1, B, zero=False), Poly(rP, levelP + 1, B, zero=False)) for _ in range(2)]
        evaluator.GadgetProductHoistedLazyDigits(levelQ, decomp, evk, share.start, share.stop, ctQP)
        self.AllReduceSumPolys([ctQP[0][0], ctQP[0][1], ctQP[1][0], ctQP[1][1]], [rQ, rP, rQ, rP], transport)
        evaluator.ModDown(levelQ, levelP, ctQP, ct)

    def shard(self, n_items: int) -> range:
        """Ciphertext b goes to rank b mod world (independent units, embarrassingly parallel)."""
        return range(self.rank, n_items, self.world)

    def close(self):
        if self._rccl is not None:
            from ._lib import load
            load().he_rccl_comm_destroy(self._rccl[1])
            self._rccl = None
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None
        if self._rccl_abandoned:
            # a thread of this process never came back from the RCCL rendezvous: the interpreter's orderly shutdown (and RCCL's own
            # exit handlers) could wait for it for ever.  Leave through os._exit when the program ends -- with the status it ends with.
            import atexit
            import sys
            status = {"code": 0}
            orig_exit, orig_hook = sys.exit, sys.excepthook

            def exit_recording(code=0):
                status["code"] = code if isinstance(code, int) else (0 if code is None else 1)
                orig_exit(code)

            def hook_recording(et, ev, tb):
                # a run that dies on an uncaught exception (a failed verification assert, ...) must not leave with status 0
                status["code"] = ev.code if isinstance(ev, SystemExit) and isinstance(ev.code, int) else 1
                orig_hook(et, ev, tb)

            sys.exit = exit_recording
            sys.excepthook = hook_recording

            def hard_exit():
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(status["code"])

            atexit.register(hard_exit)

"""Host-side mirror of the reference's ``ring`` package surface for the hot path
(ring.Ring, ring.Poly, ring.BasisExtender), backed by libhering's HIP kernels.

Method names and argument order follow the reference (outputs last, ``level``
implicit in ``Ring.AtLevel``); every method cites the Go method it mirrors.
Polynomials are device-resident batches ``[batch][limbs][N]`` (``Poly``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import H, check, load, u64p

BINOPS = {
    "Add": 0, "AddLazy": 1, "Sub": 2, "SubLazy": 3,
    "MulCoeffsBarrett": 4, "MulCoeffsBarrettLazy": 5, "MulCoeffsBarrettThenAdd": 6,
    "MulCoeffsBarrettThenAddLazy": 7,
    "MulCoeffsMontgomery": 8, "MulCoeffsMontgomeryLazy": 9, "MulCoeffsMontgomeryLazyThenNeg": 10,
    "MulCoeffsMontgomeryThenAdd": 11, "MulCoeffsMontgomeryThenAddLazy": 12,
    "MulCoeffsMontgomeryLazyThenAddLazy": 13,
    "MulCoeffsMontgomeryThenSub": 14, "MulCoeffsMontgomeryThenSubLazy": 15,
    "MulCoeffsMontgomeryLazyThenSubLazy": 16,
}
UNOPS = {"Neg": 0, "Reduce": 1, "ReduceLazy": 2, "MForm": 3, "MFormLazy": 4, "IMForm": 5}
SCALAROPS = {"AddScalar": 0, "SubScalar": 1, "MulScalar": 2, "MulScalarThenAdd": 3, "MulScalarThenSub": 4}


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _words(x: int) -> np.ndarray:
    if x < 0:
        raise ValueError("negative big integers must be reduced by the caller")
    w = []
    while True:
        w.append(x & 0xFFFFFFFFFFFFFFFF)
        x >>= 64
        if x == 0:
            break
    return np.array(w, dtype=np.uint64)


class Graph:
    """A captured sequence of calls (he_graph_*).  Objects created inside the `with` block keep their buffers for the life of
    the graph: hold on to the ones whose results you want to read after a replay."""

    def __init__(self, ctx):
        self.ctx, self.h = ctx, None

    def __enter__(self):
        check(load().he_graph_begin(self.ctx.h))
        return self

    def __exit__(self, et, ev, tb):
        h = H()
        rc = load().he_graph_end(self.ctx.h, C.byref(h))
        if rc == 0 and et is not None:
            load().he_graph_destroy(h.value)  # the body raised: drop what was recorded (a live graph pins the context's scratch)
        elif rc == 0:
            self.h = h.value
        elif et is None:
            check(rc)
        return False

    def launch(self):
        check(load().he_graph_launch(self.h))

    def nodes(self) -> int:
        n = C.c_int()
        check(load().he_graph_nodes(self.h, C.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "h", None):
            load().he_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count() -> int:
    """HIP devices visible to this process (he_device_count)."""
    n = C.c_int()
    check(load().he_device_count(C.byref(n)))
    return int(n.value)


def device_pci_bus_id(device_id: int = 0):
    """PCI bus id of a HIP device ("0000:05:00.0"), lower case as sysfs spells it; None when the runtime cannot tell."""
    buf = C.create_string_buffer(64)
    if load().he_debug_device_pci_bus_id(int(device_id), buf, 64) != 0:
        return None
    return buf.value.decode().lower() or None


class Context:
    """One HIP device + stream (one per process/GPU)."""

    def __init__(self, device_id: int = 0):
        h = H()
        check(load().he_ctx_create(device_id, C.byref(h)))
        self.h = h.value
        self.device_id = device_id

    def close(self):
        if getattr(self, "h", None):
            load().he_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(load().he_ctx_sync(self.h))

    def SetCoalescing(self, max_batch: int = 64, window_us: int = 30):
        """he_ctx_set_coalescing (include/hering.h): concurrent single-ciphertext calls of ANY operator on this context -- one OS
        thread per ciphertext, the reference's own parallel mode (b.RunParallel) -- are gathered into batched launches over the
        callers' own polynomials.  max_batch <= 1 switches it off."""
        check(load().he_ctx_set_coalescing(self.h, max_batch, window_us))

    def SetDeferred(self, depth: int = 8):
        """he_ctx_set_deferred (include/hering.h): queued calls return as soon as they are filed; the context's dispatcher thread
        launches them (a thread's calls in order, up to `depth` pending per thread).  Failures of a deferred launch surface at the
        next Sync().  depth = 0 switches back to calls that return once launched."""
        check(load().he_ctx_set_deferred(self.h, depth))

    def CoalescingStats(self) -> dict:
        out = (C.c_uint64 * 4)()
        check(load().he_ctx_coalescing_stats(self.h, out))
        return {"calls": int(out[0]), "launches": int(out[1]), "largest_batch": int(out[2]), "one_by_one": int(out[3])}

    def capture(self):
        """`with ctx.capture() as g: <calls>` records the calls made on this context into a replayable hipGraph
        (he_graph_begin / he_graph_end, include/hering.h); `g.launch()` enqueues the whole sequence at once."""
        return Graph(self)

    def timer_start(self):
        check(load().he_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(load().he_timer_stop(self.h, C.byref(ms)))
        return float(ms.value)

    def device_info(self):
        out = (C.c_uint64 * 4)()
        check(load().he_device_info(self.h, out))
        return dict(cus=int(out[0]), lds_per_cu=int(out[1]), clock_khz=int(out[2]), hbm_bytes=int(out[3]))

    def prof_begin(self):
        check(load().he_prof_begin(self.h))

    def prof_end(self) -> dict:
        """{kernel name: (launches, total ms)} since prof_begin()."""
        n = 32
        counts, ms, nk = (C.c_int * n)(), (C.c_float * n)(), C.c_int()
        check(load().he_prof_end(self.h, n, counts, ms, C.byref(nk)))
        return {load().he_prof_kernel_name(i).decode(): (int(counts[i]), float(ms[i])) for i in range(nk.value) if counts[i]}

    def prof_end_bytes(self) -> dict:
        """{kernel name: (launches, total ms, algorithmic bytes)} since prof_begin() (hering_debug.h)."""
        n = 32
        counts, ms, by, nk = (C.c_int * n)(), (C.c_float * n)(), (C.c_double * n)(), C.c_int()
        check(load().he_prof_end_bytes(self.h, n, counts, ms, by, C.byref(nk)))
        return {load().he_prof_kernel_name(i).decode(): (int(counts[i]), float(ms[i]), float(by[i])) for i in range(nk.value) if counts[i]}

    def alg_bytes(self, reset: bool = False):
        """(bytes with the key charged per batch entry, bytes with the key read once per call) of the primitives called on this
        context since the last reset, by SURVEY.md section 8(d)'s per-primitive formulas (hering_debug.h)."""
        out = (C.c_double * 2)()
        check(load().he_alg_bytes(self.h, int(reset), out))
        return float(out[0]), float(out[1])

    def alg_valu(self, reset: bool = False):
        """he_alg_valu (hering_debug.h): [integer-class multiplies, double-precision-class multiplies, butterflies among each]"""
        out = (C.c_double * 4)()
        check(load().he_alg_valu(self.h, int(reset), out))
        return [float(x) for x in out]

    def probe_modmul(self, iters=256) -> float:
        out = C.c_double()
        check(load().he_probe_modmul(self.h, iters, C.byref(out)))
        return float(out.value)

    def probe_modmul_f64(self, iters=256) -> float:
        out = C.c_double()
        check(load().he_probe_modmul_f64(self.h, iters, C.byref(out)))
        return float(out.value)


class Poly:
    """Device-resident ring.Poly batch (ring/poly.go:13): [batch][limbs][N] uint64."""

    def __init__(self, ring: "Ring", n_limbs: int | None = None, batch: int = 1, zero: bool = True):
        """zero=False: contents unspecified (a result / temporary the next operation overwrites in full)"""
        self.ring = ring
        self.n_limbs = ring.MaxLevel() + 1 if n_limbs is None else n_limbs
        self.batch = batch
        self.N = ring.N
        h = H()
        check((load().he_poly_alloc if zero else load().he_poly_alloc_scratch)(ring.h, self.n_limbs, batch, C.byref(h)))
        self.h = h.value

    def free(self):
        if getattr(self, "h", None):
            load().he_poly_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def DeviceBuffer(self):
        """(device pointer, bytes) of the words [batch][n_limbs][N]; the context's stream is drained first."""
        ptr, nb = C.c_void_p(), C.c_size_t()
        check(load().he_poly_device_buffer(self.h, C.byref(ptr), C.byref(nb)))
        return ptr.value, nb.value

    def Level(self):
        return self.n_limbs - 1

    def upload(self, arr) -> "Poly":
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(self.batch, self.n_limbs, self.N)
        check(load().he_poly_upload(self.h, _p(a), a.size))
        return self

    def download(self) -> np.ndarray:
        out = np.empty((self.batch, self.n_limbs, self.N), dtype=np.uint64)
        check(load().he_poly_download(self.h, _p(out), out.size))
        return out

    def get(self, level=None) -> np.ndarray:
        """[limbs, N] of batch entry 0 (or [batch, limbs, N] when batch > 1), limbs 0..level."""
        a = self.download()
        if level is not None:
            a = a[:, : level + 1]
        return a[0] if self.batch == 1 else a

    def upload_limb(self, b, limb, row):
        r = np.ascontiguousarray(row, dtype=np.uint64)
        assert r.size == self.N
        check(load().he_poly_upload_limb(self.h, b, limb, _p(r)))

    def download_limb(self, b, limb) -> np.ndarray:
        out = np.empty(self.N, dtype=np.uint64)
        check(load().he_poly_download_limb(self.h, b, limb, _p(out)))
        return out

    def MarshalBinary(self, b: int = 0) -> bytes:
        """ring.Poly.MarshalBinary (ring/poly.go:167) of batch entry b."""
        from . import wire
        return wire.poly_marshal(self.download()[b])

    def UnmarshalBinary(self, data: bytes, b: int = 0):
        """ring.Poly.UnmarshalBinary (ring/poly.go:176) into batch entry b."""
        from . import wire
        a = wire.poly_unmarshal(data)
        if a.shape != (self.n_limbs, self.N):
            raise ValueError(f"polynomial shape {a.shape} does not match ({self.n_limbs}, {self.N})")
        for i in range(self.n_limbs):
            self.upload_limb(b, i, a[i])

    def CopyLvl(self, level, src: "Poly"):
        check(load().he_poly_copy(self.h, src.h, level))

    def CopyBatch(self, level, dst_b0: int, src: "Poly", src_b0: int, nb: int):
        """batch entries [src_b0, src_b0 + nb) of src -> entries [dst_b0, dst_b0 + nb) of self (limbs 0..level)"""
        check(load().he_poly_copy_batch(self.h, dst_b0, src.h, src_b0, nb, level))

    def Zero(self):
        check(load().he_poly_zero(self.h))


class Ring:
    """ring.Ring (ring/ring.go:71), standard type.  ``AtLevel`` returns a view sharing tables."""

    def __init__(self, ctx: Context, N: int, moduli, _parent: "Ring | None" = None, _level: int | None = None,
                 conjugate_invariant: bool = False):
        """ring.NewRing / ring.NewRingFromType (ring/ring.go:207,267)."""
        self.ctx = ctx
        self.N = N
        self.moduli = [int(m) for m in moduli]
        self.conjugate_invariant = conjugate_invariant if _parent is None else _parent.conjugate_invariant
        if _parent is not None:
            self.h, self._owner, self.level = _parent.h, _parent._owner, _level
            return
        logN = N.bit_length() - 1
        if N <= 0 or (1 << logN) != N:
            raise _lib.HeringError(-4, "invalid ring degree: must be a power of 2")
        arr = (C.c_uint64 * len(self.moduli))(*self.moduli)
        h = H()
        check(load().he_ring_create_type(ctx.h, logN, int(conjugate_invariant), arr, len(self.moduli), C.byref(h)))
        self.h = h.value
        self._owner = self
        self.level = len(self.moduli) - 1

    def close(self):
        if self._owner is self and getattr(self, "h", None):
            load().he_ring_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- structure (ring/ring.go:175-213)
    def AtLevel(self, level: int) -> "Ring":
        if level < 0:
            raise ValueError("level cannot be negative")
        if level > self.MaxLevel():
            raise ValueError("level cannot be larger than max level")
        return Ring(self.ctx, self.N, self.moduli, _parent=self, _level=level)

    def Level(self):
        return self.level

    def MaxLevel(self):
        return len(self.moduli) - 1

    def ModuliChain(self):
        return list(self.moduli)

    def NthRoot(self):
        return (4 if self.conjugate_invariant else 2) * self.N

    def NewPoly(self, batch: int = 1) -> Poly:
        return Poly(self, self.level + 1, batch)

    def constant(self, limb, which) -> int:
        out = C.c_uint64()
        check(load().he_ring_constant(self.h, limb, which, C.byref(out)))
        return int(out.value)

    def roots(self, limb, backward=False) -> np.ndarray:
        out = np.empty(self.N, dtype=np.uint64)
        check(load().he_ring_roots(self.h, limb, int(backward), _p(out)))
        return out

    # -- NTT (ring/ntt.go:127-152)
    def NTT(self, p1: Poly, p2: Poly):
        check(load().he_ntt(self.h, self.level, p1.h, p2.h))

    def NTTLazy(self, p1: Poly, p2: Poly):
        check(load().he_ntt_lazy(self.h, self.level, p1.h, p2.h))

    def INTT(self, p1: Poly, p2: Poly):
        check(load().he_intt(self.h, self.level, p1.h, p2.h))

    def INTTLazy(self, p1: Poly, p2: Poly):
        check(load().he_intt_lazy(self.h, self.level, p1.h, p2.h))

    # -- ring.NumberTheoreticTransformer on host slices (ring/ntt.go:17-22), one limb
    def _ntt_host(self, limb, backward, lazy, p1):
        a = np.ascontiguousarray(p1, dtype=np.uint64)
        out = np.empty(self.N, dtype=np.uint64)
        check(load().he_subring_ntt_host(self.h, limb, int(backward), int(lazy), _p(a), _p(out)))
        return out

    def Forward(self, limb, p1):
        return self._ntt_host(limb, False, False, p1)

    def ForwardLazy(self, limb, p1):
        return self._ntt_host(limb, False, True, p1)

    def Backward(self, limb, p1):
        return self._ntt_host(limb, True, False, p1)

    def BackwardLazy(self, limb, p1):
        return self._ntt_host(limb, True, True, p1)

    # -- coefficient-wise (ring/operations.go:11-377)
    def binop(self, name: str, p1: Poly, p2: Poly, p3: Poly):
        check(load().he_binop(self.h, self.level, BINOPS[name], p1.h, p2.h, p3.h))

    def unop(self, name: str, p1: Poly, p2: Poly):
        check(load().he_unop(self.h, self.level, UNOPS[name], p1.h, p2.h))

    def scalarop(self, name: str, p1: Poly, scalar: int, p2: Poly):
        check(load().he_scalarop(self.h, self.level, SCALAROPS[name], p1.h, scalar, p2.h))

    def MulRNSScalarMontgomery(self, p1: Poly, scalar, p2: Poly):
        sc = np.ascontiguousarray(scalar, dtype=np.uint64)
        check(load().he_mul_rns_scalar_montgomery(self.h, self.level, p1.h, _p(sc), p2.h))

    def _nonneg(self, scalar: int) -> int:
        """big.Int.Mod is Euclidean: a negative scalar acts as its non-negative residue in every limb"""
        scalar = int(scalar)
        if scalar < 0:
            m = 1
            for q in self.ModuliChain()[: self.level + 1]:
                m *= int(q)
            scalar %= m
        return scalar

    def AddScalarBigint(self, p1: Poly, scalar: int, p2: Poly):
        w = _words(self._nonneg(scalar))
        check(load().he_add_scalar_bigint(self.h, self.level, p1.h, _p(w), len(w), p2.h))

    def SubScalarBigint(self, p1: Poly, scalar: int, p2: Poly):
        w = _words(self._nonneg(scalar))
        check(load().he_sub_scalar_bigint(self.h, self.level, p1.h, _p(w), len(w), p2.h))

    def MulScalarBigint(self, p1: Poly, scalar: int, p2: Poly):
        w = _words(self._nonneg(scalar))
        check(load().he_mul_scalar_bigint(self.h, self.level, p1.h, _p(w), len(w), p2.h))

    # -- rescale (ring/scaling.go)
    def MulScalarBigintThenAdd(self, p1: Poly, scalar: int, p2: Poly):
        w = _words(self._nonneg(scalar))
        check(load().he_mul_scalar_bigint_then_add(self.h, self.level, p1.h, _p(w), len(w), p2.h))

    # Ring.{Add,Sub,Mul}DoubleRNSScalar[ThenAdd] (ring/operations.go:166-184, 249-268)
    def _double(self, op, p1: Poly, scalar0, scalar1, p2: Poly):
        s0 = np.ascontiguousarray(scalar0, dtype=np.uint64)
        s1 = np.ascontiguousarray(scalar1, dtype=np.uint64)
        if s0.size < self.level + 1 or s1.size < self.level + 1:
            raise ValueError("RNS scalars need one residue per limb")
        check(load().he_double_rns_scalarop(self.h, self.level, op, p1.h, _p(s0), _p(s1), p2.h))

    def AddDoubleRNSScalar(self, p1: Poly, scalar0, scalar1, p2: Poly):
        self._double(0, p1, scalar0, scalar1, p2)

    def SubDoubleRNSScalar(self, p1: Poly, scalar0, scalar1, p2: Poly):
        self._double(1, p1, scalar0, scalar1, p2)

    def MulDoubleRNSScalar(self, p1: Poly, scalar0, scalar1, p2: Poly):
        self._double(2, p1, scalar0, scalar1, p2)

    def MulDoubleRNSScalarThenAdd(self, p1: Poly, scalar0, scalar1, p2: Poly):
        self._double(3, p1, scalar0, scalar1, p2)

    def EvalPolyScalar(self, p1s, scalar: int, p2: Poly):
        """Ring.EvalPolyScalar (ring/operations.go:271)"""
        p2.CopyLvl(self.level, p1s[-1])
        for i in range(len(p1s) - 1, 0, -1):
            self.MulScalar(p2, scalar, p2)
            self.Add(p2, p1s[i - 1], p2)

    def Shift(self, p1: Poly, k: int, p2: Poly):
        """Ring.Shift (ring/operations.go:279)"""
        check(load().he_shift(self.h, self.level, p1.h, int(k), p2.h))

    def MultByMonomial(self, p1: Poly, k: int, p2: Poly):
        """Ring.MultByMonomial (ring/operations.go:307)"""
        check(load().he_mult_by_monomial(self.h, self.level, p1.h, int(k), p2.h))

    def MulByVectorMontgomery(self, p1: Poly, vector: Poly, p2: Poly):
        """Ring.MulByVectorMontgomery (ring/operations.go:363); `vector`: a batch-1 polynomial whose limb 0 is the vector"""
        check(load().he_mul_by_vector_montgomery(self.h, self.level, p1.h, vector.h, 0, p2.h))

    def MulByVectorMontgomeryThenAddLazy(self, p1: Poly, vector: Poly, p2: Poly):
        check(load().he_mul_by_vector_montgomery(self.h, self.level, p1.h, vector.h, 1, p2.h))

    def AutomorphismNTT(self, pin: Poly, galel: int, pout: Poly):
        """Ring.AutomorphismNTT (ring/automorphism.go:38)"""
        self.AutomorphismNTTWithIndex(pin, self.AutomorphismNTTIndex(galel), pout)

    def DivRoundByLastModulusNTT(self, p0: Poly, p1: Poly):
        check(load().he_div_round_by_last_modulus_ntt(self.h, self.level, p0.h, p1.h))

    def DivRoundByLastModulus(self, p0: Poly, p1: Poly):
        check(load().he_div_round_by_last_modulus(self.h, self.level, p0.h, p1.h))

    def DivFloorByLastModulusNTT(self, p0: Poly, p1: Poly):
        check(load().he_div_floor_by_last_modulus_ntt(self.h, self.level, p0.h, p1.h))

    def DivFloorByLastModulus(self, p0: Poly, p1: Poly):
        check(load().he_div_floor_by_last_modulus(self.h, self.level, p0.h, p1.h))

    def DivRoundByLastModulusManyNTT(self, nb: int, p0: Poly, p1: Poly):
        check(load().he_div_round_by_last_modulus_many_ntt(self.h, self.level, nb, p0.h, p1.h))

    def DivRoundByLastModulusMany(self, nb: int, p0: Poly, p1: Poly):
        check(load().he_div_round_by_last_modulus_many(self.h, self.level, nb, p0.h, p1.h))

    def DivFloorByLastModulusManyNTT(self, nb: int, p0: Poly, p1: Poly):
        check(load().he_div_floor_by_last_modulus_many_ntt(self.h, self.level, nb, p0.h, p1.h))

    def DivFloorByLastModulusMany(self, nb: int, p0: Poly, p1: Poly):
        check(load().he_div_floor_by_last_modulus_many(self.h, self.level, nb, p0.h, p1.h))

    # -- automorphism (ring/automorphism.go)
    def AutomorphismNTTIndex(self, galel: int) -> "AutomorphismIndex":
        return AutomorphismIndex(self, galel)

    def AutomorphismNTTWithIndex(self, pin: Poly, index: "AutomorphismIndex", pout: Poly):
        check(load().he_automorphism_ntt_with_index(self.h, self.level, pin.h, index.h, pout.h))

    def AutomorphismNTTWithIndexThenAddLazy(self, pin: Poly, index: "AutomorphismIndex", pout: Poly):
        check(load().he_automorphism_ntt_with_index_then_add_lazy(self.h, self.level, pin.h, index.h, pout.h))

    def Automorphism(self, pin: Poly, galel: int, pout: Poly):
        check(load().he_automorphism(self.h, self.level, pin.h, galel, pout.h))


def _named(name, table, kind):
    if kind == "bin":
        def f(self, p1, p2, p3):
            self.binop(name, p1, p2, p3)
    elif kind == "un":
        def f(self, p1, p2):
            self.unop(name, p1, p2)
    else:
        def f(self, p1, scalar, p2):
            self.scalarop(name, p1, scalar, p2)
    f.__name__ = name
    f.__doc__ = f"Ring.{name} (ring/operations.go)"
    return f


for _n in BINOPS:
    setattr(Ring, _n, _named(_n, BINOPS, "bin"))
for _n in UNOPS:
    setattr(Ring, _n, _named(_n, UNOPS, "un"))
for _n in SCALAROPS:
    setattr(Ring, _n, _named(_n, SCALAROPS, "sc"))


class AutomorphismIndex:
    """Device copy of ring.AutomorphismNTTIndex (ring/automorphism.go:12)."""

    def __init__(self, ring: Ring, galel: int):
        h = H()
        check(load().he_automorphism_index_create(ring.h, galel, C.byref(h)))
        self.h = h.value
        self.N = ring.N

    def __del__(self):
        try:
            if getattr(self, "h", None):
                load().he_automorphism_index_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def download(self) -> np.ndarray:
        out = np.empty(self.N, dtype=np.uint64)
        check(load().he_automorphism_index_download(self.h, _p(out)))
        return out


class BasisExtender:
    """ring.BasisExtender (ring/basis_extension.go:14)."""

    def __init__(self, ringQ: Ring, ringP: Ring):
        self.ringQ, self.ringP = ringQ, ringP
        h = H()
        check(load().he_basis_extender_create(ringQ.h, ringP.h, C.byref(h)))
        self.h = h.value

    def __del__(self):
        try:
            if getattr(self, "h", None):
                load().he_basis_extender_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def ModUpQtoP(self, levelQ, levelP, polQ: Poly, polP: Poly):
        check(load().he_modup_q_to_p(self.h, levelQ, levelP, polQ.h, polP.h))

    def ModUpPtoQ(self, levelP, levelQ, polP: Poly, polQ: Poly):
        check(load().he_modup_p_to_q(self.h, levelP, levelQ, polP.h, polQ.h))

    def ModDownQPtoQ(self, levelQ, levelP, p1Q: Poly, p1P: Poly, p2Q: Poly):
        check(load().he_moddown_qp_to_q(self.h, levelQ, levelP, p1Q.h, p1P.h, p2Q.h))

    def ModDownQPtoQNTT(self, levelQ, levelP, p1Q: Poly, p1P: Poly, p2Q: Poly):
        check(load().he_moddown_qp_to_q_ntt(self.h, levelQ, levelP, p1Q.h, p1P.h, p2Q.h))

    def ModDownQPtoP(self, levelQ, levelP, p1Q: Poly, p1P: Poly, p2P: Poly):
        check(load().he_moddown_qp_to_p(self.h, levelQ, levelP, p1Q.h, p1P.h, p2P.h))

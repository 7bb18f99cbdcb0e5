"""ctypes wrapper around the CPU ORACLE (oracle/lattigo_oracle.c).

TEST INFRASTRUCTURE ONLY.  Nothing under ``lattigo_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do -- as the checker / timed CPU baseline.

Polynomials are numpy ``uint64`` arrays of shape ``[limbs, N]`` (C-contiguous),
the contiguous form of the reference's ``ring.Poly.Coeffs`` (ring/poly.go:13).
Method names mirror the reference's (``Ring.NTT``, ``MulCoeffsMontgomery`` ...).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblattigo_oracle.so")

u64p = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few seconds)."""
    src = os.path.join(_HERE, "lattigo_oracle.c")
    hdr = os.path.join(_HERE, "lattigo_oracle.h")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


class _Evk(C.Structure):
    _fields_ = [("beta", C.c_int), ("nQk", C.c_int), ("nPk", C.c_int), ("q", u64p), ("p", u64p),
                ("pw2", C.c_int), ("nj", C.c_int * 64)]


def _declare(L):
    vp = C.c_void_p
    u64 = C.c_uint64
    i = C.c_int
    L.lo_last_error.restype = C.c_char_p
    for name in ("lo_mform", "lo_mform_lazy"):
        getattr(L, name).restype = u64
        getattr(L, name).argtypes = [u64, u64, u64p]
    for name in ("lo_imform", "lo_imform_lazy"):
        getattr(L, name).restype = u64
        getattr(L, name).argtypes = [u64, u64, u64]
    L.lo_gen_mred_constant.restype = u64
    L.lo_gen_mred_constant.argtypes = [u64]
    L.lo_gen_bred_constant.argtypes = [u64, u64p]
    for name in ("lo_mred", "lo_mred_lazy"):
        getattr(L, name).restype = u64
        getattr(L, name).argtypes = [u64, u64, u64, u64]
    for name in ("lo_bred_add", "lo_bred_add_lazy"):
        getattr(L, name).restype = u64
        getattr(L, name).argtypes = [u64, u64, u64p]
    for name in ("lo_bred", "lo_bred_lazy"):
        getattr(L, name).restype = u64
        getattr(L, name).argtypes = [u64, u64, u64, u64p]
    L.lo_cred.restype = u64
    L.lo_cred.argtypes = [u64, u64]
    L.lo_modexp.restype = u64
    L.lo_modexp.argtypes = [u64, u64, u64]
    L.lo_is_prime.restype = i
    L.lo_is_prime.argtypes = [u64]
    L.lo_ring_new.restype = vp
    L.lo_ring_new.argtypes = [i, u64p, i]
    L.lo_ring_new_type.restype = vp
    L.lo_ring_new_type.argtypes = [i, u64p, i, i]
    L.lo_ring_free.argtypes = [vp]
    L.lo_ring_roots_fwd.restype = u64p
    L.lo_ring_roots_fwd.argtypes = [vp, i]
    L.lo_ring_roots_bwd.restype = u64p
    L.lo_ring_roots_bwd.argtypes = [vp, i]
    L.lo_ring_constants.argtypes = [vp, i, u64p]
    L.lo_ring_rescale_constant.restype = u64
    L.lo_ring_rescale_constant.argtypes = [vp, i, i]
    L.lo_gen_moduli.restype = i
    L.lo_gen_moduli.argtypes = [i, C.POINTER(i), i, C.POINTER(i), i, u64p, u64p]
    for name in ("lo_ntt", "lo_ntt_lazy", "lo_intt", "lo_intt_lazy"):
        getattr(L, name).argtypes = [vp, i, u64p, u64p]
    L.lo_binop.argtypes = [vp, i, i, u64p, u64p, u64p]
    L.lo_unop.argtypes = [vp, i, i, u64p, u64p]
    L.lo_scalarop.argtypes = [vp, i, i, u64p, u64, u64p]
    L.lo_mul_rns_scalar_montgomery.argtypes = [vp, i, u64p, u64p, u64p]
    for name in ("lo_add_scalar_bigint", "lo_sub_scalar_bigint", "lo_mul_scalar_bigint"):
        getattr(L, name).argtypes = [vp, i, u64p, u64p, i, u64p]
    for name in ("lo_div_floor_by_last_modulus_ntt", "lo_div_floor_by_last_modulus",
                 "lo_div_round_by_last_modulus_ntt", "lo_div_round_by_last_modulus"):
        getattr(L, name).argtypes = [vp, i, u64p, u64p]
    for name in ("lo_div_round_by_last_modulus_many_ntt", "lo_div_round_by_last_modulus_many",
                 "lo_div_floor_by_last_modulus_many_ntt", "lo_div_floor_by_last_modulus_many"):
        getattr(L, name).argtypes = [vp, i, i, u64p, u64p]
    L.lo_automorphism_ntt_index.argtypes = [i, u64, u64, u64p]
    L.lo_automorphism_ntt_with_index.argtypes = [vp, i, u64p, u64p, u64p]
    L.lo_automorphism_ntt_with_index_then_add_lazy.argtypes = [vp, i, u64p, u64p, u64p]
    L.lo_automorphism.argtypes = [vp, i, u64p, u64, u64p]
    L.lo_basis_extender_new.restype = vp
    L.lo_basis_extender_new.argtypes = [vp, vp]
    L.lo_basis_extender_free.argtypes = [vp]
    L.lo_modup_q_to_p.argtypes = [vp, i, i, u64p, u64p]
    L.lo_modup_p_to_q.argtypes = [vp, i, i, u64p, u64p]
    for name in ("lo_moddown_qp_to_q", "lo_moddown_qp_to_q_ntt", "lo_moddown_qp_to_p"):
        getattr(L, name).argtypes = [vp, i, i, u64p, u64p, u64p]
    L.lo_decomposer_new.restype = vp
    L.lo_decomposer_new.argtypes = [vp, vp]
    L.lo_decomposer_free.argtypes = [vp]
    L.lo_decompose_and_split.argtypes = [vp, i, i, i, i, u64p, u64p, u64p]
    L.lo_evaluator_new.restype = vp
    L.lo_evaluator_new.argtypes = [vp, vp]
    L.lo_evaluator_free.argtypes = [vp]
    L.lo_base_rns_decomposition_vector_size.restype = i
    L.lo_base_rns_decomposition_vector_size.argtypes = [i, i]
    ep = C.POINTER(_Evk)
    L.lo_decompose_ntt.argtypes = [vp, i, i, i, u64p, i, u64p, u64p]
    L.lo_gadget_product_lazy.argtypes = [vp, i, u64p, ep, u64p, u64p]
    L.lo_gadget_product_hoisted_lazy.argtypes = [vp, i, u64p, u64p, ep, u64p, u64p]
    L.lo_gadget_product.argtypes = [vp, i, u64p, ep, u64p]
    L.lo_gadget_product_hoisted.argtypes = [vp, i, u64p, u64p, ep, u64p]
    L.lo_moddown_ntt.argtypes = [vp, i, i, u64p, u64p, u64p]
    L.lo_relinearize.argtypes = [vp, i, u64p, ep, u64p]
    L.lo_automorphism_ct.argtypes = [vp, i, u64p, u64, ep, u64p]
    L.lo_automorphism_hoisted.argtypes = [vp, i, u64p, u64p, u64p, u64, ep, u64p]
    L.lo_automorphism_hoisted_lazy.argtypes = [vp, i, u64p, u64p, u64p, u64, ep, u64p, u64p]
    L.lo_ckks_mul_relin.argtypes = [vp, i, u64p, u64p, ep, i, u64p]
    L.lo_bgv_mul_relin.argtypes = [vp, i, u64, u64p, u64p, ep, i, u64p]
    L.lo_rescale.argtypes = [vp, i, i, i, u64p, u64p]
    L.lo_bench_bgv_mul_relin.argtypes = [vp, i, u64, u64p, u64p, ep, i, C.c_double, u64p]
    L.lo_bench_bgv_mul_relin.restype = C.c_double
    L.lo_bench_op.argtypes = [vp, i, i, u64, u64, u64p, u64p, ep, i, C.c_double, i, i, u64p]
    L.lo_bench_op.restype = C.c_double
    L.lo_batch_op.argtypes = [vp, i, i, u64, u64, u64p, u64p, ep, i, i, u64p]


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(u64p)


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _brc(q):
    b = (C.c_uint64 * 2)()
    lib().lo_gen_bred_constant(q, b)
    return b


# ---- scalar helpers (ring/modular_reduction.go) ------------------------------
def GenBRedConstant(q):
    b = _brc(q)
    return [int(b[0]), int(b[1])]


def GenMRedConstant(q):
    return int(lib().lo_gen_mred_constant(q))


def MForm(a, q):
    return int(lib().lo_mform(a, q, _brc(q)))


def MFormLazy(a, q):
    return int(lib().lo_mform_lazy(a, q, _brc(q)))


def IMForm(a, q):
    return int(lib().lo_imform(a, q, GenMRedConstant(q)))


def MRed(x, y, q):
    return int(lib().lo_mred(x, y, q, GenMRedConstant(q)))


def MRedLazy(x, y, q):
    return int(lib().lo_mred_lazy(x, y, q, GenMRedConstant(q)))


def BRed(x, y, q):
    return int(lib().lo_bred(x, y, q, _brc(q)))


def BRedLazy(x, y, q):
    return int(lib().lo_bred_lazy(x, y, q, _brc(q)))


def BRedAdd(a, q):
    return int(lib().lo_bred_add(a, q, _brc(q)))


def BRedAddLazy(a, q):
    return int(lib().lo_bred_add_lazy(a, q, _brc(q)))


def ModExp(x, e, p):
    return int(lib().lo_modexp(x, e, p))


def IsPrime(x):
    return bool(lib().lo_is_prime(x))


def GenModuli(log_nth_root, logq, logp):
    """core/rlwe/params.go:811 -- returns (q, p) lists."""
    lq = (C.c_int * len(logq))(*logq)
    lp = (C.c_int * max(1, len(logp)))(*logp)
    q = (C.c_uint64 * len(logq))()
    p = (C.c_uint64 * max(1, len(logp)))()
    rc = lib().lo_gen_moduli(log_nth_root, lq, len(logq), lp, len(logp), q, p)
    if rc != 0:
        raise ValueError(lib().lo_last_error().decode())
    return [int(x) for x in q], [int(p[i]) for i in range(len(logp))]


def AutomorphismNTTIndex(N, nthroot, galel):
    idx = np.empty(N, dtype=np.uint64)
    lib().lo_automorphism_ntt_index(N, nthroot, galel, _p(idx))
    return idx


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


def _words(x: int):
    w = []
    while True:
        w.append(x & 0xFFFFFFFFFFFFFFFF)
        x >>= 64
        if x == 0:
            break
    return np.array(w, dtype=np.uint64)


class Ring:
    """Restated ring.Ring (ring/ring.go:71) -- standard (negacyclic) type only."""

    def __init__(self, N: int, moduli, conjugate_invariant: bool = False):
        self.N = N
        self.moduli = [int(m) for m in moduli]
        self.conjugate_invariant = conjugate_invariant
        arr = (C.c_uint64 * len(self.moduli))(*self.moduli)
        self._h = lib().lo_ring_new_type(N, arr, len(self.moduli), int(conjugate_invariant))
        if not self._h:
            raise ValueError(lib().lo_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().lo_ring_free(self._h)
            self._h = None

    def MaxLevel(self):
        return len(self.moduli) - 1

    def NthRoot(self):
        return (4 if self.conjugate_invariant else 2) * self.N  # ring/ring.go:254,261

    def NewPoly(self, level=None):
        level = self.MaxLevel() if level is None else level
        return np.zeros((level + 1, self.N), dtype=np.uint64)

    def constants(self, i):
        out = (C.c_uint64 * 7)()
        lib().lo_ring_constants(self._h, i, out)
        return dict(q=int(out[0]), qinv=int(out[1]), brc=(int(out[2]), int(out[3])), ninv=int(out[4]),
                    primroot=int(out[5]), mask=int(out[6]))

    def roots_forward(self, i):
        n = 2 * self.N if self.conjugate_invariant else self.N
        return np.ctypeslib.as_array(lib().lo_ring_roots_fwd(self._h, i), shape=(n,)).copy()

    def roots_backward(self, i):
        n = 2 * self.N if self.conjugate_invariant else self.N
        return np.ctypeslib.as_array(lib().lo_ring_roots_bwd(self._h, i), shape=(n,)).copy()

    def rescale_constant(self, j, i):
        return int(lib().lo_ring_rescale_constant(self._h, j, i))

    # -- NTT (ring/ntt.go:127-152)
    def _ntt(self, fn, p1, level):
        p1 = _c(p1)
        level = p1.shape[0] - 1 if level is None else level
        out = np.zeros_like(p1)
        getattr(lib(), fn)(self._h, level, _p(p1), _p(out))
        return out

    def NTT(self, p1, level=None):
        return self._ntt("lo_ntt", p1, level)

    def NTTLazy(self, p1, level=None):
        return self._ntt("lo_ntt_lazy", p1, level)

    def INTT(self, p1, level=None):
        return self._ntt("lo_intt", p1, level)

    def INTTLazy(self, p1, level=None):
        return self._ntt("lo_intt_lazy", p1, level)

    # -- coefficient-wise (ring/operations.go)
    def binop(self, name, p1, p2, p3=None, level=None):
        p1, p2 = _c(p1), _c(p2)
        level = p1.shape[0] - 1 if level is None else level
        out = np.zeros_like(p1) if p3 is None else _c(p3).copy()
        lib().lo_binop(self._h, level, BINOPS[name], _p(p1), _p(p2), _p(out))
        return out

    def unop(self, name, p1, level=None):
        p1 = _c(p1)
        level = p1.shape[0] - 1 if level is None else level
        out = np.zeros_like(p1)
        lib().lo_unop(self._h, level, UNOPS[name], _p(p1), _p(out))
        return out

    def scalarop(self, name, p1, scalar, p2=None, level=None):
        p1 = _c(p1)
        level = p1.shape[0] - 1 if level is None else level
        out = np.zeros_like(p1) if p2 is None else _c(p2).copy()
        lib().lo_scalarop(self._h, level, SCALAROPS[name], _p(p1), scalar, _p(out))
        return out

    def MulRNSScalarMontgomery(self, p1, scalar, level=None):
        p1 = _c(p1)
        level = p1.shape[0] - 1 if level is None else level
        sc = _c(scalar)
        out = np.zeros_like(p1)
        lib().lo_mul_rns_scalar_montgomery(self._h, level, _p(p1), _p(sc), _p(out))
        return out

    def _bigint(self, fn, p1, scalar, level):
        p1 = _c(p1)
        level = p1.shape[0] - 1 if level is None else level
        scalar = int(scalar)
        if scalar < 0:  # big.Int.Mod is Euclidean: a negative scalar is its non-negative residue in every limb
            m = 1
            for q in self.moduli[: level + 1]:
                m *= q
            scalar %= m
        w = _words(scalar)
        out = np.zeros_like(p1)
        getattr(lib(), fn)(self._h, level, _p(p1), _p(w), len(w), _p(out))
        return out

    def AddScalarBigint(self, p1, scalar, level=None):
        return self._bigint("lo_add_scalar_bigint", p1, scalar, level)

    def SubScalarBigint(self, p1, scalar, level=None):
        return self._bigint("lo_sub_scalar_bigint", p1, scalar, level)

    def MulScalarBigint(self, p1, scalar, level=None):
        return self._bigint("lo_mul_scalar_bigint", p1, scalar, level)

    # -- the remaining ring/operations.go methods, restated on top of the per-limb C primitives
    def _limb_ring(self, i):
        if not hasattr(self, "_limb_rings"):
            self._limb_rings = {}
        if i not in self._limb_rings:
            self._limb_rings[i] = Ring(self.N, [self.moduli[i]])
        return self._limb_rings[i]

    def _double(self, name, p1, scalar0, scalar1, p2=None):
        """Ring.{Add,Sub,Mul}DoubleRNSScalar[ThenAdd] (ring/operations.go:166-184, 249-268): scalar0[i] on the
        coefficients [0, N/2) of limb i, scalar1[i] on [N/2, N) (Ring.MulScalar* take the Montgomery form inside)"""
        p1 = _c(p1)
        out = np.zeros_like(p1) if p2 is None else _c(p2).copy()
        h = self.N >> 1
        for i in range(p1.shape[0]):
            r = self._limb_ring(i)
            acc = None if p2 is None else out[i:i + 1]
            a = r.scalarop(name, p1[i:i + 1], int(scalar0[i]), acc)
            b = r.scalarop(name, p1[i:i + 1], int(scalar1[i]), acc)
            out[i, :h], out[i, h:] = a[0, :h], b[0, h:]
        return out

    def AddDoubleRNSScalar(self, p1, scalar0, scalar1):
        return self._double("AddScalar", p1, scalar0, scalar1)

    def SubDoubleRNSScalar(self, p1, scalar0, scalar1):
        return self._double("SubScalar", p1, scalar0, scalar1)

    def MulDoubleRNSScalar(self, p1, scalar0, scalar1):
        return self._double("MulScalar", p1, scalar0, scalar1)

    def MulDoubleRNSScalarThenAdd(self, p1, scalar0, scalar1, p2):
        return self._double("MulScalarThenAdd", p1, scalar0, scalar1, p2)

    def MulScalarBigintThenAdd(self, p1, scalar, p2):
        """ring/operations.go:240"""
        p1 = _c(p1)
        out = _c(p2).copy()
        for i in range(p1.shape[0]):
            out[i] = self._limb_ring(i).scalarop("MulScalarThenAdd", p1[i:i + 1], int(scalar) % self.moduli[i], out[i:i + 1])[0]
        return out

    def EvalPolyScalar(self, p1s, scalar):
        """ring/operations.go:271: Horner in the ring"""
        out = _c(p1s[-1]).copy()
        for i in range(len(p1s) - 1, 0, -1):
            out = self.binop("Add", self.scalarop("MulScalar", out, scalar), p1s[i - 1])
        return out

    def Shift(self, p1, k):
        """ring/operations.go:279 -> utils.RotateSliceAllocFree (utils/slices.go:77): out[j] = in[(j + k) mod N]"""
        p1 = _c(p1)
        n = p1.shape[1]
        k %= n
        return np.concatenate([p1[:, k:], p1[:, :k]], axis=1)

    def MultByMonomial(self, p1, k):
        """ring/operations.go:307-359, loop for loop (incl. the q - 0 = q representative)"""
        p1 = _c(p1)
        N = self.N
        shift = (k + (N << 1)) % (N << 1)
        if shift == 0:
            return p1.copy()
        q = np.array(self.moduli[: p1.shape[0]], dtype=np.uint64)[:, None]
        tmpx = p1.copy() if shift < N else q - p1
        shift %= N
        out = np.empty_like(p1)
        out[:, :shift] = q - tmpx[:, N - shift:]
        out[:, shift:] = tmpx[:, : N - shift]
        return out

    def MulByVectorMontgomery(self, p1, vector, p2=None):
        """ring/operations.go:363 (p2 given: MulByVectorMontgomeryThenAddLazy :370)"""
        p1 = _c(p1)
        v = np.ascontiguousarray(np.broadcast_to(_c(vector), p1.shape))
        if p2 is None:
            return self.binop("MulCoeffsMontgomery", p1, v)
        return self.binop("MulCoeffsMontgomeryThenAddLazy", p1, v, p2)

    def AutomorphismNTT(self, pin, galel):
        """ring/automorphism.go:38"""
        return self.AutomorphismNTTWithIndex(pin, self.AutomorphismNTTIndex(galel))

    # -- rescale (ring/scaling.go); input has level+1 limbs, output level+1-nb limbs
    def _div(self, fn, p0, nb=None):
        p0 = _c(p0)
        level = p0.shape[0] - 1
        out = np.zeros_like(p0)
        if nb is None:
            getattr(lib(), fn)(self._h, level, _p(p0), _p(out))
            return out[:level]
        getattr(lib(), fn)(self._h, level, nb, _p(p0), _p(out))
        return out[: level + 1 - nb]

    def DivFloorByLastModulusNTT(self, p0):
        return self._div("lo_div_floor_by_last_modulus_ntt", p0)

    def DivFloorByLastModulus(self, p0):
        return self._div("lo_div_floor_by_last_modulus", p0)

    def DivRoundByLastModulusNTT(self, p0):
        return self._div("lo_div_round_by_last_modulus_ntt", p0)

    def DivRoundByLastModulus(self, p0):
        return self._div("lo_div_round_by_last_modulus", p0)

    def DivRoundByLastModulusManyNTT(self, nb, p0):
        return self._div("lo_div_round_by_last_modulus_many_ntt", p0, nb)

    def DivRoundByLastModulusMany(self, nb, p0):
        return self._div("lo_div_round_by_last_modulus_many", p0, nb)

    def DivFloorByLastModulusManyNTT(self, nb, p0):
        return self._div("lo_div_floor_by_last_modulus_many_ntt", p0, nb)

    def DivFloorByLastModulusMany(self, nb, p0):
        return self._div("lo_div_floor_by_last_modulus_many", p0, nb)

    # -- automorphism (ring/automorphism.go)
    def AutomorphismNTTIndex(self, galel):
        return AutomorphismNTTIndex(self.N, self.NthRoot(), galel)

    def AutomorphismNTTWithIndex(self, pin, index):
        pin = _c(pin)
        out = np.zeros_like(pin)
        lib().lo_automorphism_ntt_with_index(self._h, pin.shape[0] - 1, _p(pin), _p(_c(index)), _p(out))
        return out

    def AutomorphismNTTWithIndexThenAddLazy(self, pin, index, pout):
        pin = _c(pin)
        out = _c(pout).copy()
        lib().lo_automorphism_ntt_with_index_then_add_lazy(self._h, pin.shape[0] - 1, _p(pin), _p(_c(index)), _p(out))
        return out

    def Automorphism(self, pin, galel):
        pin = _c(pin)
        out = np.zeros_like(pin)
        lib().lo_automorphism(self._h, pin.shape[0] - 1, _p(pin), galel, _p(out))
        return out


class BasisExtender:
    """Restated ring.BasisExtender (ring/basis_extension.go:14)."""

    def __init__(self, ringQ: Ring, ringP: Ring):
        self.ringQ, self.ringP = ringQ, ringP
        self._h = lib().lo_basis_extender_new(ringQ._h, ringP._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().lo_basis_extender_free(self._h)
            self._h = None

    def ModUpQtoP(self, levelQ, levelP, polQ):
        polQ = _c(polQ)
        out = np.zeros((levelP + 1, self.ringQ.N), dtype=np.uint64)
        lib().lo_modup_q_to_p(self._h, levelQ, levelP, _p(polQ), _p(out))
        return out

    def ModUpPtoQ(self, levelP, levelQ, polP):
        polP = _c(polP)
        out = np.zeros((levelQ + 1, self.ringQ.N), dtype=np.uint64)
        lib().lo_modup_p_to_q(self._h, levelP, levelQ, _p(polP), _p(out))
        return out

    def _md(self, fn, levelQ, levelP, p1Q, p1P, nout):
        p1Q, p1P = _c(p1Q), _c(p1P)
        out = np.zeros((nout, self.ringQ.N), dtype=np.uint64)
        getattr(lib(), fn)(self._h, levelQ, levelP, _p(p1Q), _p(p1P), _p(out))
        return out

    def ModDownQPtoQ(self, levelQ, levelP, p1Q, p1P):
        return self._md("lo_moddown_qp_to_q", levelQ, levelP, p1Q, p1P, levelQ + 1)

    def ModDownQPtoQNTT(self, levelQ, levelP, p1Q, p1P):
        return self._md("lo_moddown_qp_to_q_ntt", levelQ, levelP, p1Q, p1P, levelQ + 1)

    def ModDownQPtoP(self, levelQ, levelP, p1Q, p1P):
        return self._md("lo_moddown_qp_to_p", levelQ, levelP, p1Q, p1P, levelP + 1)


class Decomposer:
    """Restated ring.Decomposer (ring/basis_extension.go:313)."""

    def __init__(self, ringQ: Ring, ringP: Ring):
        self.ringQ, self.ringP = ringQ, ringP
        self._h = lib().lo_decomposer_new(ringQ._h, ringP._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().lo_decomposer_free(self._h)
            self._h = None

    def DecomposeAndSplit(self, levelQ, levelP, nbPi, digit, p0Q):
        p0Q = _c(p0Q)
        N = self.ringQ.N
        p1Q = np.zeros((levelQ + 1, N), dtype=np.uint64)
        p1P = np.zeros((levelP + 1, N), dtype=np.uint64)
        lib().lo_decompose_and_split(self._h, levelQ, levelP, nbPi, digit, _p(p0Q), _p(p1Q), _p(p1P))
        return p1Q, p1P


class EvaluationKey:
    """GadgetCiphertext with BaseTwoDecomposition = 0 (core/rlwe/gadgetciphertext.go:19).

    q: [beta, 2, nQk, N], p: [beta, 2, nPk, N] -- NTT + Montgomery form."""

    def __init__(self, q: np.ndarray, p: np.ndarray, pw2: int = 0, nj=None):
        self.q, self.p = _c(q), _c(p)
        assert self.q.ndim == 4 and self.p.ndim == 4 and self.q.shape[:2] == self.p.shape[:2]
        self.pw2 = pw2
        self.nj = list(nj) if nj is not None else [1] * 64
        assert pw2 == 0 or sum(self.nj) == self.q.shape[0]
        arr = (C.c_int * 64)(*(self.nj + [0] * (64 - len(self.nj))))
        self._s = _Evk(self.q.shape[0], self.q.shape[2], self.p.shape[2], _p(self.q), _p(self.p), pw2, arr)

    def LevelQ(self):
        return self.q.shape[2] - 1

    def LevelP(self):
        return self.p.shape[2] - 1

    def ref(self):
        return C.byref(self._s)


def BaseRNSDecompositionVectorSize(levelQ, levelP):
    return int(lib().lo_base_rns_decomposition_vector_size(levelQ, levelP))


class Evaluator:
    """Restated rlwe.Evaluator hot path (core/rlwe/evaluator*.go), NTT-domain ciphertexts."""

    def __init__(self, ringQ: Ring, ringP: Ring | None):
        """ringP = None: parameters without special primes (levelP = -1; base-2 gadget keys only, as the reference's own
        P-less test set, core/rlwe/test_params.go:36-46)"""
        self.ringQ, self.ringP = ringQ, ringP
        self._h = lib().lo_evaluator_new(ringQ._h, ringP._h if ringP is not None else None)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().lo_evaluator_free(self._h)
            self._h = None

    def DecomposeNTT(self, levelQ, levelP, nbPi, c2, c2_is_ntt=True):
        c2 = _c(c2)
        N = self.ringQ.N
        beta = BaseRNSDecompositionVectorSize(levelQ, levelP)
        dq = np.zeros((beta, levelQ + 1, N), dtype=np.uint64)
        dp = np.zeros((beta, levelP + 1, N), dtype=np.uint64)
        lib().lo_decompose_ntt(self._h, levelQ, levelP, nbPi, _p(c2), int(c2_is_ntt), _p(dq), _p(dp))
        return dq, dp

    def GadgetProductLazy(self, levelQ, cx, evk: EvaluationKey):
        cx = _c(cx)
        N = self.ringQ.N
        ctQ = np.zeros((2, levelQ + 1, N), dtype=np.uint64)
        ctP = np.zeros((2, evk.LevelP() + 1, N), dtype=np.uint64)
        lib().lo_gadget_product_lazy(self._h, levelQ, _p(cx), evk.ref(), _p(ctQ), _p(ctP))
        return ctQ, ctP

    def GadgetProductHoistedLazy(self, levelQ, dq, dp, evk: EvaluationKey):
        N = self.ringQ.N
        ctQ = np.zeros((2, levelQ + 1, N), dtype=np.uint64)
        ctP = np.zeros((2, evk.LevelP() + 1, N), dtype=np.uint64)
        lib().lo_gadget_product_hoisted_lazy(self._h, levelQ, _p(_c(dq)), _p(_c(dp)), evk.ref(), _p(ctQ), _p(ctP))
        return ctQ, ctP

    def GadgetProduct(self, levelQ, cx, evk: EvaluationKey):
        cx = _c(cx)
        levelQ = min(levelQ, evk.LevelQ())
        ct = np.zeros((2, levelQ + 1, self.ringQ.N), dtype=np.uint64)
        lib().lo_gadget_product(self._h, levelQ, _p(cx), evk.ref(), _p(ct))
        return ct

    def GadgetProductHoisted(self, levelQ, dq, dp, evk: EvaluationKey):
        ct = np.zeros((2, levelQ + 1, self.ringQ.N), dtype=np.uint64)
        lib().lo_gadget_product_hoisted(self._h, levelQ, _p(_c(dq)), _p(_c(dp)), evk.ref(), _p(ct))
        return ct

    def ModDown(self, levelQ, levelP, ctQ, ctP):
        ct = np.zeros((2, levelQ + 1, self.ringQ.N), dtype=np.uint64)
        lib().lo_moddown_ntt(self._h, levelQ, levelP, _p(_c(ctQ)), _p(_c(ctP)), _p(ct))
        return ct

    def Relinearize(self, ct_in, rlk: EvaluationKey):
        ct_in = _c(ct_in)
        level = ct_in.shape[1] - 1
        out = np.zeros((2, level + 1, self.ringQ.N), dtype=np.uint64)
        lib().lo_relinearize(self._h, level, _p(ct_in), rlk.ref(), _p(out))
        return out

    def Automorphism(self, ct_in, galel, gk: EvaluationKey):
        ct_in = _c(ct_in)
        level = ct_in.shape[1] - 1
        out = np.zeros_like(ct_in)
        lib().lo_automorphism_ct(self._h, level, _p(ct_in), galel, gk.ref(), _p(out))
        return out

    def AutomorphismHoisted(self, ct_in, dq, dp, galel, gk: EvaluationKey):
        ct_in = _c(ct_in)
        level = ct_in.shape[1] - 1
        out = np.zeros_like(ct_in)
        lib().lo_automorphism_hoisted(self._h, level, _p(ct_in), _p(_c(dq)), _p(_c(dp)), galel, gk.ref(), _p(out))
        return out

    def AutomorphismHoistedLazy(self, levelQ, ct_in0, dq, dp, galel, gk: EvaluationKey):
        N = self.ringQ.N
        outQ = np.zeros((2, levelQ + 1, N), dtype=np.uint64)
        outP = np.zeros((2, gk.LevelP() + 1, N), dtype=np.uint64)
        lib().lo_automorphism_hoisted_lazy(self._h, levelQ, _p(_c(ct_in0)), _p(_c(dq)), _p(_c(dp)), galel, gk.ref(),
                                           _p(outQ), _p(outP))
        return outQ, outP

    # scheme glue
    def CKKSMulRelin(self, op0, op1, rlk: EvaluationKey | None, relin: bool):
        op0, op1 = _c(op0), _c(op1)
        level = op0.shape[1] - 1
        out = np.zeros((2 if relin else 3, level + 1, self.ringQ.N), dtype=np.uint64)
        lib().lo_ckks_mul_relin(self._h, level, _p(op0), _p(op1), rlk.ref() if rlk else None, int(relin), _p(out))
        return out

    def BGVMulRelin(self, t, op0, op1, rlk: EvaluationKey | None, relin: bool):
        op0, op1 = _c(op0), _c(op1)
        level = op0.shape[1] - 1
        out = np.zeros((2 if relin else 3, level + 1, self.ringQ.N), dtype=np.uint64)
        lib().lo_bgv_mul_relin(self._h, level, t, _p(op0), _p(op1), rlk.ref() if rlk else None, int(relin), _p(out))
        return out

    def BenchBGVMulRelin(self, t, op0, op1, rlk: EvaluationKey, nthreads: int, seconds: float):
        """bench.py's cpu_baseline leg: `nthreads` OS threads (a C-level pthread loop, no Python in the timed region) each
        repeat BGVMulRelin on this evaluator until `seconds` have passed; returns (ops finished, elapsed seconds)."""
        op0, op1 = _c(op0), _c(op1)
        level = op0.shape[1] - 1
        counts = np.zeros(nthreads, dtype=np.uint64)
        dt = lib().lo_bench_bgv_mul_relin(self._h, level, t, _p(op0), _p(op1), rlk.ref(), nthreads, float(seconds), _p(counts))
        return int(counts.sum()), float(dt)

    def BenchOp(self, kind: str, op0, op1=None, key: EvaluationKey | None = None, t: int = 0, gal: int = 0, nthreads: int = 1,
                seconds: float = 1.0, pin: bool = True, private_copy: bool = True):
        """The generalised timed loop (lo_bench_op): kind "bgv_mulrelin" | "rotate" | "ckks_mul_rescale"; returns (ops, seconds)."""
        kinds = {"bgv_mulrelin": 0, "rotate": 1, "ckks_mul_rescale": 2}
        op0 = _c(op0)
        op1 = _c(op1) if op1 is not None else None
        level = op0.shape[1] - 1
        counts = np.zeros(nthreads, dtype=np.uint64)
        dt = lib().lo_bench_op(self._h, kinds[kind], level, t, gal, _p(op0), _p(op1) if op1 is not None else None,
                               key.ref() if key else None, nthreads, float(seconds), int(pin), int(private_copy), _p(counts))
        return int(counts.sum()), float(dt)

    def BatchOp(self, kind: str, op0, op1=None, key: EvaluationKey | None = None, t: int = 0, gal: int = 0, nthreads: int = 0):
        """Every entry of a batch through one operation (lo_batch_op): op0 / op1 [B][2][L][N] -> [B][2][L][N] ("bgv_mulrelin",
        "rotate") or [B][3][L-1][N] ("ckks_mul_rescale"), entries dealt to `nthreads` OS threads (default: the CPUs this process
        may use).  The checker for whole timed batches (bench.py, tests/test_gpu_headline.py)."""
        kinds = {"bgv_mulrelin": 0, "rotate": 1, "ckks_mul_rescale": 2}
        op0 = _c(op0)
        op1 = _c(op1) if op1 is not None else None
        B, level = op0.shape[0], op0.shape[2] - 1
        if nthreads <= 0:
            try:
                nthreads = len(os.sched_getaffinity(0))
            except (AttributeError, OSError):
                nthreads = os.cpu_count() or 1
            nthreads = min(nthreads, 32)
        shape = (B, 3, level, self.ringQ.N) if kind == "ckks_mul_rescale" else (B, 2, level + 1, self.ringQ.N)
        out = np.zeros(shape, dtype=np.uint64)
        lib().lo_batch_op(self._h, kinds[kind], level, t, gal, _p(op0), _p(op1) if op1 is not None else None,
                          key.ref() if key else None, B, nthreads, _p(out))
        return out

    def Rescale(self, ct, nb=1):
        ct = _c(ct)
        degree, level = ct.shape[0] - 1, ct.shape[1] - 1
        out = np.zeros((degree + 1, level + 1 - nb, self.ringQ.N), dtype=np.uint64)
        lib().lo_rescale(self.ringQ._h, level, degree, nb, _p(ct), _p(out))
        return out

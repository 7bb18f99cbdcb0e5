"""ctypes binding of libhering.so (include/hering.h).  Fails loudly when the HIP
extension is missing: there is no Python or CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("HERING_LIB") or os.path.join(_HERE, "libhering.so")  # HERING_LIB: A/B-test another build
_INC = os.path.join(os.path.dirname(_HERE), "include")
_HDRS = [os.path.join(_INC, "hering.h"), os.path.join(_INC, "hering_debug.h")]

H = C.c_uint64
u64p = C.POINTER(C.c_uint64)


class HeringError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libhering error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    return _SO


def declared_symbols() -> list[str]:
    """Every function include/*.h declares (used by the CPU-side export test)."""
    src = "".join(open(h).read() for h in _HDRS)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(he_[a-z0-9_]+)\s*\(", src)))


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise HeringError(-3, f"{_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


def check(rc: int):
    if rc != 0:
        raise HeringError(rc, load().he_last_error().decode())


def _declare(L):
    i, sz = C.c_int, C.c_size_t
    HP = C.POINTER(H)
    L.he_last_error.restype = C.c_char_p
    L.he_version.restype = C.c_char_p
    L.he_prof_kernel_name.restype = C.c_char_p
    L.he_prof_kernel_name.argtypes = [C.c_int]
    sig = {
        "he_device_count": [C.POINTER(i)], "he_debug_device_pci_bus_id": [i, C.c_char_p, i], "he_ctx_create": [i, HP], "he_ctx_destroy": [H], "he_ctx_sync": [H], "he_timer_start": [H],
        "he_timer_stop": [H, C.POINTER(C.c_float)], "he_device_info": [H, u64p],
        "he_ring_create": [H, i, u64p, i, HP], "he_ring_create_type": [H, i, i, u64p, i, HP], "he_ring_destroy": [H], "he_ring_constant": [H, i, i, u64p],
        "he_ring_roots": [H, i, i, u64p],
        "he_poly_alloc": [H, i, i, HP], "he_poly_alloc_scratch": [H, i, i, HP], "he_poly_free": [H],
        "he_poly_shape": [H, C.POINTER(i), C.POINTER(i), C.POINTER(i)],
        "he_poly_upload": [H, u64p, sz], "he_poly_download": [H, u64p, sz],
        "he_poly_upload_limb": [H, i, i, u64p], "he_poly_download_limb": [H, i, i, u64p],
        "he_poly_copy": [H, H, i], "he_poly_copy_batch": [H, i, H, i, i, i], "he_poly_zero": [H],
        "he_ntt": [H, i, H, H], "he_ntt_lazy": [H, i, H, H], "he_intt": [H, i, H, H], "he_intt_lazy": [H, i, H, H],
        "he_subring_ntt_host": [H, i, i, i, u64p, u64p],
        "he_binop": [H, i, i, H, H, H], "he_unop": [H, i, i, H, H], "he_scalarop": [H, i, i, H, C.c_uint64, H],
        "he_mul_rns_scalar_montgomery": [H, i, H, u64p, H],
        "he_add_scalar_bigint": [H, i, H, u64p, i, H], "he_sub_scalar_bigint": [H, i, H, u64p, i, H],
        "he_mul_scalar_bigint": [H, i, H, u64p, i, H], "he_mul_scalar_bigint_then_add": [H, i, H, u64p, i, H],
        "he_double_rns_scalarop": [H, i, i, H, u64p, u64p, H], "he_shift": [H, i, H, i, H],
        "he_mult_by_monomial": [H, i, H, i, H], "he_mul_by_vector_montgomery": [H, i, H, H, i, H],
        "he_add": [H, i, H, H, H], "he_sub": [H, i, H, H, H], "he_neg": [H, i, H, H], "he_reduce": [H, i, H, H],
        "he_mform": [H, i, H, H], "he_imform": [H, i, H, H],
        "he_mul_coeffs_montgomery": [H, i, H, H, H], "he_mul_coeffs_montgomery_then_add": [H, i, H, H, H],
        "he_mul_coeffs_montgomery_lazy": [H, i, H, H, H],
        "he_mul_coeffs_montgomery_lazy_then_add_lazy": [H, i, H, H, H],
        "he_div_round_by_last_modulus_ntt": [H, i, H, H], "he_div_round_by_last_modulus": [H, i, H, H],
        "he_div_floor_by_last_modulus_ntt": [H, i, H, H], "he_div_floor_by_last_modulus": [H, i, H, H],
        "he_div_round_by_last_modulus_many_ntt": [H, i, i, H, H], "he_div_round_by_last_modulus_many": [H, i, i, H, H],
        "he_div_floor_by_last_modulus_many_ntt": [H, i, i, H, H], "he_div_floor_by_last_modulus_many": [H, i, i, H, H],
        "he_rescale_polys": [H, i, i, i, HP, HP],
        "he_automorphism_index_create": [H, C.c_uint64, HP], "he_automorphism_index_destroy": [H],
        "he_automorphism_index_download": [H, u64p],
        "he_automorphism_ntt_with_index": [H, i, H, H, H],
        "he_automorphism_ntt_with_index_then_add_lazy": [H, i, H, H, H],
        "he_automorphism": [H, i, H, C.c_uint64, H],
        "he_basis_extender_create": [H, H, HP], "he_basis_extender_destroy": [H],
        "he_modup_q_to_p": [H, i, i, H, H], "he_modup_p_to_q": [H, i, i, H, H],
        "he_moddown_qp_to_q": [H, i, i, H, H, H], "he_moddown_qp_to_q_ntt": [H, i, i, H, H, H],
        "he_moddown_qp_to_p": [H, i, i, H, H, H],
        "he_eval_moddown_qp_to_q_ntt": [H, i, i, H, H, H], "he_evaluator_create": [H, H, HP], "he_evaluator_destroy": [H],
        "he_evk_create": [H, i, i, i, u64p, u64p, HP], "he_evk_destroy": [H],
        "he_evk_device_buffer": [H, C.POINTER(C.c_void_p), C.POINTER(sz)], "he_evk_commit": [H], "he_evk_download": [H, u64p, sz],
        "he_evk_create_base2": [H, i, C.POINTER(i), i, i, i, u64p, u64p, HP],
        "he_decompose_and_split": [H, i, i, i, i, H, H, H],
        "he_decomp_create": [H, i, HP], "he_decomp_destroy": [H],
        "he_decomp_download_limb": [H, i, i, i, i, u64p],
        "he_decompose_ntt": [H, i, i, i, H, i, H],
        "he_gadget_product_lazy": [H, i, H, H, H, H, H, H],
        "he_gadget_product_hoisted_lazy": [H, i, H, H, H, H, H, H],
        "he_gadget_product_hoisted_lazy_digits": [H, i, H, H, i, i, H, H, H, H],
        "he_poly_device_buffer": [H, C.POINTER(C.c_void_p), C.POINTER(sz)],
        "he_moddown": [H, i, i, H, H, H, H, H, H],
        "he_gadget_product": [H, i, H, H, H, H], "he_gadget_product_hoisted": [H, i, H, H, H, H],
        "he_relinearize": [H, i, H, H, H, H, H, H],
        "he_automorphism_ct": [H, i, H, H, C.c_uint64, H, H, H],
        "he_automorphism_hoisted": [H, i, H, H, C.c_uint64, H, H, H],
        "he_automorphism_hoisted_lazy": [H, i, H, H, C.c_uint64, H, H, H, H, H],
        "he_centered_lift": [H, i, H, i, i, H, i, H],
        "he_decomp_fill": [H, i, i, H, H],
        "he_lintrans_mul_sum": [H, i, i, i, HP, HP, HP, HP, HP, HP, HP, i, H, H, H, H],
        "he_lintrans_giant_step": [H, i, H, H, C.c_uint64, H, H, H, H, H, H, i],
        "he_ckks_mul_relin": [H, i, H, H, H, H, H, H, H, H],
        "he_bgv_mul_relin": [H, i, C.c_uint64, H, H, H, H, H, H, H, H],
        "he_probe_modmul": [H, i, C.POINTER(C.c_double)], "he_probe_modmul_f64": [H, i, C.POINTER(C.c_double)],
        "he_prof_begin": [H], "he_prof_end": [H, i, C.POINTER(i), C.POINTER(C.c_float), C.POINTER(i)],
        "he_prof_end_bytes": [H, i, C.POINTER(i), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(i)],
        "he_alg_bytes": [H, i, C.POINTER(C.c_double)], "he_alg_valu": [H, i, C.POINTER(C.c_double)],
        "he_graph_begin": [H], "he_graph_end": [H, HP], "he_graph_launch": [H], "he_graph_nodes": [H, C.POINTER(i)],
        "he_graph_destroy": [H],
        "he_rccl_available": [C.POINTER(i)], "he_rccl_unique_id": [C.POINTER(C.c_uint8)], "he_rccl_comm_create": [H, C.POINTER(C.c_uint8), i, i, HP],
        "he_rccl_comm_destroy": [H], "he_rccl_comm_ranks": [H, C.POINTER(i)], "he_evk_broadcast": [H, H, i],
        "he_poly_all_reduce_sum": [H, H],
        "he_evaluator_set_coalescing": [H, i, i], "he_evaluator_coalescing_stats": [H, u64p],
        "he_ctx_set_coalescing": [H, i, i], "he_ctx_set_deferred": [H, i], "he_ctx_coalescing_stats": [H, u64p],
        "he_debug_queue_counters": [H, u64p], "he_debug_queue_op_stats": [H, u64p], "he_debug_queue_inject_failure": [H],
        "he_debug_concurrent_mul_relin": [i, i, i, i, i, C.c_uint64, HP, HP, HP, HP, HP, HP, HP, HP, HP, C.POINTER(C.c_double)],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    if os.environ.get("HERING_CALL_STATS"):  # diagnosis: histogram of the ABI calls of a run, written at exit (tools/)
        import atexit
        import collections
        import json
        counts = collections.Counter()

        def wrap(name, fn):
            def call(*a):
                counts[name] += 1
                return fn(*a)
            return call
        for name in sig:
            setattr(L, name, wrap(name, getattr(L, name)))
        atexit.register(lambda: json.dump(dict(counts.most_common()), open(os.environ["HERING_CALL_STATS"], "w"), indent=1))


# ---- recording of the ABI calls of a run (diagnostics: the program of he_debug_replay, include/hering_debug.h) ----------------
# name -> (function number of csrc/replay.cpp, argument kinds): h handle, i immediate, O created handle (byref), A u64 array,
# H handle array; "+n" appends the immediate n (a variant selector the replayer's entry takes)
_TRACE_FNS = {
    "he_poly_alloc": (0, "hiiO"), "he_poly_alloc_scratch": (1, "hiiO"), "he_poly_free": (2, "h"), "he_poly_copy": (3, "hhi"),
    "he_poly_copy_batch": (4, "hihiii"), "he_poly_zero": (5, "h"),
    "he_ntt": (6, "hihh"), "he_ntt_lazy": (7, "hihh"), "he_intt": (8, "hihh"), "he_intt_lazy": (9, "hihh"),
    "he_binop": (10, "hiihhh"), "he_unop": (11, "hiihh"), "he_scalarop": (12, "hiihih"),
    "he_mul_rns_scalar_montgomery": (13, "hihAh"), "he_add_scalar_bigint": (14, "hihAih"), "he_sub_scalar_bigint": (15, "hihAih"),
    "he_mul_scalar_bigint": (16, "hihAih"), "he_mul_scalar_bigint_then_add": (17, "hihAih"), "he_double_rns_scalarop": (18, "hiihAAh"),
    "he_shift": (19, "hihih"), "he_mult_by_monomial": (20, "hihih"), "he_mul_by_vector_montgomery": (21, "hihhih"),
    "he_div_round_by_last_modulus_many_ntt": (22, "hiihh+3"), "he_div_round_by_last_modulus_many": (22, "hiihh+1"),
    "he_div_floor_by_last_modulus_many_ntt": (22, "hiihh+2"), "he_div_floor_by_last_modulus_many": (22, "hiihh+0"),
    "he_rescale_polys": (23, "hiiiHH"), "he_automorphism_index_create": (24, "hiO"), "he_automorphism_index_destroy": (25, "h"),
    "he_automorphism_ntt_with_index": (26, "hihhh"), "he_automorphism_ntt_with_index_then_add_lazy": (27, "hihhh"),
    "he_automorphism": (28, "hihih"), "he_modup_q_to_p": (29, "hiihh"), "he_modup_p_to_q": (30, "hiihh"),
    "he_moddown_qp_to_q": (31, "hiihhh+0"), "he_moddown_qp_to_q_ntt": (31, "hiihhh+1"), "he_moddown_qp_to_p": (31, "hiihhh+2"),
    "he_eval_moddown_qp_to_q_ntt": (32, "hiihhh"), "he_decomp_create": (33, "hiO"), "he_decomp_destroy": (34, "h"),
    "he_decompose_ntt": (35, "hiiihih"), "he_gadget_product_lazy": (36, "hihhhhhh"), "he_gadget_product_hoisted_lazy": (37, "hihhhhhh"),
    "he_moddown": (38, "hiihhhhhh"), "he_gadget_product": (39, "hihhhh"), "he_gadget_product_hoisted": (40, "hihhhh"),
    "he_relinearize": (41, "hihhhhhh"), "he_automorphism_ct": (42, "hihhihhh"), "he_automorphism_hoisted": (43, "hihhihhh"),
    "he_automorphism_hoisted_lazy": (44, "hihhihhhhh"), "he_centered_lift": (45, "hihiihih"), "he_decomp_fill": (46, "hiihh"),
    "he_lintrans_mul_sum": (47, "hiiiHHHHHHHihhhh"), "he_ckks_mul_relin": (48, "hihhhhhhhh"), "he_bgv_mul_relin": (49, "hiihhhhhhhh"),
    "he_lintrans_giant_step": (50, "hihhihhhhhhi"),
}
# length of the arrays of a call: (function, argument index) -> index of the argument holding it (+1 for "level" arguments)
_TRACE_LEN = {("he_mul_rns_scalar_montgomery", 3): (1, 1), ("he_add_scalar_bigint", 3): (4, 0), ("he_sub_scalar_bigint", 3): (4, 0),
              ("he_mul_scalar_bigint", 3): (4, 0), ("he_mul_scalar_bigint_then_add", 3): (4, 0), ("he_double_rns_scalarop", 4): (1, 1),
              ("he_double_rns_scalarop", 5): (1, 1), ("he_rescale_polys", 4): (3, 0), ("he_rescale_polys", 5): (3, 0)}
_TRACE_LEN.update({("he_lintrans_mul_sum", k): (3, 0) for k in range(4, 11)})
# calls a replay has no use for: they read, wait or account, and do not change what the replayed calls see
_TRACE_IGNORE = {"he_ctx_sync", "he_last_error", "he_alg_bytes", "he_timer_start", "he_timer_stop", "he_poly_download", "he_poly_shape",
                 "he_poly_download_limb", "he_prof_begin", "he_prof_end", "he_prof_end_bytes", "he_ctx_coalescing_stats",
                 "he_evaluator_coalescing_stats", "he_version", "he_device_info", "he_ring_constant", "he_ring_roots", "he_decomp_download_limb"}
_trace = None


def _val(x):
    v = getattr(x, "value", x)
    return int(v) & 0xFFFFFFFFFFFFFFFF


def trace_begin():
    """Start recording every ABI call this process makes (the single-threaded run of a driver) as a program for he_debug_replay.
    Calls outside the replayer's table (uploads, object creation other than polynomials / hoisting buffers / index tables) make
    trace_end fail: the recorded window must be the steady-state part of a run."""
    global _trace
    L = load()
    if _trace is not None:
        raise RuntimeError("a trace is already being recorded")
    _trace = {"words": [], "bad": [], "orig": {}}
    import inspect  # noqa: F401

    def wrap(name, fn):
        spec = _TRACE_FNS.get(name)

        def call(*a):
            rc = fn(*a)
            if spec is None:
                if name not in _TRACE_IGNORE:
                    _trace["bad"].append(name)
                return rc
            fid, kinds = spec
            extra = None
            if "+" in kinds:
                kinds, extra = kinds.split("+")
            w = [fid, len(kinds) + (1 if extra is not None else 0)]
            for k, (kind, x) in enumerate(zip(kinds, a)):
                if kind == "h":
                    v = _val(x)
                    w += [1, v] if v else [5]
                elif kind == "i":
                    w += [0, _val(x)]
                elif kind == "O":
                    w += [2, int(x._obj.value)]
                else:
                    src, plus = _TRACE_LEN[(name, k)]
                    n = int(_val(a[src])) + plus
                    if x is None or n <= 0:
                        w += [5] if x is None else [3 if kind == "A" else 4, 0]
                    else:
                        w += [3 if kind == "A" else 4, n] + [int(x[j]) & 0xFFFFFFFFFFFFFFFF for j in range(n)]
            if extra is not None:
                w += [0, int(extra)]
            _trace["words"] += w
            return rc
        return call

    for name in declared_symbols():
        if hasattr(L, name) and name not in ("he_last_error", "he_version", "he_prof_kernel_name", "he_debug_replay"):
            _trace["orig"][name] = getattr(L, name)
            setattr(L, name, wrap(name, _trace["orig"][name]))


def trace_end():
    """Stop recording; returns the program (numpy uint64 words) of the calls made since trace_begin."""
    global _trace
    import numpy as np
    L = load()
    t, _trace = _trace, None
    for name, fn in t["orig"].items():
        setattr(L, name, fn)
    if t["bad"]:
        raise RuntimeError(f"calls the replayer does not know were made while recording: {sorted(set(t['bad']))}")
    return np.array(t["words"], dtype=np.uint64)


def replay(ctx_handle, program, n_threads: int, rounds: int, subst_from, subst_to, watch):
    """he_debug_replay: returns (wall seconds, [n_threads][len(watch)] handles of the watched results of the last round)."""
    import numpy as np
    L = load()
    L.he_debug_replay.argtypes = [H, u64p, C.c_size_t, C.c_int, C.c_int, u64p, C.c_int, u64p, u64p, C.c_int, u64p, C.POINTER(C.c_double),
                                  C.c_char_p, C.c_size_t]
    L.he_debug_replay.restype = C.c_int
    program = np.ascontiguousarray(program, dtype=np.uint64)
    sf = np.ascontiguousarray(subst_from, dtype=np.uint64)
    st = np.ascontiguousarray(subst_to, dtype=np.uint64).reshape(n_threads, len(sf))
    wt = np.ascontiguousarray(watch, dtype=np.uint64)
    out = np.zeros((n_threads, max(len(wt), 1)), dtype=np.uint64)
    wall = C.c_double()
    err = C.create_string_buffer(512)
    p = lambda a: a.ctypes.data_as(u64p)
    rc = L.he_debug_replay(ctx_handle, p(program), program.size, n_threads, rounds, p(sf), len(sf), p(st), p(wt), len(wt), p(out),
                           C.byref(wall), err, 512)
    if rc != 0:
        raise HeringError(rc, err.value.decode() or "he_debug_replay failed")
    return float(wall.value), out[:, : len(wt)]

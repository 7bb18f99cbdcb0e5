"""The compiled host mirror include/hering.hpp (C++17, header-only: the reference's ring.Ring / ring.BasisExtender / rlwe.Evaluator
method names over the C ABI) and its parity program tests/cpp/parity.cpp.

CPU side: the program compiles with g++ -Wall -Wextra against the header -- i.e. every wrapper's call of the C ABI type-checks --
links both in-tree shared objects, and without a device fails loudly (HE_EDEVICE, "no CPU fallback").
GPU side (-m gpu): the program's ring / key-switch / MulRelin / Rescale / error-behaviour cases, all against the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "parity")


def _build():
    if not os.path.exists(os.path.join(ROOT, "lattigo_amd", "libhering.so")):
        pytest.skip("libhering.so not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "warning" not in r.stderr, r.stderr


def test_cpp_mirror_compiles_links_and_refuses_to_run_without_a_device():
    _build()
    r = subprocess.run([EXE, "--link-only"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "libhering" in r.stdout
    if " 0 HIP device(s)" in r.stdout:
        assert "no CPU fallback" in r.stdout and "code -3" in r.stdout
        r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
        assert r.returncode == 3, r.stdout + r.stderr  # the full program refuses as well


def test_header_mirrors_every_ring_operation_of_the_abi():
    """every he_binop / he_unop / he_scalarop selector of hering.h has its reference-named method in hering.hpp"""
    import re
    h = open(os.path.join(ROOT, "include", "hering.h")).read()
    hpp = open(os.path.join(ROOT, "include", "hering.hpp")).read()
    sels = re.findall(r"^\s+(HE_[A-Z_]+)[ ,=]", h, flags=re.M)
    ops = [s for s in sels if not s.endswith("_COUNT") and s not in ("HE_OK", "HE_EINVAL", "HE_EHANDLE", "HE_EDEVICE", "HE_EPARAM", "HE_ENOMEM")]
    assert len(ops) >= 28
    special = {"MFORM": "MForm", "MFORM_LAZY": "MFormLazy", "IMFORM": "IMForm"}
    for s in ops:  # HE_MUL_COEFFS_MONTGOMERY_LAZY -> a method named MulCoeffsMontgomeryLazy, as ring/operations.go names it
        name = special.get(s[3:]) or "".join(w.capitalize() for w in s[3:].split("_"))
        assert re.search(r"void %s\(" % name, hpp), f"{s}: no method {name} in hering.hpp"
    # and every entry point the header declares is called by some method of the mirror (the compiler checks each call's types)
    decl = set(re.findall(r"\b(he_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", h, flags=re.S)))
    assert len(decl) >= 110
    missing = sorted(d for d in decl if not re.search(r"\b%s\b" % d, hpp))
    assert not missing, missing


@pytest.mark.gpu
def test_cpp_parity_program():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS:" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["c3", "c2"])
def test_compiled_host_concurrent_callers(workload):
    """tests/cpp/run_parallel.cpp: std::thread per ciphertext over ONE hering::Evaluator (the reference's b.RunParallel pattern from
    a compiled host, public interface only) -- c3: BGV MulRelin at the headline shape; c2: CKKS Mul + Rescale, four interface calls
    per operation, all of them through the context's submission queue.  EVERY caller's last result equals the oracle's."""
    import json
    _build()
    # (sync_each, deferred depth): deferred = he_ctx_set_deferred, the calls return once filed
    for sync_each, deferred in (("0", "0"), ("1", "0"), ("0", "4"), ("1", "4")):
        r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "run_parallel"), "8", "6", sync_each, "1", workload, "64", "30", deferred],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["verified"] is True and d["K"] == 8 and d["ops_per_s"] > 0 and d["verified_callers"] == "8/8" and d["workload"] == workload
        assert d["deferred_depth"] == int(deferred)

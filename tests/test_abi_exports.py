"""CPU-side check of the drop-in boundary: libhering.so builds (hipcc cross-compiles for
gfx950 without a GPU), loads, and exports every symbol include/hering.h declares.
No compute calls here -- those are the `-m gpu` parity tests."""
import os

import pytest

import __graft_entry__ as graft
from lattigo_amd import _lib


@pytest.fixture(scope="module")
def lib():
    graft.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    syms = _lib.declared_symbols()
    assert len(syms) > 70
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_version_and_error_string(lib):
    assert b"gfx950" in lib.he_version()
    assert isinstance(lib.he_last_error(), bytes)


def test_no_cpu_fallback_without_device(lib):
    """Without a HIP device context creation must fail loudly (HE_EDEVICE), never fall back."""
    import ctypes as C
    h = C.c_uint64()
    rc = lib.he_ctx_create(0, C.byref(h))
    if rc == 0:  # running on a GPU box
        lib.he_ctx_destroy(h)
        pytest.skip("GPU present")
    assert rc == -3 and b"no CPU fallback" in lib.he_last_error()


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "lattigo_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/ by construction", "") or f == "host_math.h", f


def test_go_shim_matches_the_header():
    """go/hering/*.go (the cgo side of the boundary, shipped as source: no Go toolchain in the image): every C.he_* call names
    a declared entry point with the declared number of arguments, and the seven rlwe.EvaluatorProvider methods plus the
    schemes.Evaluator set are defined (tools/check_go_abi.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_go_abi.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_cpu_quota_probe_reads_cgroup_files(tmp_path, monkeypatch):
    """bench.py reports the container's CPU quota with the CPU baseline (the thread sweep is meaningless beyond it)."""
    import builtins
    import bench
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("1600000 100000\n")
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    assert bench.cpu_quota_cores() == 16.0

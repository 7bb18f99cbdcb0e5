"""The fallback launch sequences under the driver's suite (`-m gpu`): DESIGN.md section 9 lists the run-time switches that take
single fusions out of the pipelines (HERING_NO_*).  The default sequence is what every other test exercises; here the headline
shape (BGV MulRelin logN = 15, 12 + 3 limbs, B = 256, every entry x every limb against the oracle), the key-switch suite's Rotate
test and the queue's every-operator test are re-run in a subprocess (the switches are read once per process) with the fusions
switched off in the three pairs of DESIGN.md -- so that every fallback sequence the library can take is bit-exact too."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

PAIRS = [("HERING_NO_AUTO_SCATTER", "HERING_NO_TENSOR_EPILOGUE"), ("HERING_NO_MAC_EPILOGUE", "HERING_NO_PROD_PROLOGUE"),
         ("HERING_NO_FAST_MODUP", "HERING_NO_LEAN_INV_ROWS")]
TESTS = ["tests/test_gpu_headline.py::test_full_size_config3_bgv_mulrelin_logN15[256]", "tests/test_gpu_rlwe.py::test_rotate",
         "tests/test_gpu_coalesce.py::test_every_operator_entry_point_coalesces[13-0]",
         "tests/test_gpu_coalesce.py::test_every_operator_entry_point_coalesces[13-8]",
         "tests/test_gpu_coalesce.py::test_key_switches_coalesce_too[16]"]


@pytest.mark.parametrize("pair", PAIRS, ids=["+".join(s.replace("HERING_NO_", "no_").lower() for s in p) for p in PAIRS])
def test_fusions_off_in_pairs(pair):
    env = dict(os.environ)
    for name in pair:
        env[name] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + TESTS, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail

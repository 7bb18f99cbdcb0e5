#!/usr/bin/env python3
"""Golden digest of the full-size functional bootstrap (tests/test_gpu_c5_bootstrap.py::test_c5_functional_bootstrap), produced on
the CPU ORACLE backend: run_functional_bootstrap(**C5) with ctx = None runs the same drivers on oracle/ (the C restatement of the
reference's ring / rlwe arithmetic) with the same seed, and the SHA-256 of the refreshed ciphertext's words is written to
tests/golden/c5_bootstrap_digest.json.  The GPU test then requires the DEVICE run's ciphertext to have that digest: bit-for-bit
equality with the oracle at the reference's default bootstrapping shape, not only a precision bound.

    python tests/golden/gen_c5_digest.py        (a few minutes on one core: ~50 s for the bootstrap, the rest key generation)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests.bootstrap_fixtures import run_functional_bootstrap  # noqa: E402
from tests.test_gpu_c5_bootstrap import C5  # noqa: E402

out = run_functional_bootstrap(**C5)
assert out["backend"] == "oracle"
keep = {k: out[k] for k in ("ct_sha256", "seed", "output_level", "max_slot_error", "mean_precision_bits", "precision_bits", "backend")}
json.dump(keep, open(os.path.join(ROOT, "tests", "golden", "c5_bootstrap_digest.json"), "w"), indent=1)
print(json.dumps(keep))

#!/usr/bin/env python3
"""Golden digest of bench.py's c5 workload -- the operation trace of one CKKS bootstrap at the N16QP1546H192H32 shape with
synthetic keys and DFT diagonals (tools/bootstrap_c5_shape.py) -- produced on the CPU ORACLE backend: build(None, 1) runs the same
drivers over oracle/ (the C restatement of the reference's ring / rlwe arithmetic, oracle/circuits.py) with the same seed, and the
SHA-256 of the refreshed ciphertext's words goes to tests/golden/c5_trace_digest.json.  bench.py --workload c5 gives every batch
entry that same input and requires every entry of the device's output to have this digest (`verified`).

    python tests/golden/gen_c5_trace_digest.py        (about a minute on one core)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bootstrap_c5_shape as C5  # noqa: E402

t0 = time.time()
run, info = C5.build(None, 1)
res = run()
out = dict(C5.trace_digest(res, device=False), seed="0x1A77160 + 5", backend="oracle", cpu_seconds=round(time.time() - t0, 1), **info)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "c5_trace_digest.json"), "w"), indent=1)
print(json.dumps(out))

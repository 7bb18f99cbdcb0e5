#!/usr/bin/env python3
"""Random parameter shapes through the key-switch suite's full-size check (tests/test_gpu_rlwe.py::_full_size_check: GadgetProduct,
Rotate, the hoisted forms, CKKS MulRelin and Rescale at the two top levels, batch of 2, every limb against the oracle):
ring degree, chain length, number and size of special primes and the arithmetic class of every modulus are drawn at random, so
that digit widths that do not divide the chain, single-limb chains, mixed classes inside one digit etc. are all met.

    python tools/fuzz_shapes.py [--seconds 240] [--seed 1]        # on a GPU box; exit status 1 when a shape fails
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattigo_amd as la  # noqa: E402
from tests.test_gpu_rlwe import _full_size_check  # noqa: E402

QBITS = [36, 40, 45, 45, 45, 46, 50, 55, 55, 58, 60]
PBITS = [45, 55, 55, 60, 61, 61]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="logN 13..16, chains of up to 14 limbs, up to 6 special primes")
    a = ap.parse_args()
    rng = np.random.Generator(np.random.PCG64(a.seed))
    ctx = la.Context(0)
    t0, n, bad = time.time(), 0, []
    while time.time() - t0 < a.seconds:
        if a.big:
            logN = int(rng.choice([13, 14, 15, 15, 16]))
            nq = int(rng.integers(3, 15))
            np_ = int(rng.integers(1, 7))
        else:
            logN = int(rng.choice([10, 11, 12, 13, 13, 14, 15, 16]))
            nq = int(rng.integers(2, 11 if logN < 15 else 8))
            np_ = int(rng.integers(1, 5))
        logq = [int(rng.choice([50, 55, 58, 60]))] + [int(rng.choice(QBITS)) for _ in range(nq - 1)]
        logp = [int(rng.choice(PBITS)) for _ in range(np_)]
        seed = int(rng.integers(1, 1 << 30))
        rot = bool(rng.integers(0, 2))
        tag = f"logN={logN} logq={logq} logp={logp} seed={seed} rotate={rot}"
        try:
            _full_size_check(ctx, logN, logq, logp, seed, rot)
            print("ok  ", tag, flush=True)
        except Exception:  # noqa: BLE001
            bad.append(tag)
            print("FAIL", tag, flush=True)
            traceback.print_exc()
        n += 1
    print(f"{n} shapes, {len(bad)} failed", flush=True)
    for b in bad:
        print("  ", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Extract the known-answer vectors of the reference's ring/ntt_test.go:10-89
(`testVector`: N in {16..512}, two 60-bit primes, full input `poly` and expected
`polyNTT`) into tests/golden/ntt_kat.json.

Run in the build container only (needs /root/reference); the JSON it writes is
committed so the GPU box never reads the reference tree.

    python tests/golden/gen_ntt_kat.py [/root/reference]
"""
import json
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = open(os.path.join(ref, "ring", "ntt_test.go")).read()
start = src.index("var testVector")
end = src.index("func TestNTT")
body = src[start:end]

# every case: N, Qis, Poly{[][]uint64{ {..},{..} }}, Poly{[][]uint64{ {..},{..} }}
case_re = re.compile(
    r"\{\s*(\d+),\s*\[\]uint64\{([^}]*)\},\s*"
    r"Poly\{\[\]\[\]uint64\{\s*\{([^}]*)\},\s*\{([^}]*)\},?\s*\}\},\s*"
    r"Poly\{\[\]\[\]uint64\{\s*\{([^}]*)\},\s*\{([^}]*)\},?\s*\}\},\s*\}",
    re.S,
)


def ints(s):
    return [int(x) for x in re.findall(r"\d+", s)]


cases = []
for m in case_re.finditer(body):
    N = int(m.group(1))
    qis = ints(m.group(2))
    poly = [ints(m.group(3)), ints(m.group(4))]
    ntt = [ints(m.group(5)), ints(m.group(6))]
    assert all(len(r) == N for r in poly + ntt), (N, [len(r) for r in poly + ntt])
    cases.append({"N": N, "Qis": qis, "poly": poly, "polyNTT": ntt})

assert len(cases) == 6, len(cases)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ntt_kat.json")
with open(out, "w") as f:
    json.dump({"source": "tuneinsight/lattigo v6.2.0 ring/ntt_test.go:10-89", "cases": cases}, f)
print("wrote", out, [c["N"] for c in cases])

#!/usr/bin/env python3
"""tests/golden/ntt_kat.json (the reference's own NTT known-answer vectors, ring/ntt_test.go:10-89, extracted by
tests/golden/gen_ntt_kat.py) as a C++ include for tests/cpp/parity.cpp:  python gen_kat_inc.py > ntt_kat.inc"""
import json
import os

d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "ntt_kat.json")))
print("// generated from tests/golden/ntt_kat.json by tests/cpp/gen_kat_inc.py -- do not edit")
print("struct NttKat { int logN; std::vector<uint64_t> Qis, poly, polyNTT; };  // poly / polyNTT: [limb][N] flattened")
print("static const std::vector<NttKat> kNttKat = {")
for c in d["cases"]:
    flat = lambda rows: ", ".join(f"{int(v)}ull" for r in rows for v in r)
    print("    {%d, {%s}, {%s}, {%s}}," % (c["N"].bit_length() - 1, ", ".join(f"{int(q)}ull" for q in c["Qis"]), flat(c["poly"]), flat(c["polyNTT"])))
print("};")

#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; `--pmc X --kernel-trace --output-format csv`) of bench.py into
profiles/r01_pmc_traffic.json: HBM bytes per launch per kernel family.

Corrections applied as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: counters are in
KiB; FETCH_SIZE reports half of the bytes of wide coalesced streaming reads (x2); WRITE_SIZE is used as reported.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/bench_counter_collection.csv \
                                gpurun_out/pmc_WRITE_SIZE/bench_counter_collection.csv BATCH > profiles/r01_pmc_traffic.json
"""
import collections
import csv
import json
import sys

FAMILY = [("ntt_mac_f64_kernel", "ntt_mac_f64"), ("ntt_rows_f64_kernel<12, false>", "ntt_rows_fwd_f64"), ("ntt_rows_f64_kernel<12, true>", "ntt_rows_inv_f64"),
          ("ntt_rows_kernel<12, false", "ntt_rows_fwd"), ("ntt_rows_kernel<12, true", "ntt_rows_inv"),
          ("modup_fused_kernel", "modup"), ("ks_inner_kernel", "ks_inner"), ("tensor_kernel", "tensor"), ("ew_kernel", "ew")]


def fam(name):
    for pat, f in FAMILY:
        if pat in name:
            return f
    return None


def load(path):
    d = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        f = fam(r["Kernel_Name"])
        if f:
            d[f][0] += 1
            d[f][1] += float(r["Counter_Value"])
            d[f][2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {"batch": int(sys.argv[3]), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py",
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128 B request)", "kernels": {}}
for f in fetch:
    n = fetch[f][0]
    fb = 2 * fetch[f][1] / n * 1024
    wb = write[f][1] / max(write[f][0], 1) * 1024 if f in write else 0.0
    out["kernels"][f] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                         "avg_launch_us_profiled": fetch[f][2] / n / 1e3}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; `--pmc X --kernel-trace --output-format csv`) of bench.py into
profiles/rNN_pmc_traffic[_WORKLOAD].json: HBM bytes per launch and per step for every kernel family, and per unit of work.

Corrections applied as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: counters are in
KiB; FETCH_SIZE reports half of the bytes of wide coalesced streaming reads (x2); WRITE_SIZE is used as reported.

    python tools/pmc_summary.py FETCH.csv WRITE.csv BATCH [WORKLOAD STEPS] > profiles/r03_pmc_traffic.json
STEPS = launches of bench.py's step() the profiled command made (warmup + timed + the profiling leg), for the per-step sums.
"""
import collections
import csv
import json
import re
import sys

FAMILY = [(r"ntt_mac_f64(_dma)?_kernel", "ntt_mac_f64"), (r"ntt_rows_f64_kernel<\d+, false", "ntt_rows_fwd_f64"),
          (r"ntt_rows_f64_kernel<\d+, true", "ntt_rows_inv_f64"), (r"ntt_rows_kernel<\d+, false", "ntt_rows_fwd"),
          (r"ntt_rows_kernel<\d+, true", "ntt_rows_inv"), (r"ntt_cols_kernel<\d+, false\b", "ntt_cols_fwd"),
          (r"ntt_cols_kernel<\d+, true\b", "ntt_cols_inv"), (r"modup_fused_kernel|modup_kernel", "modup"), (r"center_copy", "center_copy"),
          (r"ks_inner_kernel", "ks_inner"), (r"tensor_kernel", "tensor"), (r"ew_kernel", "ew"), (r"gather_kernel|shift_kernel", "gather"),
          (r"diag_mac_kernel", "diag_mac"), (r"build_index", "build_index"), (r"automorphism_coeff", "automorphism_coeff"),
          (r"mask_spread", "mask_spread"), (r"ci_fold|ci_ref", "ci_fold"), (r"key_to_f64", "key_to_f64")]


def fam(name):
    for pat, f in FAMILY:
        if re.search(pat, name):
            return f
    return None


UNMATCHED = collections.Counter()  # library kernels (namespace he::) no pattern claims: a renamed kernel must not vanish from the sums


def load(path):
    d = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        f = fam(r["Kernel_Name"])
        if f:
            d[f][0] += 1
            d[f][1] += float(r["Counter_Value"])
            d[f][2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        elif "he::" in r["Kernel_Name"] and not re.search(r"tab_fill|probe", r["Kernel_Name"]):
            UNMATCHED[re.sub(r"\(.*$", "", r["Kernel_Name"])] += 1
    return d


fetch, write = load(sys.argv[1]), load(sys.argv[2])
batch = int(sys.argv[3])
workload = sys.argv[4] if len(sys.argv) > 4 else "c3"
steps = int(sys.argv[5]) if len(sys.argv) > 5 else None
out = {"batch": batch, "workload": workload, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py",
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128 B request)", "kernels": {}}
total = 0.0
for f in fetch:
    n = fetch[f][0]
    fb = 2 * fetch[f][1] / n * 1024
    wb = write[f][1] / max(write[f][0], 1) * 1024 if f in write else 0.0
    out["kernels"][f] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                         "avg_launch_us_profiled": fetch[f][2] / n / 1e3}
    if f != "key_to_f64":  # key conversion happens once, at key upload
        total += (fb + wb) * n
    if steps:
        out["kernels"][f]["launches_per_step"] = n / steps
        out["kernels"][f]["hbm_bytes_per_step"] = (fb + wb) * n / steps
if steps:
    out["steps_profiled"] = steps
    out["hbm_bytes_per_step"] = total / steps
    out["hbm_MiB_per_unit"] = total / steps / batch / 2**20
if UNMATCHED:
    out["unmatched_kernels"] = dict(UNMATCHED)
    sys.stderr.write(f"pmc_summary: kernels of the library that no family pattern matches (their bytes are NOT in the sums): {dict(UNMATCHED)}\n")
print(json.dumps(out, indent=1))

"""Summarise the two SQ counter passes of tools/pmc_sq.sh into profiles/rNN_sq_counters.json.
usage: python tools/sq_summary.py gpurun_out/pmc_sqb/b_counter_collection.csv gpurun_out/pmc_lds/b_counter_collection.csv > out.json
Counter values are per launch, summed over the shader engines; the summary keeps per-launch means per kernel family."""
import collections
import csv
import json
import sys

FAM = [("modup_fused", "modup"), ("ntt_mac_f64", "ntt_mac_f64"), ("ntt_rows_f64_kernel<12, false", "ntt_rows_fwd_f64"),
       ("ntt_rows_f64_kernel<12, true", "ntt_rows_inv_f64"), ("ntt_rows_kernel<12, false", "ntt_rows_fwd"),
       ("ntt_rows_kernel<12, true", "ntt_rows_inv"), ("tensor_kernel", "tensor"), ("ks_inner", "ks_inner")]


def fam(name):
    for key, f in FAM:
        if key in name:
            return f
    return None


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        f = fam(r["Kernel_Name"])
        if f:
            per[(f, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for (f, _), c in per.items():
        for k, v in c.items():
            acc[f][k].append(v)
    return {f: {k: sum(v) / len(v) for k, v in c.items()} for f, c in acc.items()}


sqb, lds = load(sys.argv[1]), load(sys.argv[2])
out = {"source": "rocprofv3 --pmc (two passes, tools/pmc_sq.sh) on bench.py --steps 3, default batch; values are per-launch means summed over the 32 shader engines",
       "notes": "SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES saturates at 8 (32 SIMDs per engine, 4-cycle issue): valu_util = that ratio / 8",
       "kernels": {}}
for f in [x[1] for x in FAM]:
    if f not in sqb:
        continue
    c = sqb[f]
    k = {"valu_util": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"] / 8, 3),
         "wait_any_frac_of_wave_cycles": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
         "valu_insts_per_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"])}
    if f in lds and lds[f].get("SQ_INSTS_LDS", 0) > 0:
        l = lds[f]
        k["lds_bank_conflict_frac"] = round(l["SQ_LDS_BANK_CONFLICT"] / max(l["SQ_ACTIVE_INST_LDS"], 1), 3)
        k["lds_insts_per_launch"] = round(l["SQ_INSTS_LDS"])
    out["kernels"][f] = k
print(json.dumps(out, indent=1))

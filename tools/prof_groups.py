#!/usr/bin/env python3
"""Per (kernel, grid) groups of a rocprofv3 counter-collection csv: launches, mean duration, mean counter values and a few
derived figures (VALU instructions per wave, issue utilisation).
usage: python tools/prof_groups.py gpurun_out/pmc_TAG/b_counter_collection.csv [name filter]"""
import collections
import csv
import re
import sys

rows = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
        continue
    name = re.sub(r"^void he::|\(he::.*$", "", r["Kernel_Name"])
    key = (name, r["Grid_Size"], r["Workgroup_Size"], r["Dispatch_Id"])
    rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
    rows[key]["_dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if "End_Timestamp" in r else 0.0
    rows[key]["_vgpr"] = float(r.get("VGPR_Count", r.get("Arch_VGPR_Count", 0)) or 0)
groups = collections.defaultdict(list)
for (name, grid, wg, _), c in rows.items():
    groups[(name, grid, wg)].append(c)
for (name, grid, wg), cs in sorted(groups.items(), key=lambda kv: -sum(c["_dur_us"] for c in kv[1])):
    n = len(cs)
    mean = {k: sum(c.get(k, 0.0) for c in cs) / n for k in cs[0]}
    line = f"{name[:60]:60s} grid={grid:>9s} wg={wg:>4s} n={n:3d} dur={mean['_dur_us']:8.1f}us"
    if "SQ_WAVES" in mean and mean["SQ_WAVES"] > 0:
        w = mean["SQ_WAVES"]
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
            if k in mean:
                line += f" {k[9:]}/w={mean[k] / w:8.1f}"
        if "SQ_WAVE_CYCLES" in mean:
            line += f" wavecyc/w={mean['SQ_WAVE_CYCLES'] / w:9.0f}"
    for k in ("SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INST_CYCLES_VMEM", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F64"):
        if k in mean:
            line += f" {k}={mean[k]:.3g}"
    print(line)

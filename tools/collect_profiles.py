#!/usr/bin/env python3
"""Copy the summaries of tools/round_artifacts.sh from gpurun_out/ (scratch) into profiles/ (tracked): python tools/collect_profiles.py r02"""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(G, "bench_final.json"), os.path.join(P, f"{tag}_bench.json"))
shutil.copy(os.path.join(G, "bench_configs.jsonl"), os.path.join(P, f"{tag}_bench_other_configs.jsonl"))
stats = [f for f in os.listdir(os.path.join(G, "prof_final")) if f.endswith("kernel_stats.csv")]
shutil.copy(os.path.join(G, "prof_final", stats[0]), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
for probe in ("instr_probe", "hbm_probe"):
    if os.path.exists(os.path.join(G, probe + ".txt")):
        shutil.copy(os.path.join(G, probe + ".txt"), os.path.join(P, f"{tag}_{probe}.txt"))
# PMC traffic per workload (tools/round_artifacts.sh: bench.py --steps S --warmup 1 --no-kernel-timing)
for w, steps in (("c2", 5), ("c3", 5), ("c4", 5), ("c5", 2)):
    f = os.path.join(G, f"pmc_FETCH_SIZE_{w}", "bench_counter_collection.csv")
    wr = os.path.join(G, f"pmc_WRITE_SIZE_{w}", "bench_counter_collection.csv")
    if not (os.path.exists(f) and os.path.exists(wr)):
        print("no PMC pass for", w)
        continue
    import bench
    batch = bench.WORKLOADS[w][1]
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f, wr, str(batch), w, str(1 + steps)], text=True)
    d = json.loads(out)
    json.dump(d, open(os.path.join(P, f"{tag}_pmc_traffic.json" if w == "c3" else f"{tag}_pmc_traffic_{w}.json"), "w"), indent=1)
for w in ("c2", "c4", "c5"):
    d = os.path.join(G, f"stats_{w}")
    if os.path.isdir(d):
        st = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("kernel_stats.csv")]
        if st:
            shutil.copy(st[0], os.path.join(P, f"{tag}_kernel_stats_{w}.csv"))
if os.path.exists(os.path.join(G, "r05_ntt_pmc.txt")):
    shutil.copy(os.path.join(G, "r05_ntt_pmc.txt"), os.path.join(P, f"{tag}_ntt_pmc_traffic.txt"))


def sq_summary(passes, workload, out_name):
    groups = {}
    for t in passes:
        f = [os.path.join(dp, x) for dp, _, fs in os.walk(os.path.join(G, f"pmc_{t}")) for x in fs if x.endswith("counter_collection.csv")]
        if not f:
            return
        rows = collections.defaultdict(dict)
        for r in csv.DictReader(open(f[0])):
            key = (r["Kernel_Name"].split("(")[0].replace("void he::", ""), r["Grid_Size"], r["Dispatch_Id"])
            rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
        agg = collections.defaultdict(list)
        for (name, grid, _), c in rows.items():
            agg[(name, grid) if workload == "c3" else (name, "")].append(c)
        for (name, grid), cs in agg.items():
            m = {k: sum(c.get(k, 0.0) for c in cs) / len(cs) for k in cs[0]}
            m["_launches"] = len(cs)
            groups.setdefault(f"{name} grid={grid}" if grid else name, {}).update(m)
    summary = {"workload": workload, "source": "rocprofv3 --pmc (tools/round_artifacts.sh) on bench.py; per-launch means, summed over the shader engines"
               + ("" if workload == "c3" else "; launches of one kernel instantiation pooled over the grids of the trace"),
               "notes": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles per wave; valu_issue_share = VALU instructions per wave / wave-quad-cycles per wave",
               "launch_groups": {}}
    for k, m in sorted(groups.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1].get("_launches", 1)):
        if m.get("SQ_WAVES", 0) < 1000:
            continue
        w = m["SQ_WAVES"]
        e = {"launches": m.get("_launches"), "waves": round(w), "valu_insts_per_wave": round(m.get("SQ_INSTS_VALU", 0) / w, 1),
             "wave_quad_cycles_per_wave": round(m.get("SQ_WAVE_CYCLES", 0) / w),
             "valu_issue_share_of_wave_time": round(m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3),
             "wait_any_share": round(m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3),
             "wait_inst_any_share": round(m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3)}
        summary["launch_groups"][k] = e
    json.dump(summary, open(os.path.join(P, out_name), "w"), indent=1)


for w in ("c2", "c4", "c5"):
    sq_summary((f"sq1_{w}",), w, f"{tag}_sq_counters_{w}.json")

# SQ passes: per (kernel, grid) launch group
groups = {}
for t in ("sq1", "sq2"):
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(os.path.join(G, f"pmc_{t}", "b_counter_collection.csv"))):
        key = (r["Kernel_Name"].split("(")[0].replace("void he::", ""), r["Grid_Size"], r["Dispatch_Id"])
        rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
    agg = collections.defaultdict(list)
    for (name, grid, _), c in rows.items():
        agg[(name, grid)].append(c)
    for (name, grid), cs in agg.items():
        m = {k: sum(c.get(k, 0.0) for c in cs) / len(cs) for k in cs[0]}
        groups.setdefault(f"{name} grid={grid}", {}).update(m)
summary = {"source": "rocprofv3 --pmc (two passes, tools/round_artifacts.sh) on bench.py --steps 3; per-launch means, summed over the shader engines",
           "notes": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles per wave; valu_issue_share = VALU instructions per wave / wave-quad-cycles per wave",
           "launch_groups": {}}
for k, m in sorted(groups.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if m.get("SQ_WAVES", 0) < 1000:
        continue
    w = m["SQ_WAVES"]
    e = {"waves": round(w), "valu_insts_per_wave": round(m.get("SQ_INSTS_VALU", 0) / w, 1),
         "wave_quad_cycles_per_wave": round(m.get("SQ_WAVE_CYCLES", 0) / w),
         "valu_issue_share_of_wave_time": round(m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3),
         "wait_any_share": round(m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3),
         "wait_inst_any_share": round(m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3),
         "salu_per_wave": round(m.get("SQ_INSTS_SALU", 0) / w, 1), "lds_per_wave": round(m.get("SQ_INSTS_LDS", 0) / w, 1),
         "vmem_rd_per_wave": round(m.get("SQ_INSTS_VMEM_RD", 0) / w, 1), "vmem_wr_per_wave": round(m.get("SQ_INSTS_VMEM_WR", 0) / w, 1),
         "smem_per_wave": round(m.get("SQ_INSTS_SMEM", 0) / w, 1)}
    if m.get("SQ_ACTIVE_INST_LDS", 0) > 0:
        e["lds_bank_conflict_share"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_ACTIVE_INST_LDS"], 3)
    summary["launch_groups"][k] = e
json.dump(summary, open(os.path.join(P, f"{tag}_sq_counters.json"), "w"), indent=1)
print("profiles written:", sorted(f for f in os.listdir(P) if f.startswith(tag)))

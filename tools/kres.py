"""Per-kernel resource usage from a `hipcc -Rpass-analysis=kernel-resource-usage` log: python tools/kres.py LOG [filter]"""
import re, subprocess, sys

KEYS = [("VGPRs", "V"), ("AGPRs", "A"), ("TotalSGPRs", "S"), ("VGPRs Spill", "vsp"), ("SGPRs Spill", "ssp"),
        ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "lds"), ("ScratchSize [bytes/lane]", "scr")]
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out, cur = [], None
for ln in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        out.append(cur)
        continue
    for key, _ in KEYS:
        m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", ln)
        if m and cur is not None:
            cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(o["name"] for o in out), capture_output=True, text=True).stdout.split("\n")
for o, name in zip(out, names):
    name = re.sub(r"\(he::\w+\)$", "", name).replace("void he::", "")
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} " + " ".join(f"{tag}{o.get(key)}" for key, tag in KEYS))
for l in txt.splitlines():
    if "error" in l:
        print(l)

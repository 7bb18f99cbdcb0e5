#!/usr/bin/env python3
"""Idle gaps of the device during the c5 replay (round 5 diagnosis): reads a rocprofv3 --kernel-trace csv, takes the LAST `wall`
seconds of it (the timed run is the last thing the probe does) and reports busy time, idle time by gap size, and which kernels
follow the long gaps.  python tools/c5_gap_analysis.py <kernel_trace.csv> <wall seconds>"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
wall = float(sys.argv[2])
t_end = max(r[1] for r in rows)
t0 = t_end - int(wall * 1e9)
rows = [r for r in rows if r[0] >= t0]
busy = 0; cur_end = rows[0][0]; gaps = []
for s, e, n in rows:
    if s > cur_end:
        gaps.append((s - cur_end, n)); cur_end_prev = cur_end
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
span = cur_end - rows[0][0]
print(f"kernels {len(rows)}  span {span/1e6:.1f} ms  busy {busy/1e6:.1f} ms ({busy/span:.2f})  kernel-time sum {sum(e-s for s,e,_ in rows)/1e6:.1f} ms")
for lo, hi in ((0, 5e3), (5e3, 20e3), (20e3, 100e3), (100e3, 1e6), (1e6, 1e12)):
    g = [x for x, _ in gaps if lo <= x < hi]
    print(f"gaps {lo/1e3:7.0f}-{hi/1e3:9.0f} us: {len(g):6d}  total {sum(g)/1e6:8.1f} ms")
after = collections.Counter()
for x, n in gaps:
    if x >= 20e3: after[n[:70]] += x
print("kernels that follow gaps >= 20 us (by idle time):")
for n, x in after.most_common(12): print(f"  {x/1e6:8.1f} ms  {n}")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    by[n][0] += 1; by[n][1] += e - s
print("kernels by time (count, ms, mean us):")
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]: print(f"  {c:6d} {t/1e6:8.1f} {t/c/1e3:8.1f}  {n[:90]}")

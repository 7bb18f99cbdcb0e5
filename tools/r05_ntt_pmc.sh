# stand-alone Ring.NTT: HBM traffic per limb-transform from the PMC counters ("FETCH + WRITE = 2 limb +- 5 %"
# is the one-pass target; the two-pass transform shipped here moves 4):  bash tools/r05_ntt_pmc.sh -> gpurun_out/r05_ntt_pmc.txt
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r05_nttpmc_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/r05_nttpmc_$C -o n -- python $R/tools/ntt_prof.py > $R/gpurun_out/r05_nttpmc_$C.log 2>&1
done
python - <<PY > $R/gpurun_out/r05_ntt_pmc.txt
import csv, collections, glob
tot = collections.defaultdict(lambda: [0.0, 0])
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/r05_nttpmc_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if "ntt_" not in r["Kernel_Name"]: continue
            k = (r["Kernel_Name"].split("(")[0].replace("void he::", ""), r["Grid_Size"], C)
            tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
print("kernel, grid, counter, launches, KiB per launch (FETCH_SIZE: x2 for gfx950 per the microarch guide)")
for k, (v, n) in sorted(tot.items()):
    print(k[0], k[1], k[2], n, round(v / n * (2 if k[2] == "FETCH_SIZE" else 1), 1))
PY
cat $R/gpurun_out/r05_ntt_pmc.txt | head -40

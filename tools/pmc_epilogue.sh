R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $R/gpurun_out/pmc_epi_$v
  HERING_NO_TENSOR_EPILOGUE=$v timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_epi_$v -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-verify > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for v in (0,1):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"{R}/gpurun_out/pmc_epi_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            key = "rows_f64_fwd" if "ntt_rows_f64_kernel<12, false" in n else "rows_int_fwd" if "ntt_rows_kernel<12, false" in n else "tensor" if "tensor" in n else None
            if key: acc[(key, r["Grid_Size"])].append(float(r["Counter_Value"]))
    print("NO_TENSOR_EPILOGUE=%d" % v, {k: round(sum(x)/len(x)*64*2/1e6) for k,x in sorted(acc.items())}, "MB fetched per launch")
PY

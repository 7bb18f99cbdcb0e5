# usage (GPU box): bash tools/mac_counters.sh  -> gpurun_out/macctr_<variant>_<pass>/ ; mean counters of the ntt_mac_f64 launches
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"
P2="SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES"
P3="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_ACTIVE_INST_ANY"
for v in r8 r16; do
  if [ $v = r16 ]; then export HERING_MAC_R16=1; else unset HERING_MAC_R16; fi
  i=0
  for P in "$P1" "$P2" "$P3"; do i=$((i+1))
    rm -rf $R/gpurun_out/macctr_${v}_$i
    timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/macctr_${v}_$i -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-verify > /dev/null 2>&1
  done
done
python - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for v in ("r16","r8"):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"{R}/gpurun_out/macctr_{v}_*/**/*counter_collection.csv", recursive=True):
        per=collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if "ntt_mac_f64" in r["Kernel_Name"]:
                per[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
        for c in per.values():
            for k,x in c.items(): acc[k].append(x)
    m={k:sum(x)/len(x) for k,x in acc.items()}
    print(v, {k: round(x) for k,x in sorted(m.items())})
PY

# rocprofv3 --kernel-trace --stats summaries (and one SQ counter pass) for the secondary workloads c2 / c4 / c5
# (VERDICT r4 item 3): bash tools/r05_stats_other.sh   -> gpurun_out/r05_stats_<w>/, gpurun_out/r05_sq_<w>/
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
BARGS="--no-cpu-baseline --no-verify --no-ntt --no-b1 --no-concurrent --no-kernel-timing"
for W in c2 c4 c5; do
  S=10; [ $W = c5 ] && S=3
  rm -rf $R/gpurun_out/r05_stats_$W $R/gpurun_out/r05_sq_$W
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_stats_$W -o bench -- python $R/bench.py --workload $W --steps $S --warmup 2 $BARGS > $R/gpurun_out/r05_stats_$W.log 2>&1
  # keep only the summaries (the raw kernel trace of c5 is tens of MiB)
  find $R/gpurun_out/r05_stats_$W -name '*kernel_trace.csv' -delete
done
for W in c4 c5; do
  S=3; [ $W = c5 ] && S=1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r05_sq_$W -o b -- python $R/bench.py --workload $W --steps $S --warmup 1 $BARGS > $R/gpurun_out/r05_sq_$W.log 2>&1
  find $R/gpurun_out/r05_sq_$W -name '*kernel_trace.csv' -delete
done
ls $R/gpurun_out/r05_stats_c2/* $R/gpurun_out/r05_sq_c4/* | head

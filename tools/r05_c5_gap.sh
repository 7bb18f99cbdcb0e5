# device-busy share and launch count of the c5 replay (K = 16) in both modes of the queue: rocprofv3 kernel trace of the probe, analysed
# on the box (the trace is tens of MiB); summaries -> gpurun_out/r05_c5_replay_gaps.txt
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_c5_replay_gaps.txt; : > $OUT
for M in deferred blocking; do
  rm -rf /tmp/kt_$M
  HERING_C5_ONLY=$M rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$M -o kt -- python $GRAFT_REPO_ROOT/tools/c5_replay_probe.py 16 > /tmp/kt_$M.log 2>&1
  grep "^{" /tmp/kt_$M.log >> $OUT
  W=$(grep "^{" /tmp/kt_$M.log | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['wall_s'])")
  F=$(find /tmp/kt_$M -name "*kernel_trace.csv" | head -1)
  echo "== $M: the last $W s of the kernel trace (the timed run: 16 callers x 3 bootstraps)" >> $OUT
  python3 $GRAFT_REPO_ROOT/tools/c5_gap_analysis.py $F $W >> $OUT
done

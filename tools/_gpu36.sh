cd /tmp; export TMPDIR=/tmp
for M in deferred; do
  rm -rf /tmp/kt_$M
  HERING_C5_ONLY=$M rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$M -o kt -- python $GRAFT_REPO_ROOT/tools/c5_replay_probe.py 16 > /tmp/kt_$M.log 2>&1
  grep "^{" /tmp/kt_$M.log | cut -c1-300; grep -i "error\|Traceback" -A3 /tmp/kt_$M.log | head -20
  W=$(grep "^{" /tmp/kt_$M.log | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['wall_s'])")
  F=$(find /tmp/kt_$M -name "*kernel_trace.csv" | head -1)
  echo "== $M wall $W file $F"
  python3 $GRAFT_REPO_ROOT/tools/c5_gap_analysis.py $F $W
done

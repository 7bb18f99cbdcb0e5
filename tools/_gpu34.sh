f() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], d['workload'],'K',d['K'],'sync',d['sync_each'],'depth',d['deferred_depth'],'mb',d['mean_batch'],'ops',round(d['ops_per_s']),d['verified_callers'])
" "$1"; }
for A in 4 16; do
  export HERING_QUEUE_AHEAD=$A
  for d in 0 8; do
    timeout 120 tests/cpp/run_parallel 64 96 0 1 c3 64 30 $d | f "ahead=$A"
    timeout 120 tests/cpp/run_parallel 64 96 1 1 c3 64 30 $d | f "ahead=$A"
    timeout 120 tests/cpp/run_parallel 64 96 0 1 c2 64 30 $d | f "ahead=$A"
    timeout 120 tests/cpp/run_parallel 16 96 0 1 c2 64 30 $d | f "ahead=$A"
  done
done
for A in 4 16; do
HERING_QUEUE_AHEAD=$A timeout 600 python tools/c5_replay_probe.py 16 2>&1 | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ahead=$A c5 K16 blocking',d['coalesced'],'deferred',d['deferred']['coalesced'],'lone',d['lone_caller'],d['deferred']['lone_caller'],d['verified'])"
done

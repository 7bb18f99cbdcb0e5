f() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['workload'],'K',d['K'],'iters',d['calls_per_caller'],'maxb',d['max_batch'],'depth',d['deferred_depth'],'mb',d['mean_batch'],'ops',round(d['ops_per_s']),d['verified_callers'])
"; }
for d in 32 64 128; do
  timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 64 100 $d | f
  timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 256 100 $d | f
  timeout 120 tests/cpp/run_parallel 16 2000 0 1 c2 64 100 $d | f
  timeout 120 tests/cpp/run_parallel 32 2000 0 1 c2 128 100 $d | f
done

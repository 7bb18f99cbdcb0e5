f() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['workload'],'K',d['K'],'iters',d['calls_per_caller'],'sync',d['sync_each'],'depth',d['deferred_depth'],'mb',d['mean_batch'],'ops',round(d['ops_per_s']),d['verified_callers'])
"; }
for d in 0 8 16; do
  timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 64 100 $d | f
  timeout 120 tests/cpp/run_parallel 16 2000 0 1 c2 64 100 $d | f
  timeout 120 tests/cpp/run_parallel 32 1000 0 1 c2 64 100 $d | f
done
timeout 120 tests/cpp/run_parallel 128 500 0 1 c2 128 100 8 | f
timeout 120 tests/cpp/run_parallel 64 300 0 1 c3 64 30 8 | f
timeout 120 tests/cpp/run_parallel 64 300 0 1 c3 64 30 0 | f
timeout 120 tests/cpp/run_parallel 16 600 0 1 c3 64 30 8 | f
timeout 120 tests/cpp/run_parallel 16 600 0 1 c3 64 30 0 | f
timeout 120 tests/cpp/run_parallel 4 1000 0 1 c3 64 30 8 | f
timeout 120 tests/cpp/run_parallel 4 1000 0 1 c3 64 30 0 | f

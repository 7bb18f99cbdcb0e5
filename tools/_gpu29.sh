set -x
timeout 600 python -m pytest tests/test_gpu_coalesce.py tests/test_cpp_host.py -x -q 2>&1 | tail -5
for d in 0 4 8; do for K in 16 64; do timeout 120 tests/cpp/run_parallel $K 96 0 1 c2 64 30 $d; done; done
for d in 0 8; do timeout 120 tests/cpp/run_parallel 64 96 0 1 c3 64 30 $d; timeout 120 tests/cpp/run_parallel 64 96 1 1 c3 64 30 $d; done
timeout 600 python tools/c5_replay_probe.py 16 2>&1 | tail -3

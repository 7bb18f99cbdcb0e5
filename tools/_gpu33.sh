timeout 600 python -m pytest tests/test_gpu_coalesce.py tests/test_cpp_host.py -x -q 2>&1 | tail -3
for K in 16 64; do timeout 120 tests/cpp/run_parallel $K 96 0 1 c2 64 30 8 | cut -c100-; done
timeout 120 tests/cpp/run_parallel 64 96 0 1 c3 64 30 8 | cut -c100-
timeout 120 tests/cpp/run_parallel 64 96 1 1 c3 64 30 8 | cut -c100-
HERING_QUEUE_TIMING=1 HERING_REPLAY_PROFILE=1 timeout 600 python tools/c5_replay_probe.py 16 2>&1 | grep "counters\|^{" | cut -c1-700

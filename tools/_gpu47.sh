timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_cpp_host.py -x -q 2>&1 | tail -3
g() { grep -o "\"K\": [0-9]*\|max_batch.*"; }
timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 256 100 128 2>&1 | g
timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 64 100 128 2>&1 | g
timeout 120 tests/cpp/run_parallel 32 2000 0 1 c2 128 100 128 2>&1 | g
timeout 120 tests/cpp/run_parallel 16 2000 0 1 c2 64 100 128 2>&1 | g
timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 256 100 16 2>&1 | g
timeout 120 tests/cpp/run_parallel 64 300 0 1 c3 64 30 8 2>&1 | g
timeout 120 tests/cpp/run_parallel 64 300 1 1 c3 64 30 8 2>&1 | g
timeout 600 python tools/c5_replay_probe.py 16 2>&1 | grep "^{" | cut -c1-900

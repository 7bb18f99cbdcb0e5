timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 256 100 128 2>&1 | grep -o "mean_batch.*"
timeout 120 tests/cpp/run_parallel 16 2000 0 1 c2 64 100 128 2>&1 | grep -o "mean_batch.*"
timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 256 100 0 2>&1 | grep -o "mean_batch.*"
timeout 120 tests/cpp/run_parallel 64 300 0 1 c3 64 30 8 2>&1 | grep -o "mean_batch.*"

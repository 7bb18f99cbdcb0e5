export HERING_QUEUE_DEBUG=1
timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 64 100 8 2>&1 | cut -c1-400
timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 256 100 8 2>&1 | cut -c1-400
timeout 120 tests/cpp/run_parallel 16 2000 0 1 c2 64 100 8 2>&1 | cut -c1-400

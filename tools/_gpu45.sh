export HERING_QUEUE_DEBUG=1
timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 256 100 128 2>&1 | cut -c1-400
timeout 120 tests/cpp/run_parallel 32 2000 0 1 c2 128 100 128 2>&1 | cut -c1-400
HERING_QUEUE_TIMING=1 timeout 120 tests/cpp/run_parallel 64 1000 0 1 c2 256 100 128 2>&1 | cut -c1-400

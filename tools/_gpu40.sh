timeout 900 python -m pytest tests/test_gpu_coalesce.py tests/test_cpp_host.py -x -q 2>&1 | tail -5
HERING_REPLAY_PROFILE=1 timeout 600 python tools/c5_replay_probe.py 16 2>&1 | grep "batches by\|^{" | cut -c1-1300

HERING_QUEUE_TIMING=1 HERING_REPLAY_PROFILE=1 timeout 600 python tools/c5_replay_probe.py 16 2>&1 | tail -12

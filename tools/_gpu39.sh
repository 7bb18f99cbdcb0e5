HERING_C5_ONLY=deferred HERING_REPLAY_PROFILE=1 timeout 600 python tools/c5_replay_probe.py 16 2>&1 | grep "batches by\|^{" | cut -c1-1200
HERING_C5_ONLY=blocking HERING_REPLAY_PROFILE=1 timeout 600 python tools/c5_replay_probe.py 16 2>&1 | grep "batches by\|^{" | cut -c1-1200

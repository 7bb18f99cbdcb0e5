#!/usr/bin/env python3
"""The c5 replay alone (bench.py concurrent_c5 at one K), for profiling: python tools/c5_replay_probe.py [K]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lattigo_amd as la
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = la.Context(0)
print(json.dumps(bench.concurrent_c5(la, ctx, ks=(K,), rounds=3)))

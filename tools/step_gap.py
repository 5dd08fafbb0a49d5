#!/usr/bin/env python3
"""Cost of the boundary between consecutive ca3d_step(K) calls: K-step batches as captured graphs against the same
batches launched kernel by kernel, timed over >= 50 ms (the question behind bench.py's `reps`)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
e = Engine(0)
e.configure(G)
e.set_rule_strings()
e.upload_state(host.random_fill(host.words_per_buffer(G)))
for K in (8, 20, 64, 256, 1024):
    for mode, gmin in (("graph", 1), ("eager", 1024)):
        e.set_option("graph_min", gmin)
        e.step(K); e.step(K); e.synchronize()
        reps = max(1, int(0.05 / (K * 6e-6)))
        t0 = time.perf_counter()
        for _ in range(reps):
            e.step(K)
        e.synchronize()
        dt = time.perf_counter() - t0
        print(f"G {G} K {K:5d} {mode}: {dt / (K * reps) * 1e6:7.3f} us/step over {reps} batches", flush=True)

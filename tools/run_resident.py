#!/usr/bin/env python3
"""Resident multi-step kernel against the per-step kernels: 512^3, default rule, batches of K steps back to back."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

G = 512
e = Engine(0)
e.set_option("stats", 0)
e.configure(G)
e.set_rule_strings()
e.upload_state(host.random_fill(host.words_per_buffer(G)))
for K in (8, 20, 64, 256, 1024, 4096):
    for res in (1, 0):
        e.set_option("resident", res)
        e.step(K); e.synchronize()
        reps = max(1, int(0.05 / (K * 5e-6)))
        t0 = time.perf_counter()
        for _ in range(reps):
            e.step(K)
        e.synchronize()
        dt = time.perf_counter() - t0
        print(f"K {K:5d} resident {res}: {dt / (K * reps) * 1e6:7.3f} us/step  frac {0.25 * G ** 3 / (dt / (K * reps)) / 8e12:.3f}  {e.info().kernel_name.decode()}", flush=True)

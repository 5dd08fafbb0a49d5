#!/usr/bin/env python3
"""Resident class kernel (clustered rule-set) against the per-step kernels at 512^3 and 256^3, batches of K steps back to back."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

grids = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [512, 256]
e = Engine(0)
e.set_option("stats", 0)
for G in grids:
    e.configure(G)
    e.set_rule_strings("moore", "5-7", "4-7", "4", "3-5", "3", "2-4")
    e.upload_state(host.random_fill(host.words_per_buffer(G)))
    for K in (20, 256, 2048):
        for res in (1, 0):
            e.set_option("resident", res)
            e.step(K); e.synchronize()
            reps = max(1, int(0.1 / (K * 8e-6)))
            t0 = time.perf_counter()
            for _ in range(reps):
                e.step(K)
            e.synchronize()
            dt = time.perf_counter() - t0
            print(f"G {G} K {K:5d} resident {res}: {dt / (K * reps) * 1e6:7.3f} us/step  frac {0.25 * G ** 3 / (dt / (K * reps)) / 8e12:.3f}  {e.info().kernel_name.decode()}", flush=True)
    e.set_option("resident", 1)

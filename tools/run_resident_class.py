#!/usr/bin/env python3
"""Clustered rule-set at 512^3: resident class kernel against the per-step kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402
G = 512
e = Engine(0)
e.set_option("stats", 0)
e.configure(G)
e.set_rule_strings("moore", "5-7", "4-7", "4", "3-5", "3", "2-4")
e.upload_state(host.random_fill(host.words_per_buffer(G)))
for K in (20, 256, 1024):
    for res in (1, 0):
        e.set_option("resident", res)
        e.step(K); e.synchronize()
        reps = max(1, int(0.05 / (K * 10e-6)))
        t0 = time.perf_counter()
        for _ in range(reps):
            e.step(K)
        e.synchronize()
        dt = (time.perf_counter() - t0) / (K * reps)
        print(f"K {K:5d} resident {res}: {dt * 1e6:7.3f} us/step  frac {0.25 * G ** 3 / dt / 8e12:.3f}  {e.info().kernel_name.decode()}", flush=True)

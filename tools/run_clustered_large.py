#!/usr/bin/env python3
"""Clustered rule-set through the per-step class kernels (launcher's own choice of form) at 1024^3 and 2048^3."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

grids = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024, 2048]
e = Engine(0)
e.set_option("stats", 0)
for G in grids:
    e.configure(G)
    e.set_rule_strings("moore", "5-7", "4-7", "4", "3-5", "3", "2-4")
    e.upload_state(host.random_fill(host.words_per_buffer(G)))
    steps = int(os.environ.get("STEPS", 256 if G <= 1024 else 32))
    e.step(steps); e.synchronize()
    t0 = time.perf_counter()
    e.step(steps); e.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"G {G}: {dt * 1e6:8.2f} us/step  frac {0.25 * G ** 3 / dt / 8e12:.3f}  {e.info().kernel_name.decode()}", flush=True)

#!/usr/bin/env python3
"""Timing of the class kernels on the clustered rule-set: rolling-window form (each depth) against the plain form."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

grids = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [512, 1024]
e = Engine(0)
e.set_option("stats", 0)
for G in grids:
    e.configure(G)
    e.set_rule_strings("moore", "5-7", "4-7", "4", "3-5", "3", "2-4")
    e.upload_state(host.random_fill(host.words_per_buffer(G)))
    steps = max(16, int(2e-2 / (G ** 3 / 1.2e13)))
    e.set_option("resident", 0)  # the per-step kernels are what is compared here
    for roll, tile, z in ((0, 0, 0), (1, 0, 0), (1, 0, 8), (1, 0, 15), (1, 0, 30), (1, 1, 0), (1, 1, 8), (1, 1, 16), (1, 1, 15), (1, 1, 30), (1, 2, 8), (1, 2, 16), (1, 3, 8), (1, 3, 16)):
        e.set_option("roll", roll)
        e.set_option("roll_tile", tile)
        e.set_option("roll_z", z)
        e.step(steps); e.synchronize()
        t0 = time.perf_counter()
        e.step(steps); e.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"G {G} roll {roll} tile {tile} z {z}: {dt * 1e6:8.2f} us/step  frac {0.25 * G ** 3 / dt / 8e12:.3f}  {e.info().kernel_name.decode()}", flush=True)
    e.set_option("roll", 1); e.set_option("roll_z", 0)

#!/usr/bin/env python3
"""Timing aid: one rank's share of a Z-slabbed grid on one GPU (ghosts are NOT refreshed: timing only)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cellularautomatons3d_amd import Engine, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--ghost", type=int, default=8)
ap.add_argument("--batches", type=int, default=200)
ap.add_argument("--phased", type=int, default=0, help="1: edge phase + interior phase per batch")
a = ap.parse_args()
G, nz = a.grid, a.grid // a.world
e = Engine(0)
e.configure_slab(G, nz, nz, a.ghost)
e.set_rule_strings()
e.upload_state(host.random_fill((G // 32) * G * nz))
def batch():
    if a.phased:
        e.slab_step_phase(a.ghost, 1)
        e.slab_step_phase(a.ghost, 2)
    else:
        e.slab_step(a.ghost)


for _ in range(10):
    batch()
e.synchronize()
t0 = time.perf_counter()
for _ in range(a.batches):
    batch()
t_host = time.perf_counter() - t0
e.synchronize()
t = time.perf_counter() - t0
steps = a.batches * a.ghost
print(f"slab {G}^3/{a.world} ghost {a.ghost} phased {a.phased}: {t / steps * 1e6:.2f} us/step (host enqueue {t_host / steps * 1e6:.2f} us/step); "
      f"ideal 1/{a.world} of the 1-GPU step would be the target")

#!/usr/bin/env python3
"""Timing aid: a 1024^3 grid on ONE GPU as P slabs driven by the group handle (ca3d_group_*: resident slab kernel per slab, ghost planes by
device copies) against the plain full-grid engine's per-step kernels.   tools/run_group_one_gpu.py [P] [ghost] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, EngineGroup, host  # noqa: E402

G = 1024
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 512
st = host.random_fill(host.words_per_buffer(G), seed=5)
with EngineGroup([0] * P) as g:
    g.configure(G, K)
    g.set_rule_strings()
    g.upload_state(st)
    g.step(K * 2)
    g.synchronize()
    t0 = time.perf_counter()
    g.step(steps)
    g.synchronize()
    dt = time.perf_counter() - t0
    print(f"group of {P} slabs on one GPU, ghost {K}: {dt / steps * 1e6:.2f} us per 1024^3 step = {G ** 3 * steps / dt / 1e12:.1f} Tcells/s  ({g.kernel_name(0)})")
e = Engine(0)
e.configure(G)
e.set_rule_strings()
e.upload_state(st)
e.step(64)
e.synchronize()
t0 = time.perf_counter()
e.step(steps)
e.synchronize()
dt = time.perf_counter() - t0
print(f"one full-grid engine: {dt / steps * 1e6:.2f} us per step = {G ** 3 * steps / dt / 1e12:.1f} Tcells/s  ({e.info().kernel_name.decode()})")

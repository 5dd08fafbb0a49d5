#!/usr/bin/env python3
"""Timing aid: the 512^3 resident row-pair kernel alone, long batches (us per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402
e = Engine(0); e.set_option("stats", 0); G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
e.configure(G); e.set_rule_strings(); e.upload_state(host.random_fill(host.words_per_buffer(G)))
K = 4096
e.step(K); e.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); e.step(K); e.synchronize(); best = min(best, (time.perf_counter() - t0) / K * 1e6)
print(f"G {G} K {K}: {best:.4f} us/step  {e.info().kernel_name.decode()}")

#!/usr/bin/env python3
"""Resident multi-step kernel against the per-step kernels: 512^3 / 256^3, default rule, batches of K steps back to back, both
z splits (1: a thread owns all planes of its column; 2: half of them, twice the threads — four waves per SIMD) and tile heights."""
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
    e.set_rule_strings()
    e.upload_state(host.random_fill(host.words_per_buffer(G)))
    for K in (20, 256, 4096):
        for res, rows, zs in ((1, 32, 2), (1, 32, 1), (1, "pair", 1), (1, 16, 2), (1, 16, 1), (0, 32, 1)):
            if G == 256 and rows in (16, "pair"):
                continue
            e.set_option("resident", res)
            pair = int(rows == "pair")
            e.set_option("resident_pair", pair)
            rows = 32 if pair else rows
            if G == 512:
                e.set_option("resident_rows", rows)
            e.set_option("resident_zsplit", zs)
            e.step(K); e.synchronize()
            reps = max(1, int(0.05 / (K * 5e-6)))
            t0 = time.perf_counter()
            for _ in range(reps):
                e.step(K)
            e.synchronize()
            dt = time.perf_counter() - t0
            print(f"G {G} K {K:5d} resident {res} rows {rows} pair {pair} zsplit {zs}: {dt / (K * reps) * 1e6:7.3f} us/step  {e.info().kernel_name.decode()}", flush=True)
    e.set_option("resident", 1)

#!/usr/bin/env python3
"""Per-step kernels over the grids the reference UI offers (any multiple of 32 from 32 to 1024, main_pathtraced.js:268-279, 675-693;
"1000" becomes 992): one JSON line per (grid, rule) — kernel that ran, us per step (HIP events around the batch on the engine's
stream), algorithmic GB/s (0.25 B per cell-step) and its fraction of the 8 TB/s HBM peak. Resident kernels off: this is the path
a one-step-per-frame host takes.   tools/grid_matrix.py [--grids 64,96,...] [--out FILE]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

RULES = {"default": (), "clustered": ("moore", "5-7", "4-7", "4", "3-5", "3", "2-4")}
ap = argparse.ArgumentParser()
ap.add_argument("--grids", default="64,96,128,256,384,512,640,768,992,1024")
ap.add_argument("--rules", default="default,clustered")
ap.add_argument("--out", default="")
ap.add_argument("--option", action="append", default=[])
a = ap.parse_args()
e = Engine(0)
lines = []
for G in (int(g) for g in a.grids.split(",")):
    for rule in a.rules.split(","):
        e.configure(G)
        e.set_rule_strings(*RULES[rule])
        e.set_option("resident", 0)
        for o in a.option:
            k, v = o.split("=")
            e.set_option(k, int(v))
        e.upload_state(host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001))
        steps = max(64, min(4096, int(2e9 / G ** 3)))
        e.step(steps)
        e.synchronize()
        best = 1e30
        for _ in range(3):
            e.step(steps)
            e.synchronize()
            best = min(best, e.stats().gpu_ms * 1e3 / steps)
        gbs = 0.25 * G ** 3 / (best * 1e-6) / 1e9
        rec = {"grid": G, "rule": rule, "kernel": e.info().kernel_name.decode(), "us_per_step": round(best, 3), "gcells_per_s": round(G ** 3 / best / 1e3, 1),
               "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)}, "steps": steps}
        print(json.dumps(rec), flush=True)
        lines.append(rec)
if a.out:
    with open(a.out, "w") as f:
        for r in lines:
            f.write(json.dumps(r) + "\n")

#!/usr/bin/env python3
"""Tuning: the tile form of the rolling-window kernel at one more depth (CA3D_ROLL_ZX=<planes per thread>) against the automatic choice
(Z = 16) on the clustered rule-set; state compared word for word with the automatic choice's after the same steps."""
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
depths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 10, 11, 12, 20, 22, 24, 0]
seed = host.random_fill(host.words_per_buffer(G))
ref = None
for zx in depths:
    if zx: os.environ["CA3D_ROLL_ZX"] = str(zx)
    else: os.environ.pop("CA3D_ROLL_ZX", None)
    e = Engine(0)
    e.set_option("stats", 0)
    e.configure(G)
    e.set_option("resident", 0)
    e.set_rule_strings("moore", "5-7", "4-7", "4", "3-5", "3", "2-4")
    e.upload_state(seed)
    e.step(5); e.synchronize()
    crc = zlib.crc32(e.read_state().tobytes())
    if ref is None: ref = crc
    steps = max(16, int(4e-2 / (G ** 3 / 1.2e13)))
    best = 1e9
    for rep in range(3):
        e.step(steps); e.synchronize()
        t0 = time.perf_counter()
        e.step(steps); e.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    print(f"G {G} zx {zx:2d}: {best * 1e6:8.2f} us/step  frac {0.25 * G ** 3 / best / 8e12:.3f}  {e.info().kernel_name.decode()}  state {'same' if crc == ref else 'DIFFERENT'}", flush=True)
    e.close()

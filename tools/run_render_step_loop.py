#!/usr/bin/env python3
"""A host that steps between converged frames (ca3d_step(1); ca3d_render without target arrays; ...): every frame waits for the step before
it, which waited for the frame before that — nothing is ever in flight beside a frame, so the frame pipeline must cost such a loop
nothing (its frames take the whole chip: ca3d_api.cpp, `beside`). ms per iteration with the pipeline on and off, and the pure render loop."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

G, W, H, spp, n = 512, 1920, 1080, 4, 200
e = Engine(0)
e.configure(G)
e.set_rule_strings("von neumann", "2,4", "1,3,5")
cells = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)
u = host.uniform_block(W, H, host.orbit_camera())
for pipe in (0, 1, 0, 1):
    e.set_option("render_pipeline", pipe)
    for stepping in (True, False):
        e.upload_state(cells)  # every block walks through the same states
        for _ in range(8):
            if stepping:
                e.step(1)
            e.render(u, W, H, spp, readback=False)
        e.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if stepping:
                e.step(1)
            e.render(u, W, H, spp, readback=False)
        e.synchronize()
        print(f"render_pipeline {pipe}, {'step + frame' if stepping else 'frame only  '}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per iteration", flush=True)
e.close()

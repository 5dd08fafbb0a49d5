#!/usr/bin/env python3
"""Timing aid: sparse volumes through the scheduled and the plain renderer (empty-space skipping active)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

G = 512
W, H = 1920, 1080
e = Engine(0)
e.configure(G)
e.set_rule_strings()
scenes = [("single seed + 30 steps, default pose", None, host.camera_matrix()),
          ("random density 2^-13, oblique pose", 12, host.orbit_camera()),
          ("random density 2^-15, oblique pose", 14, host.orbit_camera())]
for name, rounds, cam in scenes:
    if rounds is None:
        e.upload_state(host.initial_state(G))
        e.step(30)
    else:
        e.upload_state(host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=rounds))
    u = host.uniform_block(W, H, cam)
    for sched in ((1,) if "--trace" in sys.argv else (1, 0)):
        e.set_option("render_sched", sched)
        e.render(u, W, H, 4, readback=False)
        e.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            e.render(u, W, H, 4, readback=False)
        e.synchronize()
        dt = (time.perf_counter() - t0) / 10
        st = e.render_stats()
        print(f"{name}, sched {sched}: {dt * 1e3:.3f} ms, visits per primary {st.primary_cell_visits / st.primary_rays:.2f}, shadow rays {st.shadow_rays}", flush=True)
    if "--trace" in sys.argv:
        break

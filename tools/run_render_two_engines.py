#!/usr/bin/env python3
"""Probe: how much of a frame's time is idle chip that a SECOND frame in flight could use? Two engines (each its own stream and scratch)
draw the bench's dense scene alternately from one host thread; frames per second of the pair against one engine alone."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

G = 512
cells = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)
engs = []
for i in range(3):
    e = Engine(0)
    e.configure(G)
    e.set_rule_strings()
    e.upload_state(cells)
    engs.append(e)
for size, spp in (("1920x1080", 4), ("3840x2160", 4), ("1920x1080", 1)):
    W, H = (int(v) for v in size.split("x"))
    u = host.uniform_block(W, H, host.orbit_camera())
    for n in (1, 2, 3):
        use = engs[:n]
        for e in use:
            e.render(u, W, H, spp, readback=False); e.render(u, W, H, spp, readback=False)
        for e in use: e.synchronize()
        frames = 60
        t0 = time.perf_counter()
        for f in range(frames):
            use[f % n].render(u, W, H, spp, readback=False)
        for e in use: e.synchronize()
        dt = time.perf_counter() - t0
        print(f"{size} @ {spp} spp, {n} frame(s) in flight: {dt / frames * 1e3:.3f} ms per frame", flush=True)

#!/usr/bin/env python3
"""Timing aid: the bench's dense render scene through the three forms of the converged-frame renderer, same engine, same frame
(plain kernel; in-wave scheduled kernel; ray-stream passes) + a bit-equality check between them."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cellularautomatons3d_amd import Engine, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--density-rounds", type=int, default=4)
ap.add_argument("--check", type=int, default=1)
a = ap.parse_args()
W, H = (int(v) for v in a.size.lower().split("x"))
eng = Engine(0)
eng.configure(a.grid)
eng.set_rule_strings()
eng.upload_state(host.random_fill(host.words_per_buffer(a.grid), seed=0xCA3D0001, and_rounds=a.density_rounds))
u = host.uniform_block(W, H, host.orbit_camera())
ref = None
for name, opts in (("plain", dict(render_sched=0)), ("in-wave", dict(render_sched=1, render_stream=0)), ("stream", dict(render_sched=1, render_stream=1)),
                   ("stream+check", dict(render_sched=1, render_stream=1, render_stream_check=1))):
    if name == "stream+check" and not a.check:
        continue
    for k, v in {"render_stream_check": 0, **opts}.items():
        eng.set_option(k, v)
    out = eng.render(u, W, H, a.spp)
    st = eng.render_stats()
    key = (st.shadow_rays, st.primary_cell_visits, st.shadow_cell_visits)
    if ref is None:
        ref = (out, key)
    same = all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(out, ref[0])) and key == ref[1]
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.frames):
        eng.render(u, W, H, a.spp, readback=False)
    eng.synchronize()
    dt = time.perf_counter() - t0
    st = eng.render_stats()
    print(f"{name:13s} {W}x{H} @ {a.spp} spp: {dt / a.frames * 1e3:.3f} ms per frame (kernel {st.gpu_ms:.3f} ms), same frame as plain: {same}, stats {key}", flush=True)
eng.close()

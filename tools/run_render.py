#!/usr/bin/env python3
"""Timing / profiling aid: the bench's dense render scene only (512^3, hashed fill density 2^-5, oblique pose), N frames of
one size, nothing else on the GPU — so that a rocprofv3 pass over this command holds one kind of dispatch."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--sched", type=int, default=1)
ap.add_argument("--density-rounds", type=int, default=4)
ap.add_argument("--option", action="append", default=[], help="name=value engine options")
ap.add_argument("--literal", type=int, default=0, help="1: the reference's own frame (render_mode 1), one sample per pixel, elapsedTime advancing")
a = ap.parse_args()
W, H = (int(v) for v in a.size.lower().split("x"))
eng = Engine(0)
eng.configure(a.grid)
eng.set_rule_strings()
eng.upload_state(host.random_fill(host.words_per_buffer(a.grid), seed=0xCA3D0001, and_rounds=a.density_rounds))
eng.set_option("render_sched", a.sched)
for o in a.option:
    k, v = o.split("=")
    eng.set_option(k, int(v))
vm = host.orbit_camera()
if a.literal:
    a.spp = 1
    eng.set_render_mode(True)
    eng.reset_render_history()
frame_u = lambda i: host.uniform_block(W, H, vm, elapsed_time=0.5 + 0.01 * i, prev_view_mat=vm) if a.literal else host.uniform_block(W, H, vm)
for i in range(3):
    eng.render(frame_u(i), W, H, a.spp, readback=False)
eng.synchronize()
t0 = time.perf_counter()
for i in range(a.frames):
    eng.render(frame_u(10 + i), W, H, a.spp, readback=False)
eng.synchronize()
dt = time.perf_counter() - t0
st = eng.render_stats()
print(f"{W}x{H} @ {a.spp} spp, sched {a.sched}: {dt / a.frames * 1e3:.3f} ms per frame (kernel {st.gpu_ms:.3f} ms), "
      f"{(st.primary_rays + st.shadow_rays) * a.frames / dt / 1e6:.0f} Mray/s, primary {st.primary_rays} shadow {st.shadow_rays}, "
      f"visits per primary {st.primary_cell_visits / max(1, st.primary_rays):.2f} per shadow {st.shadow_cell_visits / max(1, st.shadow_rays):.2f}")
eng.close()

#!/usr/bin/env python3
"""Two converged frames in flight (option render_pipeline) against one: ms per frame of the bench's dense scene and of the start-up scene,
and the frame the pipelined engine leaves in its targets compared byte for byte with a frame read back through host pointers (never pipelined)."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
G = 512
e = Engine(0)
e.configure(G)
e.set_rule_strings()
dense = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)
for name in ("dense", "start-up"):
    if name == "dense":
        e.upload_state(dense)
        pose = host.orbit_camera()
    else:
        e.upload_state(host.initial_state(G)); e.step(30)
        pose = host.camera_matrix()
    for size, spp in (("1920x1080", 4), ("3840x2160", 4), ("1920x1080", 1)):
        W, H = (int(v) for v in size.split("x"))
        u = host.uniform_block(W, H, pose)
        ref = e.render(u, W, H, spp)  # with host pointers: never pipelined
        for pipe in (0, 1, 0, 1):
            e.set_option("render_pipeline", pipe)
            for _ in range(3): e.render(u, W, H, spp, readback=False)
            e.synchronize()
            frames = 60
            t0 = time.perf_counter()
            for _ in range(frames): e.render(u, W, H, spp, readback=False)
            e.synchronize()
            dt = time.perf_counter() - t0
            st = e.render_stats()
            got = []
            for which in (0, 1, 2):
                ptr, nbytes = e.render_target(which)
                e.synchronize()
                host_buf = np.empty(nbytes, dtype=np.uint8)
                assert hip.hipMemcpy(ctypes.c_void_p(host_buf.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) == 0
                got.append(host_buf)
            same = all(np.array_equal(g, np.ascontiguousarray(r).view(np.uint8).ravel()) for g, r in zip(got, ref))
            print(f"{name:8s} {size} @ {spp} spp, pipeline {pipe}: {dt / frames * 1e3:.3f} ms per frame (last frame's kernels {st.gpu_ms:.3f} ms), targets {'same' if same else 'DIFFERENT'}", flush=True)
e.close()

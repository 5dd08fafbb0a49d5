"""Diagnostics: which form of the converged frame differs from the plain kernel on a grid that is not a power of two."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib as ol
from cellularautomatons3d_amd import Engine, host

G = int(sys.argv[1]) if len(sys.argv) > 1 else 96
W, H, spp = 320, 180, 4
cells = host.random_fill(host.words_per_buffer(G), seed=7, and_rounds=4)
u = host.uniform_block(W, H, host.orbit_camera())
r = ol.Rules.from_strings()
with Engine(0) as eng:
    eng.configure(G)
    eng.set_rules(r.main, r.edges, r.corners, r.survive, r.born)
    eng.upload_state(cells)
    eng.set_option("render_sched", 0)
    p0, l0, d0 = eng.render(u, W, H, spp)
    olight = ol.render(cells, G, u, W, H, spp)[0]
    print("plain vs oracle", (np.abs(l0.astype(np.float32)[..., :3] - olight[..., :3]).max(-1) <= 2e-3).mean())
    eng.set_option("render_sched", 1)
    for stream, check, bricks in ((0, 0, 1), (1, 0, 0), (1, 1, 0), (1, 0, 1), (1, 1, 1)):
        eng.set_option("render_stream", stream)
        eng.set_option("render_frame_bricks", bricks)
        try:
            eng.set_option("render_stream_check", check)
            p, l, d = eng.render(u, W, H, spp)
            print("stream", stream, "check", check, "bricks", bricks, "pixels differing from plain:", int((p != p0).any(-1).sum()))
        except Exception as e:
            print("stream", stream, "check", check, "bricks", bricks, "error:", e)

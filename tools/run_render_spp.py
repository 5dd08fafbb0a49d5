import sys, time
sys.path.insert(0, ".")
import torch
from cellularautomatons3d_amd import Engine, host
e = Engine(0); G = 512
e.configure(G); e.set_rule_strings()
W, H = 1920, 1080
u = host.uniform_block(W, H, host.orbit_camera())
for scene, cells in (("dense", host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)), ("seed+30", None)):
    if cells is None:
        e.upload_state(host.initial_state(G)); e.step(30)
    else:
        e.upload_state(cells)
    for spp in (1, 4):
        for sched in (0, 1):
            e.set_option("render_sched", sched)
            e.render(u, W, H, spp, readback=False); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8): e.render(u, W, H, spp, readback=False)
            torch.cuda.synchronize()
            print(scene, "spp", spp, "sched", sched, "ms/frame %.3f" % ((time.perf_counter() - t0) / 8 * 1e3))

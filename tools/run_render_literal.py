import sys, time
sys.path.insert(0, ".")
import torch
from cellularautomatons3d_amd import Engine, host
e = Engine(0); G = 512
e.configure(G); e.set_rule_strings()
W, H = 1920, 1080
vm = host.orbit_camera()
for scene, cells in (("dense", host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)), ("seed+30", None)):
    if cells is None:
        e.upload_state(host.initial_state(G)); e.step(30)
    else:
        e.upload_state(cells)
    e.set_render_mode(True)
    for i in range(3): e.render(host.uniform_block(W, H, vm, elapsed_time=0.5 + 0.01 * i, prev_view_mat=vm), W, H, 1, readback=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(8): e.render(host.uniform_block(W, H, vm, elapsed_time=0.6 + 0.01 * i, prev_view_mat=vm), W, H, 1, readback=False)
    torch.cuda.synchronize()
    print(scene, "literal frame ms %.3f" % ((time.perf_counter() - t0) / 8 * 1e3), "kernel ms %.3f" % e.render_stats().gpu_ms)
    e.set_render_mode(False)

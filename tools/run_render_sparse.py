import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from cellularautomatons3d_amd import Engine, host
e = Engine(0)
G = 512
e.configure(G); e.set_rule_strings()
e.upload_state(host.initial_state(G))
e.step(30)
W, H = 1920, 1080
for name, vm in (("default pose", host.camera_matrix()), ("oblique", host.orbit_camera())):
    u = host.uniform_block(W, H, vm)
    e.render(u, W, H, 4, readback=False); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): e.render(u, W, H, 4, readback=False)
    torch.cuda.synchronize()
    st = e.render_stats()
    print(name, "ms/frame %.3f" % ((time.perf_counter() - t0) / 5 * 1e3), "visits/primary %.1f" % (st.primary_cell_visits / st.primary_rays), "shadow rays", st.shadow_rays)

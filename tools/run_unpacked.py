import sys, numpy as np
sys.path.insert(0, ".")
from cellularautomatons3d_amd import Engine, host, LAYOUT_UNPACKED
e = Engine(0)
for G in (256, 512):
    e.configure(G, LAYOUT_UNPACKED)
    e.set_rule_strings()
    st = (host.random_fill(G**3 // 1, seed=1) & 1).astype(np.uint32)
    e.upload_state(st)
    e.step(2); e.synchronize()
    e.step(10); s = e.stats()
    us = s.gpu_ms * 1e3 / 10
    print(G, "us/step", us, "GB/s algorithmic", 8.0 * G**3 / us / 1e3)

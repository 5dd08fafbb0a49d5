import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as ol
from cellularautomatons3d_amd import Engine, host
from gpu_common import rules, set_rules
G = 2048
e = Engine(0)
e.configure(G)
st = host.random_fill(host.words_per_buffer(G), seed=7)
for name in ("default", "clustered"):
    r = rules(name)
    set_rules(e, r)
    e.upload_state(st)
    e.step(2); e.synchronize()
    t0 = time.perf_counter(); e.step(8); e.synchronize(); dt = (time.perf_counter() - t0) / 8
    e.upload_state(st); e.step(2)
    got = e.read_state()
    want = ol.packed_run(G, st, r, 2)
    print(name, e.info().kernel_name.decode(), "match", bool(np.array_equal(got, want)), "us/step %.1f" % (dt * 1e6), "frac %.3f" % (0.25 * G**3 / dt / 8e12), flush=True)

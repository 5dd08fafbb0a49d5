import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as ol
from cellularautomatons3d_amd import Engine, host
e = Engine(0); e.set_option("stats", 0)
G = 64
e.configure(G); e.set_rule_strings(); st = host.random_fill(host.words_per_buffer(G)); e.upload_state(st)
e.step(37)
print("parity 37 steps:", np.array_equal(e.read_state(), ol.packed_run(G, st, ol.Rules.from_strings(), 37)))
for res in (1, 0):
    e.set_option("resident", res)
    for K in (8, 256, 4096):
        e.step(K); e.synchronize()
        reps = max(1, int(0.05 / (K * 2e-6)))
        t0 = time.perf_counter()
        for _ in range(reps): e.step(K)
        e.synchronize()
        print("64^3 resident", res, "K", K, round((time.perf_counter() - t0) / (K * reps) * 1e6, 3), "us/step", e.info().kernel_name.decode())

import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from cellularautomatons3d_amd import Engine, host
from gpu_common import RULESETS, rules, set_rules
e = Engine(0)
for G in (384, 640):
    for name in RULESETS:
        r = rules(name); e.configure(G); set_rules(e, r)
        st = host.random_fill(host.words_per_buffer(G), seed=G, and_rounds=2)
        e.upload_state(st); e.step(1); w1 = ol.packed_step(G, st, r)
        ok1 = np.array_equal(e.read_state(), w1)
        e.step(4); ok2 = np.array_equal(e.read_state(), ol.packed_run(G, w1, r, 4))
        print(G, name, e.info().kernel_name.decode(), ok1, ok2, flush=True)

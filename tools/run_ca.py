#!/usr/bin/env python3
"""Small driver for profiling: runs the packed CA step with the given options (no oracle, no timing harness)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--fused", type=int, default=1)
ap.add_argument("--graph", type=int, default=0)
ap.add_argument("--rule", default="default")
ap.add_argument("--variant", type=int, default=0)
a = ap.parse_args()
e = Engine(0)
e.configure(a.grid)
if a.rule == "default":
    e.set_rule_strings()
else:
    e.set_rule_strings("moore", "5-7", "4-7", "4", "3-5", "3", "2-4")
e.set_option("fused", a.fused)
e.set_option("variant", a.variant)
e.set_option("graph", a.graph)
e.upload_state(host.random_fill(host.words_per_buffer(a.grid)))
e.step(a.steps)
e.synchronize()
e.step(a.steps)
e.synchronize()
s = e.stats()
print("us/step", s.gpu_ms * 1e3 / a.steps, "launches", s.kernel_launches, e.info().kernel_name.decode())

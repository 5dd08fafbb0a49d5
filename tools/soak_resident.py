#!/usr/bin/env python3
"""Soak of the resident multi-step kernels against the per-step kernels (which the test suite pins to the oracle): two engines
step the same seeded state, one through the resident path (batches of uneven lengths, queued and flushed), one kernel per
step; the full states are compared after every round. Start-up rule at 512^3, von Neumann B2,4 / S1,3,5 at 512^3 / 256^3 / 64^3 (the one-workgroup form), clustered rule-set at 512^3 / 256^3."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellularautomatons3d_amd import Engine, host  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CLUSTERED = ("moore", "5-7", "4-7", "4", "3-5", "3", "2-4")
VN_B24_S135 = ("von neumann", "2,4", "1,3,5")  # keeps changing (the start-up rule B1,3 / S0-6 freezes from a random fill)
total_bad = 0
for G, rule, per_round in ((512, (), 20000), (512, VN_B24_S135, 20000), (256, VN_B24_S135, 40000), (64, VN_B24_S135, 100000), (512, CLUSTERED, 4000), (256, CLUSTERED, 12000)):
    a, b = Engine(0), Engine(0)
    for e in (a, b):
        e.configure(G)
        e.set_rule_strings(*rule)
        e.upload_state(host.random_fill(host.words_per_buffer(G), seed=2026, and_rounds=0 if not rule else 1))
    b.set_option("resident", 0)
    name = a.info().kernel_name.decode()
    t0 = time.perf_counter()
    steps = 0
    bad = 0
    for r in range(rounds):
        left = per_round + r  # odd and even totals
        k = 0
        while left:
            n = min(left, (8, 9, 33, 257, 2048, 1000)[k % 6])
            a.step(n)
            left -= n
            k += 1
        b.step(per_round + r)
        steps += per_round + r
        if not np.array_equal(a.read_state(), b.read_state()):
            bad += 1
            print(f"MISMATCH G {G} {name} after {steps} steps", flush=True)
            break
    total_bad += bad
    final = a.read_state()
    live = int(np.unpackbits(final.view(np.uint8)).sum())
    print(f"G {G} {name}: {steps} steps in {rounds} rounds, {'ok' if not bad else 'FAILED'}, {live} live cells at the end ({live / G ** 3:.3f}), recovered launches {a.recovered_launches()}, {time.perf_counter() - t0:.1f} s", flush=True)
    a.close(); b.close()
sys.exit(1 if total_bad else 0)

#!/usr/bin/env python3
"""Experiment: P Z-slabs of ONE grid on ONE GPU, each on its own stream, stepping K sub-steps between ghost refreshes
(device copies). Independent dependency chains running concurrently could hide each other's kernel boundaries.
Timing only (correctness of this schedule is covered by tests/test_gpu_slab.py)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cellularautomatons3d_amd import Engine, host, slab  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--parts", type=int, default=2)
ap.add_argument("--ghost", type=int, default=16)
ap.add_argument("--batches", type=int, default=64)
a = ap.parse_args()
G, P, K = a.grid, a.parts, a.ghost
pw = (G // 32) * G
full = host.random_fill(host.words_per_buffer(G))
engs, streams = [], []
for k in range(P):
    e = Engine(0)
    z0, nz = slab.slab_bounds(G, P, k)
    e.configure_slab(G, z0, nz, K)
    e.set_rule_strings()
    s = torch.cuda.Stream()
    e.set_stream(s.cuda_stream)
    e.upload_state(full[z0 * pw:(z0 + nz) * pw])
    engs.append(e)
    streams.append(s)
names = {"send_low": 0, "send_high": 1, "recv_low": 2, "recv_high": 3}
plans = [slab.halo_plan(k, P) for k in range(P)]


def batch():
    regs = [{n: slab.device_tensor(*e.slab_region(i), 0) for n, i in names.items()} for e in engs]
    evs = []
    for k in range(P):  # everybody's previous batch must be complete before ghosts move
        ev = torch.cuda.Event()
        ev.record(streams[k])
        evs.append(ev)
    for k in range(P):
        with torch.cuda.stream(streams[k]):
            for ev in evs:
                streams[k].wait_event(ev)
            # pull my ghosts from the neighbours' send regions
            if plans[k].recv_high_from is not None:
                regs[k]["recv_high"].copy_(regs[plans[k].recv_high_from]["send_low"])
            if plans[k].recv_low_from is not None:
                regs[k]["recv_low"].copy_(regs[plans[k].recv_low_from]["send_high"])
    evs2 = []
    for k in range(P):  # a neighbour must have pulled my send regions before I overwrite them
        ev = torch.cuda.Event()
        ev.record(streams[k])
        evs2.append(ev)
    for k in range(P):
        for ev in evs2:
            streams[k].wait_event(ev)
        engs[k].slab_step(K)


for _ in range(4):
    batch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.batches):
    batch()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{G}^3 as {P} concurrent slabs, ghost {K}: {dt / (a.batches * K) * 1e6:.2f} us per step (host enqueue {t_host / (a.batches * K) * 1e6:.2f})")

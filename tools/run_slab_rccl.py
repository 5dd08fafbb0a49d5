#!/usr/bin/env python3
"""Timing aid: one rank's share of a Z-slabbed grid with the exchange going through RCCL on ONE GPU (a one-rank
group; the wrap message is sent to itself with ncclSend / ncclRecv). Measures the software cost of the exchange
path and how much of it the overlapped schedule hides — not the xGMI link."""
import argparse
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cellularautomatons3d_amd import host, slab  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--planes", type=int, default=128, help="owned planes of the slab (the grid is NOT covered: timing only)")
ap.add_argument("--ghost", type=int, default=16)
ap.add_argument("--batches", type=int, default=100)
ap.add_argument("--native", type=int, default=0, help="1: the exchange inside the engine (ca3d_slab_run) instead of torch.distributed")
ap.add_argument("--rule", default="default")
ap.add_argument("--resident", type=int, default=1, help="0: per-step slab kernels only")
a = ap.parse_args()
s = socket.socket()
s.bind(("127.0.0.1", 0))
port = s.getsockname()[1]
s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
G = a.grid
for overlap in (False, True):
    if a.native:
        se = slab.NativeSlabEngine(G, 0, 1, ghost=a.ghost, device=0, overlap=overlap)
    else:
        se = slab.SlabEngine(G, 0, 1, ghost=a.ghost, device=0, overlap=overlap, loopback=True)
    # shrink the slab to one rank's share: same kernels and message sizes as rank k of G / planes ranks
    se.engine.configure_slab(G, 0, a.planes, a.ghost)
    se.z0, se.nz = 0, a.planes
    if a.rule == "clustered":
        se.engine.set_rule_strings("moore", "5-7", "4-7", "4", "3-5", "3", "2-4")
    else:
        se.engine.set_rule_strings()
    se.engine.set_stream(se.stream.cuda_stream)
    se.engine.set_option("stats", 0)
    se.engine.set_option("resident", a.resident)
    se.upload_state(host.random_fill((G // 32) * G * a.planes))
    se.run(a.ghost * 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    se.run(a.ghost * a.batches)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"slab {a.planes} planes of {G}^2, ghost {a.ghost}, RCCL loopback ({'engine' if a.native else 'torch'} transport), overlap {overlap}, {se.engine.info().kernel_name.decode()}: {dt / (a.ghost * a.batches) * 1e6:.2f} us/step "
          f"({dt / a.batches * 1e6:.1f} us per batch incl. one exchange of 2 x {a.ghost * G * G // 8 / 2**20:.1f} MiB; host enqueue {t_host / a.batches * 1e6:.1f} us per batch)")
    se.close()
dist.destroy_process_group()

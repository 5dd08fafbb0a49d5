"""Multi-rank Z-slab path on CPU: world_size 2 and 4 over gloo, product halo plan + exchange code, oracle
sub-steps; the gathered slabs must equal the single-grid oracle bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
from cellularautomatons3d_amd import LAYOUT_PACKED32, LAYOUT_UNPACKED, host, slab


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, G, ghost, steps, rule_kw, layout, out_dir, overlap=False):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from slab_oracle_backend import OracleSlab

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rules = ol.Rules.from_strings(**rule_kw)
        if layout == LAYOUT_PACKED32:
            full = host.random_fill(host.words_per_buffer(G), seed=1234)
            pw = (G // 32) * G
        else:
            full = (host.random_fill(G ** 3, seed=1234) & 1).astype(np.uint32)
            pw = G * G
        sl = OracleSlab(G, rank, world, ghost, rules, layout, overlap=overlap)
        sl.upload(full[sl.z0 * pw:(sl.z0 + sl.nz) * pw])
        sl.run(steps)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), sl.owned())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ghost,steps", [(2, 1, 3), (2, 3, 7), (4, 2, 5)])
def test_packed_slabs_over_gloo(tmp_path, world, ghost, steps):
    G = 64
    rule_kw = dict(neighbourhood="moore", born="5-7", survive="4-7", born_edges="4", survive_edges="3-5",
                   born_corners="3", survive_corners="2-4")
    mp.spawn(_worker, args=(world, _free_port(), G, ghost, steps, rule_kw, LAYOUT_PACKED32, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    want = ol.packed_run(G, host.random_fill(host.words_per_buffer(G), seed=1234), ol.Rules.from_strings(**rule_kw), steps)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("world,ghost,steps", [(2, 3, 8), (2, 8, 9)])
def test_packed_slabs_over_gloo_overlapped_schedule(tmp_path, world, ghost, steps):
    """Edge zones -> exchange -> interior (the schedule SlabEngine.run / ca3d_slab_step_phase use), world 2 over
    gloo; ghost 8 on 32-plane slabs is too thin to split in the full batches and splits in the 1-step tail."""
    G = 64
    rule_kw = dict(neighbourhood="moore", born="5-7", survive="4-7", born_edges="4", survive_edges="3-5",
                   born_corners="3", survive_corners="2-4")
    mp.spawn(_worker, args=(world, _free_port(), G, ghost, steps, rule_kw, LAYOUT_PACKED32, str(tmp_path), True), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    want = ol.packed_run(G, host.random_fill(host.words_per_buffer(G), seed=1234), ol.Rules.from_strings(**rule_kw), steps)
    np.testing.assert_array_equal(got, want)


def test_batch_ranges_partition_and_dependencies():
    """The three zones tile the whole shrinking range; every zone's inputs at sub-step s were produced at s-1 by a
    zone that is finished before it (edges before interior), and are not overwritten in between (ping-pong)."""
    for nz, K in [(128, 16), (32, 3), (16, 4), (12, 5), (8, 1)]:
        for dead in (False, True):
            for n in range(1, K + 1):
                prev = None
                for s in range(1, n + 1):
                    r = slab.batch_ranges(nz, K, n, s, dead)
                    lo, hi = r["all"]
                    assert (lo, hi) == (max(s, K if dead else 0), nz + 2 * K - s)
                    assert r["low"][0] == lo and r["high"][1] == hi
                    assert r["low"][1] == r["interior"][0] and r["interior"][1] == r["high"][0]
                    if prev is not None:
                        # an edge chain only needs its own previous range (one plane wider on the open side)
                        assert r["low"][1] + 1 <= prev["low"][1] or r["low"] == r["all"]
                        assert r["high"][0] - 1 >= prev["high"][0] or r["high"][0] == r["high"][1]
                        # the interior needs one plane of each edge from sub-step s-1 ...
                        if r["interior"][0] < r["interior"][1]:
                            assert prev["low"][0] <= r["interior"][0] - 1 < prev["low"][1]
                            assert prev["high"][0] <= r["interior"][1] < prev["high"][1]
                    # ... which later edge sub-steps writing the same buffer (s+1, s+3, ...) leave alone
                    for s2 in range(s + 2, n + 1, 2):
                        r2 = slab.batch_ranges(nz, K, n, s2, dead)
                        if r2["interior"][0] < r2["interior"][1]:
                            assert r2["low"][1] <= r["low"][1] - 2 and r2["high"][0] >= r["high"][0] + 2
                    prev = r
                end = slab.batch_ranges(nz, K, n, n, dead)
                assert end["low"][0] <= K and (end["low"][1] >= 2 * K or end["low"] == end["all"])  # planes sent down are final
                assert end["all"][1] >= nz + K


def test_unpacked_slabs_over_gloo(tmp_path):
    G, world, ghost, steps = 16, 2, 2, 5
    rule_kw = dict(neighbourhood="moore", born="5-7", survive="4-9")
    mp.spawn(_worker, args=(world, _free_port(), G, ghost, steps, rule_kw, LAYOUT_UNPACKED, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    r = ol.Rules.from_strings(**rule_kw)
    cur = (host.random_fill(G ** 3, seed=1234) & 1).astype(np.uint32)
    for _ in range(steps):
        cur = ol.unpacked_step(G, cur, r.main, r.survive, r.born)
    np.testing.assert_array_equal(got, cur)


def test_halo_plan_shapes():
    # packed: open at the bottom, closed at the top
    p0, p3 = slab.halo_plan(0, 4), slab.halo_plan(3, 4)
    assert p0 == slab.HaloPlan(send_low_to=3, send_high_to=1, recv_low_from=None, recv_high_from=1)
    assert p3 == slab.HaloPlan(send_low_to=2, send_high_to=None, recv_low_from=2, recv_high_from=0)
    assert slab.halo_plan(1, 4) == slab.HaloPlan(0, 2, 0, 2)
    # every send has exactly one matching receive
    for layout in (LAYOUT_PACKED32, LAYOUT_UNPACKED):
        for world in (1, 2, 3, 8):
            plans = [slab.halo_plan(r, world, layout) for r in range(world)]
            sends = sorted([(r, p.send_low_to, "hi") for r, p in enumerate(plans) if p.send_low_to is not None]
                           + [(r, p.send_high_to, "lo") for r, p in enumerate(plans) if p.send_high_to is not None])
            recvs = sorted([(p.recv_high_from, r, "hi") for r, p in enumerate(plans) if p.recv_high_from is not None]
                           + [(p.recv_low_from, r, "lo") for r, p in enumerate(plans) if p.recv_low_from is not None])
            assert sends == recvs
    # unpacked: a ring
    assert slab.halo_plan(0, 4, LAYOUT_UNPACKED) == slab.HaloPlan(3, 1, 3, 1)
    assert slab.slab_bounds(1024, 8, 3) == (384, 128)
    with pytest.raises(ValueError):
        slab.slab_bounds(96, 5, 0)


def test_single_rank_slab_self_wrap():
    # world == 1: the high ghost is a local copy of the first planes; no process group needed.
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from slab_oracle_backend import OracleSlab

    G = 64
    rules = ol.Rules.from_strings()
    full = host.random_fill(host.words_per_buffer(G), seed=4)
    sl = OracleSlab(G, 0, 1, 2, rules)
    sl.upload(full)
    sl.run(5)
    np.testing.assert_array_equal(sl.owned(), ol.packed_run(G, full, rules, 5))

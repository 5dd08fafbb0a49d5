"""Slab mode of the HIP engine on one GPU: P engines in one process stand for P ranks, ghosts are moved with
device copies that follow the product halo plan; the result must equal the single-grid run and the oracle."""
import numpy as np
import pytest

import oracle_lib as ol
from cellularautomatons3d_amd import LAYOUT_PACKED32, LAYOUT_UNPACKED, Ca3dError, host, slab
from gpu_common import rules, set_rules

pytestmark = pytest.mark.gpu


def _run_slabs_overlapped(G, P, K, steps, r, layout, full):
    """The overlapped schedule on one GPU: every engine runs its edge phase, then ALL interior phases are enqueued
    and the ghost copies run on another stream at the same time — the interiors must neither read nor disturb the
    ghost planes being written."""
    import torch
    from cellularautomatons3d_amd import SLAB_PHASE_EDGES, SLAB_PHASE_INTERIOR, Engine

    pw = (G // 32) * G if layout == LAYOUT_PACKED32 else G * G
    engs = []
    for k in range(P):
        e = Engine(0)
        z0, nz = slab.slab_bounds(G, P, k)
        e.configure_slab(G, z0, nz, K, layout)
        set_rules(e, r)
        e.upload_state(full[z0 * pw:(z0 + nz) * pw])
        engs.append(e)
    names = {"send_low": 0, "send_high": 1, "recv_low": 2, "recv_high": 3}
    copy_stream = torch.cuda.Stream()

    def exchange():
        regs = [{n: slab.device_tensor(*e.slab_region(i), 0) for n, i in names.items()} for e in engs]
        with torch.cuda.stream(copy_stream):
            for rk in range(P):
                plan = slab.halo_plan(rk, P, layout)
                if plan.send_low_to is not None:
                    regs[plan.send_low_to]["recv_high"].copy_(regs[rk]["send_low"])
                if plan.send_high_to is not None:
                    regs[plan.send_high_to]["recv_low"].copy_(regs[rk]["send_high"])

    for e in engs:
        e.synchronize()
    exchange()
    torch.cuda.synchronize()
    left = steps
    while left > 0:
        k = min(K, left)
        for e in engs:
            e.slab_step_phase(k, SLAB_PHASE_EDGES)
        for e in engs:
            e.synchronize()
        for e in engs:
            e.slab_step_phase(k, SLAB_PHASE_INTERIOR)  # asynchronous: runs while the copies below are in flight
        exchange()
        torch.cuda.synchronize()
        left -= k
    out = np.concatenate([e.read_state() for e in engs])
    for e in engs:
        e.close()
    return out


def _run_slabs(G, P, K, steps, r, layout, full):
    import torch
    from cellularautomatons3d_amd import Engine

    pw = (G // 32) * G if layout == LAYOUT_PACKED32 else G * G
    engs = []
    for k in range(P):
        e = Engine(0)
        z0, nz = slab.slab_bounds(G, P, k)
        e.configure_slab(G, z0, nz, K, layout)
        set_rules(e, r)
        e.upload_state(full[z0 * pw:(z0 + nz) * pw])
        engs.append(e)
    names = {"send_low": 0, "send_high": 1, "recv_low": 2, "recv_high": 3}
    left = steps
    while left > 0:
        k = min(K, left)
        for e in engs:
            e.synchronize()
        regs = [{n: slab.device_tensor(*e.slab_region(i), 0) for n, i in names.items()} for e in engs]
        for rk in range(P):
            plan = slab.halo_plan(rk, P, layout)
            if plan.send_low_to is not None:
                regs[plan.send_low_to]["recv_high"].copy_(regs[rk]["send_low"])
            if plan.send_high_to is not None:
                regs[plan.send_high_to]["recv_low"].copy_(regs[rk]["send_high"])
        torch.cuda.synchronize()
        for e in engs:
            e.slab_step(k)
        left -= k
    out = np.concatenate([e.read_state() for e in engs])
    for e in engs:
        e.close()
    return out


@pytest.mark.parametrize("P,K,steps", [(1, 2, 5), (2, 1, 3), (2, 4, 9), (4, 3, 7), (8, 2, 4)])
@pytest.mark.parametrize("name", ["default", "clustered"])
def test_packed_slabs_equal_full_grid(P, K, steps, name):
    G = 128
    r = rules(name)
    full = host.random_fill(host.words_per_buffer(G), seed=31)
    got = _run_slabs(G, P, K, steps, r, LAYOUT_PACKED32, full)
    np.testing.assert_array_equal(got, ol.packed_run(G, full, r, steps))


@pytest.mark.parametrize("G,P,K,steps,name", [(128, 2, 4, 9, "default"), (128, 4, 3, 7, "clustered"), (128, 8, 2, 5, "default"),
                                                (128, 4, 8, 17, "default"), (512, 4, 6, 13, "default"), (512, 8, 16, 33, "vn_b24_s135"),
                                                (384, 4, 3, 7, "clustered"), (384, 2, 4, 9, "default"), (640, 4, 5, 6, "edges_main")])
def test_packed_slabs_overlapped_schedule(G, P, K, steps, name):
    """Edge phase -> exchange concurrent with the interior phase: equals the full grid (the (128, 4, 8) case is too
    thin to split: the edge phase then runs whole batches). The 384 / 640 cases: slabs of a grid whose rows are 3 / 5 uint4 (the
    rolling-window kernel's whole-rows-per-wave form on plane ranges with a z offset, two ranges per launch, no wrap)."""
    r = rules(name)
    full = host.random_fill(host.words_per_buffer(G), seed=77)
    got = _run_slabs_overlapped(G, P, K, steps, r, LAYOUT_PACKED32, full)
    np.testing.assert_array_equal(got, ol.packed_run(G, full, r, steps))


def test_unpacked_slabs_overlapped_schedule():
    G, P, K, steps = 128, 2, 3, 7
    r = rules("default")
    full = (host.random_fill(G ** 3, seed=3) & 1).astype(np.uint32)
    got = _run_slabs_overlapped(G, P, K, steps, r, LAYOUT_UNPACKED, full)
    cur = full
    for _ in range(steps):
        cur = ol.unpacked_step(G, cur, r.main, r.survive, r.born)
    np.testing.assert_array_equal(got, cur)


@pytest.mark.parametrize("G,P,K,steps,name,overlapped", [
    (1024, 8, 1, 3, "default", False),      # BASELINE configs[3]: 1024^3 over 8 slabs, one-voxel halo every step
    (1024, 8, 16, 33, "default", False),    # the deep-halo schedule the bench runs (exchange every K steps)
    (1024, 8, 16, 18, "clustered", True),   # class kernels, two-range edge launches + interior
    (2048, 8, 8, 9, "clustered", True),     # BASELINE configs[4]: 2048^3 clustered, halo overlapped with compute
])
def test_packed_slabs_at_the_baseline_widths(G, P, K, steps, name, overlapped):
    """Slab mode at the grid widths BASELINE's multi-GPU configurations name: the slab kernels (`ca_packed_vn<3|4, ..>`,
    the class kernels' two-range launches) on 1024- and 2048-wide planes, P = 8 slabs on one GPU with device copies
    following the product halo plan, against the oracle's full-grid run."""
    r = rules(name)
    full = host.random_fill(host.words_per_buffer(G), seed=1000 + G + K)
    run = _run_slabs_overlapped if overlapped else _run_slabs
    got = run(G, P, K, steps, r, LAYOUT_PACKED32, full)
    want = ol.packed_run(G, full, r, steps)
    assert np.array_equal(got, want)  # (assert_array_equal would format 1 GiB arrays on failure)


@pytest.mark.parametrize("P,K,steps,name", [(8, 16, 25, "default"), (8, 32, 41, "vn_b24_s135"), (4, 16, 25, "default"), (4, 8, 19, "vn_b24_s135")])
def test_resident_slab_kernel(P, K, steps, name):
    """The slab form of the resident kernel (ca_resident_kernel.inc): a rank's share of a 1024^3 grid — 128 owned planes
    + 2 K ghost planes — runs a batch of K sub-steps in ONE launch, tiles in registers. P = 8 engines on one GPU, ghosts
    moved by device copies after every batch (product halo plan); batches shorter than resident_min and the per-step
    kernels (option off) must give the same state; all against the oracle's full-grid run. Rank 7's high ghost holds the
    copy of global plane 0 (dead plane below it, a run-time plane index inside a tile); rank 0's low ghost is never valid.
    P = 4 (round 4): 256 + 2 K planes, 36 / 34 planes per tile layer — the form that reads the rows either side of a thread's own four
    planes at a time inside its main pass."""
    from cellularautomatons3d_amd import Engine

    G = 1024
    r = rules(name)
    full = host.random_fill(host.words_per_buffer(G), seed=4242 + K)
    probe = Engine(0)
    probe.configure_slab(G, 0, G // P, K)
    set_rules(probe, r)
    assert probe.info().kernel_name == b"ca_resident_slab_vn(jit)", probe.info().kernel_name
    probe.close()
    got = _run_slabs(G, P, K, steps, r, LAYOUT_PACKED32, full)
    want = ol.packed_run(G, full, r, steps)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("overlap", [True, False])
def test_slab_engine_over_rccl_loopback(overlap):
    """The product's exchange through the real transport on one GPU: a one-rank RCCL group, the wrap message (rank
    0's first planes -> its own high ghost) sent to itself with ncclSend / ncclRecv on the engine's device memory,
    posted between the edge and interior phases and waited for on the engine's stream."""
    import socket

    import torch
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        G, K, steps = 256, 4, 14
        r = rules("default")
        full = host.random_fill(host.words_per_buffer(G), seed=91)
        se = slab.SlabEngine(G, 0, 1, ghost=K, device=0, overlap=overlap, loopback=True)
        set_rules(se.engine, r)
        se.upload_state(full)
        se.run(steps)
        se.engine.synchronize()
        got = se.engine.read_state()
        want = ol.packed_run(G, full, r, steps)
        np.testing.assert_array_equal(got, want)
        # the renderer's volume gather through ncclAllGather on the engines' device buffers
        sr = slab.SlabRenderer(se)
        sr.gather_volume()
        torch.cuda.synchronize()
        vol = slab.device_tensor(*sr.full.device_buffer(0), 0).cpu().numpy().view(np.uint32)
        np.testing.assert_array_equal(vol, want)
        sr.close()
        se.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap,name,G,K,steps", [(True, "default", 256, 4, 14), (False, "default", 256, 4, 14), (True, "clustered", 256, 3, 7), (False, "clustered", 256, 3, 7),
                                                     (True, "default", 1024, 16, 33)])
def test_native_rccl_transport_loopback(overlap, name, G, K, steps):
    """The halo exchange inside the engine (ca3d_slab_comm_init / ca3d_slab_run: libca3d.so loads librccl and issues the
    grouped ncclSend / ncclRecv itself) on a one-rank communicator: the wrap message (rank 0's first planes -> its own high
    ghost) goes through RCCL on the engine's memory, unsplit and under the interior phase; then ncclAllGather of the
    volume into a full-grid engine (ca3d_slab_gather)."""
    from cellularautomatons3d_amd import Engine

    r = rules(name)
    full = host.random_fill(host.words_per_buffer(G), seed=93)
    se = slab.NativeSlabEngine(G, 0, 1, ghost=K, device=0, overlap=overlap)
    try:
        set_rules(se.engine, r)
        se.upload_state(full)
        se.run(steps)
        se.run(1)
        se.engine.synchronize()
        want = ol.packed_run(G, full, r, steps + 1)
        assert np.array_equal(se.engine.read_state(), want)
        ci = se.engine.comm_info()  # what the communicator itself reports (bench.py's "rccl" block on every N > 1 line)
        assert ci["comm_ranks"] == 1 and ci["comm_rank"] == 0 and ci["comm_device"] == 0 and ci["device"] == 0
        assert len(ci["pci_bus_id"].split(":")) == 3, ci
        with Engine(0) as vol:
            assert vol.comm_info()["comm_ranks"] == -1 and vol.comm_info()["pci_bus_id"] == ci["pci_bus_id"]  # no communicator: the device only
            vol.configure(G)
            vol.set_rule_strings()
            vol.upload_state(np.zeros(host.words_per_buffer(G), dtype=np.uint32))
            se.engine.slab_gather(vol)
            se.engine.synchronize()
            assert np.array_equal(vol.read_state(), want)
    finally:
        se.close()


def test_slab_renderer_single_rank_equals_plain_render():
    """SlabRenderer (volume gather + band render + band gather) on a one-rank chain: the frame of the stepped slab
    state equals the frame a plain engine renders of the same state."""
    from cellularautomatons3d_amd import Engine

    G, K, steps, W, H = 128, 4, 6, 240, 136
    r = rules("default")
    full = host.random_fill(host.words_per_buffer(G), seed=17, and_rounds=2)
    se = slab.SlabEngine(G, 0, 1, ghost=K, device=0)
    set_rules(se.engine, r)
    se.upload_state(full)
    se.run(steps)
    sr = slab.SlabRenderer(se)
    u = host.uniform_block(W, H, host.orbit_camera())
    frame = sr.render(u, W, H, 4)
    state = se.engine.read_state()
    sr.close()
    se.close()
    np.testing.assert_array_equal(state, ol.packed_run(G, full, r, steps))
    e = Engine(0)
    try:
        e.configure(G)
        set_rules(e, r)
        e.upload_state(state)
        want, _, _ = e.render(u, W, H, 4)
    finally:
        e.close()
    np.testing.assert_array_equal(frame, want)
    assert slab.band_rows(1080, 8, 0) == (0, 128) and slab.band_rows(1080, 8, 7) == (944, 1080)
    assert [slab.band_rows(40, 4, k) for k in range(4)] == [(0, 0), (0, 16), (16, 32), (32, 40)]


def test_slab_phase_order_is_enforced():
    from cellularautomatons3d_amd import SLAB_PHASE_EDGES, SLAB_PHASE_INTERIOR, Ca3dError, Engine

    e = Engine(0)
    try:
        e.configure_slab(128, 0, 64, 2)
        e.set_rule_strings()
        e.upload_state(host.random_fill((128 // 32) * 128 * 64))
        with pytest.raises(Ca3dError):
            e.slab_step_phase(2, SLAB_PHASE_INTERIOR)  # no edge phase pending
        e.slab_step_phase(2, SLAB_PHASE_EDGES)
        with pytest.raises(Ca3dError):
            e.slab_step(1)  # an edge phase is pending
        with pytest.raises(Ca3dError):
            e.slab_step_phase(1, SLAB_PHASE_INTERIOR)  # different length
        e.slab_step_phase(2, SLAB_PHASE_INTERIOR)
        assert e.info().step == 2
    finally:
        e.close()


@pytest.mark.parametrize("name", ["default", "vn_b24_s135", "clustered"])
def test_packed_slabs_512_two_planes_per_thread(name):
    """512^3 slabs run the kernels' 2 / 4-planes-per-thread variants over plane ranges of every parity (the
    shrinking ghost zones): the shifted last z-run must reproduce the full grid."""
    G, P, K, steps = 512, 4, 3, 7
    r = rules(name)
    full = host.random_fill(host.words_per_buffer(G), seed=5)
    got = _run_slabs(G, P, K, steps, r, LAYOUT_PACKED32, full)
    np.testing.assert_array_equal(got, ol.packed_run(G, full, r, steps))


@pytest.mark.parametrize("G", [32, 128])
def test_unpacked_slabs_equal_full_grid(G):
    P, K, steps = 2, 2, 5
    r = ol.Rules.from_strings("moore", "5-7", "4-9")
    full = (host.random_fill(G ** 3, seed=8) & 1).astype(np.uint32)
    got = _run_slabs(G, P, K, steps, r, LAYOUT_UNPACKED, full)
    cur = full
    for _ in range(steps):
        cur = ol.unpacked_step(G, cur, r.main, r.survive, r.born)
    np.testing.assert_array_equal(got, cur)


def test_slab_engine_single_rank_uses_product_exchange():
    G = 256
    r = rules("clustered")
    full = host.random_fill(host.words_per_buffer(G), seed=2)
    se = slab.SlabEngine(G, 0, 1, ghost=4)
    set_rules(se.engine, r)
    se.upload_state(full)
    se.run(10)
    got = se.engine.read_state()
    se.close()
    np.testing.assert_array_equal(got, ol.packed_run(G, full, r, 10))


def test_slab_step_limits():
    from cellularautomatons3d_amd import Ca3dError, Engine

    e = Engine(0)
    try:
        with pytest.raises(Ca3dError):
            e.configure_slab(128, 0, 64, 0)
        with pytest.raises(Ca3dError):
            e.configure_slab(128, 96, 64, 1)
        e.configure_slab(128, 0, 64, 2)
        set_rules(e, rules("default"))
        e.upload_state(np.zeros(64 * 128 * 4, dtype=np.uint32))
        with pytest.raises(Ca3dError):
            e.slab_step(3)
        with pytest.raises(Ca3dError):
            e.step(1)
    finally:
        e.close()


@pytest.mark.parametrize("overlap", ["on", "off"])
def test_bench_two_ranks_on_one_gpu_gloo_rehearsal(overlap):
    """bench.py's N > 1 path end to end with two processes sharing GPU 0 and gloo as a stand-in transport:
    slab engines, product halo plan, exchange every `ghost` steps, final state checked against the oracle."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "21", "--warmup", "5",
           "--grid", "256", "--ghost", "4", "--backend", "gloo", "--device-map", "0,0", "--check", "--no-cpu-baseline", "--overlap", overlap, "--min-seconds", "0",
           "--multi-render", "--render-size", "640x360", "--render-frames", "2"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["oracle_match"] is True and d["scaling"] == "strong"
    # what the ranks really ran on, from the engines: two ranks, ONE device here (a real 2-GPU run must show two bus ids)
    assert d["rccl"]["ranks"] == 2 and d["rccl"]["distinct_devices"] == 1 and len(d["rccl"]["devices"]) == 2
    assert d["rccl"]["devices"][0]["pci_bus_id"] == d["rccl"]["devices"][1]["pci_bus_id"] != ""
    assert d["rccl"]["devices"][0]["pid"] != d["rccl"]["devices"][1]["pid"]
    assert d["render"]["frame_match"] is True and d["render"]["value"] > 0  # the two ranks' bands == one GPU's frame
    assert d["roofline"]["kernel"].startswith("ca_packed_vn")


def test_bench_four_ranks_rehearsal_picks_the_ghost_depth():
    """What the driver's `--gpus 4` run does, rehearsed with four processes sharing GPU 0 over gloo: 1024^3 default rule, no ghost depth
    given -> bench.py picks 16 (auto_ghost: the depth that keeps a quarter of the grid on the resident slab kernel), every rank's slab is
    verified against the oracle past one exchange. The resident kernels themselves are OFF here: a resident launch needs every CU of
    the device, and four processes on one GPU would each hold part of it until their waits time out (that form at this depth is covered
    in one process by test_resident_slab_kernel and test_engine_group_single_thread_split)."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "32", "--warmup", "16", "--resident", "0",
           "--backend", "gloo", "--device-map", "0,0,0,0", "--no-cpu-baseline", "--min-seconds", "0", "--no-schedule-compare"]
    import time

    t_run = time.perf_counter()
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 4 and d["config"]["grid"] == 1024 and "ghost 16" in d["config"]["parallelism"]
    assert d["verified"]["oracle_match"] is True
    assert d["roofline"]["kernel"].startswith("ca_packed_vn"), d["roofline"]["kernel"]
    # the oracle leg of the verification is what a `--gpus N` run spends most of its untimed time on: printed, and bounded — the driver's
    # run at N = 8 (1024^3 x 40 steps on cores / 8 threads per rank) must stay well inside its 600 s limit
    print(f"bench --gpus 4 rehearsal: oracle leg {d['verified']['oracle_seconds']} s on {d['verified']['oracle_threads_per_rank']} threads per rank, "
          f"{d['verified']['steps']} steps; whole run {time.perf_counter() - t_run:.1f} s")
    assert d["verified"]["oracle_seconds"] < 120.0 and time.perf_counter() - t_run < 400.0


def test_bench_plain_invocation_spawns_its_ranks():
    """`python bench.py --gpus 2 ...` with no launcher around it: the parent starts the two ranks itself (before it has
    touched the GPU) and rank 0's JSON line is the only stdout; same rehearsal set-up as above (gloo, both ranks on
    GPU 0), final state checked against the oracle."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "3", "--grid", "256", "--ghost", "4",
           "--backend", "gloo", "--device-map", "0,0", "--check", "--no-cpu-baseline", "--min-seconds", "0"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["oracle_match"] is True and d["reps"] == 1 and d["config"]["grid"] == 256


@pytest.mark.parametrize("name,G,P,K,steps", [("default", 256, 4, 8, (8, 13, 3)), ("clustered", 256, 8, 4, (9, 4)), ("default", 1024, 8, 16, (33,)),
                                              ("default", 1024, 4, 16, (21,)), ("life2d", 128, 2, 5, (11,))])
def test_engine_group_single_thread_split(name, G, P, K, steps):
    """`ca3d_group_*` (SURVEY 8(b)'s `ca3d_create(device_ids, n_devices, ...)`): ONE host thread drives P slab engines — here all on
    GPU 0, ghost planes as device copies ordered by events — through batches of <= K sub-steps; the full-grid state equals the
    oracle's after every call. 1024^3 over 8 slabs of 128 + 2 x 16 planes is BASELINE configs[3]'s split (the resident slab
    kernel runs the batches); over 4 slabs of 256 + 2 x 16 planes it is the resident slab kernel's 36-planes-per-tile-layer form."""
    from cellularautomatons3d_amd import EngineGroup

    rr = rules(name)
    st = host.random_fill(host.words_per_buffer(G), seed=777, and_rounds=1)
    with EngineGroup([0] * P) as g:
        g.configure(G, K)
        g.set_rules(rr.main, rr.edges, rr.corners, rr.survive, rr.born)
        g.upload_state(st)
        want = st
        for n in steps:
            g.step(n)
            want = ol.packed_run(G, want, rr, n)
            np.testing.assert_array_equal(g.read_state(), want, err_msg=f"after {n} more steps")
        if G == 1024:
            assert g.kernel_name(3).startswith("ca_resident_slab"), g.kernel_name(3)


def test_engine_group_unpacked_ring_and_rccl_transport():
    """The legacy layout is a true ring (both ghosts of every rank are filled); the RCCL transport of a group — communicators
    from ncclCommInitAll, every exchange one ncclGroupStart / End — with the one rank a single GPU allows."""
    from cellularautomatons3d_amd import EngineGroup, LAYOUT_UNPACKED

    G, P, K = 64, 4, 3
    rr = rules("default")
    cells = (host.random_fill(G * G * G // 32, seed=5).view(np.uint8)[:, None] >> np.arange(8, dtype=np.uint8) & 1).reshape(-1).astype(np.uint32)
    with EngineGroup([0] * P) as g:
        g.configure(G, K, LAYOUT_UNPACKED)
        g.set_rules(rr.main, rr.edges, rr.corners, rr.survive, rr.born)
        g.upload_state(cells)
        g.step(7)
        want = cells
        for _ in range(7):
            want = ol.unpacked_step(G, want, rr.main, rr.survive, rr.born)
        np.testing.assert_array_equal(g.read_state(), want)
    G = 256
    st = host.random_fill(host.words_per_buffer(G), seed=778)
    with EngineGroup([0]) as g:
        g.configure(G, 8)
        g.set_rules(rr.main, rr.edges, rr.corners, rr.survive, rr.born)
        g.set_option("transport", 1)
        g.upload_state(st)
        g.step(19)
        np.testing.assert_array_equal(g.read_state(), ol.packed_run(G, st, rr, 19))
    with EngineGroup([0, 0]) as g, pytest.raises(Ca3dError, match="one device per slab"):
        g.set_option("transport", 1)


def test_engine_group_renders_the_frame_in_bands():
    """ca3d_group_render: the full volume gathered to every rank, each renders its band of image rows — bit-identical to one
    engine's frame."""
    from cellularautomatons3d_amd import Engine, EngineGroup

    G, P = 256, 4
    rr = rules("default")
    st = host.random_fill(host.words_per_buffer(G), seed=779, and_rounds=4)
    W, H = 640, 360
    u = host.uniform_block(W, H, host.orbit_camera())
    with EngineGroup([0] * P) as g, Engine(0) as e:
        g.configure(G, 4)
        g.set_rules(rr.main, rr.edges, rr.corners, rr.survive, rr.born)
        g.upload_state(st)
        g.step(6)
        e.configure(G)
        e.set_rules(rr.main, rr.edges, rr.corners, rr.survive, rr.born)
        e.upload_state(st)
        e.step(6)
        a = g.render(u, W, H, 4)
        b = e.render(u, W, H, 4)
        for x, y, what in zip(a, b, ("presentation", "light", "depth")):
            np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8), err_msg=what)
        assert a[0][..., :3].any()
        g.step(3)  # stepping goes on after a frame (the gather and the next batch are ordered)
        e.step(3)
        np.testing.assert_array_equal(g.read_state(), e.read_state())


def _device_count():
    import torch

    return torch.cuda.device_count()


@pytest.mark.parametrize("devices", [[0, 1], [0, 1, 0, 1]])
@pytest.mark.parametrize("transport", [0, 1])
def test_engine_group_across_real_devices(devices, transport):
    """The group on more than ONE device — peer access, hipMemcpyPeerAsync between devices, cross-device event ordering, ncclCommInitAll
    with n > 1, the cross-device gather of ca3d_group_render: none of it is reachable on a one-GPU box (every other group test repeats
    device 0), so this test runs wherever at least two GPUs are visible and is skipped elsewhere. Until it has run on hardware the
    multi-device path is unverified (README, INTEGRATION)."""
    if _device_count() < 2:
        pytest.skip("needs two visible GPUs")
    if transport == 1 and len(set(devices)) != len(devices):
        pytest.skip("the RCCL transport takes one device per slab")
    from cellularautomatons3d_amd import Engine, EngineGroup

    G, K = 256, 8
    rr = rules("default")
    st = host.random_fill(host.words_per_buffer(G), seed=4242, and_rounds=3)
    W, H = 320, 180
    u = host.uniform_block(W, H, host.orbit_camera())
    with EngineGroup(devices) as g, Engine(0) as e:
        g.configure(G, K)
        g.set_rules(rr.main, rr.edges, rr.corners, rr.survive, rr.born)
        g.set_option("transport", transport)
        g.upload_state(st)
        e.configure(G)
        e.set_rules(rr.main, rr.edges, rr.corners, rr.survive, rr.born)
        e.upload_state(st)
        want = st
        for n in (K, 2 * K + 3, 1):
            g.step(n)
            want = ol.packed_run(G, want, rr, n)
            np.testing.assert_array_equal(g.read_state(), want, err_msg=f"after {n} more steps")
        e.step(3 * K + 4)
        a = g.render(u, W, H, 4)
        b = e.render(u, W, H, 4)
        for x, y, what in zip(a, b, ("presentation", "light", "depth")):
            np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8), err_msg=what)

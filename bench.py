#!/usr/bin/env python3
"""Headline benchmark: Gcells/s of the bit-packed CA step (BASELINE.json metric), one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A "step" is one CA step (one dispatch of the reference's compute pass) over the whole grid. N = 1 runs the 512^3
packed grid the metric is quoted on; N > 1 runs the 1024^3 grid Z-slabbed over the ranks (per-GPU cell count at
N = 8 equals the N = 1 workload) with the ghost planes exchanged by RCCL. The state is resident in HBM before the
timed region; the timed region is bracketed by barrier + synchronize and the max over ranks is taken.

Prints ONE JSON line on rank 0 with `roofline` (algorithmic bytes per launch / measured launch duration against
the 8 TB/s HBM peak) and, at N = 1, `cpu_baseline` (the CPU oracle timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

RULES = {
    "default": dict(neighbourhood="von neumann", born="1,3", survive="0-6"),
    "clustered": dict(neighbourhood="moore", born="5-7", survive="4-7", born_edges="4", survive_edges="3-5",
                      born_corners="3", survive_corners="2-4"),
    "vn_b24_s135": dict(neighbourhood="von neumann", born="2,4", survive="1,3,5"),  # a von Neumann rule without a pre-built kernel
    "life2d": dict(neighbourhood="moore 2D", born="3", survive="2,3"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--grid", type=int, default=0, help="grid edge (default: 512 at N=1, 1024 at N>1)")
    ap.add_argument("--rule", choices=sorted(RULES), default="default")
    ap.add_argument("--density-rounds", type=int, default=0, help="AND rounds of the hashed fill: density 2^-(1+r)")
    ap.add_argument("--ghost", type=int, default=32, help="ghost planes per side = steps between halo exchanges (N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-render", action="store_true", help="skip the renderer leg (N=1)")
    ap.add_argument("--multi-render", action="store_true",
                    help="N>1: also time the frame shared between the ranks (volume all-gather + bands of image rows); "
                         "off by default so that the scaling run times the CA step alone")
    ap.add_argument("--render-size", default="1920x1080")
    ap.add_argument("--render-spp", type=int, default=4)
    ap.add_argument("--render-frames", type=int, default=10)
    ap.add_argument("--overlap", choices=["auto", "on", "off"], default="auto",
                    help="N>1: run the halo exchange under the interior phase of each batch (auto: by slab size, see slab.py)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="gloo = rehearsal transport through host memory")
    ap.add_argument("--device-map", default="", help="comma list: GPU index per rank (default: LOCAL_RANK)")
    ap.add_argument("--check", action="store_true", help="verify the final state against the oracle (small grids)")
    return ap.parse_args()


def cpu_baseline(G, rule_kw, seconds):
    """The CPU oracle (a port: the reference has no CPU path) on this host's cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from cellularautomatons3d_amd import host

    threads = min(os.cpu_count() or 1, 16)
    r = ol.Rules.from_strings(**rule_kw)
    st = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001)
    st = ol.packed_step(G, st, r, threads)  # thread-pool start-up + page faults outside the timing
    t0 = time.perf_counter()
    n = 0
    while True:
        st = ol.packed_step(G, st, r, threads)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 400:
            break
    return {"value": round(G ** 3 * n / dt / 1e9, 4), "unit": "Gcells/s", "cores": threads, "kind": "port",
            "sample": f"{n} steps of the {G}^3 packed grid, rule '{rule_kw.get('neighbourhood')}', oracle/ca_oracle.c word-parallel form, {dt:.1f} s"}


def cpu_baseline_js(seconds=4.0, G=256):
    """the JavaScript CPU stepper (oracle/js_stepper.js), single thread, on this host."""
    import shutil
    import subprocess

    node = shutil.which("node")
    if not node:
        return None
    try:
        r = subprocess.run([node, os.path.join(ROOT, "oracle", "js_stepper.js"), "bench", str(G), str(seconds)],
                           capture_output=True, text=True, timeout=120)
        d = json.loads(r.stdout)
        return {"value": round(d["gcells_s"], 4), "unit": "Gcells/s", "cores": 1, "kind": "port",
                "sample": f"{d['steps']} steps of the {G}^3 packed grid, default rule, oracle/js_stepper.js on node {d['node']}, "
                          f"{d['seconds']:.1f} s; host has {d['cpus']} x {d['cpu_model']}"}
    except Exception as e:  # baseline only: never fail the bench for it
        return {"error": str(e)}


def render_leg(eng, G, a):
    """Second half of BASELINE's metric: Mray/s of the volume renderer at 1080p, 4 spp, on the same grid size.
    Volume = hashed fill of density 2^-5 (dense silhouette), oblique bench pose; rays = primary +
    shadow rays traced; frames stay on the device (no read-back in the timed region)."""
    import torch

    from cellularautomatons3d_amd import host

    W, H = (int(v) for v in a.render_size.lower().split("x"))
    cells = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)
    eng.upload_state(cells)
    u = host.uniform_block(W, H, host.orbit_camera())
    for _ in range(2):
        eng.render(u, W, H, a.render_spp, readback=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gpu_ms = 0.0
    for _ in range(a.render_frames):
        eng.render(u, W, H, a.render_spp, readback=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.render_stats()
    rays = st.primary_rays + st.shadow_rays
    dense = {"metric": "Mray/s path-trace 1080p", "value": round(rays * a.render_frames / dt / 1e6, 2), "unit": "Mray/s",
            "ms_per_frame": round(dt * 1e3 / a.render_frames, 4), "kernel_ms": round(st.gpu_ms, 4),
            "primary_rays": int(st.primary_rays), "shadow_rays": int(st.shadow_rays),
            "cell_visits_per_primary_ray": round(st.primary_cell_visits / max(1, st.primary_rays), 2),
            "cell_visits_per_shadow_ray": round(st.shadow_cell_visits / max(1, st.shadow_rays), 2),
            "config": {"workload": f"{G}^3 packed volume, hashed fill density 2^-5, {W}x{H} @ {a.render_spp} spp, oblique pose "
                                   "(0.6 rad about (1,1,0), distance 1.4), exact DDA walk + shadow ray + Cook-Torrance"}}
    # the reference UI's own start-up scene (SURVEY 8(d) "sparse"): the single seed evolved 30 steps, default pose
    eng.upload_state(host.initial_state(G))
    eng.step(30)
    us = host.uniform_block(W, H, host.camera_matrix())
    eng.render(us, W, H, a.render_spp, readback=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.render_frames):
        eng.render(us, W, H, a.render_spp, readback=False)
    torch.cuda.synchronize()
    dts = time.perf_counter() - t0
    sts = eng.render_stats()
    dense["sparse_scene"] = {"ms_per_frame": round(dts * 1e3 / a.render_frames, 4),
                             "value": round((sts.primary_rays + sts.shadow_rays) * a.render_frames / dts / 1e6, 2), "unit": "Mray/s",
                             "cell_visits_per_primary_ray": round(sts.primary_cell_visits / max(1, sts.primary_rays), 2),
                             "workload": f"{G}^3, single seed after 30 default-rule steps, default pose, {W}x{H} @ {a.render_spp} spp, "
                                         "empty-space skipping over two levels of occupancy blocks"}
    return dense


def render_leg_multi(se, G, a, world, rank, barrier):
    """N > 1: the same scene rendered by all ranks together: each frame = all-gather of the packed
    volume over RCCL + every rank's band of image rows + gather of the bands on rank 0. With --check rank 0 also
    renders the frame alone and the two must be identical."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from cellularautomatons3d_amd import Engine, host, slab

    W, H = (int(v) for v in a.render_size.lower().split("x"))
    cells = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)
    pw = (G // 32) * G
    se.upload_state(cells[se.z0 * pw:(se.z0 + se.nz) * pw])
    sr = slab.SlabRenderer(se)
    u = host.uniform_block(W, H, host.orbit_camera())
    frame = sr.render(u, W, H, a.render_spp)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.render_frames):
        frame = sr.render(u, W, H, a.render_spp)
    barrier()
    dt = time.perf_counter() - t0
    y0, y1 = slab.band_rows(H, world, rank)
    rays = 0
    if y1 > y0:
        st = sr.full.render_stats()
        rays = int(st.primary_rays + st.shadow_rays)
    t = torch.tensor([rays], dtype=torch.int64, device="cuda" if a.backend == "nccl" else "cpu")
    dist.all_reduce(t)
    out = None
    if rank == 0:
        out = {"metric": "Mray/s path-trace 1080p", "value": round(int(t.item()) * a.render_frames / dt / 1e6, 2), "unit": "Mray/s",
               "ms_per_frame": round(dt * 1e3 / a.render_frames, 4),
               "config": {"workload": f"{G}^3 packed volume all-gathered to every rank, hashed fill density 2^-5, {W}x{H} @ {a.render_spp} spp, "
                                      f"bands of image rows over {world} ranks, frame assembled on rank 0 (read-back included)"}}
        if a.check:
            solo = Engine(se.device)
            solo.configure(G)
            solo.set_rule_strings()
            solo.upload_state(cells)
            want, _, _ = solo.render(u, W, H, a.render_spp)
            solo.close()
            out["frame_match"] = bool(np.array_equal(frame, want))
    sr.close()
    return out


def copy_ceiling_gbs():
    """Measured device-to-device copy rate (read + write bytes per second) of a 1 GiB buffer: the practical HBM
    ceiling SURVEY 8(d) asks to be reported next to the 8 TB/s vendor peak."""
    import torch

    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty(n, dtype=torch.uint8, device="cuda")
    a.zero_()
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def pmc_traffic(kernel, G):
    """HBM bytes per launch from a committed rocprofv3 PMC run (profiles/pmc_traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(p):
        return None
    try:
        d = json.load(open(p))
        return d.get(f"{kernel}@{G}", {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    from cellularautomatons3d_amd import Engine, host, slab

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE {world}")
    if a.device_map:
        local_rank = int(a.device_map.split(",")[rank])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    G = a.grid or (512 if world == 1 else 1024)
    rule_kw = RULES[a.rule]
    b, s = host.recalculate_rules_values(rule_kw.get("born", "1,3"), rule_kw.get("survive", "0-6"), rule_kw.get("born_edges", "27"),
                                          rule_kw.get("survive_edges", "27"), rule_kw.get("born_corners", "27"), rule_kw.get("survive_corners", "27"))
    offs = (host.NEIGHBOURHOOD_MAP[rule_kw["neighbourhood"]], host.NEIGHBOURHOOD_MAP["edges"], host.NEIGHBOURHOOD_MAP["corners"])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    full = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=a.density_rounds)
    pw = (G // 32) * G
    if world == 1:
        eng = Engine(local_rank)
        eng.configure(G)
        eng.set_rules(*offs, s, b)
        eng.upload_state(full)
        run = eng.step
        core = eng
    else:
        se = slab.SlabEngine(G, rank, world, ghost=a.ghost, device=local_rank, host_staging=a.backend == "gloo", overlap={"auto": "auto", "on": True, "off": False}[a.overlap])
        se.engine.set_rules(*offs, s, b)
        se.upload_state(full[se.z0 * pw:(se.z0 + se.nz) * pw])
        run = se.run
        core = se.engine

    if world == 1:
        eng.set_option("graph_prepare", max(a.steps, a.warmup))  # graph capture / instantiation stays out of the timed region
    if a.warmup > 0:
        run(a.warmup)
    barrier()
    t0 = time.perf_counter()
    run(a.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = core.stats()
    info = core.info()
    kernel = info.kernel_name.decode()
    cells = float(G) ** 3
    value = cells * a.steps / dt / 1e9

    ok = None
    if a.check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as np
        import oracle_lib as ol

        want = ol.packed_run(G, full, ol.Rules.from_strings(**rule_kw), a.warmup + a.steps)
        got = core.read_state()
        lo = 0 if world == 1 else se.z0 * pw
        ok = bool(np.array_equal(got, want[lo:lo + got.size]))
        if world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if a.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())

    multi_render = None
    if world > 1 and a.multi_render:
        multi_render = render_leg_multi(se, G, a, world, rank, barrier)

    if rank == 0:
        # dominant kernel: HIP events on the engine's stream around the last step batch (get_stats), divided by the
        # launches in it; algorithmic bytes per launch = 0.25 B x cells the launch updates.
        launch_ms = st.gpu_ms / max(1, st.kernel_launches)
        bytes_per_launch = st.algorithmic_bytes / max(1, st.kernel_launches)
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        out = {
            "metric": "Gcells/s CA step at 512^3" if G == 512 else f"Gcells/s CA step at {G}^3",
            "value": round(value, 3), "unit": "Gcells/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt * 1e3 / a.steps, 6), "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{G}^3 uint32-packed grid, rule '{a.rule}' ({rule_kw['neighbourhood']} B{rule_kw['born']}/S{rule_kw['survive']}), "
                                   f"hashed fill seed 0xCA3D0001 density {2.0 ** -(1 + a.density_rounds):g}, one CA step per bench step",
                       "grid": G, "layout": "packed32", "rule": a.rule,
                       "parallelism": "1 GPU" if world == 1 else f"z-slab x{world}, ghost {a.ghost} planes, RCCL send/recv every {a.ghost} steps" + (" overlapped with the interior phase" if se.overlap else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(kernel, G),
                         "kernel": kernel, "launch_us": round(launch_ms * 1e3, 3),
                         "algorithmic_bytes_per_launch": bytes_per_launch, "launches_timed": int(st.kernel_launches)},
        }
        if world == 1:
            ceiling = copy_ceiling_gbs()
            out["roofline"]["copy_ceiling"] = round(ceiling, 1)  # GB/s, measured here: 1 GiB device-to-device copy, read + write
            out["roofline"]["frac_of_copy_ceiling"] = round(achieved / ceiling, 4)
        if ok is not None:
            out["oracle_match"] = ok
        if world == 1 and not a.no_render:
            out["render"] = render_leg(eng, G, a)
        if multi_render is not None:
            out["render"] = multi_render
            if multi_render.get("frame_match") is False:
                ok = False
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(G, rule_kw, a.cpu_seconds)
            out["cpu_baseline_js"] = cpu_baseline_js()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if ok is False:
        raise SystemExit("state does not match the oracle")


if __name__ == "__main__":
    main()

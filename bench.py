#!/usr/bin/env python3
"""Headline benchmark: Gcells/s of the bit-packed CA step (BASELINE.json metric), one process per GPU.

  python bench.py --gpus N --steps K --warmup W          (N > 1: spawns its N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W             (the same, launched from outside)

A "step" is one CA step (one dispatch of the reference's compute pass, main_pathtraced.js:1796-1809) over the whole
grid. The batch of K steps is repeated (`reps`) until at least 1 s is timed (every CA leg). At N = 1 the K-step calls are ENCODED
(`ca3d_set_option("queue")`) and handed to the GPU 2048 steps at a time (`ca3d_flush`) — what the reference does with its
command encoder and one queue.submit (main_pathtraced.js:1833-1850) — so the value does not depend on K: the resident
multi-step kernel runs a submission as one launch. `--submit call` makes every K-step call its own submission.
--config picks the BASELINE configuration:
  3 (default at N = 1)  512^3, default rule, + the 1920x1080 @ 4 spp render leg          BASELINE configs[2]
  4 (default at N > 1)  1024^3, default rule, Z-slabs over the ranks, RCCL halo exchange   BASELINE configs[3]
  5                     2048^3, clustered rule-set, halo overlapped with compute, + the 3840x2160 frame   configs[4]
Configs 4 and 5 at N = 1 run the same grid on one GPU: the base of the scaling curve (the N = 1 line of config 3 also
carries it as `scaling_base`, measured in the same run). The state is resident in HBM before the timed region; the
timed region is bracketed by barrier + synchronize and the max over ranks is taken.

Prints ONE JSON line on rank 0 with `roofline` and, at N = 1, `cpu_baseline` (the CPU oracle timed on this host's cores on a
bounded sample). `roofline` names the limiter of the kernel that ran: per-step kernels read and write the state every step —
algorithmic bytes per launch / launch duration (HIP events on the engine's stream around the timed region) against the
8 TB/s HBM peak; the resident multi-step kernels keep the state on chip, what bounds them is vector-instruction issue —
wave-instructions per second (count per wave and step from the committed rocprofv3 SQ pass of this command, profiles/) against
1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction — with the algorithmic-byte rate beside it as `hbm_equivalent`.
The N = 1 line also carries: `per_step_kernels` (the same grid with every step through memory: the HBM-path fraction),
`per_call` (every K-step call its own submission), `grid_256` (BASELINE configs[1]'s grid), `grid_64` (the UI's start-up grid), `render` / `render_4k`,
`scaling_base` (1024^3 on one GPU).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
INFINITY_CACHE_BYTES = 256 << 20
MIN_TIMED_SECONDS = 1.0
VALU_PEAK_GWIPS = 1024 * 2.4 / 2.0  # G wave-instructions/s: 256 CUs x 4 SIMD-32, a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md, Wave scheduling), 2.4 GHz

RULES = {
    "default": dict(neighbourhood="von neumann", born="1,3", survive="0-6"),
    "clustered": dict(neighbourhood="moore", born="5-7", survive="4-7", born_edges="4", survive_edges="3-5",
                      born_corners="3", survive_corners="2-4"),
    "vn_b24_s135": dict(neighbourhood="von neumann", born="2,4", survive="1,3,5"),  # a von Neumann rule without a pre-built kernel
    "life2d": dict(neighbourhood="moore 2D", born="3", survive="2,3"),
}

CONFIGS = {  # BASELINE.json configs[2..4]
    3: dict(grid=512, rule="default", render_size="1920x1080"),
    4: dict(grid=1024, rule="default", render_size="1920x1080"),
    5: dict(grid=2048, rule="clustered", render_size="3840x2160", overlap="on"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=0, help="BASELINE configuration (default: 3 at N=1, 4 at N>1)")
    ap.add_argument("--grid", type=int, default=0, help="grid edge (overrides the configuration's)")
    ap.add_argument("--rule", choices=sorted(RULES), default="", help="rule-set (overrides the configuration's)")
    ap.add_argument("--density-rounds", type=int, default=0, help="AND rounds of the hashed fill: density 2^-(1+r)")
    ap.add_argument("--ghost", type=int, default=0, help="ghost planes per side = steps between halo exchanges (N>1); 0: 32, or the deepest of 32 / 16 / 8 "
                    "with which a rank's share of a 1024^3 default-rule grid still runs the resident slab kernel (8 ranks: 32; 4 ranks: 16)")
    ap.add_argument("--min-seconds", type=float, default=MIN_TIMED_SECONDS, help="repeat the batch until this much is timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-render", action="store_true", help="skip the renderer legs (N=1)")
    ap.add_argument("--no-interactive", action="store_true", help="N=1, config 3: skip the step(1) + literal-frame-per-submission leg")
    ap.add_argument("--no-scaling-base", action="store_true", help="N=1, config 3: skip the 1024^3 single-GPU leg")
    ap.add_argument("--no-per-step-leg", action="store_true", help="N=1: skip the per-step-kernel leg printed beside a resident-kernel headline")
    ap.add_argument("--no-per-call-leg", action="store_true", help="N=1: skip the per-call-submission leg printed beside a queued-submission headline")
    ap.add_argument("--no-grid-256", action="store_true", help="N=1, config 3: skip the 256^3 leg (BASELINE configs[1]'s grid)")
    ap.add_argument("--multi-render", action="store_true",
                    help="N>1: also time the frame shared between the ranks (volume all-gather + bands of image rows); "
                         "off by default (on for --config 5) so that the scaling run times the CA step alone")
    ap.add_argument("--render-size", default="")
    ap.add_argument("--render-spp", type=int, default=4)
    ap.add_argument("--render-frames", type=int, default=100)
    ap.add_argument("--overlap", choices=["auto", "on", "off"], default="",
                    help="N>1: run the halo exchange under the interior phase of each batch (auto: by slab size, see slab.py)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="gloo = rehearsal transport through host memory")
    ap.add_argument("--transport", choices=["auto", "native", "torch"], default="auto",
                    help="N>1 halo exchange: native (auto with RCCL) = ncclSend/ncclRecv issued by the engine itself; "
                         "torch = torch.distributed point-to-point from Python")
    ap.add_argument("--device-map", default="", help="comma list: GPU index per rank (default: LOCAL_RANK)")
    ap.add_argument("--resident", type=int, choices=[0, 1], default=1,
                    help="0: per-step kernels only (every step reads and writes the state through memory); 1: batches run as one launch of the resident kernel where one exists")
    ap.add_argument("--submit", choices=["queued", "call"], default="queued",
                    help="N=1: queued = the K-step calls are ENCODED and handed to the GPU --queue steps at a time (ca3d_flush; the reference's "
                         "commandEncoder + one queue.submit), so the resident kernel runs them as one launch whatever K is; call = every "
                         "K-step call is its own submission")
    ap.add_argument("--compare-submission", action="store_true",
                    help="N=1: also time the same calls under the other submission mode (`other_submission` in the JSON line; off by default so "
                         "that a profile of the default command holds launches of one length only)")
    ap.add_argument("--queue", type=int, default=2048, help="steps per submission with --submit queued")
    ap.add_argument("--no-schedule-compare", action="store_true", help="N>1: do not time the other batch schedule (overlap on / off) behind the headline")
    ap.add_argument("--check", action="store_true", help="verify the final state against the oracle (small grids)")
    ap.add_argument("--verify-steps", type=int, default=-1,
                    help="before the timed region run this many steps from the bench state through the headline path and compare the state (N>1: every "
                         "rank's slab) with the CPU oracle; a mismatch fails the run. Default: N=1 warm-up + 40 steps on grids up to 512^3; N>1 ghost + 8 steps — "
                         "past one halo exchange — on grids up to 1024^3; 0 = off. The state is uploaded again afterwards")
    a = ap.parse_args(argv)
    a.config = a.config or (3 if a.gpus == 1 else 4)
    cfg = CONFIGS[a.config]
    a.grid = a.grid or cfg["grid"]
    a.rule = a.rule or cfg["rule"]
    a.render_size = a.render_size or cfg["render_size"]
    a.overlap = a.overlap or cfg.get("overlap", "auto")
    if a.config == 5 and a.gpus > 1:
        a.multi_render = True
    if a.ghost <= 0:
        a.ghost = auto_ghost(a.grid, a.rule, a.gpus)
    if a.verify_steps < 0:
        # N = 1: warm-up + 40 steps through the headline path (one queued submission: the resident kernel where there is one) on grids
        # the oracle steps in about a second; N > 1: past one halo exchange
        a.verify_steps = (a.ghost + 8 if a.grid <= 1024 else 0) if a.gpus > 1 else (a.warmup + 40 if a.grid <= 512 else 0)
    return a


def group_leg(device: int, G: int, parts: int, ghost: int, min_seconds: float) -> dict:
    """G^3 on ONE device as `parts` slabs driven by one thread through the group handle; returns the rate and the kernel that ran."""
    from cellularautomatons3d_amd import EngineGroup, host
    steps = 16 * ghost
    with EngineGroup([device] * parts) as g:
        g.configure(G, ghost)
        g.set_rule_strings(**RULES["default"])
        g.upload_state(host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001))
        g.step(2 * ghost)
        g.synchronize()
        reps, dt = 0, 0.0
        t0 = time.perf_counter()
        while True:
            g.step(steps)
            g.synchronize()
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= min(min_seconds, 0.5) or reps >= 64:
                break
        return {"value": round(float(G) ** 3 * steps * reps / dt / 1e9, 3), "unit": "Gcells/s", "ms_per_step": round(dt * 1e3 / (steps * reps), 6),
                "slabs": parts, "ghost": ghost, "steps": steps, "reps": reps, "kernel": g.kernel_name(0), "timing": "host clock around step + synchronize",
                "note": "one GPU, the group handle's peer-copy transport between slabs of the same device"}


def auto_ghost(grid: int, rule: str, world: int) -> int:
    """Ghost depth K (planes per side = steps between exchanges) when none is asked for: 32 — the exchange costs a fixed ~45 us of
    launch latency on the GPU, DESIGN.md section 6 — unless a shallower one lets the rank's share run the resident slab kernel
    (ca_resident.hip, resident_slab_planes: 1024^3, a von Neumann table pair, 8 tile layers of an even number of planes <= 36):
    a quarter of 1024^3 is 256 + 2 K planes, resident with K = 16 (8.3 against 12.5 us per step in loopback), not with K = 32."""
    if world > 1 and grid == 1024 and rule in ("default", "vn_b24_s135") and grid % world == 0:
        for k in (32, 16, 8):
            planes = grid // world + 2 * k
            if planes % 8 == 0 and (planes // 8) % 2 == 0 and planes // 8 <= 36:
                return k
    return 32


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` from a plain shell: start the N ranks as children (torch.distributed.run), before
    anything in this process has touched the GPU, and hand back their exit code. Rank 0's JSON line is the only
    thing the children write to stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, cwd=ROOT)


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


def time_frames(eng, u, W, H, spp, frames):
    import torch

    for _ in range(8):  # every lane of the frame pipeline has drawn (and sized its scratch for) this frame before the clock starts
        eng.render(u, W, H, spp, readback=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        eng.render(u, W, H, spp, readback=False)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, eng.render_stats()


def render_pmc_record(kernel_ms):
    """What bounds the dense frame's kernels, from the committed PMC passes over the same scene (tools/pmc_render.sh ->
    profiles/r*_pmc_render.json, newest): vector-issue fraction, live lanes, L2 hit rate, bytes served by the L2 per second — for the
    passes of the ray-stream pipeline together (counters summed over ca_stream_walk2<primary>, ca_stream_shadow_rays, ca_stream_walk2<shadow>,
    ca_stream_resolve) and per pass. SURVEY 8(d): "report Mray/s and achieved GB/s from rocprof, no roofline claim beyond that"."""
    import glob

    def rates(v, cycles):
        return {"valu_issue_frac": round(v["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cycles), 4),  # wave-instructions x 2 cycles / (SIMDs x kernel cycles)
                "lanes_active_frac": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0), 4) if v.get("SQ_ACTIVE_INST_VALU") else None,
                "wave_cycles_issuing_valu_frac": round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 4),
                "wave_cycles_waiting_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4) if v.get("SQ_WAIT_ANY") else None,
                "l2_hit_rate": round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4) if v.get("TCC_HIT_sum") else None}

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_render.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        ks = [k for k, v in d.items() if "ca_stream_" in k and v.get("SQ_INSTS_VALU")] or [k for k, v in d.items() if "sched<false" in k and v.get("SQ_INSTS_VALU")]
        if not ks:
            continue
        tot = {}
        for k in ks:
            for c, x in d[k].items():
                tot[c] = tot.get(c, 0.0) + x
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of a kernel = / 8 (MI355X_MICROARCH.md, DVFS give-back); the passes run one after the other
        cycles = tot["GRBM_GUI_ACTIVE"] / 8.0
        rec = {"bound": "valu_issue at partial lane occupancy (divergent walks)", "kernels": ks, "counter_source": os.path.relpath(f, ROOT)}
        rec.update(rates(tot, cycles))
        if len(ks) > 1:
            rec["per_kernel"] = {k: dict(rates(d[k], d[k]["GRBM_GUI_ACTIVE"] / 8.0), us_at_2400MHz=round(d[k]["GRBM_GUI_ACTIVE"] / 8.0 / 2400.0, 1)) for k in ks}
        if tot.get("TCC_REQ_sum") and kernel_ms:
            rec["l2_request_gbs"] = round(tot["TCC_REQ_sum"] * 128.0 / (kernel_ms * 1e-3) / 1e9, 1)   # 128-B lines requested of the L2 per second (this run's kernel time)
        if tot.get("TCC_EA0_RDREQ_sum") and kernel_ms:
            rec["fabric_read_gbs"] = round(tot["TCC_EA0_RDREQ_sum"] * 64.0 * 2.0 / (kernel_ms * 1e-3) / 1e9, 1)  # = FETCH_SIZE, doubled as the guide prescribes
        rec["geometry"] = ("walk launches sized for the WHOLE chip — a frame alone (rocprofv3 collects counters one dispatch at a time: every frame of a "
                           "PMC pass finds the engine idle); the frames of this line's default figure run side by side on a share each: in_flight_geometry")
        # the same scene with the walk launches a frame IN FLIGHT gets (a quarter of the chip at this size): CA3D_STREAM_WGS_PCT=25 tools/pmc_render.sh
        for g in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_render_share25.json")), reverse=True):
            try:
                q = json.load(open(g))
            except Exception:
                continue
            walks = {k: v for k, v in q.items() if "ca_stream_walk" in k and v.get("SQ_ACTIVE_INST_VALU")}
            if walks:
                rec["in_flight_geometry"] = {"counter_source": os.path.relpath(g, ROOT), "walk_share_of_chip": 0.25,
                                             "per_kernel": {k: {"lanes_active_frac": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0), 4),
                                                                "valu_wave_instructions": int(v["SQ_INSTS_VALU"]),
                                                                "valu_wave_instructions_whole_chip": int(d[k]["SQ_INSTS_VALU"]) if k in d else None} for k, v in walks.items()}}
                break
        return rec
    return None


def literal_pmc_record():
    """The same for the literal frame's kernel (tools/pmc_render.sh <tag> --literal 1 -> profiles/r*_pmc_render_literal.json)."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_render_literal.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        ks = [k for k, v in d.items() if "ca_render_frame" in k and v.get("SQ_INSTS_VALU")]
        if not ks:
            continue
        v = d[ks[0]]
        cycles = v["GRBM_GUI_ACTIVE"] / 8.0
        return {"kernel": ks[0], "counter_source": os.path.relpath(f, ROOT), "valu_issue_frac": round(v["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cycles), 4),
                "lanes_active_frac": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0), 4),
                "wave_cycles_waiting_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4),
                "l2_hit_rate": round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4) if v.get("TCC_HIT_sum") else None}
    return None


def render_leg(eng, G, a, size=None, sparse=True, literal=True):
    """Second half of BASELINE's metric: Mray/s of the volume renderer at 1080p (config 5: 3840x2160), 4 spp, on the
    same grid size. Volume = hashed fill of density 2^-5 (dense silhouette), oblique bench pose;
    rays = primary + shadow rays traced; frames stay on the device (no read-back in the timed region). `literal`: also the
    reference's own per-frame workload — ONE jittered fixed-step sample per pixel with the temporal history look-ups and the
    EMA blend (fragment_main, pathtraced_fragment_clustered.wgsl:800-890; main_pathtraced.js:1775-1794), render_mode 1."""
    from cellularautomatons3d_amd import host

    W, H = (int(v) for v in (size or a.render_size).lower().split("x"))
    cells = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)
    eng.upload_state(cells)
    # The CA legs run the engine on a torch stream (torch events bracket them). The render legs hand the engine its own stream back:
    # frames are only put in flight side by side on a stream the engine owns (a caller's stream promises the caller in-order frames);
    # the timed regions here end in torch.cuda.synchronize(), which waits for every stream of the device.
    bench_stream = getattr(eng, "bench_stream", None)
    eng.use_own_stream()
    try:
        return _render_leg_on_own_stream(eng, G, a, W, H, literal, sparse)
    finally:
        if bench_stream is not None:
            eng.set_stream(bench_stream.cuda_stream)


def _render_leg_on_own_stream(eng, G, a, W, H, literal, sparse):
    from cellularautomatons3d_amd import host

    # one frame at a time first (render_pipeline 0: what a frame costs from its first kernel to its last), then the engine's default — converged
    # frames that stay on the device alternate between up to four streams and run side by side, each frame's walks on a share of the chip
    eng.set_option("render_pipeline", 0)
    dt1, st1 = time_frames(eng, host.uniform_block(W, H, host.orbit_camera()), W, H, a.render_spp, a.render_frames)
    eng.set_option("render_pipeline", 1)
    dt, st = time_frames(eng, host.uniform_block(W, H, host.orbit_camera()), W, H, a.render_spp, a.render_frames)
    in_flight = eng.render_pipeline()  # what the engine really runs (ca3d_get_render_pipeline): 0 when the runtime gave it no side-by-side streams
    rays = st.primary_rays + st.shadow_rays
    dense = {"metric": f"Mray/s path-trace {'1080p' if H == 1080 else f'{W}x{H}'}", "value": round(rays * a.render_frames / dt / 1e6, 2), "unit": "Mray/s",
             "ms_per_frame": round(dt * 1e3 / a.render_frames, 4), "kernel_ms": round(st1.gpu_ms, 4),
             "frames_in_flight": in_flight, "one_frame_at_a_time": {"ms_per_frame": round(dt1 * 1e3 / a.render_frames, 4), "value": round(rays * a.render_frames / dt1 / 1e6, 2),
                                                            "note": "ca3d_set_option(render_pipeline, 0): a frame's kernels from first to last with nothing beside them; kernel_ms is this form's"},
             "primary_rays": int(st.primary_rays), "shadow_rays": int(st.shadow_rays),
             "cell_visits_per_primary_ray": round(st.primary_cell_visits / max(1, st.primary_rays), 2),
             "cell_visits_per_shadow_ray": round(st.shadow_cell_visits / max(1, st.shadow_rays), 2),
             "config": {"workload": f"{G}^3 packed volume, hashed fill density 2^-5, {W}x{H} @ {a.render_spp} spp, oblique pose "
                                    "(0.6 rad about (1,1,0), distance 1.4), exact DDA walk + shadow ray + Cook-Torrance; ray-stream passes (render_stream.hip); "
                                    f"frames stay on the device, {in_flight} in flight (the engine's default for converged frames without host pointers)"}}
    if H == 1080 and a.render_spp == 4:
        pmc = render_pmc_record(st1.gpu_ms)
        if pmc:
            dense["pmc"] = pmc
    if literal:
        import torch

        vm = host.orbit_camera()
        eng.set_render_mode(True)
        try:
            eng.reset_render_history()
            for i in range(3):
                eng.render(host.uniform_block(W, H, vm, elapsed_time=0.5 + 0.01 * i, prev_view_mat=vm), W, H, 1, readback=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.render_frames):
                eng.render(host.uniform_block(W, H, vm, elapsed_time=0.6 + 0.01 * i, prev_view_mat=vm), W, H, 1, readback=False)
            torch.cuda.synchronize()
            dtl = time.perf_counter() - t0
            dense["literal_frame"] = {"ms_per_frame": round(dtl * 1e3 / a.render_frames, 4), "kernel_ms": round(eng.render_stats().gpu_ms, 4),
                                      "value": round(W * H * a.render_frames / dtl / 1e6, 2), "unit": "Mpixel/s",
                                      "workload": f"the reference's own frame: one jittered fixed-step sample per pixel (<= 35 primary + <= 30 shadow march samples), history "
                                                  f"look-ups, depth repair, temporal blend; same scene and pose, {W}x{H}, static camera (prev matrices = current)"}
            lp = literal_pmc_record()
            if lp and H == 1080:
                dense["literal_frame"]["pmc"] = lp
        finally:
            eng.set_render_mode(False)
    if sparse:
        # the reference UI's own start-up scene (SURVEY 8(d) "sparse"): the single seed evolved 30 steps, default pose
        eng.upload_state(host.initial_state(G))
        eng.step(30)
        eng.set_option("render_pipeline", 0)
        dts1, _ = time_frames(eng, host.uniform_block(W, H, host.camera_matrix()), W, H, a.render_spp, a.render_frames)
        eng.set_option("render_pipeline", 1)
        dts, sts = time_frames(eng, host.uniform_block(W, H, host.camera_matrix()), W, H, a.render_spp, a.render_frames)
        dense["sparse_scene"] = {"ms_per_frame": round(dts * 1e3 / a.render_frames, 4), "ms_per_frame_one_at_a_time": round(dts1 * 1e3 / a.render_frames, 4),
                                 "value": round((sts.primary_rays + sts.shadow_rays) * a.render_frames / dts / 1e6, 2), "unit": "Mray/s",
                                 "cell_visits_per_primary_ray": round(sts.primary_cell_visits / max(1, sts.primary_rays), 2),
                                 "workload": f"{G}^3, single seed after 30 default-rule steps, default pose, {W}x{H} @ {a.render_spp} spp, "
                                             "empty-space skipping over two levels of occupancy blocks, walks clipped to the live box, "
                                             "pixels outside the live box's screen rectangle filled (ca_render_background)"}
        # ... and what the bench scene (hashed fill, density 2^-5) is kind to: the same frame over THINNER volumes. Down to ~2^-12 a volume is
        # still dense by the block count (the stream passes draw it, every walk is longer); below that it is a scattered sparse volume, drawn
        # by the stream passes' block-skipping form — the renderer's worst case. 1080p frames only.
        if H == 1080:
            thin = {}
            for rounds in (8, 12):
                eng.upload_state(host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=rounds))
                dtt, stt = time_frames(eng, host.uniform_block(W, H, host.orbit_camera()), W, H, a.render_spp, max(10, a.render_frames // 4))
                n = max(10, a.render_frames // 4)
                thin[f"density_2^-{rounds + 1}"] = {"ms_per_frame": round(dtt * 1e3 / n, 4), "value": round((stt.primary_rays + stt.shadow_rays) * n / dtt / 1e6, 2), "unit": "Mray/s",
                                                   "cell_visits_per_primary_ray": round(stt.primary_cell_visits / max(1, stt.primary_rays), 2)}
            thin["note"] = ("hashed fills thinner than the headline scene's 2^-5, same pose and size, the engine's default path: 2^-9 is drawn by the stream passes "
                            "(longer walks), 2^-13 — a scattered sparse volume — by their block-skipping form (ca_stream_walk2<.., SKIP>; the in-wave scheduled kernel until late in round 5: 4.7 ms)")
            dense["thinner_scenes"] = thin
    return dense


def interactive_leg(eng, G, a, frames=200):
    """What a drop-in behind the reference's own loop gets (main_pathtraced.js:1821-1854): per animation frame ONE compute pass and ONE
    render pass in one submission — here ca3d_step(1) (a per-step kernel: a one-step submission never reaches the resident kernel) and
    ca3d_render in the literal frame mode (one jittered sample per pixel, history, EMA), 1920x1080, nothing read back, `frames` frames
    back to back; two scenes: the UI's start-up seed (evolving from step 30) and the dense bench volume (evolving from the hashed fill)."""
    import torch

    from cellularautomatons3d_amd import host

    W, H = 1920, 1080
    vm = host.orbit_camera()
    eng.set_option("queue", 0)
    eng.set_render_mode(True)
    out = {"frames": frames, "passes": "best of 3", "size": f"{W}x{H}", "per_frame": "ca3d_step(1) + ca3d_render(render_mode 1, 1 sample per pixel), one submission per frame, no read-back",
           "reference_loop": "main_pathtraced.js:1821-1854 (_computePass + _renderPass per requestAnimationFrame)"}
    try:
        for name, start in (("startup_scene", None), ("dense_scene", host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4))):
            if start is None:
                eng.upload_state(host.initial_state(G))
                eng.step(30)
            else:
                eng.upload_state(start)
            eng.reset_render_history()
            for i in range(5):
                eng.step(1)
                eng.render(host.uniform_block(W, H, vm, elapsed_time=0.3 + 0.016 * i, prev_view_mat=vm), W, H, 1, readback=False)
            torch.cuda.synchronize()
            us = [host.uniform_block(W, H, vm, elapsed_time=0.5 + 0.016 * i, prev_view_mat=vm) for i in range(frames)]  # the host's uniform upload, prepared
            dt = 1e30
            for _ in range(3):  # the best of three passes of `frames` frames: a pass is ~25 ms of wall clock, one host hiccup doubles it
                t0 = time.perf_counter()
                for i in range(frames):
                    eng.step(1)
                    eng.render(us[i], W, H, 1, readback=False)
                torch.cuda.synchronize()
                dt = min(dt, time.perf_counter() - t0)
            gpu_ms = eng.render_stats().gpu_ms
            eng.set_option("resident", 0)  # (ca3d_get_info names the kernel the rules select for LONG batches; a one-step submission is below
            step_kernel = eng.info().kernel_name.decode()  # resident_min = 8 and takes the per-step kernel: the name with the resident form off)
            eng.set_option("resident", a.resident)
            out[name] = {"ms_per_frame": round(dt * 1e3 / frames, 4), "frames_per_second": round(frames / dt, 1), "last_render_kernel_ms": round(gpu_ms, 4),
                         "step_kernel": step_kernel,
                         "render_kernels": "ca_brick_volume + ca_render_frame_bricks (render_frame.hip)"}
            if start is not None:
                # the same loop while the user holds a key (main_pathtraced.js:858-969 move the camera, :504-524 write prevViewMat /
                # prevProjViewMatInv != current): the camera orbits 0.01 rad per frame, every frame reprojects into the previous pose
                cams = [host.orbit_camera(1.4, (1.0, 1.0, 0.0), 0.6 + 0.01 * i) for i in range(frames + 1)]
                usm = [host.uniform_block(W, H, cams[i + 1], elapsed_time=0.5 + 0.016 * i, prev_view_mat=cams[i]) for i in range(frames)]
                eng.reset_render_history()
                for i in range(5):
                    eng.step(1)
                    eng.render(usm[i], W, H, 1, readback=False)
                torch.cuda.synchronize()
                dtm = 1e30
                for _ in range(3):
                    t0 = time.perf_counter()
                    for i in range(frames):
                        eng.step(1)
                        eng.render(usm[i], W, H, 1, readback=False)
                    torch.cuda.synchronize()
                    dtm = min(dtm, time.perf_counter() - t0)
                out[name + "_moving_camera"] = {"ms_per_frame": round(dtm * 1e3 / frames, 4), "frames_per_second": round(frames / dtm, 1),
                                                "last_render_kernel_ms": round(eng.render_stats().gpu_ms, 4),
                                                "camera": "orbit, 0.01 rad per frame; prevViewMat / prevProjViewMatInv = the previous frame's"}
    finally:
        eng.set_render_mode(False)
    return out


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
        out = {"metric": f"Mray/s path-trace {'1080p' if H == 1080 else f'{W}x{H}'}", "value": round(int(t.item()) * a.render_frames / dt / 1e6, 2), "unit": "Mray/s",
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


def copy_ceiling_gbs(eng):
    """Measured device-to-device copy rate (read + write bytes per second) of a 1 GiB buffer through a float4-per-lane copy
    kernel with non-temporal stores (ca3d_measure_copy): the practical HBM ceiling SURVEY 8(d) asks to be reported next to the
    8 TB/s vendor peak (MI355X_MICROARCH.md quotes 6.29 TB/s for this kind of kernel)."""
    return eng.measure_copy(1 << 30, 8)


def sq_profile(kernel, G, variant=None):
    """The committed rocprofv3 SQ pass of this command for a resident kernel (tools/profile_round4.sh -> tools/pmc_sq_reduce.py):
    newest profiles/r*_pmc_sq_<key>.json that says how many steps its launches held AND was taken on the same instruction stream —
    its `variant` (ca3d_get_kernel_variant at profile time: kernel, grid, rule hash, form options, device-source hash) must equal
    the running engine's. None when there is none: the count of another rule, form or source revision prices nothing."""
    import glob

    base = kernel.split("(")[0]
    key = {"ca_resident_vn": f"resident{G}", "ca_resident_class": f"clustered{G}", "ca_resident_slab_vn": "residentslab"}.get(base)
    if not key:
        return None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_sq_{key}.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("steps_per_launch") and d.get("SQ_INSTS_VALU") and d.get("SQ_WAVES") and variant and d.get("variant") == variant:
            return d, os.path.relpath(f, ROOT)
    return None, None


def pmc_traffic(kernel, G, steps_per_launch=1):
    """Fabric bytes per launch (FETCH_SIZE + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes) from the committed
    rocprofv3 PMC passes of this command (profiles/pmc_traffic.json; tools/profile_round.sh collects them), or None:
    counters cannot be read from inside an un-profiled run."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(p):
        return None
    try:
        d = json.load(open(p))
        for key in (f"{kernel}@{G}", f"{kernel.split('(')[0].split('<')[0]}@{G}"):  # e.g. ca_packed_class_roll<moore,E,C>(jit) -> ca_packed_class_roll@1024
            if key in d:
                e = d[key]
                if "steps_per_launch" in e:  # a resident kernel: the pass was taken at that many steps per launch
                    return round(e["hbm_bytes_per_launch"] / e["steps_per_launch"] * steps_per_launch)
                return e.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def rule_payload(rule_kw):
    from cellularautomatons3d_amd import host

    b, s = host.recalculate_rules_values(rule_kw.get("born", "1,3"), rule_kw.get("survive", "0-6"), rule_kw.get("born_edges", "27"),
                                          rule_kw.get("survive_edges", "27"), rule_kw.get("born_corners", "27"), rule_kw.get("survive_corners", "27"))
    offs = (host.NEIGHBOURHOOD_MAP[rule_kw["neighbourhood"]], host.NEIGHBOURHOOD_MAP["edges"], host.NEIGHBOURHOOD_MAP["corners"])
    return offs, s, b


def timed_region(run, stream, steps, warmup, min_seconds, barrier, world, backend, flush=None, group=1, launch_count=None):
    """W warm-up steps, then `reps` batches of K steps between barrier + synchronize; reps is chosen (the same on every
    rank) so that at least `min_seconds` are timed. `flush` submits what `run` only encoded (queued submission); `group`
    batches make one submission, the calibration runs one group and reps is a multiple of it, so every launch of the
    calibration and of the timed region has the same length. Returns (wall seconds, max over ranks; reps; HIP-event ms
    between the first and the last enqueue of the timed region on the engine's stream; batches run before the timed ones;
    kernel launches inside the timed region by `launch_count`, the engine's own counter, or None)."""
    import torch
    import torch.distributed as dist

    flush = flush or (lambda: None)
    if warmup > 0:
        run(warmup)
    flush()
    barrier()
    t0 = time.perf_counter()
    for _ in range(group):
        run(steps)  # calibration (also warm: graphs for K steps exist afterwards)
    flush()
    barrier()
    est = time.perf_counter() - t0
    reps = group * max(1, min((1 << 20) // group, int(math.ceil(min_seconds / max(est, 1e-7)))))
    if world > 1:
        t = torch.tensor([reps], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        reps = int(t.item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    n0 = launch_count() if launch_count else 0
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(reps):
        run(steps)
    flush()
    e1.record(stream)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, reps, e0.elapsed_time(e1), group, (launch_count() - n0 if launch_count else None)


def oracle_verify(eng, G, full, rule, nsteps, queued):
    """`nsteps` from the bench state through the engine exactly as the timed region drives it (queued: one submission — the resident
    kernel where one exists), compared bit for bit with the CPU oracle; the bench state is uploaded again afterwards."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol

    if queued:
        eng.set_option("queue", max(nsteps, 8))
    eng.step(nsteps)
    eng.flush()
    got = eng.read_state()
    kernel = eng.info().kernel_name.decode()
    t0 = time.perf_counter()
    want = ol.packed_run(G, full, ol.Rules.from_strings(**RULES[rule]), nsteps, min(os.cpu_count() or 1, 64))
    rec = {"oracle_match": bool(np.array_equal(got, want)), "steps": nsteps, "kernel": kernel, "oracle_seconds": round(time.perf_counter() - t0, 2),
           "how": "the bench state stepped through the headline path (same engine, same submission mode) and read back, against the CPU oracle, before the timed region"}
    eng.upload_state(full)
    return rec


def single_gpu_leg(device, G, rule, steps, warmup, min_seconds, density_rounds=0, resident=1, queue=0, verify_steps=0):
    """One GPU, whole grid: returns the numbers of the timed region plus the engine (state = the bench state advanced).
    queue > 0: the K-step calls are encoded and submitted `queue` steps at a time. verify_steps > 0: the path is checked against
    the oracle first (eng.verified)."""
    import torch

    from cellularautomatons3d_amd import Engine, host

    offs, s, b = rule_payload(RULES[rule])
    eng = Engine(device)
    eng.configure(G)
    eng.set_option("resident", resident)
    eng.set_rules(*offs, s, b)
    full = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=density_rounds)
    eng.upload_state(full)
    stream = torch.cuda.Stream(device=device)  # the engine runs on a torch-visible stream so torch events bracket its work
    eng.set_stream(stream.cuda_stream)
    eng.bench_stream = stream
    eng.set_option("stats", 0)  # no per-call event pair inside the timed region: the events below bracket all of it
    eng.set_option("graph_prepare", steps)  # graph capture / instantiation stays out of the timed region
    if warmup:
        eng.set_option("graph_prepare", warmup)

    def barrier():
        torch.cuda.synchronize()

    eng.verified = oracle_verify(eng, G, full, rule, verify_steps, queue > 0) if verify_steps > 0 else None
    group = 1
    if queue > 0:
        group = max(1, -(-queue // steps))
        eng.set_option("queue", group * steps)  # a submission = `group` whole calls
    dt, reps, ev_ms, cal, launches = timed_region(eng.step, stream, steps, warmup, min_seconds, barrier, 1, "nccl", eng.flush, group,
                                                  lambda: eng.info().launches_total)
    return eng, full, dt, reps, ev_ms, cal, launches


def roofline_block(kernel, G, bytes_per_step, total_steps, ev_ms, state_bytes, launches=None, variant=None):
    """`launches`: kernel launches in the timed region (the engine's counter); None = one per step. The resident kernel runs a
    whole submission in ONE launch: per launch = per submission (rocprof's kernel duration is a submission)."""
    launches = launches or total_steps
    steps_per_launch = total_steps / launches
    bytes_per_launch = bytes_per_step * steps_per_launch
    if steps_per_launch == int(steps_per_launch):
        steps_per_launch = int(steps_per_launch)
    launch_ms = ev_ms / max(1, launches)
    hbm_rate = bytes_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    traffic = pmc_traffic(kernel, G, steps_per_launch)
    traffic_source = None if traffic is None else ("profiles/pmc_traffic.json (entry's `round`): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                                   "(fabric requests, Infinity Cache hits included), not measured in this run")
    common = {"kernel": kernel, "launch_us": round(launch_ms * 1e3, 3), "steps_per_launch": steps_per_launch,
              "algorithmic_bytes_per_launch": bytes_per_launch, "launches_timed": int(launches),
              "timing": "HIP events on the engine's stream around the whole timed region / launches in it",
              "working_set_bytes": 2 * state_bytes}
    if kernel.startswith("ca_resident"):
        # State in registers + LDS for the whole launch; per step only the tile faces cross the fabric (2 MiB of payload at
        # 512^3). The HBM model does not describe this kernel (its algorithmic-byte rate exceeds the HBM peak): what bounds it is
        # vector-instruction issue. Instructions per launch come from the SQ pass of this command, the time from this run.
        prof, src = sq_profile(kernel, G, variant)
        out = {"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_GWIPS, "unit": "Gwaveinst/s", "frac": None,
               "traffic": traffic, "traffic_source": traffic_source, "variant": variant}
        if prof:
            valu_per_step = prof["SQ_INSTS_VALU"] / prof["steps_per_launch"]
            achieved = valu_per_step * steps_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
            out.update({"achieved": round(achieved, 2), "frac": round(achieved / VALU_PEAK_GWIPS, 4),
                        "valu_wave_instructions_per_step": round(valu_per_step), "valu_per_wave_step": round(valu_per_step / prof["SQ_WAVES"], 1),
                        "waves": int(prof["SQ_WAVES"]), "counter_source": src,
                        "peak_definition": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (SIMD-32; needs >= 2 waves per SIMD); "
                                           "half-rate instructions (v_alignbit, DPP moves) count once, so full issue sits below frac 1"})
        else:
            out["note"] = "no committed profiles/r*_pmc_sq_* pass of this kernel VARIANT (same rule, form options and device sources): instruction count unknown, no fraction"
        out.update(common)
        out["hbm_equivalent"] = {"achieved": round(hbm_rate, 2), "unit": "GB/s", "hbm_peak": HBM_PEAK_GBS,
                                 "note": "algorithmic bytes (0.25 B per cell-step) per second — a rate, not a fraction of anything: the state does not move; "
                                         "the HBM-path fraction of this grid is per_step_kernels.roofline.frac"}
        out["resident"] = "registers + LDS"
        return out
    fits = 2 * state_bytes < INFINITY_CACHE_BYTES
    out = {"bound": "hbm", "achieved": round(hbm_rate, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(hbm_rate / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source}
    out.update(common)
    out["resident"] = ("infinity cache (both ping-pong buffers fit in 256 MiB: the rate is an algorithmic-byte rate against the HBM peak, "
                       "not DRAM traffic)" if fits else "hbm (the ping-pong buffers exceed the 256 MiB Infinity Cache)")
    return out


def main():
    a = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world_env == 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(a))
    # stdout carries the JSON line and nothing else: libraries that print there (gloo's connection banner ...) are sent
    # to stderr from here on
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    from cellularautomatons3d_amd import host, slab

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = world_env
    if world != a.gpus:
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

    G = a.grid
    rule_kw = RULES[a.rule]
    pw = (G // 32) * G
    state_bytes = pw * G * 4

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    se = None
    verified = None
    single_verified = None
    if world == 1:
        queue = a.queue if a.submit == "queued" else 0
        eng, full, dt, reps, ev_ms, cal, launches = single_gpu_leg(local_rank, G, a.rule, a.steps, a.warmup, a.min_seconds, a.density_rounds, a.resident, queue, a.verify_steps)
        core = eng
        single_verified = eng.verified
    else:
        offs, s, b = rule_payload(rule_kw)
        full = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=a.density_rounds)
        native = a.backend == "nccl" and a.transport != "torch"
        if a.transport == "native" and not native:
            raise SystemExit("--transport native needs the RCCL backend")
        if native:
            # the exchange inside the engine (ncclSend / ncclRecv issued by libca3d.so); every rank must take the same
            # path, so a rank that cannot load RCCL fails the run rather than falling back alone
            se = slab.NativeSlabEngine(G, rank, world, ghost=a.ghost, device=local_rank, overlap={"auto": "auto", "on": True, "off": False}[a.overlap])
        else:
            se = slab.SlabEngine(G, rank, world, ghost=a.ghost, device=local_rank, host_staging=a.backend == "gloo",
                                 overlap={"auto": "auto", "on": True, "off": False}[a.overlap])
        if not a.resident:
            se.engine.set_option("resident", 0)  # per-step slab kernels only (rehearsals with several ranks on one GPU: a resident launch needs the whole device)
        se.engine.set_rules(*offs, s, b)
        se.upload_state(full[se.z0 * pw:(se.z0 + se.nz) * pw])
        core = se.engine
        if a.verify_steps > 0:
            # the multi-GPU path has no reference counterpart: its state must equal the single-grid oracle's. Checked here on
            # every run, across at least one halo exchange, before anything is timed.
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import numpy as np
            import oracle_lib as ol

            se.run(a.verify_steps)
            got = core.read_state()
            verify_threads = max(1, min(os.cpu_count() or 1, 128) // world)
            t_or = time.perf_counter()
            want = ol.packed_run(G, full, ol.Rules.from_strings(**rule_kw), a.verify_steps, verify_threads)
            verify_seconds = time.perf_counter() - t_or
            flag = torch.tensor([1 if np.array_equal(got, want[se.z0 * pw:(se.z0 + se.nz) * pw]) else 0], dtype=torch.int32,
                                device="cuda" if a.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            verified = bool(flag.item())
            del want, got
            se.upload_state(full[se.z0 * pw:(se.z0 + se.nz) * pw])
        dt, reps, ev_ms, cal, launches = timed_region(se.run, se.stream, a.steps, a.warmup, a.min_seconds, barrier, world, a.backend)

    total_steps = a.steps * reps
    info = core.info()
    kernel = info.kernel_name.decode()
    cells = float(G) ** 3
    value = cells * total_steps / dt / 1e9

    ok = None
    if a.check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as np
        import oracle_lib as ol

        want = ol.packed_run(G, full, ol.Rules.from_strings(**rule_kw), a.warmup + a.steps * (reps + cal))
        got = core.read_state()
        lo = 0 if world == 1 else se.z0 * pw
        ok = bool(np.array_equal(got, want[lo:lo + got.size]))
        if world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if a.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())

    schedules = None
    if world > 1 and not a.no_schedule_compare:
        # The same slabs under the OTHER batch schedule, timed right behind the headline (state carried on): overlap on = edge zones,
        # exchange posted, interior under it; off = exchange, then the whole slab in one launch. On one GPU's loopback the split form
        # measured slower (two smaller launches per step) — whether the transfer it hides pays for that is a property of the real links,
        # so every multi-GPU line carries both and the reader picks.
        first = bool(se.overlap)
        se.overlap = not first
        dt_o, reps_o, _, _, _ = timed_region(se.run, se.stream, a.steps, a.steps, a.min_seconds / 2, barrier, world, a.backend)
        se.overlap = first
        rec = lambda d, r: {"ms_per_step": round(d * 1e3 / (a.steps * r), 6), "value": round(cells * a.steps * r / d / 1e9, 3), "unit": "Gcells/s"}
        schedules = {"overlap_on" if first else "overlap_off": rec(dt, reps), "overlap_off" if first else "overlap_on": rec(dt_o, reps_o),
                     "headline": "overlap_on" if first else "overlap_off",
                     "why_headline": ("BASELINE configs[4] names the overlapped schedule" if a.config == 5 else "--overlap / slab.py's automatic choice")}
    rccl_evidence = None
    if world > 1:
        # What the run really ran on, from the engines and the communicator themselves — not from WORLD_SIZE: every rank's HIP device, PCI
        # bus id and (native transport) ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the engine's RCCL communicator, gathered to
        # rank 0. N ranks are N GPUs only if the N bus ids differ.
        mine = dict(core.comm_info(), rank=rank, local_rank=local_rank, pid=os.getpid())
        if not getattr(se, "native", False) and a.backend == "nccl":
            mine["torch_backend"] = dist.get_backend()
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if rank == 0:
            ids = [e["pci_bus_id"] for e in everyone]
            rccl_evidence = {"ranks": world, "distinct_devices": len(set(ids)), "devices": everyone,
                             "communicator": ("the engine's own (ca3d_slab_comm_init): counts below are ncclCommCount / ncclCommUserRank / ncclCommCuDevice"
                                              if getattr(se, "native", False) else f"torch.distributed process group, backend {a.backend}"),
                             "comm_ranks_agree": all(e["comm_ranks"] in (-1, world) for e in everyone)}
    multi_render = None
    if world > 1 and a.multi_render:
        multi_render = render_leg_multi(se, G, a, world, rank, barrier)

    if rank == 0:
        # dominant kernel: one launch = one step over this rank's planes; algorithmic bytes per launch = 0.25 B x the
        # cells this rank owns; launches timed = steps in the timed region (an overlapped slab batch
        # issues two smaller launches per step: the pair counts as one)
        own_cells = cells if world == 1 else float(G) * G * se.nz
        out = {
            "metric": "Gcells/s CA step at 512^3" if G == 512 else f"Gcells/s CA step at {G}^3",
            "value": round(value, 3), "unit": "Gcells/s", "n_gpus": world, "steps": a.steps, "reps": reps, "warmup": a.warmup,
            "ms_per_step": round(dt * 1e3 / total_steps, 6), "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{G}^3 uint32-packed grid, rule '{a.rule}' ({rule_kw['neighbourhood']} B{rule_kw['born']}/S{rule_kw['survive']}), "
                                   f"hashed fill seed 0xCA3D0001 density {2.0 ** -(1 + a.density_rounds):g}, one CA step per bench step, "
                                   f"{reps} batches of {a.steps} steps timed",
                       "baseline_config": a.config, "grid": G, "layout": "packed32", "rule": a.rule,
                       "state_note": ("under this rule (S0-6: a live cell never dies) a density-0.5 fill reaches a fixed point within a few steps; the kernels are "
                                      "bit-sliced with no data-dependent path, so the rate does not depend on what the cells do (`--rule vn_b24_s135` keeps "
                                      "changing: same kernel, same time; tests/test_gpu_ca_parity.py runs both for 1000 steps against the oracle)") if a.rule == "default" else None,
                       "parallelism": "1 GPU" if world == 1 else f"z-slab x{world}, ghost {a.ghost} planes, RCCL send/recv every {a.ghost} steps"
                                      + (" overlapped with the interior phase" if se.overlap else "")
                                      + (", exchange issued by the engine (ca3d_slab_run)" if getattr(se, "native", False) else ", exchange through torch.distributed")},
            "roofline": roofline_block(kernel, G, 0.25 * own_cells, total_steps, ev_ms, state_bytes, launches, core.kernel_variant()),
        }
        if world == 1:
            out["config"]["submission"] = (f"queued: ca3d_step({a.steps}) encodes, ca3d_flush submits every {cal * a.steps} steps (the reference's commandEncoder + "
                                           "queue.submit, main_pathtraced.js:1833-1850)" if queue else f"per call: every ca3d_step({a.steps}) is its own submission")
        if world == 1:
            ceiling = copy_ceiling_gbs(eng)
            out["copy_ceiling"] = {"value": round(ceiling, 1), "unit": "GB/s",
                                   "how": "1 GiB float4-per-lane device-to-device copy, non-temporal stores (ca3d_measure_copy), bytes read + bytes written per second, measured in this run"}
        if rccl_evidence is not None:
            out["rccl"] = rccl_evidence
        if schedules is not None:
            out["schedules"] = schedules
        if ok is not None:
            out["oracle_match"] = ok
        if world == 1 and single_verified is not None:
            out["verified"] = single_verified
            if not single_verified["oracle_match"]:
                ok = False
        if verified is not None:
            out["verified"] = {"oracle_match": verified, "steps": a.verify_steps, "oracle_seconds": round(verify_seconds, 2), "oracle_threads_per_rank": verify_threads,
                               "how": "every rank's slab after this many steps (past one halo exchange) against the CPU oracle's full grid, before the timed region"}
            if not verified:
                ok = False

        def per_step_leg(e, Gx, state_b):
            """The same grid through the per-step kernels (every step reads and writes the state): the fraction of the HBM roofline
            in the usual sense, next to a resident kernel's on-chip rate."""
            e.set_option("queue", 0)
            e.set_option("resident", 0)
            e.set_option("graph_prepare", 256)
            dt3, reps3, ev3, _, _ = timed_region(e.step, e.bench_stream, 256, 64, a.min_seconds, barrier, 1, "nccl")
            k3 = e.info().kernel_name.decode()
            leg = {"value": round(float(Gx) ** 3 * 256 * reps3 / dt3 / 1e9, 3), "unit": "Gcells/s", "ms_per_step": round(dt3 * 1e3 / (256 * reps3), 6),
                   "steps": 256, "reps": reps3, "roofline": roofline_block(k3, Gx, 0.25 * float(Gx) ** 3, 256 * reps3, ev3, state_b, None, e.kernel_variant())}
            leg["roofline"]["frac_of_copy_ceiling"] = round(leg["roofline"]["achieved"] / ceiling, 4)
            e.set_option("resident", a.resident)
            return leg

        def per_call_leg(e, Gx, state_b, K):
            """Every ca3d_step(K) its own submission (no ca3d_flush batching): what a host that submits K steps per frame gets."""
            e.set_option("queue", 0)
            dt2, reps2, ev2, _, l2 = timed_region(e.step, e.bench_stream, K, 0, a.min_seconds / 2, barrier, 1, "nccl", e.flush, 1,
                                                  lambda: e.info().launches_total)
            return {"submission": f"per call: every ca3d_step({K}) is its own submission", "value": round(float(Gx) ** 3 * K * reps2 / dt2 / 1e9, 3), "unit": "Gcells/s",
                    "ms_per_step": round(dt2 * 1e3 / (K * reps2), 6), "steps": K, "reps": reps2,
                    "roofline": roofline_block(e.info().kernel_name.decode(), Gx, 0.25 * float(Gx) ** 3, K * reps2, ev2, state_b, l2, e.kernel_variant())}

        resident_headline = world == 1 and kernel.startswith("ca_resident")
        if world == 1 and resident_headline and (a.compare_submission or (queue and not a.no_per_call_leg)):
            if queue:
                out["per_call"] = per_call_leg(eng, G, state_bytes, a.steps)
            if a.compare_submission:
                # the same K-step calls under the other submission mode, for the record (same engine, state carried on)
                q2 = 0 if queue else a.queue
                g2 = max(1, -(-q2 // a.steps)) if q2 else 1
                eng.set_option("queue", g2 * a.steps if q2 else 0)
                dt2, reps2, ev2, _, l2 = timed_region(eng.step, eng.bench_stream, a.steps, 0, a.min_seconds, barrier, 1, "nccl", eng.flush, g2,
                                                      lambda: eng.info().launches_total)
                out["other_submission"] = {"submission": "queued" if q2 else "per call", "value": round(cells * a.steps * reps2 / dt2 / 1e9, 3), "unit": "Gcells/s",
                                           "ms_per_step": round(dt2 * 1e3 / (a.steps * reps2), 6), "reps": reps2,
                                           "roofline": roofline_block(kernel, G, 0.25 * cells, a.steps * reps2, ev2, state_bytes, l2, eng.kernel_variant())}
        if world == 1 and resident_headline and not a.no_per_step_leg:
            out["per_step_kernels"] = per_step_leg(eng, G, state_bytes)
        if world == 1 and not resident_headline:
            out["roofline"]["frac_of_copy_ceiling"] = round(out["roofline"]["achieved"] / ceiling, 4)
        if world == 1 and not a.no_render:
            out["render"] = render_leg(eng, G, a)
            if a.config == 3 and a.render_size == "1920x1080":
                out["render_4k"] = render_leg(eng, G, a, size="3840x2160", sparse=False, literal=False)  # BASELINE configs[4]'s frame size
            if a.config == 3 and not a.no_interactive:
                out["interactive_frame"] = interactive_leg(eng, G, a)
        if world == 1 and a.config == 3 and G == 512 and not a.no_grid_256:
            # BASELINE configs[1]'s grid (256^3): the resident kernel's 256^3 form under queued submission, and the per-step kernels
            eng.close()
            e1, _, dt1, reps1, ev1, cal1, l1 = single_gpu_leg(local_rank, 256, "default", a.steps, a.warmup, a.min_seconds, 0, a.resident, a.queue if a.submit == "queued" else 0,
                                                              a.verify_steps)
            k1 = e1.info().kernel_name.decode()
            out["grid_256"] = {"grid": 256, "rule": "default", "value": round(256.0 ** 3 * a.steps * reps1 / dt1 / 1e9, 3), "unit": "Gcells/s",
                               "ms_per_step": round(dt1 * 1e3 / (a.steps * reps1), 6), "steps": a.steps, "reps": reps1,
                               "roofline": roofline_block(k1, 256, 0.25 * 256.0 ** 3, a.steps * reps1, ev1, 2 << 20, l1, e1.kernel_variant())}
            if e1.verified is not None:
                out["grid_256"]["verified"] = e1.verified
                if not e1.verified["oracle_match"]:
                    ok = False
            if k1.startswith("ca_resident") and not a.no_per_step_leg:
                out["grid_256"]["per_step_kernels"] = per_step_leg(e1, 256, 2 << 20)
            eng = e1
            # the reference UI's start-up grid (main_pathtraced.js:101): 64^3 — one workgroup holds the whole state (resident64_run)
            eng.close()
            e0, _, dt0, reps0, ev0, cal0, l0 = single_gpu_leg(local_rank, 64, "default", a.steps, a.warmup, min(a.min_seconds, 0.2), 0, a.resident, a.queue if a.submit == "queued" else 0,
                                                              a.verify_steps)
            k0 = e0.info().kernel_name.decode()
            out["grid_64"] = {"grid": 64, "rule": "default", "value": round(64.0 ** 3 * a.steps * reps0 / dt0 / 1e9, 3), "unit": "Gcells/s",
                              "ms_per_step": round(dt0 * 1e3 / (a.steps * reps0), 6), "steps": a.steps, "reps": reps0, "kernel": k0,
                              "what": "the reference UI's start-up grid; batches through the one-workgroup resident kernel (the whole state in one CU's registers)"}
            if e0.verified is not None:
                out["grid_64"]["verified"] = e0.verified
                if not e0.verified["oracle_match"]:
                    ok = False
            if k0.startswith("ca_resident") and not a.no_per_step_leg:
                e0.set_option("queue", 0)
                e0.set_option("resident", 0)
                dtp, repsp, _, _, _ = timed_region(e0.step, e0.bench_stream, 256, 64, min(a.min_seconds, 0.2), barrier, 1, "nccl")
                out["grid_64"]["per_step_kernels"] = {"ms_per_step": round(dtp * 1e3 / (256 * repsp), 6), "kernel": e0.info().kernel_name.decode()}
                e0.set_option("resident", a.resident)
            eng = e0
        if world == 1 and a.config == 3 and G == 512 and not a.no_scaling_base:
            # the single-GPU point of the multi-GPU curve, on the multi-GPU grid, in the same run (N > 1 runs 1024^3)
            eng.close()
            e2, _, dt2, reps2, ev2, _, _ = single_gpu_leg(local_rank, 1024, "default", 256, 64, a.min_seconds)
            k2 = e2.info().kernel_name.decode()
            out["scaling_base"] = {"grid": 1024, "rule": "default", "n_gpus": 1, "value": round(1024.0 ** 3 * 256 * reps2 / dt2 / 1e9, 3), "unit": "Gcells/s",
                                   "ms_per_step": round(dt2 * 1e3 / (256 * reps2), 6), "steps": 256, "reps": reps2,
                                   "roofline": roofline_block(k2, 1024, 0.25 * 1024.0 ** 3, 256 * reps2, ev2, 128 << 20, None, e2.kernel_variant()),
                                   "note": "divide the N > 1 values (config 4: 1024^3) by this, not by the 512^3 headline; "
                                           "`python bench.py --gpus 1 --config 5` gives the base of the 2048^3 clustered curve"}
            out["scaling_base"]["roofline"]["frac_of_copy_ceiling"] = round(out["scaling_base"]["roofline"]["achieved"] / ceiling, 4)
            e2.close()
            # The same grid on the same GPU as FOUR Z-slabs behind the group handle (ca3d_group_*, all slabs on this device): every slab's
            # 256 + 2 x 16 planes run their 16-step batches through the resident slab kernel, ghost planes refreshed by device copies — the
            # state crosses HBM once per batch instead of once per step. Bit-exactness of this split: tests/test_gpu_slab.py::
            # test_engine_group_single_thread_split (1024^3, four slabs).
            out["scaling_base"]["as_four_resident_slabs"] = group_leg(local_rank, 1024, 4, 16, a.min_seconds)
        if multi_render is not None:
            out["render"] = multi_render
            if multi_render.get("frame_match") is False:
                ok = False
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(min(G, 512), rule_kw, a.cpu_seconds)
            out["cpu_baseline_js"] = cpu_baseline_js(G=min(G, 512))  # the JS stepper at G in {256, 512}
            if G == 512:
                out["cpu_baseline_js_256"] = cpu_baseline_js(seconds=3.0, G=256)
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if ok is False:
        raise SystemExit("state does not match the oracle")


if __name__ == "__main__":
    main()

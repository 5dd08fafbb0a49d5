"""GPU parity of the volume renderer (C ABI ca3d_render) against the float32 CPU oracle.

Tolerance (SURVEY 8(a) row R-par, stated here as the test's contract): on >= 99.9 % of pixels linear-light RGB
abs error <= 2e-3, depth abs error <= max(1e-4, one binary16 ulp of the stored value) — the depth target is
RG16F as in the reference — and presentation <= 1/255; the remaining <= 0.1 % are silhouette pixels where a
grazing ray may pick the neighbouring cell."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from cellularautomatons3d_amd import host
from gpu_common import rules, set_rules

pytestmark = pytest.mark.gpu


@pytest.fixture(scope=lambda fixture_name, config: os.environ.get("CA3D_TEST_ENGINE_SCOPE", "module"))
def eng():
    # one engine per file by default (faster, and state carried from test to test is itself a test); CA3D_TEST_ENGINE_SCOPE=function gives
    # every test a fresh one: a test that only passes behind another shows up (round 4 found a renderer cache bug that way)
    from cellularautomatons3d_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def _compare(eng, cells, G, u, W, H, spp, rows=None, min_ok=0.999):
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(cells)
    # the plain kernel (one pixel per lane, samples in order), the in-wave scheduled one and the ray-stream passes (the default; also with
    # every decision of their interval filter checked against the slab test: a contradiction fails the call) must agree bit for bit
    eng.set_option("render_sched", 0)
    pres0, light0, depth0 = eng.render(u, W, H, spp)
    st0 = eng.render_stats()
    eng.set_option("render_sched", 1)
    for stream, check in ((0, 0), (1, 1), (1, 0)):
        eng.set_option("render_stream", stream)
        eng.set_option("render_stream_check", check)
        pres, light, depth = eng.render(u, W, H, spp)
        st1 = eng.render_stats()
        np.testing.assert_array_equal(pres, pres0)
        np.testing.assert_array_equal(light.view(np.uint16), light0.view(np.uint16))
        np.testing.assert_array_equal(depth.view(np.uint16), depth0.view(np.uint16))
        assert (st0.shadow_rays, st0.primary_cell_visits, st0.shadow_cell_visits) == (st1.shadow_rays, st1.primary_cell_visits, st1.shadow_cell_visits)
    olight, odepth, opres, oshadow = ol.render(cells, G, u, W, H, spp, rows)
    y0, y1 = rows if rows else (0, H)
    sl = slice(y0, y1)
    l = light[sl].astype(np.float32)
    d = depth[sl].astype(np.float32)
    od16 = odepth[sl].astype(np.float16).astype(np.float32)
    ulp = np.maximum(np.spacing(od16.astype(np.float16)).astype(np.float32), 1e-4)
    ok_rgb = np.abs(l[..., :3] - olight[sl][..., :3]).max(-1) <= 2e-3
    ok_depth = np.abs(d[..., 0] - od16[..., 0]) <= ulp[..., 0]
    want8 = np.rint(np.clip(opres[sl], 0, 1) * 255.0)
    ok_pres = np.abs(pres[sl].astype(np.float32) - want8).max(-1) <= 1.0
    frac = (ok_rgb & ok_depth & ok_pres).mean()
    assert frac >= min_ok, (frac, ok_rgb.mean(), ok_depth.mean(), ok_pres.mean())
    assert (light[sl][..., 3] == 1.0).all() and (depth[sl][..., 1] == 1.0).all()
    assert olight[sl][..., :3].max() > 0.05  # the scene is not empty
    if rows is None:
        st = eng.render_stats()
        assert st.primary_rays == W * H * spp
        assert abs(int(st.shadow_rays) - oshadow) <= max(4, int(0.001 * oshadow))
    return frac


@pytest.mark.parametrize("pose", ["default", "oblique"])
@pytest.mark.parametrize("spp", [1, 4])
def test_random_volume(eng, pose, spp):
    G, W, H = 64, 320, 180
    cells = host.random_fill(host.words_per_buffer(G), seed=3, and_rounds=4)
    vm = host.camera_matrix() if pose == "default" else host.orbit_camera()
    _compare(eng, cells, G, host.uniform_block(W, H, vm), W, H, spp)


@pytest.mark.parametrize("G,W,H,spp", [(96, 320, 180, 4), (160, 320, 180, 1), (992, 1920, 1080, 4)])
def test_grids_that_are_not_a_power_of_two(eng, G, W, H, spp):
    """Every multiple of 32 is a UI grid (main_pathtraced.js:268-279, 675-693; "1000" -> 992): the ray-stream walks read the bricked
    copy of such a volume too (bricks of 8^3 divide every multiple of 32; brick index by multiply-adds instead of shifts) — plain,
    scheduled and stream forms bit for bit the same, the oracle's frame within the tolerance. 992^3: a 24-row band of the 1080p frame."""
    cells = host.random_fill(host.words_per_buffer(G), seed=7, and_rounds=4)
    _compare(eng, cells, G, host.uniform_block(W, H, host.orbit_camera()), W, H, spp, rows=(520, 544) if G == 992 else None)
    assert eng.info().grid_size == G


@pytest.mark.parametrize("G,steps", [(96, 12), (160, 30), (288, 40)])
def test_sparse_scenes_on_grids_that_are_not_a_power_of_two(eng, G, steps):
    """The UI's start-up seed a few steps on, on grids that are not a power of two: the sparse-volume kernels (occupancy bits over 32 x 8 x 8
    blocks, the box of the live blocks, block jumps, the spread kernel; no second occupancy level off multiples of 128) against the
    oracle, converged frame at one and four samples and the literal frame's two forms against each other."""
    cells = ol.packed_run(G, host.initial_state(G), rules("default"), steps)
    W, H = 320, 180
    for pose in (host.camera_matrix(), host.orbit_camera(1.2, (1.0, 0.3, 0.0), 0.8)):
        for spp in (1, 4):
            _compare(eng, cells, G, host.uniform_block(W, H, pose), W, H, spp)
    eng.set_render_mode(True)
    try:
        got = {}
        vm = host.orbit_camera(1.2, (1.0, 0.3, 0.0), 0.8)
        for bricks in (0, 1):
            eng.set_option("render_frame_bricks", bricks)
            eng.reset_render_history()
            got[bricks] = [eng.render(host.uniform_block(W, H, vm, elapsed_time=0.1 + 0.07 * f, prev_view_mat=vm if f else None), W, H, 1) for f in range(3)]
        for a, b in zip(got[0], got[1]):
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
        u = host.uniform_block(W, H, vm, elapsed_time=0.1, prev_view_mat=None)
        ol_light, ol_depth, _ = ol.render_frame(cells, G, u, W, H, None, None)
        ok = np.abs(got[1][0][1].astype(np.float32)[..., :3] - ol_light.astype(np.float16).astype(np.float32)[..., :3]).max(-1) <= 2e-3
        assert ok.mean() >= 0.999, ok.mean()
    finally:
        eng.set_option("render_frame_bricks", 1)
        eng.set_render_mode(False)


def test_evolved_seed_volume(eng):
    # state after 30 steps of the default rule from the single seed (SURVEY 8(d) render input)
    G, W, H = 128, 480, 270
    cells = ol.packed_run(G, host.initial_state(G), rules("default"), 30)
    _compare(eng, cells, G, host.uniform_block(W, H, host.orbit_camera(1.2, (1.0, 0.3, 0.0), 0.8)), W, H, 1)


@pytest.mark.parametrize("G,W,H,spp", [(128, 320, 180, 4), (256, 480, 270, 1)])
def test_empty_space_skipping_on_a_sparse_volume(eng, G, W, H, spp):
    """The evolved single seed fills a few percent of the 32 x 8 x 8 blocks: walks jump over the empty ones. Against
    the oracle's cell-by-cell walk (tolerance of this file), against the unskipped GPU walk (same tolerance: re-seeded
    boundary times may tie-break differently at cell corners), and far fewer cell visits."""
    cells = ol.packed_run(G, host.initial_state(G), rules("default"), 30)
    for pose in (host.camera_matrix(), host.orbit_camera(1.2, (1.0, 0.3, 0.0), 0.8)):
        u = host.uniform_block(W, H, pose)
        _compare(eng, cells, G, u, W, H, spp)               # skipping is on by default: GPU vs oracle
        skipped = eng.render(u, W, H, spp)
        v_skip = eng.render_stats().primary_cell_visits
        eng.set_option("render_skip", 0)
        plain = eng.render(u, W, H, spp)
        v_plain = eng.render_stats().primary_cell_visits
        eng.set_option("render_skip", 1)
        same = (np.abs(skipped[1].astype(np.float32) - plain[1].astype(np.float32)).max(-1) <= 2e-3)
        assert same.mean() >= 0.999, same.mean()
        assert v_skip * 3 < v_plain, (v_skip, v_plain)


@pytest.mark.parametrize("G,rounds,W,H,spp", [(128, 12, 320, 180, 4), (256, 13, 480, 270, 1), (96, 13, 320, 180, 4), (256, 15, 320, 180, 4)])
def test_scattered_sparse_volume(eng, G, rounds, W, H, spp):
    """Live cells everywhere, but in under a quarter of the 32 x 8 x 8 blocks (a hashed fill of density 2^-(rounds + 1)): the frame of the
    block-skipping walks — walk<.., SKIP> in the plain kernel, the in-wave scheduled kernel (render_stream 0) and, the default since
    round 5, the ray-stream passes (ca_stream_walk2<.., SKIP>: live-box clip at set-up, two-level block jumps inside the stepping loop) —
    bit for bit the same from all three, cell-visit counts included, and within the file's tolerance of the oracle's cell-by-cell walk.
    The scene must really be of that kind: far fewer visits than the unskipped walk, and a live box that is not small."""
    cells = host.random_fill(host.words_per_buffer(G), seed=17 + G, and_rounds=rounds)
    for pose in (host.camera_matrix(), host.orbit_camera(1.3, (1.0, 1.0, 0.0), 0.6)):
        u = host.uniform_block(W, H, pose)
        _compare(eng, cells, G, u, W, H, spp, min_ok=0.998)
        v_skip = eng.render_stats().primary_cell_visits
        eng.set_option("render_skip", 0)
        eng.render(u, W, H, spp)
        v_plain = eng.render_stats().primary_cell_visits
        eng.set_option("render_skip", 1)
        assert v_skip * 2 < v_plain, (v_skip, v_plain)


@pytest.mark.parametrize("spp", [1, 4])
def test_small_live_box_frame(eng, spp):
    """A sparse volume whose live cells sit in a small box (the reference's start-up seed a few steps on): view rays that miss
    the box of the occupied blocks are answered without a walk and the frame is rendered one lane per sample by
    ca_render_packed_spread — bit for bit the plain kernel's frame (inside _compare), the oracle's within the tolerance, and
    almost no cell visits."""
    G, W, H = 256, 320, 180
    cells = ol.packed_run(G, host.initial_state(G), rules("default"), 12)
    for pose in (host.camera_matrix(), host.orbit_camera(1.2, (1.0, 0.3, 0.0), 0.8)):
        _compare(eng, cells, G, host.uniform_block(W, H, pose), W, H, spp)
        st = eng.render_stats()
        assert st.primary_cell_visits < 2 * st.primary_rays, (st.primary_cell_visits, st.primary_rays)
    # nothing alive at all: every ray is answered at once, only the light gizmo can show
    eng.upload_state(np.zeros(host.words_per_buffer(G), dtype=np.uint32))
    u = host.uniform_block(W, H, host.camera_matrix())
    pres, light, depth = eng.render(u, W, H, spp)
    olight, odepth, opres, _ = ol.render(np.zeros(host.words_per_buffer(G), dtype=np.uint32), G, u, W, H, spp)
    np.testing.assert_array_equal(light.astype(np.float32), olight)
    assert eng.render_stats().primary_cell_visits == 0


def test_camera_inside_volume(eng):
    G, W, H = 64, 160, 90
    cells = host.random_fill(host.words_per_buffer(G), seed=5, and_rounds=5)
    _compare(eng, cells, G, host.uniform_block(W, H, host.camera_matrix((0.1, -0.05, 0.2), (0.0, 1.0, 0.0), 0.4)), W, H, 1, min_ok=0.995)


@pytest.mark.parametrize("position,axis,angle", [((0.9, 0.0, 0.9), (0.0, 1.0, 0.0), 0.35),   # the volume crosses the left edge of the frame
                                                 ((0.0, 0.0, 0.75), (0.0, 1.0, 0.0), 3.0),    # looking away: the volume is off screen
                                                 ((0.0, 0.3, 4.0), (1.0, 0.0, 0.0), -0.05),   # far away: a rectangle of a few tiles
                                                 ((0.2, 0.6, 0.4), (1.0, 0.0, 0.0), -0.9)])   # close above a face: corners beside the camera
def test_volume_rectangle_cases(eng, position, axis, angle):
    """The scheduled kernel runs on the volume's screen rectangle and the plain kernel on the tiles around it
    (render.hip: volume_rect): partly or wholly off screen, tiny, and with corners that do not project, the frame
    stays the plain kernel's bit for bit (checked inside _compare) and the oracle's within the tolerance."""
    G, W, H = 64, 320, 180
    cells = host.random_fill(host.words_per_buffer(G), seed=21, and_rounds=3)
    u = host.uniform_block(W, H, host.camera_matrix(position, axis, angle))
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(cells)
    for spp in (1, 4):
        eng.set_option("render_sched", 0)
        pres0, light0, depth0 = eng.render(u, W, H, spp)
        st0 = eng.render_stats()
        eng.set_option("render_sched", 1)
        for stream, check in ((0, 0), (1, 1), (1, 0)):
            eng.set_option("render_stream", stream)
            eng.set_option("render_stream_check", check)
            pres, light, depth = eng.render(u, W, H, spp)
            st1 = eng.render_stats()
            np.testing.assert_array_equal(pres, pres0)
            np.testing.assert_array_equal(light.view(np.uint16), light0.view(np.uint16))
            np.testing.assert_array_equal(depth.view(np.uint16), depth0.view(np.uint16))
            assert (st0.shadow_rays, st0.primary_cell_visits, st0.shadow_cell_visits) == (st1.shadow_rays, st1.primary_cell_visits, st1.shadow_cell_visits)
    olight, odepth, opres, _ = ol.render(cells, G, u, W, H, 4)
    ok = np.abs(light.astype(np.float32)[..., :3] - olight[..., :3]).max(-1) <= 2e-3
    assert ok.mean() >= 0.995, ok.mean()


def test_material_colour_depth_overlay_and_gamma(eng):
    G, W, H = 64, 160, 90
    cells = host.random_fill(host.words_per_buffer(G), seed=6, and_rounds=4)
    u = host.uniform_block(W, H, host.orbit_camera(), materialColor=(0.8, 0.2, 0.1), showDepthOverlay=1, gamma=2.2,
                           roughness=0.6, cellSize=0.6, light=(-0.9, 0.4, 1.2, 2.0))
    _compare(eng, cells, G, u, W, H, 1)


def test_light_gizmo_and_empty_volume(eng):
    G, W, H = 32, 128, 72
    cells = np.zeros(host.words_per_buffer(G), dtype=np.uint32)
    u = host.uniform_block(W, H, light=(0.0, 0.0, 0.6, 5.0))  # light between camera and volume, on the view axis
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(cells)
    pres, light, depth = eng.render(u, W, H, 1)
    olight, odepth, opres, _ = ol.render(cells, G, u, W, H, 1)
    np.testing.assert_array_equal(light.astype(np.float32), olight)
    assert light[..., :3].max() == 1.0 and (light[..., :3].sum(-1) > 0).sum() < 40  # only the gizmo is white
    assert eng.render_stats().shadow_rays == 0


def test_derived_buffers_follow_the_state(eng):
    """The occupancy bits and the bricked copy of the volume are rebuilt only when the state has changed since the frame before
    (step count, uploads, buffers handed out): frames of a changing state must never show the old one — dense and sparse scenes, both
    frame modes, a same-sized second upload, stepping between frames."""
    G, W, H = 64, 160, 90
    u = host.uniform_block(W, H, host.orbit_camera())
    eng.configure(G)
    set_rules(eng, rules("default"))
    a = host.random_fill(host.words_per_buffer(G), seed=31, and_rounds=3)
    b = host.random_fill(host.words_per_buffer(G), seed=32, and_rounds=3)
    sparse = host.initial_state(G)

    def frame(mode):
        eng.set_render_mode(mode)
        try:
            if mode:
                eng.reset_render_history()
            return [x.copy() for x in eng.render(u, W, H, 1)]
        finally:
            eng.set_render_mode(False)

    def fresh(state, steps, mode):
        from cellularautomatons3d_amd import Engine

        with Engine(0) as e2:
            e2.configure(G)
            set_rules(e2, rules("default"))
            e2.upload_state(state)
            if steps:
                e2.step(steps)
            e2.set_render_mode(mode)
            if mode:
                e2.reset_render_history()
            return [x.copy() for x in e2.render(u, W, H, 1)]

    for mode in (False, True):
        eng.upload_state(a)
        f1 = frame(mode)
        f1b = frame(mode)  # same state again: the cached buffers
        for x, y in zip(f1, f1b):
            np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
        eng.upload_state(b)  # same size, same step count, same buffers: only the serial tells
        for x, y in zip(frame(mode), fresh(b, 0, mode)):
            np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
        eng.step(2)
        for x, y in zip(frame(mode), fresh(b, 2, mode)):
            np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
        eng.upload_state(sparse)  # dense -> sparse: the other set of kernels, the live box
        eng.step(3)
        for x, y in zip(frame(mode), fresh(sparse, 3, mode)):
            np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
        assert not np.array_equal(f1[0], fresh(b, 0, mode)[0])


def test_state_written_through_a_device_pointer(eng):
    """ca3d_device_buffer hands out a pointer the caller may write through (slab.SlabRenderer over the torch transports does: every
    gather_volume() copies the new state into it, the full-grid engine never steps and nothing else tells it): every later frame must
    show what is in the buffer NOW. Round 4's cache of the derived buffers keyed them on {serial, step, pointer}, the serial bumped
    only when the pointer was handed out — the second and all later frames drew the first volume."""
    from cellularautomatons3d_amd import Engine, slab

    G, W, H = 64, 160, 90
    u = host.uniform_block(W, H, host.orbit_camera())
    a = host.random_fill(host.words_per_buffer(G), seed=41, and_rounds=3)
    b = host.random_fill(host.words_per_buffer(G), seed=42, and_rounds=3)
    sparse = host.initial_state(G)
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(a)
    import torch

    vol = slab.device_tensor(*eng.device_buffer(0), 0)
    for mode in (False, True):
        eng.set_render_mode(mode)
        try:
            for state in (a, b, sparse, b):
                vol.copy_(torch.from_numpy(state.view(np.int32)).view(vol.dtype).reshape(vol.shape))
                torch.cuda.synchronize()
                if mode:
                    eng.reset_render_history()
                got = [x.copy() for x in eng.render(u, W, H, 1)]
                again = eng.render(u, W, H, 1) if not mode else got
                with Engine(0) as e2:
                    e2.configure(G)
                    set_rules(e2, rules("default"))
                    e2.upload_state(state)
                    e2.set_render_mode(mode)
                    want = e2.render(u, W, H, 1)
                for x, y, z in zip(got, want, again):
                    np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
                    np.testing.assert_array_equal(z.view(np.uint8), y.view(np.uint8))
        finally:
            eng.set_render_mode(False)
    # the pointer's validity ends with the next upload: frames of an unchanged state reuse the derived buffers again
    eng.upload_state(a)
    f1 = eng.render(u, W, H, 1)
    f2 = eng.render(u, W, H, 1)
    np.testing.assert_array_equal(f1[0], f2[0])


def test_render_follows_step_parity(eng):
    # the render pass binds buffer [step % 2] (main_pathtraced.js:1788)
    G, W, H = 64, 96, 54
    eng.restart_sim(G, "von neumann", "1,3", "0-6")
    u = host.uniform_block(W, H)
    for n in (0, 3, 4):
        if n:
            eng.step(n if n == 3 else 1)
        _, light, _ = eng.render(u, W, H, 1)
        want, _, _, _ = ol.render(eng.read_state(), G, u, W, H, 1)
        assert np.abs(light.astype(np.float32) - want).max() <= 2e-3


def test_full_hd_band_at_512(eng):
    # BASELINE config 3 shape: 512^3 volume, 1920x1080, 4 spp; the oracle renders a 24-row band.
    G, W, H = 512, 1920, 1080
    cells = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)
    _compare(eng, cells, G, host.uniform_block(W, H, host.orbit_camera()), W, H, 4, rows=(520, 544))


def test_indirect_lighting_mode(eng):
    """The reference's calculateIndirectLighting (pathtraced_fragment_clustered.wgsl:307-377; its call is commented out at
    :424): an optional mode here ("render_indirect"), same tolerance as the direct frame, and it must change the frame
    (a dense volume has lit neighbours next to most visible faces)."""
    G, W, H = 64, 320, 180
    cells = host.random_fill(host.words_per_buffer(G), seed=9, and_rounds=2)
    u = host.uniform_block(W, H, host.orbit_camera())
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(cells)
    _, direct, _ = eng.render(u, W, H, 1)
    eng.set_option("render_indirect", 1)
    try:
        for spp in (1, 4):
            pres, light, depth = eng.render(u, W, H, spp)
            olight, odepth, opres, _ = ol.render(cells, G, u, W, H, spp, indirect=True)
            ok = np.abs(light.astype(np.float32)[..., :3] - olight[..., :3]).max(-1) <= 2e-3
            ok &= np.abs(pres.astype(np.float32) - np.rint(np.clip(opres, 0, 1) * 255.0)).max(-1) <= 1.0
            assert ok.mean() >= 0.999, ok.mean()
            if spp == 1:
                changed = np.abs(light.astype(np.float32)[..., :3] - direct.astype(np.float32)[..., :3]).max(-1) > 4e-3
                assert changed.mean() > 0.005, changed.mean()
    finally:
        eng.set_option("render_indirect", 0)


def test_4k_band_at_512(eng):
    # BASELINE config 5's render leg: 3840x2160 @ 4 spp (fragment_main, pathtraced_fragment_clustered.wgsl:800-890,
    # once per pixel of a 4K target); the oracle renders a 16-row band through the middle of the volume's silhouette.
    G, W, H = 512, 3840, 2160
    cells = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)
    _compare(eng, cells, G, host.uniform_block(W, H, host.orbit_camera()), W, H, 4, rows=(1040, 1056))


def test_4k_band_on_the_evolved_2048_grid(eng):
    # config 5 as a whole: the 2048^3 grid under the clustered rule-set, then the 4K frame of that state (1 spp band
    # against the oracle; the oracle steps the same state on the CPU)
    G, W, H = 2048, 3840, 2160
    r = rules("clustered")
    st = host.random_fill(host.words_per_buffer(G), seed=11, and_rounds=2)
    want = ol.packed_run(G, st, r, 1)
    eng.configure(G)
    set_rules(eng, r)
    eng.upload_state(st)
    eng.step(1)
    assert np.array_equal(eng.read_state(), want)
    u = host.uniform_block(W, H, host.orbit_camera())
    pres, light, depth = eng.render(u, W, H, 1, rows=(1072, 1088))
    olight, odepth, opres, _ = ol.render(want, G, u, W, H, 1, (1072, 1088))
    sl = slice(1072, 1088)
    ok = (np.abs(light[sl].astype(np.float32)[..., :3] - olight[sl][..., :3]).max(-1) <= 2e-3)
    od16 = odepth[sl].astype(np.float16).astype(np.float32)
    ulp = np.maximum(np.spacing(od16.astype(np.float16)).astype(np.float32), 1e-4)
    ok &= np.abs(depth[sl].astype(np.float32)[..., 0] - od16[..., 0]) <= ulp[..., 0]
    ok &= np.abs(pres[sl].astype(np.float32) - np.rint(np.clip(opres[sl], 0, 1) * 255.0)).max(-1) <= 1.0
    assert ok.mean() >= 0.999, ok.mean()
    assert olight[sl][..., :3].max() > 0.05
    eng.configure(32)  # release the 2 GiB


def test_row_bands_stitch_into_the_full_frame(eng):
    """A frame shared between GPUs is rendered in bands of image rows: the bands of one engine, stitched,
    are the full frame bit for bit; rows outside a band are left untouched."""
    G, W, H = 64, 200, 150
    cells = host.random_fill(host.words_per_buffer(G), seed=5, and_rounds=3)
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(cells)
    u = host.uniform_block(W, H, host.orbit_camera())
    for spp in (4, 1):  # 16-row tiles at 4 samples per pixel, 32-row tiles at 1: bands start on multiples of 16 in both
        pres, light, depth = eng.render(u, W, H, spp)
        got = np.zeros_like(pres)
        for y0, y1 in ((0, 48), (48, 64), (64, 150)):
            p, l, d = eng.render(u, W, H, spp, rows=(y0, y1))
            got[y0:y1] = p[y0:y1]
            np.testing.assert_array_equal(l[y0:y1].view(np.uint16), light[y0:y1].view(np.uint16))
            np.testing.assert_array_equal(d[y0:y1].view(np.uint16), depth[y0:y1].view(np.uint16))
            assert eng.render_stats().primary_rays == W * (y1 - y0) * spp
        np.testing.assert_array_equal(got, pres)
    from cellularautomatons3d_amd import Ca3dError
    with pytest.raises(Ca3dError):
        eng.render(u, W, H, 1, rows=(8, 64))     # a band starts on a tile row
    with pytest.raises(Ca3dError):
        eng.render(u, W, H, 1, rows=(160, 176))  # empty for this target
    eng.render(u, W, H, 1)                       # rows=None restores the whole frame


def test_render_errors(eng):
    from cellularautomatons3d_amd import Ca3dError

    eng.configure(64)
    set_rules(eng, rules("default"))
    eng.upload_state(host.initial_state(64))
    u = host.uniform_block(64, 64)
    with pytest.raises(Ca3dError):
        eng.render(u, 64, 64, 3)
    with pytest.raises(Ca3dError):
        eng.render(u, 0, 64, 1)
    with pytest.raises(ValueError):
        eng.render(u[:100], 64, 64, 1)


def test_literal_frame_mode_tracks_the_oracle(eng):
    """N3: the reference's per-frame process (jittered marches, history reads, EMA) — GPU and oracle run in lock-step,
    each feeding on its own history, and must stay together."""
    G, W, H = 32, 160, 90
    cells = host.random_fill(host.words_per_buffer(G), seed=11, and_rounds=4)
    vm = host.orbit_camera(1.3, (1.0, 0.4, 0.0), 0.7)
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(cells)
    eng.set_render_mode(True)
    try:
        eng.render(host.uniform_block(W, H, vm), W, H, 1)  # allocates the targets
        eng.reset_render_history()
        pl = pd = None
        for f in range(12):
            u = host.uniform_block(W, H, vm, elapsed_time=0.1 + 0.137 * f, prev_view_mat=vm if f else None)
            pres, light, depth = eng.render(u, W, H, 1)
            ol_light, ol_depth, ol_pres = ol.render_frame(cells, G, u, W, H, pl, pd)
            pl, pd = ol_light.astype(np.float16).astype(np.float32), ol_depth.astype(np.float16).astype(np.float32)
            ok = (np.abs(light.astype(np.float32)[..., :3] - pl[..., :3]).max(-1) <= 2e-3) & \
                 (np.abs(depth.astype(np.float32)[..., 0] - pd[..., 0]) <= 2e-3)
            assert ok.mean() >= 0.999, (f, ok.mean())
        want8 = np.rint(np.clip(ol_pres, 0, 1) * 255.0)
        assert (np.abs(pres.astype(np.float32) - want8).max(-1) <= 1).mean() >= 0.999
        # and the accumulated frame sits on the converged one
        eng.set_render_mode(False)
        _, limit, _ = eng.render(host.uniform_block(W, H, vm), W, H, 1)
        assert np.abs(light.astype(np.float32)[..., :3] - limit.astype(np.float32)[..., :3]).mean() < 0.01
        with pytest.raises(Exception):
            eng.set_render_mode(True)
            eng.render(u, W, H, 4)
    finally:
        eng.set_render_mode(False)


def _moving_camera(f):
    """A camera path that takes every branch of R6 / R10 (checked on the oracle side below): an orbit of 0.04 rad per frame (frames
    0-9: reprojected uvs on screen, cell identity decides), a pan that puts most of the volume off screen (10), the pan back (11: the
    points that come back into view reproject OUTSIDE [0,1]^2), a jump to a pose inside the volume (12) and a turn there (13)."""
    if f < 10:
        return host.orbit_camera(1.3, (1.0, 0.4, 0.0), 0.7 + 0.04 * f)
    if f == 10:
        return host.camera_matrix((0.1, 0.05, 1.2), (0, 1, 0), 0.85)
    if f == 11:
        return host.camera_matrix((0.1, 0.05, 1.2), (0, 1, 0), 0.1)
    if f == 12:
        return host.camera_matrix((0.15, 0.0, 0.45), (0, 1, 0), 0.0)
    return host.camera_matrix((0.15, 0.02, 0.5), (0, 1, 0), 0.3)


@pytest.mark.parametrize("G,W,H,seed", [(32, 160, 90, 11), (256, 640, 360, 0xCA3D0001), (96, 320, 180, 5)])
def test_literal_frame_under_a_moving_camera(eng, G, W, H, seed):
    """R6 / R10 with previous matrices that DIFFER from the current ones (main_pathtraced.js:504-524 writes them on every frame the
    user moves, :858-969): getReprojectedUV (:473-487), the prevCameraPos reconstruction and the out-of-range-uv / cell-identity
    branches of mixWithReprojectedColor (:429-471), estimateLikelyDepth's repair (:743-798). GPU and oracle run 14 frames in
    lock-step, each on its own history; the oracle reports which branch every pixel took, and the test requires each branch to have
    been taken by many pixels — with a static camera `uv outside` cannot fire. Both forms of the frame (statement by statement /
    batched march over bricks; 96^3 = a grid that is not a power of two) must also agree bit for bit under motion.

    The bar is R-par's (this file's header): >= 99.9 % of pixels within 2e-3 linear RGB and 2e-3 depth (one binary16 ulp of the stored
    depth is 9.8e-4 above 1). Rounds 3-4 asked only for 99 % within 4e-3 and did not say what the rest was; counted here (and
    printed): on the three cases NO pixel of the 14 frames is outside the tolerance — the march's decisions (sample cells, repairs,
    blends) come out the same on both sides, the jitter hash included (both evaluate sin in double precision and round to f32:
    render_device.inc n1rand, render_oracle.c n1rand). Should a pixel ever differ, it must be one where the two sides made a different
    DECISION (its depth differs too), not one whose shading arithmetic drifted: asserted below."""
    cells = host.random_fill(host.words_per_buffer(G), seed=seed, and_rounds=4)
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(cells)
    eng.set_render_mode(True)
    try:
        frames = {}
        for bricks in (0, 1):
            eng.set_option("render_frame_bricks", bricks)
            eng.render(host.uniform_block(W, H, _moving_camera(0)), W, H, 1)
            eng.reset_render_history()
            seq, prev = [], None
            for f in range(14):
                vm = _moving_camera(f)
                u = host.uniform_block(W, H, vm, elapsed_time=0.1 + 0.137 * f, prev_view_mat=prev)
                seq.append(eng.render(u, W, H, 1))
                prev = vm
            frames[bricks] = seq
        for a, b in zip(frames[0], frames[1]):
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
        pl = pd = prev = None
        taken = dict(repair=0, outside=0, differs=0, blended=0, blended_lit=0)
        bad = decided = 0
        for f in range(14):
            vm = _moving_camera(f)
            u = host.uniform_block(W, H, vm, elapsed_time=0.1 + 0.137 * f, prev_view_mat=prev)
            prev = vm
            ol_light, ol_depth, _, br = ol.render_frame(cells, G, u, W, H, pl, pd, branches=True)
            pl, pd = ol_light.astype(np.float16).astype(np.float32), ol_depth.astype(np.float16).astype(np.float32)
            _, light, depth = frames[1][f]
            ok_rgb = np.abs(light.astype(np.float32)[..., :3] - pl[..., :3]).max(-1) <= 2e-3
            ok_d = np.abs(depth.astype(np.float32)[..., 0] - pd[..., 0]) <= 2e-3
            ok = ok_rgb & ok_d
            assert ok.mean() >= 0.999, (f, ok.mean())
            bad += int((~ok).sum())
            decided += int((~ok_d).sum())
            if f:
                taken["repair"] += int(((br & ol.BRANCH_DEPTH_REPAIR) != 0).sum())
                taken["outside"] += int(((br & ol.BRANCH_UV_OUTSIDE) != 0).sum())
                taken["differs"] += int(((br & ol.BRANCH_CELL_DIFFERS) != 0).sum())
                taken["blended"] += int(((br & ol.BRANCH_BLENDED) != 0).sum())
                taken["blended_lit"] += int(((br & (ol.BRANCH_BLENDED | ol.BRANCH_LIT)) == (ol.BRANCH_BLENDED | ol.BRANCH_LIT)).sum())
        px = W * H
        assert taken["outside"] > 0.1 * px and taken["differs"] > px and taken["blended_lit"] > 0.1 * px and taken["repair"] > 0.005 * px, taken
        print(f"moving camera {G}^3 {W}x{H}: branches {taken}; pixels outside the tolerance {bad} of {14 * px} "
              f"({bad / (14 * px):.5f}), of them with another depth (a different march decision) {decided}")
        assert decided >= 0.8 * bad or bad < 20, (bad, decided)
    finally:
        eng.set_option("render_frame_bricks", 1)
        eng.set_render_mode(False)


def test_literal_frame_batched_march_at_256(eng):
    """The literal frame's two forms — ca_render_frame_packed (fragment_main statement by statement) and the batched march over the
    bricked volume (render_frame.hip, the default) — produce the same frames bit for bit, each feeding on its own history, and at
    256^3 / 640 x 360 they track the oracle's lock-step frames like the small case above."""
    G, W, H = 256, 640, 360
    cells = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)
    vm = host.orbit_camera()
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(cells)
    eng.set_render_mode(True)
    try:
        frames = {}
        for bricks in (0, 1):
            eng.set_option("render_frame_bricks", bricks)
            eng.render(host.uniform_block(W, H, vm), W, H, 1)
            eng.reset_render_history()
            seq = []
            for f in range(5):
                u = host.uniform_block(W, H, vm, elapsed_time=0.2 + 0.093 * f, prev_view_mat=vm if f else None)
                out = eng.render(u, W, H, 1)
                st = eng.render_stats()
                seq.append((out, (st.shadow_rays, st.primary_cell_visits, st.shadow_cell_visits)))
            frames[bricks] = seq
        for (a, sa), (b, sb) in zip(frames[0], frames[1]):
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
            assert sa == sb
        pl = pd = None
        for f in range(5):
            u = host.uniform_block(W, H, vm, elapsed_time=0.2 + 0.093 * f, prev_view_mat=vm if f else None)
            ol_light, ol_depth, ol_pres = ol.render_frame(cells, G, u, W, H, pl, pd)
            pl, pd = ol_light.astype(np.float16).astype(np.float32), ol_depth.astype(np.float16).astype(np.float32)
            (pres, light, depth), _ = frames[1][f]
            ok = (np.abs(light.astype(np.float32)[..., :3] - pl[..., :3]).max(-1) <= 2e-3) & \
                 (np.abs(depth.astype(np.float32)[..., 0] - pd[..., 0]) <= 2e-3)
            assert ok.mean() >= 0.999, (f, ok.mean())
        assert pl[..., :3].max() > 0.05
    finally:
        eng.set_option("render_frame_bricks", 1)
        eng.set_render_mode(False)


@pytest.mark.parametrize("where", ["centre", "corner", "face", "two_clusters", "empty"])
def test_literal_frame_forms_agree_on_sparse_scenes(eng, where):
    """The literal frame's two forms (statement by statement / batched march over the bricked volume) on scenes that are mostly empty:
    small clusters in the middle of the volume, in a corner and on a face (sample cells wrap at the volume's faces), two clusters far
    apart, no live cell at all; the camera outside, close, and inside the volume — the same frames and the same sample counts. (Written
    for a march clipped to the box of the live bricks — round 4, measured: 0.135 -> 0.123 ms per interactive frame on the start-up
    scene, 0.127 -> 0.137 on a dense one, the box costs a reduce launch per frame — not kept; the cases stay.)"""
    G, W, H = 128, 320, 180
    rng = np.random.default_rng(17)
    dense = np.zeros((G, G, G), dtype=np.uint8)  # [z][y][x]
    def cluster(z, y, x, n=12):
        dense[z:z + n, y:y + n, x:x + n] = rng.integers(0, 2, size=(n, n, n), dtype=np.uint8)
    if where == "centre":
        cluster(58, 60, 56)
    elif where == "corner":
        cluster(0, 0, 0)
        cluster(G - 12, G - 12, G - 12)
    elif where == "face":
        cluster(50, 0, 60)
    elif where == "two_clusters":
        cluster(20, 24, 28)
        cluster(90, 96, 84)
    cells = np.packbits(dense.reshape(G, G, G // 32, 32), axis=-1, bitorder="little").view(np.uint32).ravel()
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(cells)
    eng.set_render_mode(True)
    try:
        for dist, angle in ((1.6, 0.6), (0.9, 2.1), (0.3, 1.0)):  # outside, close, inside the volume
            vm = host.orbit_camera(distance=dist, angle_rad=angle)
            got = {}
            for bricks in (0, 1):
                eng.set_option("render_frame_bricks", bricks)
                eng.reset_render_history()
                seq = []
                for f in range(3):
                    u = host.uniform_block(W, H, vm, elapsed_time=0.1 + 0.07 * f, prev_view_mat=vm if f else None)
                    out = eng.render(u, W, H, 1)
                    st = eng.render_stats()
                    seq.append((out, (st.shadow_rays, st.primary_cell_visits, st.shadow_cell_visits)))
                got[bricks] = seq
            for (a, sa), (b, sb) in zip(got[0], got[1]):
                for x, y in zip(a, b):
                    np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8), err_msg=f"{where} distance {dist}")
                assert sa == sb, (where, dist, sa, sb)
            if where != "empty" and dist > 1.0:
                assert got[1][0][1][0] > 0  # the cluster is in view: shadow rays were cast
    finally:
        eng.set_option("render_frame_bricks", 1)
        eng.set_render_mode(False)


def test_converged_frame_right_after_literal_frames_in_a_fresh_engine():
    """An engine whose FIRST frames are drawn in the literal mode (what a drop-in behind main_pathtraced.js does) and which is then
    asked for a converged frame of the same state: the occupancy bits the converged kernels consult have never been built — the
    literal mode does not use them — and must not be taken for current (round 4's first cache of the derived buffers did: the frame
    came out black). Also the other way round, and with the volume off screen in between (no bricks are built for such a frame)."""
    from cellularautomatons3d_amd import Engine

    G, W, H = 64, 160, 90
    cells = host.random_fill(host.words_per_buffer(G), seed=5, and_rounds=3)
    vm = host.orbit_camera(1.3, (1.0, 0.4, 0.0), 0.7)
    away = host.camera_matrix((0.0, 0.0, 3.0), (0.0, 1.0, 0.0), 3.14159265)  # behind the camera: the volume's rectangle is empty
    u = host.uniform_block(W, H, vm)
    with Engine(0) as ref:
        ref.configure(G)
        set_rules(ref, rules("default"))
        ref.upload_state(cells)
        want = ref.render(u, W, H, 4)
        ref.set_render_mode(True)
        want_lit = ref.render(u, W, H, 1)
    assert want[1].astype(np.float32)[..., :3].max() > 0.05
    with Engine(0) as e:
        e.configure(G)
        set_rules(e, rules("default"))
        e.upload_state(cells)
        e.set_render_mode(True)
        for f in range(3):
            e.render(host.uniform_block(W, H, vm, elapsed_time=0.1 * f), W, H, 1)
        e.set_render_mode(False)
        got = e.render(u, W, H, 4)
        for x, y in zip(got, want):
            np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
    with Engine(0) as e:
        e.configure(G)
        set_rules(e, rules("default"))
        e.upload_state(cells)
        e.render(host.uniform_block(W, H, away), W, H, 4)  # nothing of the volume on screen
        got = e.render(u, W, H, 4)
        for x, y in zip(got, want):
            np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
        e.set_render_mode(True)
        e.reset_render_history()
        lit = e.render(u, W, H, 1)
        for x, y in zip(lit, want_lit):
            np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))


def test_legacy_renderer_over_the_unpacked_volume(eng):
    """R-legacy: shaders/pathtraced_fragment.wgsl — one u32 per cell, reflect-based shading with 1/d^2 attenuation,
    OCCLUSION_FACTOR 0.095, gamma 2.2 — same converged-frame definition and tolerance."""
    from cellularautomatons3d_amd import LAYOUT_UNPACKED

    G, W, H = 64, 320, 180
    packed = host.random_fill(host.words_per_buffer(G), seed=3, and_rounds=4)
    cells = np.zeros(G ** 3, dtype=np.uint32)
    bits = np.unpackbits(packed.view(np.uint8), bitorder="little")
    cells[:] = bits  # word w bit b -> x = 32 * (w % C) + b: the unpacked index order is the same
    for pose in (host.camera_matrix(), host.orbit_camera()):
        u = host.uniform_block(W, H, pose, light=(0.721, 1.0, 1.0, 1.5))
        eng.configure(G, LAYOUT_UNPACKED)
        set_rules(eng, rules("default"))
        eng.upload_state(cells)
        pres, light, depth = eng.render(u, W, H, 1)
        olight, odepth, opres, _ = ol.render(cells, G, u, W, H, 1, legacy=True)
        ok = (np.abs(light.astype(np.float32)[..., :3] - olight[..., :3]).max(-1) <= 2e-3) & \
             (np.abs(depth.astype(np.float32)[..., 0] - odepth[..., 0].astype(np.float16).astype(np.float32)) <= 2e-3) & \
             (np.abs(pres.astype(np.float32) - np.rint(np.clip(opres, 0, 1) * 255.0)).max(-1) <= 1)
        assert ok.mean() >= 0.999, ok.mean()
        assert olight[..., :3].max() > 0.05
    # same geometry as the packed renderer sees
    eng.configure(G)
    set_rules(eng, rules("default"))
    eng.upload_state(packed)
    _, _, d2 = eng.render(u, W, H, 1)
    assert (np.abs(d2.astype(np.float32)[..., 0] - depth.astype(np.float32)[..., 0]) <= 2e-3).mean() >= 0.999


def _targets(eng):
    """The engine's three render targets as they stand on the device (after everything enqueued)."""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    out = []
    for which in (0, 1, 2):
        ptr, nbytes = eng.render_target(which)
        eng.synchronize()
        buf = np.empty(nbytes, dtype=np.uint8)
        assert hip.hipMemcpy(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) == 0
        out.append(buf)
    return out


def test_frames_in_flight(eng):
    """Converged frames that stay on the device alternate between up to four streams of the engine (option render_pipeline, default on; 2 - 4:
    that many), so that one frame's walks fill the idle tails of the others'. What must not change: the targets after a run of such frames are those of the LAST
    frame asked for — camera moving from frame to frame, the state stepped and uploaded between frames (a frame in flight reads the state it
    was asked for; the step behind it waits) — byte for byte the frame an engine without the pipeline leaves there."""
    G, W, H, spp = 128, 640, 360, 4
    eng.configure(G)
    set_rules(eng, rules("default"))
    cells = host.random_fill(host.words_per_buffer(G), seed=11, and_rounds=4)
    poses = [host.orbit_camera(1.4 + 0.02 * i, (1.0, 1.0, 0.0), 0.6 + 0.05 * i) for i in range(7)]

    def run(pipe):
        eng.set_option("render_pipeline", pipe)
        eng.upload_state(cells)
        got = []
        # a run of frames, the camera moving: the last one counts
        for vm in poses:
            eng.render(host.uniform_block(W, H, vm), W, H, spp, readback=False)
        got.append(_targets(eng))
        st = eng.render_stats()
        got.append([np.array([st.primary_rays, st.shadow_rays, st.primary_cell_visits, st.shadow_cell_visits], dtype=np.uint64).view(np.uint8)])
        # steps between frames: every frame shows the state of its moment, the last one the last state
        for i, vm in enumerate(poses):
            eng.render(host.uniform_block(W, H, vm), W, H, spp, readback=False)
            eng.step(1 + (i & 1))
        eng.render(host.uniform_block(W, H, poses[0]), W, H, spp, readback=False)
        got.append(_targets(eng))
        # a batch of steps right behind a frame in flight: the frame in the targets is the one of the state BEFORE them
        eng.render(host.uniform_block(W, H, poses[5]), W, H, spp, readback=False)
        eng.step(24)
        got.append(_targets(eng))
        # an upload right behind a frame in flight, then two frames
        eng.upload_state(cells[::-1].copy())
        eng.render(host.uniform_block(W, H, poses[1]), W, H, spp, readback=False)
        eng.render(host.uniform_block(W, H, poses[2]), W, H, spp, readback=False)
        got.append(_targets(eng))
        # a frame read back through host pointers behind frames in flight (it joins them), and a literal frame behind those
        eng.render(host.uniform_block(W, H, poses[3]), W, H, spp, readback=False)
        pres, light, depth = eng.render(host.uniform_block(W, H, poses[4]), W, H, spp)
        got.append([pres.view(np.uint8).ravel(), light.view(np.uint8).ravel(), depth.view(np.uint8).ravel()])
        return got

    try:
        a = run(0)
        assert eng.render_pipeline() == 0
        deep = {}
        for pipe in (1, 2, 3):  # the default depth (four frames in flight at this size), two, three
            deep[pipe] = run(pipe)
            # (the depth in use is what the runtime's hardware queues allow: streams that do not run side by side are not used as lanes)
            want, got = (4 if pipe == 1 else pipe), eng.render_pipeline()  # (default depth: four frames up to 24 M samples a frame)
            assert got == want or (got < want and got != 1), (pipe, got)
    finally:
        eng.set_option("render_pipeline", 1)
    for pipe, b in deep.items():
        for fa, fb in zip(a, b):
            for x, y in zip(fa, fb):
                np.testing.assert_array_equal(x, y, err_msg=f"render_pipeline {pipe}")
    assert a[0][1].any()  # the scene is not empty

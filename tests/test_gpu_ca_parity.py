"""GPU parity: the HIP CA step, called through the C ABI, against the CPU oracle — bit-exact."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from cellularautomatons3d_amd import Ca3dError, host
from gpu_common import RULESETS, rules, set_rules

pytestmark = pytest.mark.gpu


@pytest.fixture(scope=lambda fixture_name, config: os.environ.get("CA3D_TEST_ENGINE_SCOPE", "module"))
def eng():
    # one engine per file by default (faster, and state carried from test to test is itself a test); CA3D_TEST_ENGINE_SCOPE=function gives
    # every test a fresh one: a test that only passes behind another shows up (round 4 found a renderer cache bug that way)
    from cellularautomatons3d_amd import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("G", [32, 64, 96, 128, 256])
@pytest.mark.parametrize("name", list(RULESETS))
def test_one_and_many_steps_random_fill(eng, G, name):
    r = rules(name)
    eng.configure(G)
    for jit in (1, 0):  # run-time compiled rule specialisation (where one applies), then the pre-built kernels
        eng.set_option("jit", jit)
        set_rules(eng, r)
        # every grid has a kernel compiled for the rule at run time (power-of-two grids from 128 up: the uint4 kernels; the others:
        # the rows kernel); the start-up rule's is pre-built on power-of-two grids from 256 up
        # (info() names the kernel LONG batches get: at 64^3 a von Neumann table rule has the one-workgroup resident kernel — pre-built for
        # the start-up rule; the one- and four-step batches below run the per-step kernels, as everywhere)
        res64 = G == 64 and name in ("default", "vn_b24_s135")
        if res64:
            want = b"ca_resident_vn" if name == "default" else (b"ca_resident_vn(jit)" if jit else None)  # (another table pair without the run-time compiler: none)
            assert want is None or eng.info().kernel_name == want, eng.info().kernel_name
            eng.set_option("resident", 0)
        assert (b"(jit)" in eng.info().kernel_name) == (bool(jit) and not (name == "default" and G >= 256))
        assert (b"ca_packed_rows" in eng.info().kernel_name) == (bool(jit) and G < 128)
        if res64:
            eng.set_option("resident", 1)
        for rounds in (0, 3):
            st = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001 + G, and_rounds=rounds)
            eng.upload_state(st)
            eng.step(1)
            want1 = ol.packed_step(G, st, r)
            np.testing.assert_array_equal(eng.read_state(), want1)
            eng.step(4)
            np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, want1, r, 4))
            assert eng.info().step == 5
    eng.set_option("jit", 1)


@pytest.mark.parametrize("G", [256, 512])
def test_vn_truth_table_kernel_random_tables(eng, G):
    """ca_packed_vn evaluates the rule as two 8-entry truth tables: sweep random tables (the run-time dispatch) and
    the pre-built default-rule specialisation, one and several steps, against the oracle."""
    rng = np.random.default_rng(1234 + G)
    eng.configure(G)
    st = host.random_fill(host.words_per_buffer(G), seed=77 + G)
    tables = [(0x7F, 0x0A), (0x00, 0x7F), (0x7F, 0x00), (0x55, 0x2A)] + [tuple(int(x) for x in rng.integers(0, 128, 2)) for _ in range(4)]
    for lut_s, lut_b in tables:
        survive = ",".join(str(k) for k in range(7) if lut_s >> k & 1)
        born = ",".join(str(k) for k in range(7) if lut_b >> k & 1)
        r = ol.Rules.from_strings(neighbourhood="von neumann", born=born, survive=survive)
        want = ol.packed_run(G, st, r, 3)
        for jit in (1, 0):  # the run-time compiled specialisation, then the pre-built table dispatch
            eng.set_option("jit", jit)
            set_rules(eng, r)
            eng.upload_state(st)
            prebuilt = (lut_s & 0x7F, lut_b & 0x7F) == (0x7F, 0x0A)
            resident = jit or prebuilt  # (256^3 and 512^3) batches of >= 8 steps would take the resident kernel; 3 steps do not
            stem = b"ca_resident_vn" if resident else b"ca_packed_vn"
            assert eng.info().kernel_name == stem + (b"(jit)" if jit and not prebuilt else b"")
            eng.step(3)
            np.testing.assert_array_equal(eng.read_state(), want, err_msg=f"S={lut_s:#x} B={lut_b:#x} jit={jit}")
        eng.set_option("jit", 1)


def test_random_rules_through_the_rule_compilers(eng):
    """Random rule strings over all six main neighbourhoods with random edges / corners rule-sets (including counts no
    neighbourhood can reach, the "27" convention and empty lists): the run-time compiled truth-table kernels, the
    pre-built cube-program kernels and the generic kernel all against the oracle."""
    rng = np.random.default_rng(20260614)
    G = 128
    st = host.random_fill(host.words_per_buffer(G), seed=404)
    eng.configure(G)
    hoods = ["von neumann", "von neumann 2D", "moore", "moore 2D", "edges", "corners"]

    def rule_string(max_count, p_empty=0.15):
        if rng.random() < p_empty:
            return "27" if rng.random() < 0.5 else ""
        ks = sorted(set(int(k) for k in rng.integers(0, max_count + 3, size=int(rng.integers(1, 6)))))
        return ",".join(str(k) for k in ks)

    for case in range(14):
        kw = dict(neighbourhood=hoods[case % 6], born=rule_string(26, 0.0), survive=rule_string(26, 0.05),
                  born_edges=rule_string(12, 0.4), survive_edges=rule_string(12, 0.4),
                  born_corners=rule_string(8, 0.4), survive_corners=rule_string(8, 0.4))
        r = ol.Rules.from_strings(**kw)
        want = ol.packed_run(G, st, r, 2)
        for jit, variant in ((1, 0), (0, 0), (0, 1)):
            eng.set_option("jit", jit)
            eng.set_option("variant", variant)
            set_rules(eng, r)
            eng.upload_state(st)
            eng.step(2)
            np.testing.assert_array_equal(eng.read_state(), want, err_msg=f"{kw} jit={jit} variant={variant} {eng.info().kernel_name}")
    eng.set_option("jit", 1)
    eng.set_option("variant", 0)


VN_TABLE_PAIRS = [  # (born, survive) over the von Neumann count 0 .. 6, chosen by what vn_next (ca_bitops.inc) does with them
    ("1,3", "0-6"), ("1,5", "1,3"), ("3", "1,3,5"), ("1,3,5", ""),            # one answer for every EVEN count: the carry out of plane 0 is dropped
    ("2,4", "1,3,5"), ("0,2", "0,1,3,5,6"), ("4,6", "2"), ("0,2,4,6", "0,6"),  # one answer for every ODD count: the carry is the first adder's sum
    ("1,2", "2,3"), ("2", "1-3"), ("3,4,5", "0,1,6"), ("6", "5"),             # neither: the full adder tree
    ("1,3,5", "1,3,5"), ("2,3,6", "2,3,6"), ("4-6", "4-6"), ("0-3", "4-6"),  # tables that need one or two count planes only
]


@pytest.mark.parametrize("born,survive", VN_TABLE_PAIRS)
def test_von_neumann_tables_through_every_branch_of_vn_next(eng, born, survive):
    """The run-time compiled von Neumann kernels drop what a rule's tables make unnecessary (count planes nobody reads, the carry out of
    plane 0 when one parity of counts is answered alike — vn_same_for_parity): rules from each of those classes, two steps of the
    per-step kernel at 128^3 and a 24-step resident batch at 256^3, dense and sparse, against the oracle."""
    r = ol.Rules.from_strings("von neumann", born, survive)
    for G, steps in ((128, 2), (256, 24)):
        eng.configure(G)
        set_rules(eng, r)
        for rounds in (0, 3):
            st = host.random_fill(host.words_per_buffer(G), seed=7 * G + rounds, and_rounds=rounds)
            eng.upload_state(st)
            eng.step(steps)
            np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st, r, steps), err_msg=f"B{born}/S{survive} at {G} ({eng.info().kernel_name})")


# every rule-set at 96 / 160 / 384; at 768 (3.5 s per case: the oracle) one rule per kernel form — von Neumann, clustered, a 2D rule, a wide
# Moore table, an edges-only rule-set: the forms 768 selects are the ones 384 selects for the other rules (both are multiples of 128)
@pytest.mark.parametrize("G,name", [(G, n) for G in (96, 160, 384) for n in RULESETS] + [(768, n) for n in ("default", "clustered", "life2d", "moore_wide", "vn_edges_only")])
def test_rows_kernel_on_grids_that_are_not_powers_of_two(eng, G, name):
    """The grids the reference UI offers are the multiples of 32 up to 1024 (main_pathtraced.js:268-279, 675-693). Those that are not
    powers of two run the rows kernel (ca_packed_rows_kernel.inc, compiled for grid and rule at run time): against the oracle, and
    against the kernels that served them before (option rows 0: the class kernel's pre-built form on multiples of 128, else the
    generic kernel)."""
    r = rules(name)
    eng.configure(G)
    set_rules(eng, r)
    # rows of whole uint4 that are not a power of two of them (384, 768): every 3D rule takes the rolling-window kernel's
    # whole-rows-per-wave form (roll_step_np2); the 2D rules keep the class kernel's run-time compiled np2 form on the larger grids
    # (measured faster there: ca_packed.hip, rows_kernel_applies) and the rows kernel on the smaller; every other grid: rows
    flat = name in ("vn2d", "life2d")  # counts from the cell's own plane only: nothing to gain from a window along z
    if G % 128 == 0 and not flat:
        want = b"ca_packed_roll_np2(jit)"
    elif G % 128 != 0 or G < 768:
        want = b"ca_packed_rows(jit)"
    else:
        want = None  # the class kernel, run-time compiled
    name_now = eng.info().kernel_name
    assert (name_now == want if want else (b"class" in name_now and b"(jit)" in name_now)), name_now
    for rounds in (0, 4):
        st = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001 + G, and_rounds=rounds)
        eng.upload_state(st)
        eng.step(1)
        want1 = ol.packed_step(G, st, r)
        np.testing.assert_array_equal(eng.read_state(), want1)
        eng.step(5)  # batches of more than one step: z-runs of several planes, graph replay
        want6 = ol.packed_run(G, want1, r, 5)
        np.testing.assert_array_equal(eng.read_state(), want6)
    # ... and the kernels that served these grids before: without the rolling form (rows / class np2), then without the rows kernel too
    try:
        for off in ("roll", "rows"):
            eng.set_option(off, 0)
            assert b"roll_np2" not in eng.info().kernel_name and (off == "roll" or b"rows" not in eng.info().kernel_name)
            eng.upload_state(st)
            eng.step(6)
            np.testing.assert_array_equal(eng.read_state(), want6, err_msg=eng.info().kernel_name.decode())
    finally:
        eng.set_option("roll", 1)
        eng.set_option("rows", 1)


@pytest.mark.parametrize("G,name,z", [(640, "clustered", 8), (640, "moore_wide", 4), (896, "clustered", 2), (384, "edges_main", 8), (384, "clustered", 4),
                                      (896, "vn_corners_only", 0)])
def test_roll_np2_kernel_plane_depths(eng, G, name, z):
    """roll_step_np2 at every depth it is compiled for (option roll_z forces 2 / 4 / 8 planes per thread; 0: the launcher's choice), rows of
    3, 5 and 7 uint4 (21, 12 and 9 whole rows per wave), three steps from a sparse state against the oracle."""
    r = rules(name)
    eng.configure(G)
    set_rules(eng, r)
    assert eng.info().kernel_name == b"ca_packed_roll_np2(jit)"
    st = host.random_fill(host.words_per_buffer(G), seed=G + z, and_rounds=1)
    eng.set_option("roll_z", z)
    try:
        eng.upload_state(st)
        eng.step(3)
        np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st, r, 3, 16))
    finally:
        eng.set_option("roll_z", 0)


@pytest.mark.parametrize("name", ["default", "clustered"])
def test_rows_kernel_at_992(eng, name):
    """"1000" in the reference UI becomes 992 (_gridSizeUIFormatter): rows of 31 words, two rows per wave."""
    G = 992
    r = rules(name)
    eng.configure(G)
    set_rules(eng, r)
    assert eng.info().kernel_name == b"ca_packed_rows(jit)"
    st = host.random_fill(host.words_per_buffer(G), seed=992, and_rounds=1)
    eng.upload_state(st)
    eng.step(3)
    np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st, r, 3, 16))


@pytest.mark.parametrize("G", [128, 256])
@pytest.mark.parametrize("name", list(RULESETS))
def test_generic_kernel_equals_class_kernel(eng, G, name):
    r = rules(name)
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=11)
    eng.upload_state(st)
    assert b"class" in eng.info().kernel_name or eng.info().kernel_name.startswith((b"ca_packed_vn", b"ca_resident_vn"))
    eng.step(3)
    a = eng.read_state()
    eng.set_option("variant", 1)
    try:
        eng.upload_state(st)
        assert eng.info().kernel_name == b"ca_packed_generic"
        eng.step(3)
        b = eng.read_state()
    finally:
        eng.set_option("variant", 0)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, ol.packed_run(G, st, r, 3))


def test_arbitrary_offset_lists(eng):
    rng = np.random.default_rng(3)
    for G in (32, 128):
        eng.configure(G)
        for _ in range(5):
            lists = [rng.integers(-1, 2, size=3 * int(rng.integers(0, 12))).astype(np.int32) for _ in range(3)]
            r = ol.Rules(lists[0], lists[1], lists[2], (rng.random(81) < 0.3).astype(np.uint32), (rng.random(81) < 0.3).astype(np.uint32))
            set_rules(eng, r)
            st = host.random_fill(host.words_per_buffer(G), seed=int(rng.integers(1 << 30)))
            eng.upload_state(st)
            eng.step(2)
            np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st, r, 2))


def test_lut_entry_must_equal_one(eng):
    G = 128
    r = rules("vn2d")
    r.born = r.born.copy()
    r.born[1] = 2  # not == 1: never born (compute_clustered.wgsl:232)
    eng.configure(G)
    set_rules(eng, r)
    eng.upload_state(host.cells_to_words(G, [(5, 5, 5)]))
    eng.step(1)
    assert ol.popcount(eng.read_state()) == 0


def test_default_seed_population_and_hashes(eng):
    want = {32: "8ebd1f9c c3f1930e 6b170f5a 7815260a dec86ca6 afa3b49c b79051df 23672217".split(),
            64: "eaf84574 2a89e9f6 08947442 7e303b52 44d3c10e 3ca0f014 548cafef ae46dc87".split()}
    for G in (32, 64, 128):
        eng.restart_sim(G, "von neumann", "1,3", "0-6")
        pops, hashes = [], []
        for _ in range(8):
            eng.compute_pass(1)
            st = eng.read_state()
            pops.append(ol.popcount(st))
            hashes.append("%08x" % ol.fnv1a32(st))
        assert pops == [7, 13, 43, 49, 79, 133, 259, 313]
        if G in want:
            assert hashes == want[G]


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_boundary_asymmetry(eng, axis):
    for G in (32, 128):
        eng.configure(G)
        set_rules(eng, ol.Rules.from_strings("von neumann", "1", ""))

        def cell(v):
            c = [5, 5, 5]
            c[axis] = v
            return tuple(c)

        eng.upload_state(host.cells_to_words(G, [cell(0)]))
        eng.step(1)
        st = eng.read_state()
        assert ol.popcount(st) == 6 and host.get_cell(G, st, *cell(G - 1)) == 1
        eng.upload_state(host.cells_to_words(G, [cell(G - 1)]))
        eng.step(1)
        st = eng.read_state()
        assert ol.popcount(st) == 5 and host.get_cell(G, st, *cell(0)) == 0


def test_ping_pong_buffers(eng):
    # After n steps buffer n % 2 is current and the other one holds state n-1 (main_pathtraced.js:1580-1609).
    import ctypes as C

    G = 128
    r = rules("clustered")
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=21)
    eng.upload_state(st)
    eng.step(3)
    assert eng.info().current_buffer == 1
    s2, s3 = ol.packed_run(G, st, r, 2), ol.packed_run(G, st, r, 3)
    np.testing.assert_array_equal(eng.read_state(), s3)
    import torch

    p, n = eng.device_buffer(0)
    eng.synchronize()
    other = torch.empty(n // 4, dtype=torch.int32, device="cuda:0")
    rt = C.CDLL("libamdhip64.so")
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert rt.hipMemcpy(other.data_ptr(), p, n, 3) == 0
    np.testing.assert_array_equal(other.cpu().numpy().view(np.uint32), s2)


def test_config2_256_cubed_1000_steps(eng):
    """BASELINE config 2: 256^3, default rule, 1000 steps, compared at steps 1, 2, 10, 100, 1000."""
    G = 256
    r = rules("default")
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=4)
    eng.upload_state(st)
    cur, done = st, 0
    for target in (1, 2, 10, 100, 1000):
        eng.step(target - done)
        cur = ol.packed_run(G, cur, r, target - done)
        done = target
        np.testing.assert_array_equal(eng.read_state(), cur)
    assert eng.info().step == 1000 and eng.info().current_buffer == 0


def test_config2_single_seed_200_steps(eng):
    G = 256
    eng.restart_sim(G, "von neumann", "1,3", "0-6")
    eng.step(200)
    want = ol.packed_run(G, host.initial_state(G), rules("default"), 200)
    np.testing.assert_array_equal(eng.read_state(), want)


@pytest.mark.parametrize("name", ["default", "clustered"])
def test_512_cubed_against_oracle(eng, name):
    G = 512
    r = rules(name)
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=5)
    eng.upload_state(st)
    eng.step(2)
    np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st, r, 2))


def test_512_translation_equivariance(eng):
    # Size-independent property: away from the faces the rule commutes with translation by whole words / rows /
    # planes, so a shifted seed pattern evolves into the shifted result.
    G = 512
    r = rules("clustered")
    eng.configure(G)
    set_rules(eng, r)
    rng = np.random.default_rng(1)
    cells = [(int(200 + rng.integers(0, 24)), int(200 + rng.integers(0, 24)), int(200 + rng.integers(0, 24))) for _ in range(600)]
    dx, dy, dz = 64, 37, 101

    def run(cs):
        eng.upload_state(host.cells_to_words(G, cs))
        eng.step(6)
        return eng.read_state().reshape(G, G, G // 32)

    a = run(cells)
    b = run([(x + dx, y + dy, z + dz) for (x, y, z) in cells])
    assert a.any()
    np.testing.assert_array_equal(np.roll(a, (dz, dy, dx // 32), axis=(0, 1, 2)), b)


def test_1024_cubed_one_step(eng):
    G = 1024
    r = rules("default")
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=9)
    eng.upload_state(st)
    eng.step(1)
    np.testing.assert_array_equal(eng.read_state(), ol.packed_step(G, st, r))


@pytest.mark.parametrize("name", ["default", "clustered"])
def test_2048_cubed_two_steps(eng, name):
    """BASELINE config 5's grid (1 GiB per buffer, offsets past 2^32 bits) under the default rule and under config 5's
    own clustered rule-set (all three rule-sets of compute_clustered.wgsl:192-247 live): two steps, whole state
    compared with the oracle."""
    G = 2048
    r = rules(name)
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=7)
    eng.upload_state(st)
    eng.step(2)  # (von Neumann tables: 4 planes per thread on grids past the Infinity Cache — pre-built for the start-up rule, run-time compiled otherwise)
    assert "(jit)" in eng.info().kernel_name.decode() or name == "default"
    np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st, r, 2))
    eng.configure(32)  # release the 2 GiB before the next test


@pytest.mark.parametrize("name", ["clustered", "moore_wide", "moore_b4s4", "edges_main", "corners_main", "vn_edges_only", "vn_corners_only"])
def test_rolling_window_kernel_every_depth(eng, name):
    """The rolling-window form of the class kernels (ca_packed_roll_kernel.inc: per-plane partial sums A / B / c in a
    three-plane register window) with 2, 4 and 8 planes per thread, on a full 256^3 grid (runs start on global plane 0,
    the last plane's z+1 wraps) and — launcher's own choice — at 512^3; against the oracle and against the plain class
    kernel."""
    r = rules(name)
    G = 256
    eng.configure(G)
    eng.set_option("resident", 0)  # the per-step kernels are what is tested here (long batches would take the resident class kernel)
    set_rules(eng, r)
    assert b"roll" in eng.info().kernel_name, eng.info().kernel_name
    st = host.random_fill(host.words_per_buffer(G), seed=61)
    want = ol.packed_run(G, st, r, 3)
    # both forms: every thread shifting its three rows itself (roll_tile 0) and the tile form, where a workgroup shares the
    # x-shifted rows through LDS (16 planes per thread exist in the tile form only)
    # (roll_tile 2: wave tiles — the tile form with one wave per workgroup, its rows exchanged through LDS without a barrier)
    # (roll_tile 3: two words per thread instead of four — a window of 42 registers, five or six waves per SIMD)
    for tile in (1, 0, 2, 3):
        eng.set_option("roll_tile", tile)
        for z in {1: (2, 4, 8, 16, 15, 30), 0: (2, 4, 8, 15, 30), 2: (8, 16), 3: (8, 16)}[tile]:  # 15 / 30: the looped forms (plane loop in groups of three)
            eng.set_option("roll_z", z)
            eng.upload_state(st)
            eng.step(3)
            np.testing.assert_array_equal(eng.read_state(), want, err_msg=f"roll_tile {tile} roll_z {z}")
    eng.set_option("roll_tile", 1)
    eng.set_option("roll_z", 0)
    eng.set_option("roll", 0)
    assert b"roll" not in eng.info().kernel_name
    eng.upload_state(st)
    eng.step(3)
    np.testing.assert_array_equal(eng.read_state(), want)
    eng.set_option("roll", 1)
    eng.set_option("resident", 1)


@pytest.mark.parametrize("name,G", [("clustered", 1024), ("edges_main", 512)])
def test_rolling_window_kernel_launcher_choice(eng, name, G):
    """The launcher's own choice of kernel and depth: at 1024^3 (many resident generations) the rolling-window form, at 512^3
    (one generation of 2048 waves) the plain class kernel for short batches — the resident class kernel takes the long ones."""
    r = rules(name)
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=62)
    eng.upload_state(st)
    eng.step(2)
    assert (b"roll" in eng.info().kernel_name) == (G == 1024) and (eng.info().kernel_name == b"ca_resident_class(jit)") == (G == 512)
    assert np.array_equal(eng.read_state(), ol.packed_run(G, st, r, 2))


_ORACLE_MEMO = {}


def _oracle_run(G, key, state, r, steps):
    """ol.packed_run, remembered per (grid, name of the start state, rule tables) and continued from the furthest remembered state
    at or before `steps`: the parametrised forms of one kernel family are compared with the same oracle states — a 512^3 x 1000-step
    oracle run is ~22 s of the suite each time it is recomputed."""
    memo = _ORACLE_MEMO.setdefault((G, key, tuple(r.survive), tuple(r.born)), {0: state})
    if steps not in memo:
        base = max(k for k in memo if k <= steps)
        memo[steps] = ol.packed_run(G, memo[base], r, steps - base)
    return memo[steps]


@pytest.mark.parametrize("tables,rows,zsplit", [("default", "pair", 1), ("vn_b24_s135", "pair", 1), ("default", 32, 1), ("default", 32, 2), ("vn_b24_s135", 32, 2),
                                                ("default", 16, 2), ("vn_b24_s135", 16, 1)])
def test_resident_multi_step_kernel(eng, tables, rows, zsplit):
    """The resident kernel (ca_resident_kernel.inc): K steps of a 512^3 von Neumann rule in one launch, the state in
    registers, tile faces handed over through tagged granules. Batches of several lengths back to back (the state tags
    keep counting across launches and across uploads), odd and even lengths (the result lands in either ping-pong
    buffer, the other one holds the state one step earlier), against the oracle and against the per-step kernels."""
    G = 512
    r = rules(tables)
    eng.configure(G)
    pair = rows == "pair"  # the row-pair form (the default): 32-row tiles, a thread owns two adjacent rows x 16 planes (resident_pair_run)
    eng.set_option("resident_pair", int(pair))
    rows = 32 if pair else rows
    eng.set_option("resident_rows", rows)  # tiles of 32 rows (one workgroup per CU) or 16 rows (two per CU)
    eng.set_option("resident_zsplit", zsplit)  # 2: two threads per (row, word) column, half the planes each — four waves per SIMD
    set_rules(eng, r)
    assert eng.info().kernel_name.startswith(b"ca_resident_vn")
    st = host.random_fill(host.words_per_buffer(G), seed=88)
    eng.upload_state(st)
    total = 0
    for n in (8, 9, 20):
        eng.step(n)
        prev = _oracle_run(G, "fill88", st, r, total + n - 1)
        want = _oracle_run(G, "fill88", st, r, total + n)
        total += n
        np.testing.assert_array_equal(eng.read_state(), want, err_msg=f"after a batch of {n}")
        assert eng.info().current_buffer == total % 2 and eng.info().step == total
        import torch
        from cellularautomatons3d_amd import slab
        other = slab.device_tensor(*eng.device_buffer(1 - total % 2), 0).cpu().numpy().view(np.uint32)
        np.testing.assert_array_equal(other, prev, err_msg="the other buffer holds the state one step earlier")
    # a fresh upload, a sparse state (most tiles empty: faces of zeros must still carry their tags), many steps
    st2 = host.initial_state(G)
    eng.upload_state(st2)
    eng.step(60)
    got = eng.read_state()
    eng.set_option("resident", 0)
    assert eng.info().kernel_name.startswith(b"ca_packed_vn")
    eng.upload_state(st2)
    eng.step(60)
    np.testing.assert_array_equal(got, eng.read_state())
    np.testing.assert_array_equal(got, _oracle_run(G, "seed", st2, r, 60))
    # many steps on a dense state in one launch against as many launches of the per-step kernel — 1000 for the default form (the
    # row-pair kernel) and, on the default rule, against the oracle's 1000 steps too; the forms that are off by default run 64
    long = 1000 if pair else 64
    eng.upload_state(st)
    eng.step(long)
    per_step = eng.read_state()
    eng.set_option("resident", 1)
    eng.upload_state(st)
    eng.step(long)
    assert eng.info().kernel_name.startswith(b"ca_resident_vn")
    got = eng.read_state()
    np.testing.assert_array_equal(got, per_step)
    if pair and tables == "default":
        np.testing.assert_array_equal(got, _oracle_run(G, "fill88", st, r, 1000), err_msg="1000 resident steps vs the oracle")
    assert eng.recovered_launches() == 0
    eng.set_option("resident_pair", 1)  # the defaults again
    eng.set_option("resident_rows", 32)
    eng.set_option("resident_zsplit", 1)


@pytest.mark.parametrize("name,G,n", [("default", 512, 40), ("default", 256, 33), ("clustered", 512, 16), ("clustered", 256, 20)])
def test_resident_launch_that_gives_up_is_recovered(name, G, n):
    """A resident launch only completes when all its workgroups are on the chip. Simulate one that is not (option
    "resident_fault_tile": that tile leaves at once, exactly what a workgroup stuck in the dispatcher's queue looks like to
    its neighbours) with a short timeout: the neighbours' waits expire, the launch and the resident launches queued behind
    it write nothing, and the engine re-runs their steps from the intact input through the per-step kernels — the call
    sequence ends bit-exact with the oracle, the resident path is off afterwards and can be turned on again."""
    from cellularautomatons3d_amd import Engine

    r = rules(name)
    with Engine(0) as e:
        e.configure(G)
        set_rules(e, r)
        assert e.info().kernel_name.startswith(b"ca_resident")
        st = host.random_fill(host.words_per_buffer(G), seed=404, and_rounds=1)
        e.upload_state(st)
        e.set_option("resident_timeout_us", 3000)
        e.step(9)                                 # a good launch first: the failing one starts from a rotated buffer set
        e.set_option("resident_fault_tile", 38)   # tile 37 of the next launch never shows up
        e.step(n)
        e.step(n + 1)                             # queued behind the failing one: must not touch anything
        e.step(3)                                 # a short batch (per-step kernels) behind both
        got = e.read_state()
        assert e.recovered_launches() == 1
        assert not e.info().kernel_name.startswith(b"ca_resident"), e.info().kernel_name
        total = 9 + n + n + 1 + 3
        assert e.info().step == total and e.info().current_buffer == total % 2
        want_prev = ol.packed_run(G, st, r, total - 1)
        want = ol.packed_step(G, want_prev, r)
        np.testing.assert_array_equal(got, want)
        from cellularautomatons3d_amd import slab
        other = slab.device_tensor(*e.device_buffer(1 - total % 2), 0).cpu().numpy().view(np.uint32)
        np.testing.assert_array_equal(other, want_prev, err_msg="the other buffer holds the state one step earlier")
        # the per-step kernels carry on; the path can be switched on again and works
        e.step(5)
        want = ol.packed_run(G, want, r, 5)
        np.testing.assert_array_equal(e.read_state(), want)
        e.set_option("resident", 1)
        assert e.info().kernel_name.startswith(b"ca_resident")
        e.step(12)
        np.testing.assert_array_equal(e.read_state(), ol.packed_run(G, want, r, 12))
        assert e.recovered_launches() == 1


@pytest.mark.parametrize("name,G", [(n, 512) for n in ("clustered", "moore_b4s4", "edges_main", "corners_main", "vn_edges_only", "moore_wide")]
                         + [(n, 256) for n in ("clustered", "edges_main", "corners_main", "moore_wide")])
def test_resident_class_kernel(eng, name, G):
    """The resident kernel for rules with diagonal neighbour classes (ca_resident_class_kernel.inc): 512^3 (tiles of 16 words
    x 32 rows x 32 planes) and 256^3 (8 x 32 x 8), K steps per launch, halo planes and corner rows from all eight neighbour
    tiles; dense and sparse states (faces of zeros still carry their tags), odd and even batch lengths, against the oracle
    and against the per-step kernels."""
    r = rules(name)
    eng.configure(G)
    set_rules(eng, r)
    assert eng.info().kernel_name == b"ca_resident_class(jit)", eng.info().kernel_name
    st = host.random_fill(host.words_per_buffer(G), seed=99, and_rounds=1)
    eng.upload_state(st)
    want, total = st, 0
    for n in (9, 8):
        eng.step(n)
        want = ol.packed_run(G, want, r, n)
        total += n
        assert np.array_equal(eng.read_state(), want), f"after a batch of {n}"
        assert eng.info().current_buffer == total % 2
    # boundary faces: a slab of live cells hugging the -x/-y/-z corner and the +faces (dead below, wrap above)
    edge = np.zeros(host.words_per_buffer(G), dtype=np.uint32)
    edge.reshape(G, G, G // 32)[:3, :3, :] = host.random_fill(9 * (G // 32), seed=7).reshape(3, 3, G // 32)
    edge.reshape(G, G, G // 32)[-3:, -3:, :] = host.random_fill(9 * (G // 32), seed=8).reshape(3, 3, G // 32)
    eng.upload_state(edge)
    eng.step(12)
    got = eng.read_state()
    eng.set_option("resident", 0)
    eng.upload_state(edge)
    eng.step(12)
    assert np.array_equal(got, eng.read_state())
    assert np.array_equal(got, ol.packed_run(G, edge, r, 12))
    eng.set_option("resident", 1)
    if name == "clustered":
        # a long batch in one launch against the oracle (256 steps of the live kernel's full semantics)
        eng.upload_state(st)
        eng.step(256)
        assert eng.info().kernel_name == b"ca_resident_class(jit)"
        assert np.array_equal(eng.read_state(), ol.packed_run(G, st, r, 256)), "256 resident class steps vs the oracle"


def test_batches_of_any_length_replay_as_graphs(eng):
    """A batch of n < 1024 steps is one captured graph of exactly n steps, cached per (n, start buffer): odd lengths
    alternate between the two buffers; short batches (< graph_min) are launched kernel by kernel."""
    G = 128
    r = rules("vn_b24_s135")
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=22)
    eng.upload_state(st)
    eng.set_option("graph_min", 8)  # default 128: shorter batches go kernel by kernel
    eng.set_option("graph_prepare", 21)
    total = 0
    for n in (21, 21, 3, 20, 21, 1, 37):
        eng.step(n)
        total += n
        assert eng.info().current_buffer == total % 2
    np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st, r, total))
    eng.set_option("graph_min", 2)
    eng.upload_state(st)
    for n in (2, 3, 5):
        eng.step(n)
    np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st, r, 10))
    eng.set_option("graph_min", 128)


def test_jit_failure_is_reported_and_falls_back(monkeypatch):
    """A failed run-time specialisation must not fail ca3d_set_rules, and must not be silent: the pre-built kernel is
    selected (no "(jit)" in the kernel name), ca3d_last_error / ca3d_get_jit_log carry the compiler's message, and the
    step is still bit-exact."""
    from cellularautomatons3d_amd import Engine, _capi

    G = 128
    st = host.random_fill(host.words_per_buffer(G), seed=23)
    monkeypatch.setenv("CA3D_JIT_FLAGS", "-DCA3D_JIT_MAIN=this_is_not_a_main_kind -DCA3D_TEST_BREAK=1")
    with Engine(0) as e:
        e.configure(G)
        r2 = ol.Rules.from_strings("moore", "4,9", "4,11")  # tables unique to this test: not in the module cache
        e.set_rules(r2.main, r2.edges, r2.corners, r2.survive, r2.born)  # returns normally
        assert "(jit)" not in e.info().kernel_name.decode()
        msg = _capi.load().ca3d_last_error().decode()
        assert "specialisation failed" in msg and "hiprtc" in msg
        log = e.jit_log()
        assert "error" in log.lower() and len(log) > 20
        e.upload_state(st)
        e.step(2)
        np.testing.assert_array_equal(e.read_state(), ol.packed_run(G, st, r2, 2))
        monkeypatch.delenv("CA3D_JIT_FLAGS")
        e.set_rules(r2.main, r2.edges, r2.corners, r2.survive, r2.born)
        assert "(jit)" in e.info().kernel_name.decode() and e.jit_log() == ""
        e.upload_state(st)
        e.step(2)
        np.testing.assert_array_equal(e.read_state(), ol.packed_run(G, st, r2, 2))


def test_large_graph_replay_1024_steps(eng):
    """ca3d_step replays captured graphs of 1024 and of 64 steps, then single launches: one call that uses all three
    (and one from buffer 1, which first re-aligns with a single step) against the oracle."""
    G = 128
    r = rules("default")
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=21)
    eng.upload_state(st)
    eng.set_option("graph_prepare", 2000)
    eng.step(1024 + 64 + 7)
    want = ol.packed_run(G, st, r, 1024 + 64 + 7)
    np.testing.assert_array_equal(eng.read_state(), want)
    assert eng.info().current_buffer == 1
    eng.step(1 + 1024)
    np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, want, r, 1025))


def test_graph_and_eager_agree(eng):
    G = 256
    r = rules("clustered")
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=77, and_rounds=1)
    eng.upload_state(st)
    eng.step(1)      # odd parity first: forces single steps before graph replays
    eng.step(64 + 64 + 5)
    a = eng.read_state()
    eng.set_option("graph", 0)
    try:
        eng.upload_state(st)
        eng.step(134)
        b = eng.read_state()
    finally:
        eng.set_option("graph", 1)
    np.testing.assert_array_equal(a, b)


def test_error_behaviour(eng):
    from cellularautomatons3d_amd import Engine

    e = Engine(0)
    try:
        with pytest.raises(Ca3dError) as ei:
            e.step(1)
        assert ei.value.code == -2
        with pytest.raises(Ca3dError):
            e.configure(48)  # not a multiple of 32
        e.configure(64)
        with pytest.raises(Ca3dError) as ei:
            e.upload_state(np.zeros(5, dtype=np.uint32))
        assert ei.value.code == -1
        with pytest.raises(Ca3dError) as ei:
            e.step(1)  # rules missing
        assert ei.value.code == -2
        vn = host.NEIGHBOURHOOD_MAP["von neumann"]
        b, s = host.recalculate_rules_values()
        with pytest.raises(Ca3dError):
            e.set_rules(vn[:5], vn, vn, s, b)  # not xyz triples
        with pytest.raises(Ca3dError) as ei:
            e.set_rules(np.array([2, 0, 0], dtype=np.int32), vn, vn, s, b)  # outside the 3x3x3 shell
        assert ei.value.code == -5
        e.set_rules(vn, host.NEIGHBOURHOOD_MAP["edges"], host.NEIGHBOURHOOD_MAP["corners"], s, b)
        with pytest.raises(Ca3dError):
            e.step(1)  # no state yet
        e.upload_state(host.initial_state(64))
        e.step(0)
        assert e.info().step == 0
    finally:
        e.close()


@pytest.mark.parametrize("G", [128, 256, 512])
@pytest.mark.parametrize("name", ["default", "vn2d"])
def test_fused_two_step_passes(eng, G, name):
    """Temporal blocking (2 steps per launch) must not change a single bit, for any step count and parity."""
    r = rules(name)
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=100 + G, and_rounds=1 if name == "default" else 5)
    eng.set_option("graph", 0)
    eng.set_option("fused", 1)
    try:
        eng.upload_state(st)
        want = st
        done = 0
        for n in (3, 4, 1, 5, 8):
            eng.step(n)
            want = ol.packed_run(G, want, r, n)
            done += n
            np.testing.assert_array_equal(eng.read_state(), want)
            assert eng.info().step == done and eng.info().current_buffer == done % 2
        eng.set_option("fused", 0)
        eng.upload_state(st)
        eng.step(done)
        np.testing.assert_array_equal(eng.read_state(), want)
    finally:
        eng.set_option("fused", 0)
        eng.set_option("graph", 1)


def test_fused_keeps_previous_state_in_other_buffer(eng):
    import ctypes as C

    import torch

    G = 256
    r = rules("default")
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=5, and_rounds=1)
    eng.set_option("fused", 1)
    eng.upload_state(st)
    eng.step(70)  # a graph replay (64) + a fused/single tail
    eng.set_option("fused", 0)
    assert eng.info().current_buffer == 0
    s69 = ol.packed_run(G, st, r, 69)
    np.testing.assert_array_equal(eng.read_state(), ol.packed_step(G, s69, r))
    p, n = eng.device_buffer(1)
    eng.synchronize()
    other = torch.empty(n // 4, dtype=torch.int32, device="cuda:0")
    rt = C.CDLL("libamdhip64.so")
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert rt.hipMemcpy(other.data_ptr(), p, n, 3) == 0
    np.testing.assert_array_equal(other.cpu().numpy().view(np.uint32), s69)


def test_checkpoint_resume(eng, tmp_path):
    G = 128
    r = rules("clustered")
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=41, and_rounds=1)
    eng.upload_state(st)
    eng.step(4)
    eng.save_checkpoint(tmp_path / "s4.ca3d")
    eng.step(6)
    want = eng.read_state()
    assert eng.load_checkpoint(tmp_path / "s4.ca3d") == 4
    set_rules(eng, r)
    eng.step(6)
    np.testing.assert_array_equal(eng.read_state(), want)


@pytest.mark.parametrize("tables,zsplit", [("default", 2), ("vn_b24_s135", 1), ("default", 1)])
def test_resident_kernel_at_256(eng, tables, zsplit):
    """BASELINE configs[1] (256^3, 1000 steps) through the resident kernel's 256^3 form: 256 tiles of 8 words x 32 rows x 8
    planes (rows of 8 words: two grid rows per DPP row), y faces smaller than a tile's thread count. Batches of several
    lengths, a sparse state, the other ping-pong buffer, and 1000 steps in one launch against the oracle."""
    G = 256
    r = rules(tables)
    eng.configure(G)
    eng.set_option("resident_zsplit", zsplit)
    set_rules(eng, r)
    assert eng.info().kernel_name.startswith(b"ca_resident_vn")
    st = host.random_fill(host.words_per_buffer(G), seed=256)
    eng.upload_state(st)
    want = st
    total = 0
    from cellularautomatons3d_amd import slab
    for n in (8, 9, 21, 1, 2):
        if n < 8:
            eng.set_option("resident_min", 1)  # short batches through the resident path too (one round of one step / of two)
        eng.step(n)
        prev = ol.packed_run(G, want, r, n - 1)
        want = ol.packed_step(G, prev, r)
        total += n
        np.testing.assert_array_equal(eng.read_state(), want, err_msg=f"after a batch of {n}")
        assert eng.info().current_buffer == total % 2 and eng.info().step == total
        other = slab.device_tensor(*eng.device_buffer(1 - total % 2), 0).cpu().numpy().view(np.uint32)
        np.testing.assert_array_equal(other, prev, err_msg="the other buffer holds the state one step earlier")
    eng.set_option("resident_min", 8)
    st2 = host.initial_state(G)
    eng.upload_state(st2)
    eng.step(100)
    np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st2, r, 100))
    # live cells hugging the dead -y / -z faces and the wrapping +y / +z faces, and both x ends
    edge = np.zeros(host.words_per_buffer(G), dtype=np.uint32)
    e3 = edge.reshape(G, G, G // 32)
    e3[:3, :3, :] = host.random_fill(9 * (G // 32), seed=7).reshape(3, 3, G // 32)
    e3[-3:, -3:, :] = host.random_fill(9 * (G // 32), seed=8).reshape(3, 3, G // 32)
    e3[:3, -3:, :] = host.random_fill(9 * (G // 32), seed=9).reshape(3, 3, G // 32)
    e3[-3:, :3, :] = host.random_fill(9 * (G // 32), seed=10).reshape(3, 3, G // 32)
    eng.upload_state(edge)
    for n in (9, 12):
        eng.step(n)
    np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, edge, r, 21), err_msg="boundary slabs, 21 steps")
    eng.upload_state(st)
    eng.step(1000)
    assert eng.info().kernel_name.startswith(b"ca_resident_vn")
    got = eng.read_state()
    np.testing.assert_array_equal(got, ol.packed_run(G, st, r, 1000))
    eng.set_option("resident", 0)
    try:
        assert eng.info().kernel_name.startswith(b"ca_packed_vn")
        eng.upload_state(st)
        eng.step(1000)
        np.testing.assert_array_equal(eng.read_state(), got)
    finally:
        eng.set_option("resident", 1)
        eng.set_option("resident_zsplit", 1)


@pytest.mark.parametrize("tables", ["default", "vn_b24_s135", "B2,4,5/S0,3,6"])
def test_resident_kernel_at_64(eng, tables):
    """The reference UI's start-up grid, 64^3 (main_pathtraced.js:101), through the one-workgroup resident kernel (resident64_run: the
    whole state in the registers of one CU, lane = row, a wave = four planes, y neighbours by wave-wide DPP, plane exchange between
    waves through LDS): batches of several lengths incl. one and two steps, the other ping-pong buffer, the single seed, live cells
    hugging every face (dead -x / -y / -z, wrapping +x / +y / +z — here the wraps are lane 63 -> lane 0 and wave 15 -> wave 0), and
    1000 steps in one launch against the oracle and against the per-step kernels."""
    G = 64
    if tables in RULESETS:
        r = rules(tables)
    else:
        b, s_ = tables[1:].split("/S")
        r = ol.Rules.from_strings(neighbourhood="von neumann", born=b, survive=s_)
    eng.configure(G)
    set_rules(eng, r)
    assert eng.info().kernel_name.startswith(b"ca_resident_vn"), eng.info().kernel_name
    st = host.random_fill(host.words_per_buffer(G), seed=64)
    eng.upload_state(st)
    want = st
    total = 0
    from cellularautomatons3d_amd import slab
    try:
        for n in (8, 9, 21, 1, 2):
            if n < 8:
                eng.set_option("resident_min", 1)
            eng.step(n)
            prev = ol.packed_run(G, want, r, n - 1)
            want = ol.packed_step(G, prev, r)
            total += n
            np.testing.assert_array_equal(eng.read_state(), want, err_msg=f"after a batch of {n}")
            assert eng.info().current_buffer == total % 2 and eng.info().step == total
            other = slab.device_tensor(*eng.device_buffer(1 - total % 2), 0).cpu().numpy().view(np.uint32)
            np.testing.assert_array_equal(other, prev, err_msg="the other buffer holds the state one step earlier")
    finally:
        eng.set_option("resident_min", 8)
    st2 = host.initial_state(G)
    eng.upload_state(st2)
    eng.step(100)
    np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st2, r, 100))
    edge = np.zeros(host.words_per_buffer(G), dtype=np.uint32)
    e3 = edge.reshape(G, G, G // 32)
    e3[:3, :3, :] = host.random_fill(9 * (G // 32), seed=7).reshape(3, 3, G // 32)
    e3[-3:, -3:, :] = host.random_fill(9 * (G // 32), seed=8).reshape(3, 3, G // 32)
    e3[:3, -3:, :] = host.random_fill(9 * (G // 32), seed=9).reshape(3, 3, G // 32)
    e3[-3:, :3, :] = host.random_fill(9 * (G // 32), seed=10).reshape(3, 3, G // 32)
    e3[20:40, 20:40, 0] |= 1           # cells at x == 0
    e3[20:40, 20:40, 1] |= 0x80000000  # and at x == 63
    eng.upload_state(edge)
    for n in (9, 12):
        eng.step(n)
    np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, edge, r, 21), err_msg="faces, 21 steps")
    eng.upload_state(st)
    eng.step(1000)
    assert eng.info().kernel_name.startswith(b"ca_resident_vn")
    got = eng.read_state()
    np.testing.assert_array_equal(got, ol.packed_run(G, st, r, 1000))
    eng.set_option("resident", 0)
    try:
        assert eng.info().kernel_name.startswith(b"ca_packed_rows"), eng.info().kernel_name
        eng.upload_state(st)
        eng.step(1000)
        np.testing.assert_array_equal(eng.read_state(), got)
    finally:
        eng.set_option("resident", 1)
    assert eng.recovered_launches() == 0


def test_queued_submission(eng):
    """Option "queue": ca3d_step only encodes, the steps of consecutive calls are submitted together (ca3d_flush, any call
    that looks at the state, or once `queue` steps wait) — the reference's commandEncoder + queue.submit
    (main_pathtraced.js:1833-1850). Same states as per-call submission, fewer launches; rules apply to the steps encoded
    under them; an upload drops what was never submitted."""
    G = 512
    r = rules("default")
    eng.configure(G)
    set_rules(eng, r)
    st = host.random_fill(host.words_per_buffer(G), seed=99)
    eng.upload_state(st)
    eng.set_option("queue", 64)
    try:
        l0 = eng.info().launches_total
        for _ in range(20):
            eng.step(5)  # 13 calls reach 65 >= 64 steps: one launch of the resident kernel; 35 steps stay encoded
        eng.step(3)
        info = eng.info()  # submits the remaining 38 steps
        assert info.step == 103 and info.current_buffer == 1
        assert info.launches_total - l0 == 2 and info.kernel_name.startswith(b"ca_resident_vn")
        want = ol.packed_run(G, st, r, 103)
        np.testing.assert_array_equal(eng.read_state(), want)
        # a rule change between encoded steps: each step runs under the rules it was encoded with
        r2 = rules("vn_b24_s135")
        eng.step(10)
        set_rules(eng, r2)
        eng.step(9)
        eng.flush()
        want = ol.packed_run(G, ol.packed_run(G, want, r, 10), r2, 9)
        np.testing.assert_array_equal(eng.read_state(), want)
        # short submissions fall back to the per-step kernels; an upload discards encoded steps
        eng.step(2)
        eng.upload_state(st)
        assert eng.info().step == 0
        eng.step(2)
        eng.step(1)
        np.testing.assert_array_equal(eng.read_state(), ol.packed_run(G, st, r2, 3))
    finally:
        eng.set_option("queue", 0)


def test_bench_line_single_gpu_queued_and_per_call():
    """bench.py at N = 1 as the driver runs it (a small grid): one JSON line and nothing else on stdout, the state checked
    against the oracle after warm-up + calibration + timed steps, the K-step calls submitted together (`ca3d_flush`) with the
    per-call figure beside it, `roofline` per launch of the resident kernel."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--grid", "256", "--steps", "20", "--warmup", "5", "--queue", "256",
           "--min-seconds", "0.0001", "--no-cpu-baseline", "--no-render", "--no-scaling-base", "--check", "--compare-submission"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["oracle_match"] is True and d["n_gpus"] == 1 and d["steps"] == 20 and d["unit"] == "Gcells/s"
    rf = d["roofline"]
    # a resident kernel keeps the state on chip: its limiter is vector-instruction issue, the algorithmic-byte rate is kept as a
    # rate beside it; a fraction of a peak never exceeds 1
    assert rf["kernel"].startswith("ca_resident_vn") and rf["steps_per_launch"] == 260 and rf["bound"] == "valu_issue"
    assert rf["hbm_equivalent"]["achieved"] > 0 and "frac" not in rf["hbm_equivalent"]
    if rf["frac"] is not None:
        assert 0 < rf["frac"] <= 1 and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3 and os.path.exists(os.path.join(root, rf["counter_source"]))
    assert d["config"]["submission"].startswith("queued") and d["reps"] % 13 == 0
    other = d["other_submission"]
    assert other["submission"] == "per call" and other["roofline"]["steps_per_launch"] == 20 and other["value"] > 0
    assert d["per_call"]["roofline"]["steps_per_launch"] == 20
    ps = d["per_step_kernels"]["roofline"]
    assert ps["bound"] == "hbm" and ps["kernel"].startswith("ca_packed_vn") and 0 < ps["frac"] <= 1 and abs(ps["achieved"] / ps["peak"] - ps["frac"]) < 1e-3
    assert d["copy_ceiling"]["value"] > 1000
    # the line verifies itself: warm-up + 40 steps through the headline path (one queued submission: the resident kernel) against the
    # oracle before anything is timed
    v = d["verified"]
    assert v["oracle_match"] is True and v["steps"] == 45 and v["kernel"].startswith("ca_resident_vn")
    # a resident kernel's fraction is only given against a profile of the SAME instruction stream (kernel, grid, rule, form options,
    # device sources); the variant the run had is on the line either way
    assert rf["variant"].startswith(rf["kernel"]) and ";rule=" in rf["variant"] and ";src=" in rf["variant"]
    if rf["frac"] is not None:
        assert json.load(open(os.path.join(root, rf["counter_source"])))["variant"] == rf["variant"]


def test_bench_scaling_base_names_both_single_gpu_paths():
    """The N = 1 line's `scaling_base`: 1024^3 on one GPU through the full-grid engine (per-step kernels) and, beside it, as four Z-slabs
    behind the group handle on the same device (resident slab kernel: the state crosses HBM once per 16-step batch)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--min-seconds", "0.05", "--no-cpu-baseline", "--no-render",
           "--no-per-step-leg", "--no-per-call-leg", "--no-grid-256", "--no-interactive", "--verify-steps", "0"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    sb = d["scaling_base"]
    assert sb["grid"] == 1024 and sb["roofline"]["kernel"].startswith("ca_packed_vn") and sb["value"] > 0
    g4 = sb["as_four_resident_slabs"]
    assert g4["kernel"].startswith("ca_resident_slab") and g4["slabs"] == 4 and g4["ghost"] == 16 and g4["value"] > 0


def test_bench_interactive_frame_leg():
    """bench.py's `interactive_frame`: the reference's own loop — ca3d_step(1) + one literal frame per submission (main_pathtraced.js:1821-1854)
    — at the driver's grid, with the kernels that ran; and the literal frame leg beside the converged one."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--min-seconds", "0.0001", "--no-cpu-baseline", "--no-scaling-base",
           "--no-per-step-leg", "--no-per-call-leg", "--no-grid-256", "--render-frames", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert d["verified"]["oracle_match"] is True and d["verified"]["kernel"].startswith("ca_resident_vn")
    it = d["interactive_frame"]
    for scene in ("startup_scene", "dense_scene"):
        assert 0 < it[scene]["ms_per_frame"] < 50 and it[scene]["step_kernel"].startswith("ca_packed_vn")
    assert it["frames"] == 200
    assert 0 < d["render"]["literal_frame"]["ms_per_frame"] < d["render"]["ms_per_frame"] * 2

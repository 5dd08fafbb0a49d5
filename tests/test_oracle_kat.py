"""Pins the CPU oracle: literal restatement == word-parallel form on random inputs, plus analytic known answers
for compute_clustered.wgsl semantics."""
import numpy as np
import pytest

import oracle_lib as ol
from cellularautomatons3d_amd import host

R = ol.Rules.from_strings


def test_default_rule_population_sequence():
    # centre + 6 faces, then + 6 second-shell axis cells, ... (hand-checked for steps 1-2)
    for G in (32, 64):
        st = host.initial_state(G)
        pops = []
        for _ in range(8):
            st = ol.packed_step(G, st, R())
            pops.append(ol.popcount(st))
        assert pops == [7, 13, 43, 49, 79, 133, 259, 313]


def test_default_rule_first_step_cells():
    G = 32
    c = G // 2 - 1
    st = ol.packed_step_literal(G, host.initial_state(G), R())
    alive = {(c, c, c), (c + 1, c, c), (c - 1, c, c), (c, c + 1, c), (c, c - 1, c), (c, c, c + 1), (c, c, c - 1)}
    np.testing.assert_array_equal(st, host.cells_to_words(G, alive))


def test_default_rule_fnv_anchors():
    # Survey-time hashes from an independent throw-away JS restatement.
    want = {32: "8ebd1f9c c3f1930e 6b170f5a 7815260a dec86ca6 afa3b49c b79051df 23672217".split(),
            64: "eaf84574 2a89e9f6 08947442 7e303b52 44d3c10e 3ca0f014 548cafef ae46dc87".split()}
    for G, hashes in want.items():
        st = host.initial_state(G)
        got = []
        for _ in range(8):
            st = ol.packed_step(G, st, R())
            got.append("%08x" % ol.fnv1a32(st))
        assert got == hashes


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_boundary_asymmetry(axis):
    # `<= G` in the bounds test: +faces wrap to 0, -faces are dead (compute_clustered.wgsl:104).
    G = 32
    rules = R("von neumann", "1", "")

    def cell(v):
        c = [5, 5, 5]
        c[axis] = v
        return tuple(c)

    st = ol.packed_step_literal(G, host.cells_to_words(G, [cell(0)]), rules)
    assert ol.popcount(st) == 6 and host.get_cell(G, st, *cell(1)) == 1 and host.get_cell(G, st, *cell(31)) == 1
    st = ol.packed_step_literal(G, host.cells_to_words(G, [cell(31)]), rules)
    assert ol.popcount(st) == 5 and host.get_cell(G, st, *cell(30)) == 1 and host.get_cell(G, st, *cell(0)) == 0


def test_zero_wins_at_mixed_corner():
    # P(G, -1, z) = 0: a neighbour with any component -1 is dropped even if another component wraps.
    G = 32
    rules = R("moore", "1", "")
    st = ol.packed_step_literal(G, host.cells_to_words(G, [(0, 31, 7)]), rules)
    # (31, 0, 7) has the seed as its (+1 -> wraps to 0, -1 -> dropped) neighbour: must stay dead.
    assert host.get_cell(G, st, 31, 0, 7) == 0
    # (31, 30, 7): neighbour (32 -> 0, 31, 7) counts.
    assert host.get_cell(G, st, 31, 30, 7) == 1


def test_moore_b4s4_oscillator():
    G = 32
    seed = [(15, 15, 15), (16, 15, 15), (15, 16, 15), (15, 15, 16)]
    s0 = host.cells_to_words(G, seed)
    rules = R("moore", "4", "4")
    s1 = ol.packed_step(G, s0, rules)
    s2 = ol.packed_step(G, s1, rules)
    assert ol.popcount(s1) == 4 and ol.popcount(s2) == 4
    np.testing.assert_array_equal(s2, s0)
    assert not np.array_equal(s1, s0)


def test_rule_27_is_disabled_for_edges_and_corners():
    # "27" clamps to slot 26, unreachable for 12- or 8-neighbour counts (main_pathtraced.js:129-132, 575).
    G = 32
    st = host.random_fill(host.words_per_buffer(G), and_rounds=1)
    a = ol.packed_step(G, st, R("von neumann", "1,3", "0-6", "27", "27", "27", "27"))
    b = ol.packed_step(G, st, R("von neumann", "1,3", "0-6", "", "", "", ""))
    np.testing.assert_array_equal(a, b)


def test_lut_value_must_equal_one():
    # any(result == vec3u(1)) (compute_clustered.wgsl:232): a LUT entry of 2 does not make a cell alive.
    G = 32
    r = R("von neumann", "1", "")
    r.born = r.born.copy()
    r.born[1] = 2
    st = host.cells_to_words(G, [(5, 5, 5)])
    assert ol.popcount(ol.packed_step_literal(G, st, r)) == 0
    assert ol.popcount(ol.packed_step(G, st, r)) == 0


RULESETS = [
    dict(),
    dict(neighbourhood="moore", born="5-7", survive="4-7", born_edges="4", survive_edges="3-5", born_corners="3", survive_corners="2-4"),
    dict(neighbourhood="moore 2D", born="3", survive="2,3"),
    dict(neighbourhood="von neumann 2D", born="1", survive=""),
    dict(neighbourhood="edges", born="2,6,9", survive="4,6,8-9", born_edges="2", survive_corners="1"),
    dict(neighbourhood="corners", born="1", survive="0-8", survive_edges="0", born_corners="8"),
    dict(neighbourhood="moore", born="13-14,17-19", survive="13-26", born_edges="0", survive_corners="0"),
]


@pytest.mark.parametrize("G", [32, 64, 96])
@pytest.mark.parametrize("ri", range(len(RULESETS)))
def test_fast_equals_literal(G, ri):
    if G == 96 and ri not in (0, 1):
        pytest.skip("keep the CPU suite short")
    rules = R(**RULESETS[ri])
    for rounds in (0, 2):
        st = host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001 + ri, and_rounds=rounds)
        np.testing.assert_array_equal(ol.packed_step(G, st, rules), ol.packed_step_literal(G, st, rules))


def test_fast_equals_literal_arbitrary_offset_lists():
    # The ABI takes any xyz-triple list within +-1, including duplicates and (0,0,0).
    G = 32
    rng = np.random.default_rng(7)
    for _ in range(6):
        lists = [rng.integers(-1, 2, size=3 * int(rng.integers(0, 9))).astype(np.int32) for _ in range(3)]
        survive = (rng.random(81) < 0.3).astype(np.uint32)
        born = (rng.random(81) < 0.3).astype(np.uint32)
        rules = ol.Rules(lists[0], lists[1], lists[2], survive, born)
        st = host.random_fill(host.words_per_buffer(G), seed=int(rng.integers(1 << 30)))
        np.testing.assert_array_equal(ol.packed_step(G, st, rules), ol.packed_step_literal(G, st, rules))


def test_slab_planes_equal_full_grid():
    # Z-slabs with one ghost plane each side reproduce the full-grid step, including the open bottom / wrapped
    # top of the packed kernel.
    G, P = 64, 4
    rules = R(**RULESETS[1])
    st = host.random_fill(host.words_per_buffer(G), seed=99)
    full = ol.packed_step(G, st, rules).reshape(G, -1)
    planes = st.reshape(G, -1)
    nz = G // P
    for k in range(P):
        z0 = k * nz
        idx = [(z0 - 1) % G] + list(range(z0, z0 + nz)) + [(z0 + nz) % G]
        slab = planes[idx].copy()
        if k == 0:
            slab[0] = 0xFFFFFFFF  # must be ignored: global z = -1 is dead
        out = ol.packed_step_planes(G, slab, z0 - 1, 1, nz + 1, rules).reshape(nz + 2, -1)
        np.testing.assert_array_equal(out[1:nz + 1], full[z0:z0 + nz])


def test_deep_ghost_slab_two_steps():
    # Two steps on a slab with two ghost planes per side == two full-grid steps (ghost copy of plane 0 above
    # plane G-1 evolves with a dead plane below it).
    G, P, K = 64, 2, 2
    rules = R(**RULESETS[1])
    st = host.random_fill(host.words_per_buffer(G), seed=5)
    full2 = ol.packed_run(G, st, rules, 2).reshape(G, -1)
    planes = st.reshape(G, -1)
    nz = G // P
    for k in range(P):
        z0 = k * nz
        idx = [(z0 - K + j) % G for j in range(nz + 2 * K)]
        slab = planes[idx].copy()
        L = nz + 2 * K
        lo1 = K if k == 0 else 1
        s1 = ol.packed_step_planes(G, slab, z0 - K, lo1, L - 1, rules)
        s2 = ol.packed_step_planes(G, s1, z0 - K, K, L - K, rules).reshape(L, -1)
        np.testing.assert_array_equal(s2[K:K + nz], full2[z0:z0 + nz])


def test_unpacked_toroidal_and_rules():
    # compute.wgsl: fully toroidal for power-of-two G; survive/born test uses > 0 and state == 1 / == 0.
    G = 16
    vn = host.NEIGHBOURHOOD_MAP["von neumann"]
    born, survive = host.recalculate_rules_values("1", "")
    st = np.zeros(G ** 3, dtype=np.uint32)
    st[0] = 1  # cell (0,0,0)
    out = ol.unpacked_step(G, st, vn, survive, born)
    want = np.zeros_like(st)
    for (x, y, z) in [(1, 0, 0), (G - 1, 0, 0), (0, 1, 0), (0, G - 1, 0), (0, 0, 1), (0, 0, G - 1)]:
        want[x + y * G + z * G * G] = 1
    np.testing.assert_array_equal(out, want)


def test_unpacked_non_pow2_wrap_quirk():
    # vec3u(-1) % G == 0xFFFFFFFF % G, which is G-1 only for power-of-two G (compute.wgsl:24-27, 42).
    G = 12
    vn = host.NEIGHBOURHOOD_MAP["von neumann"]
    born, survive = host.recalculate_rules_values("1", "")
    st = np.zeros(G ** 3, dtype=np.uint32)
    q = 0xFFFFFFFF % G  # = 3
    st[q] = 1  # cell (3,0,0)
    out = ol.unpacked_step(G, st, vn, survive, born)
    assert out[0] == 1  # (0,0,0) sees (3,0,0) as its -x neighbour
    assert q != G - 1

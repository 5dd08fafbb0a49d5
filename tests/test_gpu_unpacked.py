"""Legacy one-u32-per-cell kernel (shaders/compute.wgsl) through the C ABI against the oracle."""
import numpy as np
import pytest

import oracle_lib as ol
from cellularautomatons3d_amd import LAYOUT_UNPACKED, host
from gpu_common import set_rules

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G", [12, 16, 64, 128])
@pytest.mark.parametrize("kw", [dict(), dict(neighbourhood="moore", born="5-7", survive="4-9"),
                                dict(neighbourhood="moore 2D", born="3", survive="2,3")])
def test_unpacked_steps(G, kw):
    from cellularautomatons3d_amd import Engine

    r = ol.Rules.from_strings(**kw)
    with Engine(0) as e:
        e.configure(G, LAYOUT_UNPACKED)
        set_rules(e, r)
        st = (host.random_fill(G ** 3, seed=G) & 1).astype(np.uint32)
        e.upload_state(st)
        e.step(3)
        cur = st
        for _ in range(3):
            cur = ol.unpacked_step(G, cur, r.main, r.survive, r.born)
        np.testing.assert_array_equal(e.read_state(), cur)


def test_unpacked_non_binary_states_follow_the_literal_kernel():
    # raw u32 sums; `state == 1` survives, `state == 0` may be born, anything else dies (compute.wgsl:160-174)
    from cellularautomatons3d_amd import Engine

    G = 16
    r = ol.Rules.from_strings("von neumann", "1,2,3", "0-6")
    st = (host.random_fill(G ** 3, seed=3) % 3).astype(np.uint32)
    with Engine(0) as e:
        e.configure(G, LAYOUT_UNPACKED)
        set_rules(e, r)
        e.upload_state(st)
        e.step(1)
        np.testing.assert_array_equal(e.read_state(), ol.unpacked_step(G, st, r.main, r.survive, r.born))


@pytest.mark.parametrize("G", [128, 256])
@pytest.mark.parametrize("kw", [dict(), dict(neighbourhood="moore", born="5-7", survive="4-9"),
                                dict(neighbourhood="moore 2D", born="3", survive="2,3"),
                                dict(neighbourhood="edges", born="2-3", survive="1-4"),
                                dict(neighbourhood="corners", born="1", survive="0-8")])
def test_ballot_kernel(G, kw):
    """0/1 states on a power-of-two grid take the ballot-packing kernel; it must equal the literal restatement."""
    from cellularautomatons3d_amd import Engine

    r = ol.Rules.from_strings(**kw)
    with Engine(0) as e:
        e.configure(G, LAYOUT_UNPACKED)
        set_rules(e, r)
        st = (host.random_fill(G ** 3, seed=G + 1) & 1).astype(np.uint32)
        e.upload_state(st)
        e.step(3)
        assert e.info().kernel_name == b"ca_unpacked_ballot"
        cur = st
        for _ in range(3):
            cur = ol.unpacked_step(G, cur, r.main, r.survive, r.born)
        np.testing.assert_array_equal(e.read_state(), cur)
        # the literal kernel on the same input
        e.set_option("variant", 1)
        e.upload_state(st)
        e.step(3)
        assert e.info().kernel_name == b"ca_unpacked_literal"
        np.testing.assert_array_equal(e.read_state(), cur)


def test_non_binary_upload_takes_the_literal_kernel_first():
    from cellularautomatons3d_amd import Engine

    G = 128
    r = ol.Rules.from_strings("von neumann", "1,2,3", "0-6")
    st = (host.random_fill(G ** 3, seed=3) % 3).astype(np.uint32)
    with Engine(0) as e:
        e.configure(G, LAYOUT_UNPACKED)
        set_rules(e, r)
        e.upload_state(st)
        e.step(1)
        assert e.info().kernel_name == b"ca_unpacked_literal"
        s1 = ol.unpacked_step(G, st, r.main, r.survive, r.born)
        np.testing.assert_array_equal(e.read_state(), s1)
        e.step(2)
        assert e.info().kernel_name == b"ca_unpacked_ballot"  # every cell is 0 / 1 after a step
        s3 = ol.unpacked_step(G, ol.unpacked_step(G, s1, r.main, r.survive, r.born), r.main, r.survive, r.born)
        np.testing.assert_array_equal(e.read_state(), s3)


def test_graph_replay_after_a_non_binary_upload():
    """A captured graph must never replay the 0/1-only ballot kernel on a state with cell values > 1: binary upload,
    a graph-sized batch, then an upload with values > 1 and the same batch again (compute.wgsl:160-174 sums raw u32
    values and tests `== 1` / `== 0`)."""
    from cellularautomatons3d_amd import Engine

    G, n = 128, 130  # >= graph_min (128): the batch replays as a captured graph
    r = ol.Rules.from_strings("von neumann", "1,3", "0-6")
    a = (host.random_fill(G ** 3, seed=41) & 1).astype(np.uint32)
    b = (host.random_fill(G ** 3, seed=42) % 4).astype(np.uint32)  # values 0..3
    with Engine(0) as e:
        e.configure(G, LAYOUT_UNPACKED)
        e.set_rules(r.main, r.edges, r.corners, r.survive, r.born)
        for st in (a, b, a):
            e.upload_state(st)
            e.step(n)
            cur = st
            for _ in range(n):
                cur = ol.unpacked_step(G, cur, r.main, r.survive, r.born)
            np.testing.assert_array_equal(e.read_state(), cur)


@pytest.mark.parametrize("kw,rounds", [(dict(), 0), (dict(neighbourhood="moore", born="5-7", survive="4-9"), 3)])
def test_pipelined_kernel_at_512(kw, rounds):
    """512^3 in the one-u32-per-cell layout (512 MiB per buffer) takes the pipelined form of the ballot kernel — tiles of 32 planes walked in
    groups of four, the next group's rows loaded under this group's stores: two steps from a dense (start-up rule) / a sparse (a Moore rule)
    0 / 1 state against the oracle's restatement of compute.wgsl."""
    from cellularautomatons3d_amd import Engine

    G = 512
    r = ol.Rules.from_strings(**kw)
    with Engine(0) as e:
        e.configure(G, LAYOUT_UNPACKED)
        set_rules(e, r)
        st = (host.random_fill(G ** 3 // 32, seed=512 + rounds, and_rounds=rounds)[:, None] >> np.arange(32, dtype=np.uint32) & 1).astype(np.uint32).ravel()
        e.upload_state(st)
        e.step(2)
        assert e.info().kernel_name == b"ca_unpacked_ballot"
        cur = ol.unpacked_step(G, ol.unpacked_step(G, st, r.main, r.survive, r.born), r.main, r.survive, r.born)
        assert np.array_equal(e.read_state(), cur)  # (assert_array_equal would format 512 MiB arrays on failure)

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

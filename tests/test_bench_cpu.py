"""Host-side pieces of bench.py that need no GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_auto_ghost_keeps_a_share_on_the_resident_slab_kernel():
    """Ghost depth when none is given: 32, or the deepest of 32 / 16 / 8 with which a rank's share of a 1024^3 von Neumann grid still divides
    into 8 tile layers of an even number of planes <= 36 (ca_resident.hip, resident_slab_planes)."""
    assert bench.auto_ghost(1024, "default", 8) == 32   # 128 + 64 planes: 24 per layer
    assert bench.auto_ghost(1024, "default", 4) == 16   # 256 + 32: 36 per layer (with 32 ghost planes it would be 40)
    assert bench.auto_ghost(1024, "default", 2) == 32   # 512 + ...: no resident form, the default depth
    assert bench.auto_ghost(1024, "clustered", 4) == 32 and bench.auto_ghost(2048, "default", 8) == 32 and bench.auto_ghost(1024, "default", 1) == 32
    for world in (4, 8):
        k = bench.auto_ghost(1024, "vn_b24_s135", world)
        planes = 1024 // world + 2 * k
        assert planes % 8 == 0 and (planes // 8) % 2 == 0 and planes // 8 <= 36

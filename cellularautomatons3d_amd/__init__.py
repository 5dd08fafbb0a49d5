"""cellularautomatons3d_amd — MI355X-native engine for the CA-step and volume-render hot paths of
lightest/cellularautomatons3d. The product is the C-ABI library (include/ca3d.h, csrc/); this package is the
Python host-side mirror of the reference's rule / grid / step surface used by tests and bench.py.
"""
from . import host  # noqa: F401
from ._capi import (LAYOUT_PACKED32, LAYOUT_UNPACKED, SLAB_PHASE_ALL, SLAB_PHASE_EDGES, SLAB_PHASE_INTERIOR,  # noqa: F401
                    Ca3dError)


def __getattr__(name):
    if name in ("Engine", "EngineGroup"):
        from . import engine

        return getattr(engine, name)
    raise AttributeError(name)

import numpy as np

import oracle_lib as ol
from cellularautomatons3d_amd import host

RULESETS = {
    "default": dict(),
    "clustered": dict(neighbourhood="moore", born="5-7", survive="4-7", born_edges="4", survive_edges="3-5",
                      born_corners="3", survive_corners="2-4"),
    "life2d": dict(neighbourhood="moore 2D", born="3", survive="2,3"),
    "vn2d": dict(neighbourhood="von neumann 2D", born="1", survive=""),
    "edges_main": dict(neighbourhood="edges", born="2,6,9", survive="4,6,8-9", born_edges="2", survive_corners="1"),
    "corners_main": dict(neighbourhood="corners", born="1", survive="0-8", survive_edges="0", born_corners="8"),
    "moore_wide": dict(neighbourhood="moore", born="13-14,17-19", survive="13-26", born_edges="0", survive_corners="0"),
    "moore_b4s4": dict(neighbourhood="moore", born="4", survive="4"),
    "vn_b24_s135": dict(neighbourhood="von neumann", born="2,4", survive="1,3,5"),
    "vn_edges_only": dict(neighbourhood="von neumann", born="2", survive="1-3", born_edges="3-4", survive_edges="2"),
    "vn_corners_only": dict(neighbourhood="von neumann", born="2", survive="1-3", born_corners="1", survive_corners="2,4"),
}


def rules(name):
    return ol.Rules.from_strings(**RULESETS[name])


def set_rules(engine, r):
    engine.set_rules(r.main, r.edges, r.corners, r.survive, r.born)

"""The on-disk cache of run-time compiled code objects (ca_jit.cpp): a second PROCESS with the same rules compiles nothing.

`_restartSim` (main_pathtraced.js:624-637) is synchronous and cheap in the reference; here a rule edit selects kernels compiled for
the rule (hiprtc, 0.3-1 s per program, several per rule). The in-process module cache covered a repeated edit, not a new process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
from cellularautomatons3d_amd import host
from gpu_common import RULESETS

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, time, hashlib
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import oracle_lib as ol
from cellularautomatons3d_amd import Engine, host, _capi
from gpu_common import RULESETS
out = {}
with Engine(0) as eng:
    for name, G in (("clustered", 128), ("default", 512), ("clustered", 512), ("default", 96)):
        r = ol.Rules.from_strings(**RULESETS[name])
        t0 = time.perf_counter()
        eng.configure(G)
        t1 = time.perf_counter()
        eng.set_rules(r.main, r.edges, r.corners, r.survive, r.born)
        t2 = time.perf_counter()
        st = host.random_fill(host.words_per_buffer(G), seed=3)
        eng.upload_state(st)
        eng.step(9)
        out[f"{name}@{G}"] = {"configure_ms": (t1 - t0) * 1e3, "set_rules_ms": (t2 - t1) * 1e3, "kernel": eng.info().kernel_name.decode(),
                              "jit_log": eng.jit_log(), "sha": hashlib.sha256(eng.read_state().tobytes()).hexdigest()}
    # the same rules again in this process: the in-memory modules
    r = ol.Rules.from_strings(**RULESETS["clustered"])
    eng.configure(512)
    t1 = time.perf_counter()
    eng.set_rules(r.main, r.edges, r.corners, r.survive, r.born)
    out["warm_set_rules_ms"] = (time.perf_counter() - t1) * 1e3
out["jit"] = _capi.jit_stats()
print("RESULT " + json.dumps(out))
"""


def _run(cache_dir, extra_env=None):
    env = dict(os.environ, CA3D_CACHE_DIR=cache_dir)
    env.pop("CA3D_JIT_CACHE", None)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_second_process_compiles_nothing(tmp_path):
    cache = str(tmp_path / "ca3d-cache")
    cold = _run(cache)
    assert cold["jit"]["programs_compiled"] >= 4 and cold["jit"]["programs_from_disk"] == 0, cold["jit"]
    assert cold["jit"]["cache_dir"] == cache
    files = sorted(os.listdir(cache))
    assert 1 <= len(files) <= cold["jit"]["programs_compiled"] and all(f.endswith(".hsaco") for f in files), files
    warm = _run(cache)
    assert warm["jit"]["programs_compiled"] == 0 and warm["jit"]["programs_from_disk"] == cold["jit"]["programs_compiled"], warm["jit"]
    for k, v in cold.items():
        if isinstance(v, dict) and "sha" in v:
            assert warm[k]["sha"] == v["sha"] and warm[k]["kernel"] == v["kernel"], (k, v, warm[k])
            assert v["jit_log"] == "" and warm[k]["jit_log"] == ""
    print("cold:", {k: (round(v["configure_ms"] + v["set_rules_ms"], 1) if isinstance(v, dict) and "sha" in v else v) for k, v in cold.items()})
    print("warm:", {k: (round(v["configure_ms"] + v["set_rules_ms"], 1) if isinstance(v, dict) and "sha" in v else v) for k, v in warm.items()})
    # from the cache a rule edit costs the loads only (the first one of a process also pays hiprtc's / the HIP runtime's own start-up)
    assert warm["jit"]["compile_ms"] == 0.0
    assert warm["jit"]["disk_read_ms"] + warm["jit"]["load_ms"] < 0.25 * cold["jit"]["compile_ms"]
    assert warm["warm_set_rules_ms"] < 20.0, warm["warm_set_rules_ms"]
    # a damaged object (truncated; a flipped byte) is a miss, is recompiled and rewritten — never loaded
    victim = os.path.join(cache, files[0])
    blob = open(victim, "rb").read()
    open(victim, "wb").write(blob[: len(blob) // 2])
    victim2 = os.path.join(cache, files[1])
    blob2 = bytearray(open(victim2, "rb").read())
    blob2[len(blob2) // 2] ^= 0x40
    open(victim2, "wb").write(bytes(blob2))
    again = _run(cache)
    assert again["jit"]["programs_compiled"] == 2 and again["jit"]["programs_from_disk"] == len(files) - 2, again["jit"]
    for k, v in cold.items():
        if isinstance(v, dict) and "sha" in v:
            assert again[k]["sha"] == v["sha"]
    assert open(victim, "rb").read() == blob
    # kernels loaded from the cache are the kernels: the states of the process that compiled nothing equal the oracle's
    import hashlib

    for name, G in (("clustered", 128), ("default", 96)):
        r = ol.Rules.from_strings(**RULESETS[name])
        want = ol.packed_run(G, host.random_fill(host.words_per_buffer(G), seed=3), r, 9)
        assert hashlib.sha256(want.tobytes()).hexdigest() == warm[f"{name}@{G}"]["sha"], (name, G)
    # CA3D_JIT_CACHE=0: nothing read, nothing written
    off_dir = str(tmp_path / "unused")
    off = _run(off_dir, {"CA3D_JIT_CACHE": "0"})
    assert off["jit"]["cache_dir"] == "" and off["jit"]["programs_from_disk"] == 0 and off["jit"]["programs_compiled"] >= 4
    assert not os.path.exists(off_dir)

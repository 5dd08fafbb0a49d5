#!/usr/bin/env python3
"""Builds tools/ubench/resident_class_probe.hip for the clustered rule-set (twice: plain and with the per-phase cycle stamps) and
runs it: tools/run_class_probe.py [steps] [launches] [grid] [zgroups]. Binaries under tools/ubench/_build/ (git-ignored; they travel to the GPU box)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_jit_source_cpu as T  # noqa: E402

out = os.path.join(ROOT, "tools", "ubench", "_build")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "ca_jit_rule.inc"), "wb").write(T.CLUSTERED_RULE_FN)
tables = (0x000000F0, 0x000000E0, 0x0038, 0x0010, 0x0014, 0x0008)
defines = ["-DCA3D_JIT=1", "-DCA3D_JIT_MAIN=2", "-DCA3D_JIT_E=true", "-DCA3D_JIT_C=true"] + [f"-DCA3D_JIT_{n}={t}u" for n, t in zip(["TS0", "TB0", "TS1", "TB1", "TS2", "TB2"], tables)]
args = sys.argv[1:] or ["256", "3", "256", "2"]
for name, extra in (("plain", []), ("stamps", ["-DCA3D_RES_STAMPS=1"])):
    exe = os.path.join(out, f"probe_{name}")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", out, "-I", os.path.join(ROOT, "cellularautomatons3d_amd", "csrc")] + defines + extra + \
          [os.path.join(ROOT, "tools", "ubench", "resident_class_probe.hip"), "-o", exe]
    subprocess.run(cmd, check=True)
    if "--build-only" not in args:
        print(f"--- {name}", flush=True)
        subprocess.run([exe] + args, check=False)

#!/usr/bin/env python3
"""Compile one of the run-time (hiprtc) kernel programs exactly as ca_jit.cpp does — no GPU needed — and print per
entry point: VGPR / SGPR counts and the instruction mix. Usage:
  tools/jit_disasm.py roll --cvl 2 [--rule clustered] [--dump DIR]      (also: rollnp2 --cv 5, class, rclass)"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_jit_source_cpu as T  # noqa: E402

TABLES = {"clustered": (2, "true", "true", (0x000000F0, 0x000000E0, 0x0038, 0x0010, 0x0014, 0x0008)),
          "life2d": (3, "false", "false", (0x000C, 0x0008, 0, 0, 0, 0)),
          "moore": (2, "false", "false", (0x000000F0, 0x000000E0, 0, 0, 0, 0))}

ap = argparse.ArgumentParser()
ap.add_argument("program", choices=["roll", "rollnp2", "class", "rclass"])
ap.add_argument("--cvl", type=int, default=2)
ap.add_argument("--cv", type=int, default=5, help="rollnp2: uint4 per row (3, 5, 6, 7)")
ap.add_argument("--rule", default="clustered")
ap.add_argument("--dump", default="")
ap.add_argument("--extra", default="")
ap.add_argument("--synth", type=int, default=1, help="clustered rule: use the synthesised rule function (as the engine does)")
a = ap.parse_args()
main, e, c, tables = TABLES[a.rule]
defines = [b"-DCA3D_JIT_MAIN=%d" % main, b"-DCA3D_JIT_E=" + e.encode(), b"-DCA3D_JIT_C=" + c.encode()]
defines += [b"-DCA3D_JIT_%s=%du" % (n, t) for n, t in zip([b"TS0", b"TB0", b"TS1", b"TB1", b"TS2", b"TB2"], tables)]
defines += [x.encode() for x in a.extra.split()]
if a.program == "roll":
    code = T._compile(T._hiprtc(), T.ROLL_PROGRAM, b"ca3d_jit_roll.hip", defines + [b"-DCA3D_JIT_CVL=%d" % a.cvl], *([T.CLUSTERED_RULE_FN] if a.rule == "clustered" and a.synth else []))
elif a.program == "rollnp2":
    code = T._compile(T._hiprtc(), T.ROLL_PROGRAM, b"ca3d_jit_roll_np2.hip", defines + [b"-DCA3D_JIT_CV_NP2=%d" % a.cv], *([T.CLUSTERED_RULE_FN] if a.rule == "clustered" and a.synth else []))
elif a.program == "rclass":
    code = T._compile(T._hiprtc(), T.RESIDENT_CLASS_PROGRAM, b"ca3d_jit_resident_class.hip", defines, *([T.CLUSTERED_RULE_FN] if a.rule == "clustered" and a.synth else []))
else:
    code = T._compile(T._hiprtc(), T.CLASS_PROGRAM, b"ca3d_jit_class.hip", defines + [b"-DCA3D_JIT_ZR=4"])
out = a.dump or "/tmp/ca3d_jit_disasm"
os.makedirs(out, exist_ok=True)
co = os.path.join(out, "code.co")
open(co, "wb").write(code)
LLVM = "/opt/rocm/lib/llvm/bin"
notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for m in re.finditer(r"\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
    print(f"{m.group(1)}: sgpr {m.group(2)} vgpr {m.group(3)}")
for m in re.finditer(r"\.lds_size|\.private_segment_fixed_size:\s+(\d+)", notes):
    pass
dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
open(os.path.join(out, "code.s"), "w").write(dis)
cur, mix = None, collections.defaultdict(collections.Counter)
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"^\s+([a-z_0-9]+)", line)
    if cur and m:
        mix[cur][m.group(1)] += 1
for k, cnt in mix.items():
    tot = sum(cnt.values())
    valu = sum(v for n, v in cnt.items() if n.startswith("v_"))
    salu = sum(v for n, v in cnt.items() if n.startswith("s_"))
    print(f"{k}: {tot} instructions, {valu} VALU, {salu} SALU, loads {sum(v for n, v in cnt.items() if 'load' in n)}, stores {sum(v for n, v in cnt.items() if 'store' in n)}, scratch {sum(v for n, v in cnt.items() if 'scratch' in n)}")
    print("   ", ", ".join(f"{n} {v}" for n, v in cnt.most_common(14)))
print("disassembly:", os.path.join(out, "code.s"))

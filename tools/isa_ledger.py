#!/usr/bin/env python3
"""Instruction ledger of one kernel from hipcc -S output: per basic block the instructions by class (VALU / SALU / branch / LDS / vector memory /
waits), with the loop nest LLVM annotates. tools/isa_ledger.py <file.s> <mangled-name-substring> [first-line last-line]
(round 5: the per-visit ledger of the ray-stream walk, profiles/r5_render_walk_ledger.txt)."""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2]
start = next(i for i, l in enumerate(src) if re.match(r"^[A-Za-z_][\w$.]*:", l) and sub in l.split(":")[0])
end = next(i for i in range(start, len(src)) if ".amdhsa_group_segment_fixed_size" in src[i] or src[i].startswith("\t.end_amdhsa_kernel"))
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9


def cls(op):
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


blocks, cur = [], None
for n, l in enumerate(src[start:end]):
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
    if m or cur is None:
        cur = {"label": m.group(1) if m else "entry", "line": n, "note": (m.group(2) or "") if m else "", "c": {}}
        blocks.append(cur)
        if m:
            continue
    t = l.strip()
    if not t or t.startswith((";", ".", "//")):
        if "Loop Header" in t:
            cur["note"] += " " + t
        continue
    op = t.split()[0]
    k = cls(op)
    cur["c"][k] = cur["c"].get(k, 0) + 1
    if op in ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32"):
        cur["c"]["div/sqrt"] = cur["c"].get("div/sqrt", 0) + 1
print(f"kernel lines {start + 1}..{end + 1} of {sys.argv[1]}")
tot = {}
for b in blocks:
    if not (lo <= b["line"] <= hi):
        continue
    c = b["c"]
    for k, v in c.items():
        tot[k] = tot.get(k, 0) + v
    depth = re.search(r"Depth=(\d+)", b["note"])
    print(f"{b['line']:5d} {b['label']:11s} d{depth.group(1) if depth else '-'} " + " ".join(f"{k}={c[k]}" for k in ("valu", "salu", "branch", "lds", "vmem", "wait", "div/sqrt") if k in c)
          + ("   <- loop header" if "Loop Header" in b["note"] else ""))
print("total", tot)

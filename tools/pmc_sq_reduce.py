#!/usr/bin/env python3
"""Per-kernel means of a `rocprofv3 --pmc SQ_...` pass: tools/pmc_sq_reduce.py DIR KERNEL_SUBSTR [STEPS_PER_LAUNCH [BENCH_JSON]]  -> JSON on stdout.
BENCH_JSON: the bench line the profiled command printed; its roofline.variant (ca3d_get_kernel_variant: kernel, grid, rule hash, form
options, device-source hash) and the repo's HEAD are recorded, and bench.py uses the instruction count only for a run of the same variant.
STEPS_PER_LAUNCH (resident multi-step kernels): how many CA steps each profiled launch held; only launches of the most
common duration class are averaged when given (the run also holds a shorter warm-up launch).
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); ratios between them are unit-free."""
import csv, glob, json, os, sys, collections

d, kernel = sys.argv[1], sys.argv[2]
steps_per_launch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
acc = collections.defaultdict(list)
vg = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if kernel in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Grid_Size", "Workgroup_Size"):
                    if k in row and row[k] not in ("", None):
                        acc["_" + k].append(float(row[k]))
out = {"kernel": kernel, "dispatches": max((len(v) for v in acc.values()), default=0)}
if len(sys.argv) > 4:
    import subprocess
    try:
        line = [l for l in open(sys.argv[4]).read().splitlines() if l.startswith("{")][-1]
        bench = json.loads(line)
        out["variant"] = (bench.get("roofline") or {}).get("variant")
    except Exception as e:  # no variant: bench.py will not use this profile
        out["variant_error"] = str(e)
    try:
        out["git_head"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:
        pass
if steps_per_launch:
    out["steps_per_launch"] = steps_per_launch
    # keep the launches of full length: SQ_INSTS_VALU within 10 % of the largest value seen (warm-up / calibration launches are shorter)
    ref = max(acc.get("SQ_INSTS_VALU", [0]))
    keep = [i for i, x in enumerate(acc.get("SQ_INSTS_VALU", [])) if x >= 0.9 * ref]
    if keep and len(keep) < len(acc["SQ_INSTS_VALU"]):
        n = len(acc["SQ_INSTS_VALU"])
        for k in list(acc):
            if len(acc[k]) == n:
                acc[k] = [acc[k][i] for i in keep]
        out["dispatches_full_length"] = len(keep)
for k, v in sorted(acc.items()):
    out[k.lstrip("_")] = sum(v) / len(v)
w = out.get("SQ_WAVES")
if w:
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU"):
        if k in out:
            out[k + "_per_wave"] = out[k] / w
if out.get("SQ_WAVE_CYCLES"):
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in out:
            out[k + "_frac_of_wave_cycles"] = out[k] / out["SQ_WAVE_CYCLES"]
if out.get("SQ_THREAD_CYCLES_VALU") and out.get("SQ_ACTIVE_INST_VALU"):
    out["VALUUtilization_pct"] = 100.0 * out["SQ_THREAD_CYCLES_VALU"] / (out["SQ_ACTIVE_INST_VALU"] * 64.0)
print(json.dumps(out, indent=1))

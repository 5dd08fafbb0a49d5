#!/bin/bash
# One PMC pass (instruction counts, lane cycles, busy cycles) over the dense render scene: tools/pmc_render_a.sh <tag> [run_render.py arguments]
set -e -o pipefail
export TMPDIR=/tmp
tag=${1:-ra}
shift || true
out=$PWD/gpurun_out
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace -d "$out/${tag}_pmc_a" -o p -- python tools/run_render.py --frames 3 "$@" > /dev/null
python - "$out/${tag}_pmc_a/p_counter_collection.csv" <<'PY'
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "stream" not in k: continue
    k = k.replace("void ", "").replace("ca3d::(anonymous namespace)::", "").split("(ca3d")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    c = {x: y / len(n[k]) for x, y in v.items()}
    print(f"{k[:52]:52s} VALU {c['SQ_INSTS_VALU']/1e6:7.2f} M  lanes {c['SQ_THREAD_CYCLES_VALU']/max(1,c['SQ_INSTS_VALU'])/64:.3f}  busy {c['GRBM_GUI_ACTIVE']/8/2.4e3:7.1f} us(@2.4GHz)  waves {c['SQ_WAVES']:.0f}")
PY

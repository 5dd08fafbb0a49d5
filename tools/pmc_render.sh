#!/bin/bash
# PMC passes over the dense render scene alone (tools/run_render.py): tools/pmc_render.sh <tag> [run_render.py arguments]; one counter group per pass.
set -e -o pipefail
export TMPDIR=/tmp
tag=${1:-rr}
shift || true
extra="$@"
out=$PWD/gpurun_out
python tools/run_render.py $extra > $out/${tag}_base.txt
pmc() { rocprofv3 --output-format csv --pmc $2 --kernel-trace -d "$out/${tag}_pmc_$1" -o p -- python tools/run_render.py --frames 3 $extra > /dev/null; }
pmc a "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"
pmc b "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
pmc c "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
pmc d "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"

python - "$out" "$tag" <<'PY'
import csv, collections, sys, json
out, tag = sys.argv[1], sys.argv[2]
res = {}
for p in "abcde":
    import os
    if not os.path.exists(f"{out}/{tag}_pmc_{p}/p_counter_collection.csv"): continue
    rows = list(csv.DictReader(open(f"{out}/{tag}_pmc_{p}/p_counter_collection.csv")))
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]
        if "render" not in k and "stream" not in k: continue
        k = k.replace("void ", "").replace("ca3d::(anonymous namespace)::", "").split("(ca3d")[0].split("(StreamParams")[0].split("(RenderParams")[0].split("(FrameParams")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        res.setdefault(k, {}).update({c: x / len(n[k]) for c, x in v.items()})
print(json.dumps(res, indent=1))
PY

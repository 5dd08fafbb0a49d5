#!/bin/bash
# Exploratory PMC passes over an arbitrary python command (one small counter set per pass).
#   tools/pmc_probe.sh <tag> <kernel-substring> <script.py> [args...]     e.g.  vn512 ca_packed tools/run_ca.py --grid 512 --fused 0
tag=$1; kern=$2; shift 2
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled" "VALUUtilization L2CacheHit" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --output-format csv --pmc $set --kernel-trace -d "$out/p$i" -o p -- python "$@" > "$out/p$i.log" 2>&1 || echo "pass $i ($set) failed"
done
python - "$out" "$kern" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:45s} n={len(v):4d} mean={sum(v)/len(v):.4f}")
PY
find "$out" -name "*.csv" -delete

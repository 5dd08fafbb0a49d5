#!/bin/bash
# Exploratory PMC passes over the packed CA step (one small counter set per pass).
#   tools/pmc_probe.sh <tag> [run_ca.py args...]
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled" "WriteUnitStalled VALUUtilization" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "L2CacheHit LDSBankConflict" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_BUSY_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --output-format csv --pmc $set --kernel-trace -d "$out/p$i" -o p -- python tools/run_ca.py "$@" > "$out/p$i.log" 2>&1 || echo "pass $i ($set) failed"
done
python - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ca_packed" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:45s} n={len(v):4d} mean={sum(v)/len(v):.4f}")
PY
find "$out" -name "*.csv" -delete

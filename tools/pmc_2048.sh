#!/bin/bash
# Fabric traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the per-step kernels at 2048^3: default rule and clustered.
set -e -o pipefail
tag=${1:-r3_x}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-render --no-scaling-base --no-per-step-leg --no-per-call-leg --no-grid-256 --min-seconds 0.05"
pmc() { rocprofv3 --output-format csv --pmc $2 --kernel-trace -d "$out/${tag}_pmc_$1" -o p -- python3 ${@:3} > /dev/null; }
pmc fetch2048 FETCH_SIZE bench.py $Q --grid 2048 --steps 8 --warmup 2
pmc write2048 WRITE_SIZE bench.py $Q --grid 2048 --steps 8 --warmup 2
python3 tools/pmc_reduce.py "ca_packed_vn@2048" ca_packed_vn "$out/${tag}_pmc_fetch2048" "$out/${tag}_pmc_write2048" "$out/${tag}_pmc_traffic.json"
pmc fetch2048cl FETCH_SIZE bench.py $Q --grid 2048 --rule clustered --steps 8 --warmup 2
pmc write2048cl WRITE_SIZE bench.py $Q --grid 2048 --rule clustered --steps 8 --warmup 2
python3 tools/pmc_reduce.py "ca3d_jit_tile@2048" ca3d_jit_ "$out/${tag}_pmc_fetch2048cl" "$out/${tag}_pmc_write2048cl" "$out/${tag}_pmc_traffic.json"

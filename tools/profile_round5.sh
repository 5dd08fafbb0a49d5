#!/bin/bash
# Round-5 evidence on an MI355X box (from the repo root):  tools/profile_round5.sh r5_z [stats|pmc|render|matrix|all]
# Order: the PMC passes FIRST, their reductions copied into profiles/ of this checkout, THEN the bench lines under rocprofv3 --stats — a bench
# line prices a resident kernel against the committed SQ pass of its exact variant, and in round 4 the clustered 512^3 line was taken
# before its pass existed (`frac: null` beside a matching pass: an ordering artefact of the script, not of the match).
# Bench lines, rocprofv3 kernel stats, and PMC passes — one counter group per pass, never combined with --stats
# (MI355X_MICROARCH.md). The SQ passes of the resident kernels record how many steps their launches held AND the kernel variant
# the profiled run reported (roofline.variant = ca3d_get_kernel_variant): bench.py prices a resident kernel's time against
# SQ_INSTS_VALU per step only when the running engine's variant is the profile's.
set -e -o pipefail
tag=${1:-r5_x}
what=${2:-all}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-render --no-scaling-base --no-per-step-leg --no-per-call-leg --no-grid-256 --min-seconds 0.2 --verify-steps 0"
DRV="bench.py --steps 20 --warmup 5"          # the driver's own command
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
stats() { rocprofv3 --output-format csv --kernel-trace --stats -d "$out/${tag}_stats_$1" -o s -- python3 ${@:2} > "$out/${tag}_bench_$1_under_rocprof.json"; cp "$out/${tag}_stats_$1"/*/s_kernel_stats.csv "$out/${tag}_$1_kernel_stats.csv" 2>/dev/null || cp "$out/${tag}_stats_$1"/s_kernel_stats.csv "$out/${tag}_$1_kernel_stats.csv"; echo "stats $1 done"; }
pmc() { rocprofv3 --output-format csv --pmc $2 --kernel-trace -d "$out/${tag}_pmc_$1" -o p -- python3 ${@:3} > "$out/${tag}_pmc_$1_bench.json"; echo "pmc $1 done"; }
if [ "$what" = pmc ] || [ "$what" = all ]; then
  pmc sq512res "$SQ" $DRV $Q
  python3 tools/pmc_sq_reduce.py "$out/${tag}_pmc_sq512res" ca_resident_vn 2060 "$out/${tag}_pmc_sq512res_bench.json" > "$out/${tag}_pmc_sq_resident512.json"
  pmc sq256res "$SQ" $DRV $Q --grid 256
  python3 tools/pmc_sq_reduce.py "$out/${tag}_pmc_sq256res" ca_resident_vn256 2060 "$out/${tag}_pmc_sq256res_bench.json" > "$out/${tag}_pmc_sq_resident256.json"
  pmc sq512cl "$SQ" bench.py $Q --rule clustered --steps 128 --warmup 16 --queue 1024
  python3 tools/pmc_sq_reduce.py "$out/${tag}_pmc_sq512cl" ca3d_jit_resident_class 1024 "$out/${tag}_pmc_sq512cl_bench.json" > "$out/${tag}_pmc_sq_clustered512.json"
  pmc sq1024cl "$SQ" bench.py $Q --grid 1024 --rule clustered --steps 32 --warmup 8
  python3 tools/pmc_sq_reduce.py "$out/${tag}_pmc_sq1024cl" ca3d_jit_ > "$out/${tag}_pmc_sq_clustered1024.json"
  pmc fetch512res FETCH_SIZE $DRV $Q
  pmc write512res WRITE_SIZE $DRV $Q
  python3 tools/pmc_reduce.py "ca_resident_vn@512" ca_resident_vn "$out/${tag}_pmc_fetch512res" "$out/${tag}_pmc_write512res" "$out/${tag}_pmc_traffic.json"
  pmc fetch512 FETCH_SIZE bench.py $Q --resident 0 --steps 256 --warmup 32
  pmc write512 WRITE_SIZE bench.py $Q --resident 0 --steps 256 --warmup 32
  python3 tools/pmc_reduce.py "ca_packed_vn@512" ca_packed_vn "$out/${tag}_pmc_fetch512" "$out/${tag}_pmc_write512" "$out/${tag}_pmc_traffic.json"
  pmc fetch1024 FETCH_SIZE bench.py $Q --grid 1024 --steps 64 --warmup 16
  pmc write1024 WRITE_SIZE bench.py $Q --grid 1024 --steps 64 --warmup 16
  python3 tools/pmc_reduce.py "ca_packed_vn@1024" ca_packed_vn "$out/${tag}_pmc_fetch1024" "$out/${tag}_pmc_write1024" "$out/${tag}_pmc_traffic.json"
  pmc fetch1024cl FETCH_SIZE bench.py $Q --grid 1024 --rule clustered --steps 32 --warmup 8
  pmc write1024cl WRITE_SIZE bench.py $Q --grid 1024 --rule clustered --steps 32 --warmup 8
  python3 tools/pmc_reduce.py "ca3d_jit_tile@1024" ca3d_jit_ "$out/${tag}_pmc_fetch1024cl" "$out/${tag}_pmc_write1024cl" "$out/${tag}_pmc_traffic.json"
  pmc fetch2048 FETCH_SIZE bench.py $Q --grid 2048 --steps 8 --warmup 2
  pmc write2048 WRITE_SIZE bench.py $Q --grid 2048 --steps 8 --warmup 2
  python3 tools/pmc_reduce.py "ca_packed_vn@2048" ca_packed_vn "$out/${tag}_pmc_fetch2048" "$out/${tag}_pmc_write2048" "$out/${tag}_pmc_traffic.json"
  pmc fetch2048cl FETCH_SIZE bench.py $Q --grid 2048 --rule clustered --steps 8 --warmup 2
  pmc write2048cl WRITE_SIZE bench.py $Q --grid 2048 --rule clustered --steps 8 --warmup 2
  python3 tools/pmc_reduce.py "ca3d_jit_tile@2048" ca3d_jit_ "$out/${tag}_pmc_fetch2048cl" "$out/${tag}_pmc_write2048cl" "$out/${tag}_pmc_traffic.json"
  # what the bench lines below (and every later run of this checkout) price themselves against
  cp "$out/${tag}"_pmc_sq_*.json profiles/ 2>/dev/null || true
  python3 - "$out/${tag}_pmc_traffic.json" "$tag" <<'PY'
import json, sys
new = json.load(open(sys.argv[1])); p = "profiles/pmc_traffic.json"
try: d = json.load(open(p))
except Exception: d = {}
for k, v in new.items():
    v["round"] = sys.argv[2]; d[k] = v
json.dump(d, open(p, "w"), indent=1)
PY
fi
if [ "$what" = render ] || [ "$what" = all ]; then
  bash tools/pmc_render.sh ${tag}_render > "$out/${tag}_pmc_render.json" 2>/dev/null
  echo "pmc render done"
  bash tools/pmc_render.sh ${tag}_render_literal --literal 1 > "$out/${tag}_pmc_render_literal.json" 2>/dev/null
  echo "pmc render literal done"
  # the walk launches of a frame IN FLIGHT (a quarter of the chip at 1080p): counter collection runs one dispatch at a time, so the share is forced
  CA3D_STREAM_WGS_PCT=25 bash tools/pmc_render.sh ${tag}_render_share25 > "$out/${tag}_pmc_render_share25.json" 2>/dev/null
  echo "pmc render share25 done"
  rm -rf "$out"/${tag}_render_pmc_* "$out"/${tag}_render_literal_pmc_* "$out"/${tag}_render_share25_pmc_*
  cp "$out/${tag}_pmc_render.json" "$out/${tag}_pmc_render_literal.json" "$out/${tag}_pmc_render_share25.json" profiles/
fi
if [ "$what" = stats ] || [ "$what" = all ]; then
  python3 $DRV > "$out/${tag}_bench512_driver_cmd.json"
  echo "driver command done"
  stats 512_driver_cmd $DRV $Q
  stats 512_perstep bench.py $Q --resident 0 --steps 512 --warmup 64
  stats 256_driver_cmd $DRV $Q --grid 256
  stats 1024 bench.py $Q --grid 1024 --steps 128 --warmup 32
  stats 2048 bench.py $Q --grid 2048 --steps 16 --warmup 4
  stats 512_clustered bench.py $Q --rule clustered --steps 256 --warmup 32
  stats 1024_clustered bench.py $Q --grid 1024 --rule clustered --steps 64 --warmup 16
  stats 992 bench.py $Q --grid 992 --steps 64 --warmup 16
  stats render bench.py --no-cpu-baseline --no-scaling-base --no-per-step-leg --no-per-call-leg --no-grid-256 --min-seconds 0.2 --steps 64 --warmup 8 --verify-steps 0
  python3 bench.py --config 5 --steps 16 --warmup 4 --no-cpu-baseline --min-seconds 0.3 > "$out/${tag}_bench_config5_1gpu.json"
fi
if [ "$what" = matrix ] || [ "$what" = all ]; then
  python3 tools/grid_matrix.py --grids 64,96,128,160,256,288,384,512,640,768,896,992,1024 --out "$out/${tag}_grid_matrix.jsonl" > /dev/null
  echo "grid matrix done"
fi
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*agent_info.csv" -delete
echo done

#!/bin/bash
# Collect the per-round evidence on an MI355X box: bench lines, rocprofv3 kernel stats, and the two PMC passes.
#   tools/profile_round.sh r1_d        (run from the repo root; writes gpurun_out/<tag>_*)
set -e -o pipefail
tag=${1:-rX}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
B512="bench.py --no-cpu-baseline --no-render --steps 512 --warmup 64"
B1024="bench.py --no-cpu-baseline --no-render --grid 1024 --steps 128 --warmup 32"

python bench.py > "$out/${tag}_bench512_default.json"
cat "$out/${tag}_bench512_default.json"

rocprofv3 --output-format csv --kernel-trace --stats -d "$out/${tag}_stats512" -o s -- python $B512 > "$out/${tag}_bench512_under_rocprof.json"
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/${tag}_stats1024" -o s -- python $B1024 > "$out/${tag}_bench1024_under_rocprof.json"
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/${tag}_stats_render" -o s -- python bench.py --no-cpu-baseline --steps 64 --warmup 8 > "$out/${tag}_bench_render_under_rocprof.json"
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/${tag}_stats_clustered" -o s -- python $B512 --rule clustered > "$out/${tag}_bench512_clustered_under_rocprof.json"
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/${tag}_stats_unpacked" -o s -- python tools/run_unpacked.py > "$out/${tag}_unpacked.log"
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/${tag}_stats_slab" -o s -- python tools/run_slab.py --ghost 16 --batches 50 > "$out/${tag}_slab.log"
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/${tag}_stats_slab_phased" -o s -- python tools/run_slab.py --ghost 16 --batches 50 --phased 1 > "$out/${tag}_slab_phased.log"
for g in 256 1024; do python bench.py --no-cpu-baseline --no-render --grid $g --steps 1024 --warmup 128 >> "$out/${tag}_bench_matrix.jsonl"; done
for g in 256 512 1024; do python bench.py --no-cpu-baseline --no-render --grid $g --rule clustered --steps 256 --warmup 64 >> "$out/${tag}_bench_matrix.jsonl"; done
for r in vn_b24_s135 life2d; do python bench.py --no-cpu-baseline --no-render --grid 512 --rule $r --steps 1024 --warmup 128 >> "$out/${tag}_bench_matrix.jsonl"; done

rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$out/${tag}_pmc_fetch512" -o p -- python $B512 > /dev/null
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$out/${tag}_pmc_write512" -o p -- python $B512 > /dev/null
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$out/${tag}_pmc_fetch1024" -o p -- python $B1024 > /dev/null
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$out/${tag}_pmc_write1024" -o p -- python $B1024 > /dev/null
python -c "from cellularautomatons3d_amd import host; host.uniform_block(1920, 1080, host.orbit_camera()).tofile('/tmp/ca3d_u.f32')"
node cellularautomatons3d_amd/js/bench.js --uniforms /tmp/ca3d_u.f32 > "$out/${tag}_bench_node.json"
python tools/run_slab_rccl.py --ghost 32 --batches 40 2>/dev/null | grep "^slab" > "$out/${tag}_slab_rccl_loopback.txt"
python tools/run_slab_rccl.py --grid 2048 --planes 256 --ghost 16 --batches 15 2>/dev/null | grep "^slab" >> "$out/${tag}_slab_rccl_loopback.txt"
python tools/pmc_reduce.py "ca_packed_vn@512" ca_packed_vn "$out/${tag}_pmc_fetch512" "$out/${tag}_pmc_write512" "$out/${tag}_pmc_traffic.json"
python tools/pmc_reduce.py "ca_packed_vn@1024" ca_packed_vn "$out/${tag}_pmc_fetch1024" "$out/${tag}_pmc_write1024" "$out/${tag}_pmc_traffic.json"
# keep only summaries: the raw per-dispatch traces are large
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*agent_info.csv" -delete
echo done

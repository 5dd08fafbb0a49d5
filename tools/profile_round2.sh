#!/bin/bash
# Round-2 evidence on an MI355X box: bench lines, rocprofv3 kernel stats and PMC passes (one counter group per pass, as
# MI355X_MICROARCH.md prescribes; never combined with --stats).   tools/profile_round2.sh r2_a   (from the repo root)
set -e -o pipefail
tag=${1:-r2_x}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-render --no-scaling-base --no-per-step-leg"
DRV="bench.py --steps 20 --warmup 5"          # the driver's own command
python $DRV > "$out/${tag}_bench512_driver_cmd.json"
python bench.py > "$out/${tag}_bench512_default.json"
python bench.py $Q --resident 0 --steps 512 --warmup 64 > "$out/${tag}_bench512_perstep.json"
python $DRV $Q --compare-submission > "$out/${tag}_bench512_driver_cmd_both_submissions.json"
python tools/run_resident.py > "$out/${tag}_resident_batches.txt" 2>/dev/null
python tools/run_resident_class.py > "$out/${tag}_resident_class_batches.txt" 2>/dev/null
tools/ubench/resident_probe_stamps 10000 2 32 > "$out/${tag}_resident_phases.txt" 2>/dev/null || true
tools/ubench/resident_probe_nowait 10000 2 32 >> "$out/${tag}_resident_phases.txt" 2>/dev/null || true
tools/ubench/resident_probe_stamps 10000 2 256 > "$out/${tag}_resident256_phases.txt" 2>/dev/null || true
tools/ubench/resident_probe_nowait 10000 2 256 >> "$out/${tag}_resident256_phases.txt" 2>/dev/null || true
stats() { rocprofv3 --output-format csv --kernel-trace --stats -d "$out/${tag}_stats_$1" -o s -- python ${@:2} > "$out/${tag}_bench_$1_under_rocprof.json"; }
stats 512_driver_cmd $DRV $Q
stats 512_perstep bench.py $Q --resident 0 --steps 512 --warmup 64
stats 1024 bench.py $Q --grid 1024 --steps 128 --warmup 32
stats 512_clustered bench.py $Q --rule clustered --steps 256 --warmup 32
stats 1024_clustered bench.py $Q --grid 1024 --rule clustered --steps 64 --warmup 16
stats render bench.py --no-cpu-baseline --no-scaling-base --steps 64 --warmup 8
for g in 256 1024; do python bench.py $Q --grid $g --steps 1024 --warmup 128 >> "$out/${tag}_bench_matrix.jsonl"; done
for g in 256 512 1024; do python bench.py $Q --grid $g --rule clustered --steps 256 --warmup 64 >> "$out/${tag}_bench_matrix.jsonl"; done
for r in vn_b24_s135 life2d; do python bench.py $Q --grid 512 --rule $r --steps 1024 --warmup 128 >> "$out/${tag}_bench_matrix.jsonl"; done
python bench.py --config 5 --steps 16 --warmup 4 --no-cpu-baseline > "$out/${tag}_bench_config5_1gpu.json"
pmc() { rocprofv3 --output-format csv --pmc $2 --kernel-trace -d "$out/${tag}_pmc_$1" -o p -- python ${@:3} > /dev/null; }
pmc fetch512res FETCH_SIZE $DRV $Q
pmc write512res WRITE_SIZE $DRV $Q
pmc fetch512 FETCH_SIZE bench.py $Q --resident 0 --steps 256 --warmup 32
pmc write512 WRITE_SIZE bench.py $Q --resident 0 --steps 256 --warmup 32
pmc fetch512cl FETCH_SIZE bench.py $Q --rule clustered --steps 128 --warmup 16
pmc write512cl WRITE_SIZE bench.py $Q --rule clustered --steps 128 --warmup 16
pmc sq512cl "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" bench.py $Q --rule clustered --steps 128 --warmup 16
pmc sq512res "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" $DRV $Q
pmc sqrender "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" bench.py --no-cpu-baseline --no-scaling-base --steps 64 --warmup 8
python tools/pmc_reduce.py "ca_resident_vn@512" ca_resident_vn "$out/${tag}_pmc_fetch512res" "$out/${tag}_pmc_write512res" "$out/${tag}_pmc_traffic.json"
python tools/pmc_reduce.py "ca_packed_vn@512" ca_packed_vn "$out/${tag}_pmc_fetch512" "$out/${tag}_pmc_write512" "$out/${tag}_pmc_traffic.json"
python tools/pmc_reduce.py "ca_resident_class@512" ca3d_jit_resident_class "$out/${tag}_pmc_fetch512cl" "$out/${tag}_pmc_write512cl" "$out/${tag}_pmc_traffic.json"
python tools/pmc_sq_reduce.py "$out/${tag}_pmc_sq512cl" ca3d_jit_resident_class > "$out/${tag}_pmc_sq_clustered512.json"
pmc sq1024cl "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" bench.py $Q --grid 1024 --rule clustered --steps 32 --warmup 8
python tools/pmc_sq_reduce.py "$out/${tag}_pmc_sq1024cl" ca3d_jit_roll > "$out/${tag}_pmc_sq_clustered1024.json"
python tools/pmc_sq_reduce.py "$out/${tag}_pmc_sq512res" ca_resident_vn > "$out/${tag}_pmc_sq_resident512.json"
python tools/pmc_sq_reduce.py "$out/${tag}_pmc_sqrender" ca_render_packed_sched > "$out/${tag}_pmc_sq_render.json"
python -c "from cellularautomatons3d_amd import host; host.uniform_block(1920, 1080, host.orbit_camera()).tofile('/tmp/ca3d_u.f32')"
node cellularautomatons3d_amd/js/bench.js --uniforms /tmp/ca3d_u.f32 > "$out/${tag}_bench_node.json"
for n in 0 1; do python tools/run_slab_rccl.py --ghost 32 --batches 40 --native $n --resident 0 2>/dev/null | grep "^slab"; done > "$out/${tag}_slab_rccl_loopback.txt"
for k in 16 32; do python tools/run_slab_rccl.py --ghost $k --batches 40 --native 1 --resident 1 2>/dev/null | grep "^slab"; done >> "$out/${tag}_slab_rccl_loopback.txt"
python tools/run_slab_rccl.py --grid 2048 --planes 256 --ghost 16 --batches 15 --native 1 --rule clustered 2>/dev/null | grep "^slab" >> "$out/${tag}_slab_rccl_loopback.txt"
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*agent_info.csv" -delete
bash tools/pmc_render.sh ${tag}_render > "$out/${tag}_pmc_render.json" 2>/dev/null
CA3D_RENDER_TRACE=/tmp/rt.bin python tools/run_render.py --frames 3 > /dev/null 2>&1 && python tools/render_trace.py /tmp/rt.bin > "$out/${tag}_render_trace.txt"
rm -rf "$out"/${tag}_render_pmc_*
echo done

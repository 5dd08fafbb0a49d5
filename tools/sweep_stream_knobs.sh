#!/bin/bash
# Tuning sweep of the ray-stream walk passes' run-time knobs on the bench's dense scene (tools/run_render.py): the idle-lane count at which
# a wave pops prepared rays (CA3D_STREAM_POP, default 16), the pixel block of a chunk (CA3D_STREAM_LB, default 3 at 4 spp), the batched
# loop in the tail (CA3D_STREAM_TAIL_BATCH), one frame at a time and two in flight.   tools/sweep_stream_knobs.sh > out.txt
run() { echo -n "$1 | "; env $1 python3 tools/run_render.py --frames 200 --size $2 --option render_pipeline=$3 | sed 's/, primary.*//'; }
for size in 1920x1080 3840x2160; do
  for pipe in 0 1; do
    echo "== $size, render_pipeline $pipe"
    run "CA3D_X=0" $size $pipe
    for pop in 4 8 12 24 32 48; do run "CA3D_STREAM_POP=$pop" $size $pipe; done
    for lb in 2 4; do run "CA3D_STREAM_LB=$lb" $size $pipe; done
    run "CA3D_STREAM_TAIL_BATCH=0" $size $pipe
    run "CA3D_STREAM_QMAP=0" $size $pipe
  done
done

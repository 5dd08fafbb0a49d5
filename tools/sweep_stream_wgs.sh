#!/bin/bash
# Tuning: the share of the chip's wave slots one walk launch of the ray-stream passes asks for (CA3D_STREAM_WGS_PCT) against the number of
# converged frames in flight (render_pipeline), by frame size — tools/run_render.py on the bench's dense scene.
for size in 1920x1080 2560x1440 3840x2160; do
  for n in 3 4; do
    for pct in 100 67 50 34 25 17; do
      echo -n "$size lanes $n pct $pct: "; CA3D_STREAM_WGS_PCT=$pct python3 tools/run_render.py --frames 200 --size $size --option render_pipeline=$n 2>/dev/null | sed 's/, primary.*//' | sed 's/.*sched 1: //'
    done
  done
done
for pct in 100 50 34 25; do echo -n "1920x1080 1spp lanes 3 pct $pct: "; CA3D_STREAM_WGS_PCT=$pct python3 tools/run_render.py --frames 200 --spp 1 --option render_pipeline=3 2>/dev/null | sed 's/, primary.*//' | sed 's/.*sched 1: //'; done

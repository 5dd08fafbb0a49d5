for size in 1920x1080 3840x2160; do for n in 0 2 3 4; do echo -n "lanes $n: "; python3 tools/run_render.py --frames 300 --size $size --option render_pipeline=$n | sed 's/, primary.*//'; done; done
echo -n "1spp lanes 2: "; python3 tools/run_render.py --frames 300 --spp 1 --option render_pipeline=2 | sed 's/, primary.*//'
echo -n "1spp lanes 3: "; python3 tools/run_render.py --frames 300 --spp 1 --option render_pipeline=3 | sed 's/, primary.*//'
echo -n "1spp lanes 4: "; python3 tools/run_render.py --frames 300 --spp 1 --option render_pipeline=4 | sed 's/, primary.*//'

python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "sparse or skipping or evolved or small_live or power_of_two or random_volume or rectangle or inside" > gpurun_out/s3_skip_tests.log 2>&1; rc=$?; tail -15 gpurun_out/s3_skip_tests.log | cut -c1-300
python - <<EOP
import sys,time
sys.path.insert(0,".")
from cellularautomatons3d_amd import Engine, host
G,W,H=512,1920,1080
for p in (0,1):
    e=Engine(0); e.configure(G); e.set_rule_strings(); e.set_option("render_pipeline",p)
    for rounds in (12,14):
        e.upload_state(host.random_fill(host.words_per_buffer(G), seed=0xCA3D0001, and_rounds=rounds))
        u=host.uniform_block(W,H,host.orbit_camera())
        for _ in range(8): e.render(u,W,H,4,readback=False)
        e.synchronize(); t0=time.perf_counter()
        for _ in range(50): e.render(u,W,H,4,readback=False)
        e.synchronize(); st=e.render_stats(); print("pipeline %d density 2^-%d: %.3f ms per frame, visits/primary %.2f shadow rays %d"%(p,rounds+1,(time.perf_counter()-t0)/50*1e3, st.primary_cell_visits/max(1,st.primary_rays), st.shadow_rays), flush=True)
    e.close()
EOP

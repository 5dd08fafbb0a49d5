set -e -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out
python tools/run_render.py > $out/rr_base.txt
rocprofv3 -L > $out/rr_counters.txt 2>&1 || true
pmc() { rocprofv3 --output-format csv --pmc $2 --kernel-trace -d "$out/rr_pmc_$1" -o p -- python tools/run_render.py --frames 3 > /dev/null; }
pmc a "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"
pmc b "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS"
pmc c "SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACCUM_PREV_HIRES" || true
echo ok

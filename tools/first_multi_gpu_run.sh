#!/bin/bash
# What to run FIRST on a node with more than one MI355X (no round of this build has had one: every multi-GPU figure in the repo is a
# one-GPU share or a projection). In this order, each step's output says what the next may assume:
#   tools/first_multi_gpu_run.sh [N]          (N = GPUs to use, default: all visible, at most 8)
set -e -o pipefail
cd "$(dirname "$0")/.."
N=${1:-$(python3 -c "import torch; print(min(8, torch.cuda.device_count()))")}
echo "== 1. the group handle across real devices (peer copies over xGMI), parity against the oracle"
python3 -m pytest tests/test_gpu_slab.py -q -m gpu -k "across_real_devices" -s
echo "== 2. one process per GPU, engine-issued RCCL, state verified against the oracle before timing; both schedules; rccl.distinct_devices must be $N"
python3 bench.py --gpus "$N" --steps 32 --warmup 32 | tee /tmp/ca3d_first_n${N}.json | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus')}, 'verified', d['verified']['oracle_match'], 'distinct devices', d['rccl']['distinct_devices'], 'schedules', {k: v.get('ms_per_step') if isinstance(v, dict) else v for k, v in d['schedules'].items()})
assert d['verified']['oracle_match'] and d['rccl']['distinct_devices'] == d['n_gpus']"
echo "== 3. the same split driven by ONE JavaScript thread (EngineGroup), state compared with a single grid"
node cellularautomatons3d_amd/js/bench.js --gpus "$N" --check 40
echo "== 4. BASELINE configs[4]: 2048^3 clustered, overlapped halo, 3840x2160 frame shared by the ranks"
python3 bench.py --gpus "$N" --config 5 --steps 16 --warmup 8
echo "== 5. the scaling curve the driver computes: N = 1, 2, 4, 8 back to back"
for n in 1 2 4 8; do [ "$n" -le "$N" ] && python3 bench.py --gpus $n --grid 1024 --steps 32 --warmup 32 --no-render --no-cpu-baseline | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['n_gpus'], 'GPUs', d['value'], 'Gcells/s', d['ms_per_step'], 'ms per step')"; done

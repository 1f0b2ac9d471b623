#!/bin/bash
# training-step leg alone under rocprofv3 --kernel-trace --stats (round 2)
set -e
mkdir -p gpurun_out/r2/train_prof
F="--steps 1 --warmup 0 --no-cpu --no-table --no-scale-parity --no-small-ops --no-collate --no-kernel-profile"
python bench.py $F > gpurun_out/r2/train_bench.json 2> gpurun_out/r2/train_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/train_bench.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('train_mode'), indent=1))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/train_prof -o train -- python $GRAFT_REPO_ROOT/bench.py $F > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r2/train_prof/err.log

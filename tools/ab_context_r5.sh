#!/bin/bash
# Does the drop-in leg's B = 8 figure depend on which legs ran before it in the same process?  (development aid, round 5)
set -u
mkdir -p gpurun_out/r5
F="--gpus 1 --steps 2 --warmup 1 --no-table --no-small-ops --no-collate --no-kernel-profile --no-parity-grade --no-scale-parity"
show() { python -c "import sys,json; j=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', j['legs'].get('dropin_ms_per_step'), j['legs'].get('train_ms_per_step'))"; }
python bench.py $F --no-cpu --no-train --full-json gpurun_out/r5/ctx_a.json 2>/dev/null | show "dropin alone      "
python bench.py $F --no-cpu            --full-json gpurun_out/r5/ctx_b.json 2>/dev/null | show "train + dropin    "
python bench.py $F --no-train          --full-json gpurun_out/r5/ctx_c.json 2>/dev/null | show "cpu + dropin      "
python bench.py $F                     --full-json gpurun_out/r5/ctx_d.json 2>/dev/null | show "cpu+train+dropin  "
python bench.py $F --no-cpu --no-train --full-json gpurun_out/r5/ctx_e.json 2>/dev/null | show "dropin alone again"

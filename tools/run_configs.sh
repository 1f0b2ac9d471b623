#!/bin/bash
# un-profiled bench lines of configs 2-4 on one box (profiles/r3_final/bench_config{2,3,4}.json)
mkdir -p gpurun_out/r3
for c in 2 3 4; do
  python bench.py --config $c --steps 3 --no-collate --no-small-ops > gpurun_out/r3/bench_config$c.json 2> gpurun_out/r3/bench_config$c.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3/bench_config$c.json").read().strip().splitlines()[-1])
print($c, round(d["value"]), d["dtype"], round(d["encoder_mfma_frac"],3), "bf16", round(d["bf16_mode"]["value"]), round(d["bf16_mode"]["encoder_mfma_frac"],3), "cpu", round(d["cpu_baseline"]["value"],1))
PY
done

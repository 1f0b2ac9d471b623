#!/bin/bash
# un-profiled bench lines of BASELINE configs 1-4 on one box -> gpurun_out/r4/bench_config{1..4}.json (copied to profiles/r4_final/)
set -u
mkdir -p gpurun_out/r4
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_config1.json 2> gpurun_out/r4/bench_config1.err || echo "config 1 failed"
for c in 2 3 4; do
  timeout -k 10 500 python bench.py --config $c --steps 3 --no-collate --no-small-ops > gpurun_out/r4/bench_config$c.json 2> gpurun_out/r4/bench_config$c.err || echo "config $c failed"
done
python - <<PY
import json
for c in (1, 2, 3, 4):
    try:
        d = json.loads(open(f"gpurun_out/r4/bench_config{c}.json").read().strip().splitlines()[-1])
        print(c, round(d["value"]), d["dtype"], round(d["encoder_mfma_frac"], 3), "bf16", round(d["bf16_mode"]["value"]), round(d["bf16_mode"]["encoder_mfma_frac"], 3),
              "cpu", round(d["cpu_baseline"]["value"], 1), "parity", {k: v["top10_identical"] for k, v in d["config"]["parity_vs_oracle"].items() if isinstance(v, dict) and "top10_identical" in v})
    except Exception as e:
        print(c, "unreadable:", e)
PY

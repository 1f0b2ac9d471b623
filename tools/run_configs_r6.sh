#!/bin/bash
# un-profiled bench runs of BASELINE configs 1-4 on one box: the compact line (what the driver parses) -> gpurun_out/r6/bench_config{c}.line.json,
# the complete record -> gpurun_out/r6/bench_config{c}.json (both copied to profiles/r6_final/)
set -u
mkdir -p gpurun_out/r6
timeout -k 10 700 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/r6/bench_config1.json > gpurun_out/r6/bench_config1.line.json 2> gpurun_out/r6/bench_config1.err || echo "config 1 failed"
for c in 2 3 4; do
  timeout -k 10 600 python bench.py --config $c --steps 3 --no-collate --no-small-ops --full-json gpurun_out/r6/bench_config$c.json > gpurun_out/r6/bench_config$c.line.json 2> gpurun_out/r6/bench_config$c.err || echo "config $c failed"
done
python - <<PY
import json
for c in (1, 2, 3, 4):
    try:
        raw = open(f"gpurun_out/r6/bench_config{c}.line.json").read().strip().splitlines()[-1]
        d = json.loads(raw)
        pg = d["config"].get("parity_grade") or {}
        print(c, "line bytes", len(raw), "value", round(d["value"]), d["dtype"], "frac", round(d["roofline"]["frac"], 3), "enc", d["roofline"]["encoder_mfma_frac"],
              "cpu", round(d["cpu_baseline"]["value"], 1), "parity-grade", pg.get("dtype"), pg.get("candidates_per_s"), pg.get("top10_identical"), "/", pg.get("of_impressions"),
              "parity", {k: v["top10_identical"] for k, v in d["config"]["parity_vs_oracle"].items() if isinstance(v, dict) and "top10_identical" in v})
    except Exception as e:
        print(c, "unreadable:", e)
PY

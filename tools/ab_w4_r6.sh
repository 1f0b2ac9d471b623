#!/bin/bash
# Round 6: same-box A/B of the headline (bench.py timed region only) with the 4-wave GEMM (default) against the 8-wave GEMM
# (MANNER_HIP_GEMM_W4=0), alternating, f16 (the line's value) with its bf16 repeat.  Output: gpurun_out/r6/ab_w4_*.json
set -u
O=gpurun_out/r6
mkdir -p "$O"
B="python3 bench.py --steps 20 --warmup 5 --no-cpu --no-table --no-collate --no-small-ops --no-train --no-dropin --no-scale-parity --no-parity-grade"
for rep in 1 2; do
  for w4 in 8 0; do
    MANNER_HIP_GEMM_ASM=$w4 timeout -k 10 400 $B --full-json "$O/ab_w4_${w4}_rep${rep}.full.json" > "$O/ab_w4_${w4}_rep${rep}.line.json" 2> "$O/ab_w4_${w4}_rep${rep}.err" || { echo "bench w4=$w4 rep $rep FAILED"; tail -5 "$O/ab_w4_${w4}_rep${rep}.err"; exit 1; }
    python3 - "$O/ab_w4_${w4}_rep${rep}.line.json" "$w4" "$rep" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = l.get("roofline", {})
print(f"W4={sys.argv[2]} rep {sys.argv[3]}: {l['value']:.0f} cand/s  {l['ms_per_step']:.2f} ms/step  frac {r.get('frac')}  encoder_mfma_frac {r.get('encoder_mfma_frac')}  kernel_avg_us {l.get('legs', {}).get('kernel_avg_us')}")
PY
  done
done

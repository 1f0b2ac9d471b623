#!/bin/bash
# Round 6: same-box A/B of the headline (bench.py timed region only) between two settings of ONE environment switch, alternating:
#   bash tools/ab_env_r6.sh MANNER_HIP_XCD_RANGES 1 0 [reps]      ("-" = leave the variable unset)
# f16 (the line's value) with its bf16 repeat.  Output: gpurun_out/r6/ab_<VAR>_<value>_rep<n>.{line,full}.json
set -u
VAR="$1"; A="$2"; BV="$3"; REPS="${4:-2}"
O=gpurun_out/r6
mkdir -p "$O"
B="python3 bench.py --steps 20 --warmup 5 --no-cpu --no-table --no-collate --no-small-ops --no-train --no-dropin --no-scale-parity --no-parity-grade"
for rep in $(seq 1 "$REPS"); do
  for v in "$A" "$BV"; do
    F="$O/ab_${VAR}_${v}_rep${rep}"
    if [ "$v" = "-" ]; then unset "$VAR"; else export "$VAR=$v"; fi
    timeout -k 10 400 $B --full-json "$F.full.json" > "$F.line.json" 2> "$F.err" || { echo "bench $VAR=$v rep $rep FAILED"; tail -5 "$F.err"; exit 1; }
    python3 - "$F.line.json" "$VAR=$v" "$rep" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = l.get("roofline", {})
print(f"{sys.argv[2]} rep {sys.argv[3]}: {l['value']:.0f} cand/s  {l['ms_per_step']:.2f} ms/step  frac {r.get('frac')}  encoder_mfma_frac {r.get('encoder_mfma_frac')}  kernel_avg_us {l.get('legs', {}).get('kernel_avg_us')}")
PY
  done
done

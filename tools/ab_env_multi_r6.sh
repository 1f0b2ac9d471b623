#!/bin/bash
# Round 6: same-box comparison of the headline across SEVERAL values of one environment switch, round-robin:
#   bash tools/ab_env_multi_r6.sh MANNER_HIP_COL_GROUP 2 - 0 4 3      (reps, then the values; "-" = unset)
set -u
VAR="$1"; REPS="$2"; shift 2
O=gpurun_out/r6
mkdir -p "$O"
B="python3 bench.py --steps 20 --warmup 5 --no-cpu --no-table --no-collate --no-small-ops --no-train --no-dropin --no-scale-parity --no-parity-grade"
for rep in $(seq 1 "$REPS"); do
  for v in "$@"; do
    F="$O/abm_${VAR}_${v}_rep${rep}"
    if [ "$v" = "-" ]; then unset "$VAR"; else export "$VAR=$v"; fi
    timeout -k 10 400 $B --full-json "$F.full.json" > "$F.line.json" 2> "$F.err" || { echo "bench $VAR=$v rep $rep FAILED"; tail -5 "$F.err"; exit 1; }
    python3 - "$F.line.json" "$VAR=$v" "$rep" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = l.get("roofline", {})
k = l.get("legs", {}).get("kernel_avg_us", {})
print(f"{sys.argv[2]} rep {sys.argv[3]}: {l['value']:.0f} cand/s  {l['ms_per_step']:.2f} ms/step  frac {r.get('frac')}  enc {r.get('encoder_mfma_frac')}  qkv {k.get('gemm_qkv'):.1f} attn {k.get('attention'):.1f} out {k.get('gemm_out'):.1f} ffn1 {k.get('gemm_ffn1'):.1f} ffn2 {k.get('gemm_ffn2'):.1f}")
PY
  done
done

#!/bin/bash
# the headline's per-round drift (45.9 -> 44.4 -> 44.3 -> 42.2 -> 42.5 k cand/s on five different driver boxes) settled by ONE
# lease: the round-final builds of rounds 1, 3, 5 (git worktrees under _ab/, each with its own lib/) and HEAD (with the hand-scheduled GEMM and
# with MANNER_HIP_GEMM_ASM=0), alternating, timed region only, bf16 (the one arithmetic round 1 had) and f16.
set -u
O=gpurun_out/r6
mkdir -p "$O"
OUT="$O/headline_ab.txt"
: > "$OUT"
run() {  # tag dir precision env...
  local tag=$1 dir=$2 prec=$3; shift 3
  local flags="--steps 20 --warmup 5 --precision $prec --no-cpu --no-table --no-collate --no-kernel-profile"
  case $tag in r1) ;; r3) flags="$flags --no-dropin --no-scale-parity --no-small-ops --no-train";; *) flags="$flags --no-dropin --no-scale-parity --no-small-ops --no-train --no-parity-grade";; esac
  ( cd "$dir" && env "$@" timeout -k 10 300 python3 bench.py $flags 2> /dev/null | tail -1 ) > "$O/hab_line.json" || { echo "$tag $prec FAILED" | tee -a "$OUT"; return; }
  python3 - "$O/hab_line.json" "$tag" "$prec" >> "$OUT" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
extra = l.get("bf16_mode") or {}
print(f"{sys.argv[2]:>9} {sys.argv[3]:>5}: {l['value']:9.0f} cand/s  {l['ms_per_step']:8.2f} ms/step" + (f"   (bf16 repeat in the same run: {extra.get('candidates_per_s', extra.get('value', 0)):.0f})" if extra else ""))
PY
  tail -1 "$OUT"
}
for rep in 1 2; do
  echo "--- pass $rep, bf16" | tee -a "$OUT"
  run r1 _ab/r1 bf16 X=1
  run r3 _ab/r3 bf16 X=1
  run r5 _ab/r5 bf16 X=1
  run head-asm8 . bf16 MANNER_HIP_GEMM_ASM=8
  run head-asm0 . bf16 MANNER_HIP_GEMM_ASM=0
  echo "--- pass $rep, f16" | tee -a "$OUT"
  run r3 _ab/r3 f16 X=1
  run r5 _ab/r5 f16 X=1
  run head-asm8 . f16 MANNER_HIP_GEMM_ASM=8
  run head-asm0 . f16 MANNER_HIP_GEMM_ASM=0
done

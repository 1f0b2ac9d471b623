#!/bin/bash
# Round 5: the training step of the reference's default configuration (embeddings trainable) ALONE under rocprofv3
# --kernel-trace --stats: 2 warm-up + 10 timed steps with a synchronisation each + the same 10 steps back to back
# (tools/train_probe.py -> bench.train_leg); per-step kernel time = totals / 22.
set -eu
OUT="${1:-gpurun_out/r5/train_prof}"
mkdir -p "$OUT"
export TRAIN_PROBE_VARIANT=reference_default_embeddings_trainable TRAIN_PROBE_STEPS=10
python tools/train_probe.py bf16 > "$OUT/probe.json" 2> "$OUT/probe.err"
cat "$OUT/probe.json"
ROOT="$(pwd)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT" -o train -- python3 "$ROOT/tools/train_probe.py" bf16 > "$ROOT/$OUT/prof_probe.json" 2> "$ROOT/$OUT/err.log"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel ms {tot / 1e6:.2f} over 22 steps = {tot / 22e6:.2f} ms per step")
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs']) / 22e6:8.3f} ms/step {int(r['Calls']) / 22:7.1f} calls/step {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:150]}")
PY

#!/bin/bash
# kernel timeline of the headline step (two streams): union busy time vs wall; development aid, round 6
set -u
mkdir -p gpurun_out/r6/tlh
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/r6/tlh" -o headline -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-kernel-profile --no-table --no-collate --no-small-ops --no-train --no-dropin --no-parity-grade --no-scale-parity > "$R/gpurun_out/r6/tlh/run.log" 2>&1
cd "$R"
f=$(find gpurun_out/r6/tlh -name '*kernel_trace.csv' | head -1)
python3 tools/kernel_timeline.py "$f" "${1:-3000}" > gpurun_out/r6/timeline_headline.txt 2>&1
rm -rf gpurun_out/r6/tlh
head -30 gpurun_out/r6/timeline_headline.txt

#!/bin/bash
# kernel timeline of the B = 8 drop-in eval step (busy vs idle); development aid, round 6
set -u
mkdir -p gpurun_out/r6/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6/tl -o dropin -- python3 tools/dropin_probe.py "${1:-8:eval}" > gpurun_out/r6/tl/run.log 2>&1
f=$(find gpurun_out/r6/tl -name '*kernel_trace.csv' | head -1)
python3 tools/kernel_timeline.py "$f" "${2:-400}" > gpurun_out/r6/timeline_dropin.txt 2>&1
rm -rf gpurun_out/r6/tl
cat gpurun_out/r6/timeline_dropin.txt

#!/bin/bash
# Round 3: one leg of bench.py alone under rocprofv3 --kernel-trace --stats (nothing of the headline steps in the trace):
#   bash tools/profile_leg_r3.sh train [bf16]        the training-step leg (2 variants x (2 warm-up + 5 timed) steps)
#   bash tools/profile_leg_r3.sh dropin 8:train      one case of the drop-in leg (2 warm-up + 8 timed steps)
# prints the total kernel time and the top kernels; raw output under gpurun_out/r3/leg_prof.
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
LEG=${1:-train}
ARG=${2:-}
O="$GRAFT_REPO_ROOT/gpurun_out/r3/leg_prof"
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o leg -- python3 "$GRAFT_REPO_ROOT/tools/${LEG}_probe.py" $ARG > "$O/leg.json" 2> "$O/err.log"
find "$O" -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 3), "calls", sum(int(r["Calls"]) for r in rows))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms {int(r["Calls"]):6d} calls {float(r["TotalDurationNs"])/int(r["Calls"])/1e3:8.1f} us  {r["Name"][:120]}')
print(open("$O/leg.json").read()[-1200:])
PY

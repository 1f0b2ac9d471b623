#!/bin/bash
# Round 4: SQ counters of the one-pass pooler (tools/pool_probe.py) — separate --pmc passes, --kernel-trace only.
set -eu
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/r4/pool_pmc"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/tools/pool_probe.py" > "$O/probe.json" 2> "$O/stats.log" || true
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo "$set" | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/$tag" -- python3 "$R/tools/pool_probe.py" > /dev/null 2> "$O/$tag.log" || echo "pass $tag failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "pool" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pool" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
find "$O" -name "*kernel_trace.csv" -size +2M -delete

#!/bin/bash
# Round 4: the 256x384-tile A/B on one box — tools/tile_lab (main loops of both tile shapes at equal pipeline
# structure), then the production kernel's own ablations (tools/gemm_lab: abl1 = main loop only) for reference, then counters of the
# tile lab in separate --pmc passes.  Raw output under gpurun_out/r4/tile_lab.
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/r4/tile_lab"
mkdir -p "$O"
cd "$R"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imanner_amd/csrc tools/tile_lab.hip -o "$O/tile_lab"
timeout -k 10 120 "$O/tile_lab" 65536 30 | tee "$O/tile_lab.txt"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo "$set" | tr ' ' '_' | cut -c1-30)
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/$tag" -- "$O/tile_lab" 65536 4 > /dev/null 2> "$O/$tag.log" || echo "pass $tag failed"
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("$O/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            # launches alternate K = 768 / K = 3072 per kernel: report both halves (first 3 launches of a shape are check + warm-up)
            out.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v), "min": min(v), "max": max(v)}
print(json.dumps(out, indent=1))
open("$O/tile_lab_counters.json", "w").write(json.dumps(out, indent=1))
PY
find "$O" -name "*kernel_trace.csv" -size +2M -delete

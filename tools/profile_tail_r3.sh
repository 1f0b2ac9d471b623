#!/bin/bash
# Round 3, final build: the tail kernels at evaluation scale (tools/tail_probe.py: scorer on the 495 MB MIND-large table and on its
# centred half copy, pooler, dot, z-score, to_dense) — durations, then FETCH_SIZE and WRITE_SIZE in separate passes.  Raw output
# under gpurun_out/prof_r3 (what tools/collect_r3.py reads); every rocprofv3 under its own timeout.
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O="$R/gpurun_out/prof_r3"
mkdir -p "$O" && rm -rf "$O"/tail_stats "$O"/tail_fetch "$O"/tail_write      # only this script's own outputs: profile_r3.sh fills the same directory
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tail_stats -- python3 $R/tools/tail_probe.py > $O/tail_probe.json 2> $O/tail_stats.log
echo "stats done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tail_fetch -- python3 $R/tools/tail_probe.py > /dev/null 2> $O/tail_fetch.log
echo "fetch done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tail_write -- python3 $R/tools/tail_probe.py > /dev/null 2> $O/tail_write.log
echo "write done"
find $O -name "*kernel_trace.csv" -size +4M -delete
du -sh $O

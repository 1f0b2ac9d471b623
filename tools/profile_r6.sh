#!/bin/bash
# Round-6 profiling session on the GPU box (bash tools/profile_r6.sh through gpurun, from the repo root); raw output under
# gpurun_out/prof_r6, summaries are copied into profiles/r6_final by `python tools/collect_r3.py gpurun_out/prof_r6 profiles/r6_final`.
# Every rocprofv3 runs the program directly after `--` (no env / shell wrapper), counters in their own passes with --kernel-trace only.
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/prof_r6"
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export MANNER_HIP_STREAMS=1        # single stream: launch durations are not inflated by the second stream
B="python3 $R/bench.py --warmup 1 --no-cpu --no-kernel-profile --no-table --no-collate --no-small-ops --no-train --no-dropin --no-parity-grade"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- $B --steps 3 > "$O/stats.log" 2>&1; echo "stats done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- $B --steps 1 > "$O/fetch.log" 2>&1; echo "fetch done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -- $B --steps 1 > "$O/write.log" 2>&1; echo "write done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$O/sq" -- $B --steps 1 > "$O/sq.log" 2>&1; echo "sq done"
unset MANNER_HIP_STREAMS
# the tail kernels at evaluation scale: scorer on the 495 MB table, one-pass and strict pooler, dot, z-score, to_dense, phase C in one launch vs separate
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tail_stats" -- python3 "$R/tools/tail_probe.py" > "$O/tail_probe.json" 2> "$O/tail_stats.log"; echo "tail stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/tail_fetch" -- python3 "$R/tools/tail_probe.py" > /dev/null 2> "$O/tail_fetch.log"; echo "tail fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/tail_write" -- python3 "$R/tools/tail_probe.py" > /dev/null 2> "$O/tail_write.log"; echo "tail write done"
# the training step of the reference's default configuration alone (tools/train_probe.py: 2 warm-up + 10 + 10 timed steps; both streams, as it runs)
export TRAIN_PROBE_VARIANT=reference_default_embeddings_trainable TRAIN_PROBE_STEPS=10
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train" -- python3 "$R/tools/train_probe.py" bf16 > "$O/train_probe.json" 2> "$O/train.log"; echo "train done"
unset TRAIN_PROBE_VARIANT TRAIN_PROBE_STEPS
find "$O" -name "*kernel_trace.csv" -size +4M -delete
du -sh "$O"

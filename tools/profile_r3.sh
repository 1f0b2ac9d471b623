#!/bin/bash
# Round-3 profiling session on the GPU box (run through gpurun from the repo root); raw output under gpurun_out/prof_r3,
# summaries are copied into profiles/r3_final by tools/collect_r3.py afterwards.
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_r3
rm -rf $O && mkdir -p $O
B="python3 bench.py --warmup 1 --no-cpu --no-kernel-profile --no-table --no-collate --no-small-ops --no-train --no-dropin"
# 1. per-kernel durations of the headline steps, single stream so that launch durations are not inflated by the second stream
MANNER_HIP_STREAMS=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 3 > $O/stats.log 2>&1
echo "stats done"
# 2. HBM-side traffic, separate passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
MANNER_HIP_STREAMS=1 timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 1 > $O/fetch.log 2>&1
echo "fetch done"
MANNER_HIP_STREAMS=1 timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 1 > $O/write.log 2>&1
echo "write done"
# 3. SQ counters of the same command (8 SQ slots)
MANNER_HIP_STREAMS=1 timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- $B --steps 1 > $O/sq.log 2>&1
echo "sq done"
# 4. the tail kernels at evaluation scale (scorer on the 495 MB MIND-large table and on its centred half copy, pooler, dot, z-score, to_dense)
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tail_stats -- python3 tools/tail_probe.py > $O/tail_probe.json 2> $O/tail_stats.log
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tail_fetch -- python3 tools/tail_probe.py > /dev/null 2> $O/tail_fetch.log
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tail_write -- python3 tools/tail_probe.py > /dev/null 2> $O/tail_write.log
echo "tail done"
# 5. the training-step leg alone, ONE stream (no inference kernels measured under a concurrent stream)
T="python3 bench.py --steps 1 --warmup 0 --no-cpu --no-table --no-scale-parity --no-small-ops --no-collate --no-kernel-profile --no-dropin"
MANNER_HIP_STREAMS=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- $T > $O/train.log 2>&1
echo "train done"
# keep only what the collectors need: the big per-dispatch CSVs stay on the box except the counter / stats tables
find $O -name "*kernel_trace.csv" -size +4M -delete
du -sh $O

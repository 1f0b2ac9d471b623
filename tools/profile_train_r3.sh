#!/bin/bash
# Round 3: the training-step leg alone under rocprofv3 --kernel-trace --stats, ONE stream (MANNER_HIP_STREAMS=1 so that the
# inference kernels of the frozen-prefix variant are not measured under a concurrent stream — VERDICT r2 weak #11).
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r3/train_prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
F="--steps 1 --warmup 0 --no-cpu --no-table --no-scale-parity --no-small-ops --no-collate --no-kernel-profile --no-dropin"
MANNER_HIP_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o train -- python3 $GRAFT_REPO_ROOT/bench.py $F > $O/bench.json 2> $O/err.log
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -R $O | head -20

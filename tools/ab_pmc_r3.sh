#!/bin/bash
# VERDICT r2 item 3a, the counters beside the times of tools/ab_r3.sh: FETCH_SIZE / WRITE_SIZE of the encoder GEMMs for the chunk-size x
# store-policy variants (separate --pmc passes, one bench step each, single stream).  Raw: gpurun_out/prof_r3/ab_*; summary:
# profiles/r3_final/ab_chunk_nt.json (tools/collect_r3.py).
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_r3
mkdir -p $O
B="python3 bench.py --warmup 1 --steps 1 --no-cpu --no-kernel-profile --no-table --no-collate --no-small-ops --no-train --no-dropin"
for v in "65536 1" "65536 0" "32768 0"; do
  set -- $v
  tag="ct$1_nt$2"
  MANNER_HIP_NT_STORES=$2 MANNER_HIP_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/ab_fetch_$tag -- $B --chunk-tokens $1 > $O/ab_fetch_$tag.log 2>&1
  echo "$tag fetch done"
  MANNER_HIP_NT_STORES=$2 MANNER_HIP_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/ab_write_$tag -- $B --chunk-tokens $1 > $O/ab_write_$tag.log 2>&1
  # summarise on the box and drop the raw per-dispatch tables (gpurun_out/ travels back only below 64 MiB)
  python3 tools/pmc_traffic.py $O/ab_fetch_$tag $O/ab_write_$tag > $O/ab_pmc_$tag.json
  rm -rf $O/ab_fetch_$tag $O/ab_write_$tag
  echo "$tag done"
done

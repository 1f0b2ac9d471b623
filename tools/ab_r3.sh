#!/bin/bash
# Round-3 A/B of VERDICT r2 item 3a on the GPU box: chunk size x store policy of the streaming Q|K|V / FFN outputs
# (MANNER_HIP_NT_STORES=0: default-policy stores, so that a SMALL chunk's intermediates can stay in the 256 MiB Infinity
# Cache between producer and consumer), with the socket power and clocks sampled during every run.  Raw output:
# gpurun_out/r3/ab_*.json / ab_*.smi; summary: profiles/r3/ab_chunk_nt.md (tools/collect_r3.py).
mkdir -p gpurun_out/r3
F="--steps 10 --warmup 2 --no-cpu --no-table --no-scale-parity --no-small-ops --no-collate --no-train --no-dropin --no-kernel-profile"
for ct in 65536 32768 16384; do
  for nt in 1 0; do
    tag="ct${ct}_nt${nt}"
    MANNER_HIP_NT_STORES=$nt python bench.py $F --chunk-tokens $ct > gpurun_out/r3/ab_$tag.json 2> gpurun_out/r3/ab_$tag.err &
    BP=$!
    sleep 6
    : > gpurun_out/r3/ab_$tag.smi
    while kill -0 $BP 2>/dev/null; do
      rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" >> gpurun_out/r3/ab_$tag.smi
      sleep 0.4
    done
    wait $BP
    echo "$tag rc=$? $(python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/r3/ab_$tag.json').read().strip().splitlines()[-1])
    print(round(j['value']), round(j['encoder_mfma_frac'],4), 'bf16', round(j['bf16_mode']['value']), round(j['bf16_mode']['encoder_mfma_frac'],4))
except Exception as e: print('parse failed', e)
PY
)"
  done
done

F="--gpus 1 --steps 2 --warmup 1 --no-cpu --no-table --no-small-ops --no-collate --no-train --no-kernel-profile --no-parity-grade --dropin-only ${CASE:-8:eval}"
for i in 1 2 3; do
for m in 0 1; do
  export MANNER_PARAM_VIEW=$m
  python bench.py $F --full-json gpurun_out/r5/ab_pv_${m}_$i.json 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().splitlines()[-1]); print('param_view=$m', $i, j['legs']['dropin_ms_per_step'])"
done; done

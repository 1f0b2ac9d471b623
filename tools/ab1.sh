set -e
B="python bench.py --steps 4 --no-cpu --no-kernel-profile --no-collate --no-table --no-small-ops"
for v in "x" "MANNER_HIP_GEMM_CUS=128" "MANNER_HIP_GEMM_CUS=128 CH=32768" "MANNER_HIP_GEMM_CUS=192" "MANNER_HIP_GEMM_CUS=128 MANNER_HIP_STREAMS=3" "MANNER_HIP_GEMM_CUS=128 MANNER_HIP_STREAMS=4 CH=32768"; do
  CH=65536
  for kv in $v; do case $kv in CH=*) CH=${kv#CH=};; x) ;; *) export $kv;; esac; done
  echo "== $v"
  $B --chunk-tokens $CH 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['encoder_mfma_frac'],4), round(d['ms_per_step'],1))"
  unset MANNER_HIP_GEMM_CUS MANNER_HIP_STREAMS
done

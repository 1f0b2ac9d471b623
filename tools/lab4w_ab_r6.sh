#!/bin/bash
# same-box A/B of the production GEMM main loop against the 4-wave / 128x128-wave-tile main loop (tools/gemm4w_lab.hip,
# tools/gen_gemm4w_asm.py): check, times, SQ counters (separate --pmc passes per arm), rocm-smi power / sclk samples under each arm.
# Output: gpurun_out/r6/lab4w_*  -> profiles/r6_final/
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r6
mkdir -p "$O"
L=tools/bin/gemm4w_lab
timeout -k 10 300 $L 65536 check > "$O/lab4w_check.txt" 2>&1 || { echo "check FAILED"; cat "$O/lab4w_check.txt"; exit 1; }
timeout -k 10 200 $L 65536 time > "$O/lab4w_time.txt" 2>&1 || exit 1
for arm in prod asm; do
  LAB_REPS=1 LAB_ITERS=3 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS \
    --output-format csv -d "$O/pmc_sq_$arm" -- $L 65536 time $arm > "$O/pmc_sq_$arm.log" 2>&1 || { echo "pmc sq $arm failed"; tail -5 "$O/pmc_sq_$arm.log"; exit 1; }
  LAB_REPS=1 LAB_ITERS=3 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$arm" -- $L 65536 time $arm > "$O/pmc_fetch_$arm.log" 2>&1 || { echo "pmc fetch $arm failed"; exit 1; }
  LAB_REPS=1 LAB_ITERS=3 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d "$O/pmc_lds_$arm" -- $L 65536 time $arm > "$O/pmc_lds_$arm.log" 2>&1 || { echo "pmc lds $arm failed"; exit 1; }
done
python3 tools/lab4w_pmc.py "$O"/pmc_sq_prod "$O"/pmc_sq_asm "$O"/pmc_fetch_prod "$O"/pmc_fetch_asm "$O"/pmc_lds_prod "$O"/pmc_lds_asm > "$O/lab4w_pmc.json"
rm -rf "$O"/pmc_sq_* "$O"/pmc_fetch_* "$O"/pmc_lds_*
# power / clock under each arm: ~6 s of back-to-back launches, rocm-smi sampled meanwhile (read-only queries)
for arm in prod asm; do
  LAB_REPS=8 LAB_ITERS=400 $L 65536 time $arm > "$O/lab4w_power_$arm.time.txt" 2>&1 &
  BP=$!
  sleep 4
  : > "$O/lab4w_power_$arm.smi.txt"
  for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" >> "$O/lab4w_power_$arm.smi.txt"; echo "--" >> "$O/lab4w_power_$arm.smi.txt"; sleep 0.5; done
  wait $BP
done
echo "lab4w A/B done"
tail -13 "$O/lab4w_time.txt"

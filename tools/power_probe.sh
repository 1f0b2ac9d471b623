#!/bin/bash
# samples rocm-smi power / clocks while bench.py runs its timed steps (read-only queries; evidence for DESIGN §4)
mkdir -p gpurun_out/r2
rocm-smi --showpower --showclocks --showmaxpower > gpurun_out/r2/smi_idle.txt 2>&1
python bench.py --steps 40 --warmup 2 --no-cpu --no-table --no-scale-parity --no-small-ops --no-collate --no-train --no-kernel-profile > gpurun_out/r2/bench_power.json 2> gpurun_out/r2/bench_power.err &
BP=$!
sleep 12
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" >> gpurun_out/r2/smi_load.txt
  echo "--" >> gpurun_out/r2/smi_load.txt
  sleep 0.7
done
wait $BP
echo "bench rc $?"
head -30 gpurun_out/r2/smi_idle.txt
echo ==== load
head -40 gpurun_out/r2/smi_load.txt

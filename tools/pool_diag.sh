#!/bin/bash
# Development aid: builds a DIAGNOSTIC copy of the library (pool.hip with -DMANNER_POOL_DIAG) into gpurun_out/diag and runs tools/pool_diag.py on it.
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/diag
cp manner_amd/lib/libmanner_hip.so gpurun_out/diag/libmanner_hip.prod.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -Imanner_amd/csrc -DMANNER_POOL_DIAG ${POOL_DIAG_FLAGS:-} -c manner_amd/csrc/pool.hip -o gpurun_out/diag/pool.diag.o
objs=$(ls manner_amd/lib/obj/*.o | grep -v pool.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o manner_amd/lib/libmanner_hip.so $objs gpurun_out/diag/pool.diag.o
for nw in 4 8; do echo "NW=$nw"; MANNER_HIP_POOL_NW=$nw timeout -k 10 100 python tools/pool_diag.py 2>&1 | grep -v amdgpu.ids; done
cp gpurun_out/diag/libmanner_hip.prod.so manner_amd/lib/libmanner_hip.so

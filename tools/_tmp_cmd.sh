R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r6/p4_pmc"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export P4_SHAPE=ffn1 LAB_REPS=1 LAB_ITERS=2 P4_DEPHASE=0
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$O/p1" -- "$R/tools/bin/gemm_p4_lab" 65536 time > "$O/p1.log" 2>&1; echo "p1 rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d "$O/p2" -- "$R/tools/bin/gemm_p4_lab" 65536 time > "$O/p2.log" 2>&1; echo "p2 rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d "$O/p3" -- "$R/tools/bin/gemm_p4_lab" 65536 time > "$O/p3.log" 2>&1; echo "p3 rc=$?"
find "$O" -name "*kernel_trace.csv" -delete
python3 "$R/tools/pmc_kernels.py" "$O/p1" "$O/p2" "$O/p3" > "$O/summary.txt" 2>&1
tail -3 "$O/p3.log"; cat "$O/summary.txt"

#!/bin/bash
# PMC passes over the GEMM microbenchmark
export TMPDIR=/tmp
out=gpurun_out/pmc_gemm
mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1 </dev/null
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o g -- python tools/scratch/mb_gemm.py > $out/$tag.log 2>&1 </dev/null
  tail -2 $out/$tag.log
done
ls $out

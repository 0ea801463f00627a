#!/bin/bash
# PMC passes (one counter group per run) over the packed-GEMM microbenchmark at cfg3 shapes.
export TMPDIR=/tmp
tag=${1:-pmc_hl}
out=gpurun_out/$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  MB_ONLY=cfg3 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/gemm_$i -o g -- python tools/gemm_hl_microbench.py > $out/gemm_$i.log 2>&1 </dev/null
done
python tools/pmc_parse.py $out $out/summary.md gemm_hl pack_hl | grep -v "^  " | head -20

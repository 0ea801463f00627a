#!/bin/bash
# PMC passes (one counter group per run) over the cfg3 training step: wave-time split, MFMA
# pipe use, LDS and L2 figures of every kernel of the step.
# Usage (repo root, on the GPU box): bash tools/pmc_step.sh <tag> [config]
export TMPDIR=/tmp
tag=${1:-pmc_step}
cfg=${2:-cfg3}
out=gpurun_out/$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/step_$i -o g -- python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/step_$i.log 2>&1 </dev/null
done
python tools/pmc_parse.py $out $out/summary.md lstm_ gemm_hlx gemm_hlp pack_hl pack_rows ctc_ adam norm_partial fe_ gemm_splitk conv_ | grep -v "^  " | head -40

#!/bin/bash
mkdir -p gpurun_out/r2r
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x > gpurun_out/r2r/tests.log 2>&1; tail -3 gpurun_out/r2r/tests.log
for t in 128 1; do
  echo "== ASR_GEMM_HL_TILE=$t"
  ASR_GEMM_HL_TILE=$t MB_ONLY=cfg3 timeout 300 python tools/gemm_hl_microbench.py 2>&1 | grep -v "^$" | grep "fwd\|dX\|dW\|dU"
done

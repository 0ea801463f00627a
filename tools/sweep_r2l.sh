#!/bin/bash
out=gpurun_out/${1:-r2l}
mkdir -p $out
export TMPDIR=/tmp
for e in "A=1" "ASR_PIPELINE=0" "ASR_LSTM_FWD_GEN=1" "ASR_LSTM_FAST=0" "ASR_GEMM_FAST=0 REPS=8"; do
  env $e timeout 300 python tools/flaky_probe.py > $out/probe_$(echo $e | tr ' =' '__').log 2>&1 </dev/null
  tail -4 $out/probe_$(echo $e | tr ' =' '__').log
done

#!/bin/bash
# recurrence microbench with the phase profiler + the CU-starvation probe
tag=${1:-r3b}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/rec_bench.py cfg3 base ASR_LSTM_BWD_2D=0 ASR_LSTM_BWD_2D=1,ASR_LSTM_PREPOLL_B=0 \
    ASR_LSTM_BWD_2D=1,ASR_LSTM_PREPOLL_B=8 ASR_LSTM_BWD_2D=1,ASR_LSTM_FAST=0 ASR_LSTM_BWD_2D=0,ASR_LSTM_FAST=0 \
    > $out/rec_cfg3.log 2>&1 </dev/null
cat $out/rec_cfg3.log
timeout 600 python tools/rec_bench.py cfg2 --bwd base ASR_LSTM_BWD_2D=1 > $out/rec_cfg2.log 2>&1 </dev/null
cat $out/rec_cfg2.log
timeout 300 python tools/starve_probe.py > $out/starve.log 2>&1 </dev/null
cat $out/starve.log

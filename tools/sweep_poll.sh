#!/bin/bash
# Sweep of the pre-poll naps (ASR_LSTM_PREPOLL_F / _B) of the recurrent kernels.
run() { # cfg pf pb
  r=$(ASR_LSTM_PREPOLL_F=$2 ASR_LSTM_PREPOLL_B=$3 timeout 120 python tools/gpu_microbench.py $1 --lstm-only --no-stepwise 2>&1 | tail -1)
  echo "$1 pf=$2 pb=$3 :: $r"
}
for pb in 0 4 8 12 16 20; do run cfg2 12 $pb; done
for pb in 0 4 8 12 16; do run cfg3 16 $pb; done

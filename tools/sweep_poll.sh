#!/bin/bash
# Sweep of the pre-poll naps (ASR_LSTM_PREPOLL_F / _B) of the recurrent kernels.
run() { # cfg pf pb
  r=$(ASR_LSTM_PREPOLL_F=$2 ASR_LSTM_PREPOLL_B=$3 timeout 120 python tools/gpu_microbench.py $1 --lstm-only --no-stepwise 2>&1 | tail -1)
  echo "$1 pf=$2 pb=$3 :: $r"
}
for pf in 6 8 10 12 14 16 18; do run cfg2 $pf 8; done
for pb in 4 6 10; do run cfg2 12 $pb; done
for pf in 10 14 16 20 24; do run cfg3 $pf 0; done

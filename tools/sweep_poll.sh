#!/bin/bash
run() { # cfg pf rf pb rb
  r=$(ASR_LSTM_PREPOLL_F=$2 ASR_LSTM_REPOLL_F=$3 ASR_LSTM_PREPOLL_B=$4 ASR_LSTM_REPOLL_B=$5 timeout 120 python tools/gpu_microbench.py $1 --lstm-only --no-stepwise 2>&1 | tail -1)
  echo "$1 pf=$2 rf=$3 pb=$4 rb=$5 :: $r"
}
for pf in 14 16 18; do for rf in 1 4; do run cfg2 $pf $rf 0 1; done; done
for pb in 0 8 12 14; do for rb in 1 4 8; do run cfg2 16 1 $pb $rb; done; done
for pf in 12 16 20; do for rf in 1 4; do run cfg3 $pf $rf 0 1; done; done
for pb in 0 4 8; do for rb in 1 4; do run cfg3 16 1 $pb $rb; done; done

#!/bin/bash
# BPTT: default kernel vs the two-tiles-per-workgroup kernel (ASR_LSTM_PAIR_B=1) over the
# gather-issue placement and the nap before the issue.
run() { # cfg pair place nap
  r=$(ASR_LSTM_PAIR_B=$2 ASR_LSTM_PAIR_PLACE_B=$3 ASR_LSTM_PREPOLL_B=$4 timeout 120 python tools/gpu_microbench.py $1 --lstm-only --no-stepwise 2>&1 | grep "lstm mode0\|FAILED\|Error" | tail -1 | sed 's/.*bwd/bwd/')
  echo "$1 pairB=$2 place=$3 nap=$4 :: $r"
}
cfg=${1:-cfg2}
run $cfg 0 0 ${2:-8}
for place in 1 3 2; do for nap in 0 3; do run $cfg 1 $place $nap; done; done

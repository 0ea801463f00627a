#!/bin/bash
# Forward recurrence: default kernel vs the two-tiles-per-workgroup kernel (ASR_LSTM_PAIR=1)
# over the gather-issue placement and the nap before the issue / between re-polls.
run() { # cfg pair place nap repoll
  r=$(ASR_LSTM_PAIR=$2 ASR_LSTM_PAIR_PLACE=$3 ASR_LSTM_PREPOLL_F=$4 ASR_LSTM_REPOLL_F=${5:-1} timeout 120 python tools/gpu_microbench.py $1 --lstm-only --no-stepwise 2>&1 | grep "lstm mode0\|FAILED\|Error" | tail -1 | sed 's/ bwd.*//')
  echo "$1 pair=$2 place=$3 nap=$4 repoll=${5:-1} :: $r"
}
cfg=${1:-cfg2}
run $cfg 0 0 ${2:-12}
for place in 3 1; do for nap in 0 2; do for rp in 0 1 3; do run $cfg 1 $place $nap $rp; done; done; done

#!/bin/bash
tag=${1:-r3c}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/rec_bench.py cfg3 --bwd ASR_LSTM_BWD_2D=1 ASR_LSTM_BWD_2D=1,ASR_LSTM_PREPOLL_B=0 \
    ASR_LSTM_BWD_2D=1,ASR_LSTM_PREPOLL_B=2 ASR_LSTM_BWD_2D=1,ASR_LSTM_PREPOLL_B=8 ASR_LSTM_BWD_2D=1,ASR_LSTM_PREPOLL_B=12 \
    > $out/rec_cfg3.log 2>&1 </dev/null
cat $out/rec_cfg3.log
timeout 600 python tools/rec_bench.py cfg2 --bwd base ASR_LSTM_BWD_2D=1 ASR_LSTM_BWD_2D=1,ASR_LSTM_PREPOLL_B=8 > $out/rec_cfg2.log 2>&1 </dev/null
cat $out/rec_cfg2.log
timeout 300 python -m pytest -m gpu -x -q tests/test_gpu_lstm.py::test_bptt_two_dimensional_split_matches_the_one_dimensional_kernel > $out/pytest.log 2>&1 </dev/null
tail -3 $out/pytest.log

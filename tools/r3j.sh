#!/bin/bash
tag=${1:-r3j}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/rec_bench.py cfg3 --fwd base ASR_LSTM_PREPOLL_F=12 ASR_LSTM_PREPOLL_F=8 ASR_LSTM_PREPOLL_F=20 > $out/rec_cfg3.log 2>&1 </dev/null
cat $out/rec_cfg3.log
timeout 600 python tools/rec_bench.py cfg2 --fwd base ASR_LSTM_PREPOLL_F=8 ASR_LSTM_PREPOLL_F=16 > $out/rec_cfg2.log 2>&1 </dev/null
cat $out/rec_cfg2.log
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_lstm.py tests/test_gpu_model.py > $out/pytest.log 2>&1 </dev/null
tail -3 $out/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench.log 2>&1 </dev/null
python tools/bench_fields.py $out/bench.log

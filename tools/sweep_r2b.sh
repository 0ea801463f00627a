#!/bin/bash
# GPU visit: parity of the third-generation BPTT kernel, then a sweep of its switches at
# cfg3 and cfg2 (one bench.py process each; tools/bench_fields.py prints the headline).
out=gpurun_out/${1:-r2b}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_model.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $out/pytest.log 2>&1 </dev/null
tail -5 $out/pytest.log
run() {  # name, config, env...
  name=$1; cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_$name.log 2>&1 </dev/null
  echo "$name: $(tail -1 $out/bench_$name.log | python tools/bench_fields.py 2>&1 | tail -1)"
}
run c3_default cfg3 A=1
run c3_single cfg3 ASR_LSTM_PAIR_B=0
run c3_gen1 cfg3 ASR_LSTM_BWD_GEN=1
run c3_pair_pipe cfg3 ASR_PIPELINE=1
run c3_pair_place3 cfg3 ASR_LSTM_PAIR_PLACE_B=3
run c3_pair_prepoll4 cfg3 ASR_LSTM_PREPOLL_B=4
run c2_default cfg2 A=1
run c2_single cfg2 ASR_LSTM_PAIR_B=0
run c2_single_pp0 cfg2 ASR_LSTM_PAIR_B=0 ASR_LSTM_PREPOLL_B=0
run c2_single_pp4 cfg2 ASR_LSTM_PAIR_B=0 ASR_LSTM_PREPOLL_B=4
run c2_gen1 cfg2 ASR_LSTM_BWD_GEN=1
run c2_pair_place3 cfg2 ASR_LSTM_PAIR_PLACE_B=3

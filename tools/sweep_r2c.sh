#!/bin/bash
# GPU visit: parity of the third-generation forward kernel, then a sweep of its switches.
out=gpurun_out/${1:-r2c}
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
run c3_fsingle cfg3 ASR_LSTM_PAIR=0
run c3_fsingle_pp8 cfg3 ASR_LSTM_PAIR=0 ASR_LSTM_PREPOLL_F=8
run c3_fsingle_pp0 cfg3 ASR_LSTM_PAIR=0 ASR_LSTM_PREPOLL_F=0
run c3_fgen2 cfg3 ASR_LSTM_FWD_GEN=2
run c3_fpair_place3 cfg3 ASR_LSTM_PAIR_PLACE=3
run c3_fpair_place0 cfg3 ASR_LSTM_PAIR_PLACE=0
run c3_bpp4 cfg3 ASR_LSTM_PREPOLL_B=4
run c3_bpp8 cfg3 ASR_LSTM_PREPOLL_B=8
run c2_default cfg2 A=1
run c2_fsingle cfg2 ASR_LSTM_PAIR=0
run c2_fsingle_pp8 cfg2 ASR_LSTM_PAIR=0 ASR_LSTM_PREPOLL_F=8
run c2_fsingle_pp4 cfg2 ASR_LSTM_PAIR=0 ASR_LSTM_PREPOLL_F=4
run c2_fgen2 cfg2 ASR_LSTM_FWD_GEN=2
run c2_fpair_place3 cfg2 ASR_LSTM_PAIR_PLACE=3
run c2_bpp4 cfg2 ASR_LSTM_PREPOLL_B=4
run c2_nopipe cfg2 ASR_PIPELINE=0

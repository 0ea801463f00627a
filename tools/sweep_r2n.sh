#!/bin/bash
# Experiment: weight-gradient GEMMs (side stream) co-resident with the recurrences.
out=gpurun_out/${1:-r2n}
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, config, env...
  name=$1; cfg=$2; shift 2
  env "$@" timeout 200 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_$name.log 2>&1 </dev/null
  echo "$name: $(tail -1 $out/bench_$name.log | python tools/bench_fields.py 2>&1 | tail -1)"
}
run c3_base cfg3 A=1
run c3_share cfg3 ASR_LSTM_EXCL=0 ASR_SIDE_TILE=128 ASR_LDS_BALLAST=0
run c3_share_ballast cfg3 ASR_LSTM_EXCL=0 ASR_SIDE_TILE=128
run c3_all128_share cfg3 ASR_LSTM_EXCL=0 ASR_GEMM_HL_TILE=128 ASR_LDS_BALLAST=0
run c3_side128_excl cfg3 ASR_SIDE_TILE=128

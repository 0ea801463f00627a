#!/bin/bash
out=gpurun_out/${1:-r2j}
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, config, env...
  name=$1; cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_$name.log 2>&1 </dev/null
  echo "$name: $(tail -1 $out/bench_$name.log | python tools/bench_fields.py 2>&1 | tail -1)"
}
run c3_base cfg3 A=1
run c3_rec128 cfg3 ASR_REC_CUS=128
run c3_rec160 cfg3 ASR_REC_CUS=160
run c3_rec192 cfg3 ASR_REC_CUS=192
run c3_rec256 cfg3 ASR_REC_CUS=256
run c2_rec32 cfg2 ASR_REC_CUS=32
run c2_rec64 cfg2 ASR_REC_CUS=64
ASR_REC_CUS=128 timeout 600 python -m pytest tests/test_gpu_fullsize_parity.py -q -x -k "cfg3 and not exact" --timeout 600 > $out/pytest.log 2>&1 </dev/null
tail -3 $out/pytest.log

#!/bin/bash
out=gpurun_out/${1:-r2f}
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $out/pytest.log 2>&1 </dev/null
tail -15 $out/pytest.log
grep "parity\] \(mfcc\|logfbank\)" $out/pytest.log | sort -t= -k2 | tail -3
run() {  # name, config, env...
  name=$1; cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_$name.log 2>&1 </dev/null
  echo "$name: $(tail -1 $out/bench_$name.log | python tools/bench_fields.py 2>&1 | tail -1)"
}
run c3 cfg3 A=1
run c2 cfg2 A=1

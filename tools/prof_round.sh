#!/bin/bash
# One GPU-box visit: GPU test-suite, bench line, rocprofv3 kernel stats, and two PMC passes
# (FETCH_SIZE / WRITE_SIZE in their own runs, as gfx950's TCC slots require).
# Usage (from the repo root on the box): bash tools/prof_round.sh <tag>
tag=${1:-rX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > $out/pytest.log 2>&1 </dev/null
tail -3 $out/pytest.log
timeout 300 python bench.py > $out/bench.log 2>&1 </dev/null
tail -1 $out/bench.log
timeout 300 python bench.py --config cfg3 --steps 10 --warmup 3 > $out/bench_cfg3.log 2>&1 </dev/null
tail -1 $out/bench_cfg3.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/prof_stdout.log 2>&1 </dev/null
tail -1 $out/prof_stdout.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o bench -- \
      python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_$c.log 2>&1 </dev/null
  tail -1 $out/pmc_$c.log
done
find $out -name '*.csv' | head -20

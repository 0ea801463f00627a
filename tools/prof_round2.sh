#!/bin/bash
# One GPU-box visit (round 2): GPU test-suite, the default bench line (cfg3 + companions),
# rocprofv3 kernel stats of the cfg3 bench, and two PMC passes (FETCH_SIZE / WRITE_SIZE in
# their own runs, as gfx950's TCC slots require).
# Usage (from the repo root on the box): bash tools/prof_round2.sh <tag> [skip_tests]
tag=${1:-r2x}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ -z "$2" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 > $out/pytest.log 2>&1 </dev/null
  tail -3 $out/pytest.log
fi
timeout 600 python bench.py > $out/bench.log 2>&1 </dev/null
tail -1 $out/bench.log | cut -c1-600
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/prof_stdout.log 2>&1 </dev/null
tail -1 $out/prof_stdout.log | cut -c1-300
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o bench -- \
      python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc_$c.log 2>&1 </dev/null
  tail -1 $out/pmc_$c.log | cut -c1-200
done
find $out -name '*.csv' | head -20

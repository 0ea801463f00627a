#!/bin/bash
# Round-3 profile visit: rocprofv3 kernel stats of the cfg3 bench, FETCH_SIZE / WRITE_SIZE in
# separate PMC passes, and the per-kernel SQ / LDS / L2 counter groups of tools/pmc_step.sh.
# Usage (repo root, on the box): bash tools/prof_round3.sh <tag>
tag=${1:-r3p}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/prof_stdout.log 2>&1 </dev/null
tail -1 $out/prof_stdout.log | cut -c1-300
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o bench -- \
      python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc_$c.log 2>&1 </dev/null
  tail -1 $out/pmc_$c.log | cut -c1-200
done
find $out -name '*.csv' | head -20

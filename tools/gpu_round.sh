#!/bin/bash
# One GPU-box visit: the whole -m gpu suite, smoke(), the default bench line.
# Usage (repo root, on the box): bash tools/gpu_round.sh <tag> [pytest args]
tag=${1:-r3x}
shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 "$@" > $out/pytest.log 2>&1 </dev/null
tail -15 $out/pytest.log
timeout 120 python __graft_entry__.py smoke > $out/smoke.log 2>&1 </dev/null
tail -2 $out/smoke.log
timeout 600 python bench.py > $out/bench.log 2>&1 </dev/null
python tools/bench_fields.py $out/bench.log

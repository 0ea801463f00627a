#!/bin/bash
# Round-3 final evidence: kernel stats + HBM traffic of the cfg3 bench, the in-step PMC groups.
# Usage (repo root, on the box): bash tools/prof_final.sh <tag>
tag=${1:-r3z}
bash tools/prof_round3.sh $tag > gpurun_out/${tag}_prof.log 2>&1
python tools/summarize_prof.py gpurun_out/$tag/prof bench gpurun_out/$tag/kernel_stats.md "cfg3 bench, round 3 (final)" > /dev/null 2> gpurun_out/$tag/summ.err
python tools/summarize_pmc.py gpurun_out/$tag gpurun_out/$tag/hbm_traffic.md cfg3 > /dev/null 2>> gpurun_out/$tag/summ.err
cp gpurun_out/$tag/prof/bench_kernel_stats.csv gpurun_out/$tag/kernel_stats.csv
bash tools/pmc_step.sh ${tag}_pmc > gpurun_out/${tag}_pmc.log 2>&1
tail -3 gpurun_out/${tag}_prof.log; head -14 gpurun_out/$tag/kernel_stats.md; cat gpurun_out/$tag/summ.err | tail -3; tail -12 gpurun_out/${tag}_pmc.log

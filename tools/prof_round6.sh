#!/bin/bash
# Round-6 profile visit (repo root, on the GPU box): bash tools/prof_round6.sh <tag> [config]
#   1. rocprofv3 --kernel-trace --stats of the bench (13 steps, 3 of them warm-up);
#   2. FETCH_SIZE / WRITE_SIZE, each in its OWN --pmc pass (HBM traffic per kernel);
#   3. the SQ / LDS / L2 counter groups of tools/pmc_step.sh (one group per run);
# then the summaries profiles/ keeps: kernel_stats.md/.csv, hbm_traffic.md (EVERY kernel) and
# the families of profiles/pmc_traffic.json.
tag=${1:-r6z}
cfg=${2:-cfg3}
out=gpurun_out/$tag
mkdir -p $out/prof
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- \
    python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/prof/bench_stdout.log 2>$out/prof_stderr.log </dev/null
tail -1 $out/prof/bench_stdout.log | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o bench -- \
      python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc_$c.log 2>&1 </dev/null
done
python tools/summarize_prof.py $out/prof bench $out/kernel_stats.md "$cfg bench, round 6" > /dev/null 2> $out/summ.err
cp $out/prof/bench_kernel_stats.csv $out/kernel_stats.csv
python tools/summarize_pmc.py $out $out/hbm_traffic.md $cfg --json $out/pmc_traffic.json --source "profiles/r6z_bench_${cfg}_hbm_traffic.md" > /dev/null 2>> $out/summ.err
bash tools/pmc_step.sh ${tag}_pmc $cfg > $out/pmc_step.log 2>&1
head -12 $out/kernel_stats.md; tail -3 $out/summ.err

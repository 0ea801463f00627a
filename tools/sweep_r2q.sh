#!/bin/bash
# packed GEMM pipeline change: parity tests, microbench at cfg3 / cfg2 shapes, cfg3 bench
mkdir -p gpurun_out/r2q
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x > gpurun_out/r2q/tests.log 2>&1; tail -2 gpurun_out/r2q/tests.log
timeout 300 python tools/gemm_hl_microbench.py > gpurun_out/r2q/mb.log 2>&1; grep -v "^$" gpurun_out/r2q/mb.log | tail -12
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2q/bench_cfg3.json 2> gpurun_out/r2q/bench_cfg3.err
python tools/bench_fields.py < gpurun_out/r2q/bench_cfg3.json || tail -3 gpurun_out/r2q/bench_cfg3.err

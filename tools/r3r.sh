#!/bin/bash
tag=${1:-r3r}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest -m gpu -x -q -s tests/test_gpu_lstm.py::test_exact_fp32_kernels_at_the_benchmarked_widths > $out/pytest.log 2>&1 </dev/null
grep "parity\|passed\|failed\|Error" $out/pytest.log | tail -30
ASR_LSTM_PREC=0 timeout 300 python tools/rec_bench.py cfg3 base > $out/rec_exact.log 2>&1 </dev/null
grep -v amdgpu $out/rec_exact.log
ASR_LSTM_PREC=0 ASR_GEMM_PREC=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_exact.log 2>&1 </dev/null
python tools/bench_fields.py $out/bench_exact.log

#!/bin/bash
out=gpurun_out/${1:-r2h}
mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parallel.py -q --timeout 300 > $out/pytest.log 2>&1 </dev/null
tail -5 $out/pytest.log
SECONDS=0; timeout 900 python bench.py > $out/bench.log 2> $out/bench.err </dev/null; echo "bench wall ${SECONDS}s"

tail -1 $out/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('value', 'ms_per_step', 'dtype')})
print('predict', d.get('predict_latency'))
print('cfg2', {k: d['cfg2'].get(k) for k in ('value', 'ms_per_step', 'error')})
print('exact', {k: d['exact_fp32'].get(k) for k in ('value', 'ms_per_step', 'error')})
print('gemm', d['roofline_gate_gemm'])
print('roofline', d['roofline'])
print('gemm_step', d.get('roofline_gemm_step'))
print('cpu', d['cpu_baseline'])
"
tail -3 $out/bench.err

#!/bin/bash
out=gpurun_out/${1:-r2h}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_cli.py -q --timeout 600 -k "single_utterance or cli or predict or cfg5" > $out/pytest.log 2>&1 </dev/null
tail -5 $out/pytest.log
/usr/bin/time -v timeout 900 python bench.py > $out/bench.log 2> $out/bench.err </dev/null
grep -E "Elapsed|Maximum resident" $out/bench.err
tail -1 $out/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('value', 'ms_per_step', 'dtype')})
print('predict', d.get('predict_latency'))
print('cfg2', {k: d['cfg2'].get(k) for k in ('value', 'ms_per_step', 'error')})
print('exact', {k: d['exact_fp32'].get(k) for k in ('value', 'ms_per_step', 'error')})
print('gemm', d['roofline_gate_gemm'])
print('cpu', d['cpu_baseline'])
"

set -x
mkdir -p gpurun_out/r5c
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "independent" > gpurun_out/r5c/test1.log 2>&1; echo "rc=$?" >> gpurun_out/r5c/test1.log
timeout 300 python tools/rec_compact.py cfg3 > gpurun_out/r5c/rec_compact.log 2>&1
timeout 400 python bench.py --steps 10 --warmup 3 --config cfg2_n128 --no-cpu-baseline --no-extras > gpurun_out/r5c/bench_cfg2_n128.log 2>&1
ASR_BPTT_COMPACT=0 timeout 400 python bench.py --steps 10 --warmup 3 --config cfg2_n128 --no-cpu-baseline --no-extras > gpurun_out/r5c/bench_cfg2_n128_serial.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --config cfg3_conv --no-cpu-baseline --no-extras > gpurun_out/r5c/bench_conv.log 2>&1
ASR_BPTT_COMPACT=0 timeout 400 python bench.py --steps 20 --warmup 5 --config cfg3_conv --no-cpu-baseline --no-extras > gpurun_out/r5c/bench_conv_serial.log 2>&1
tail -3 gpurun_out/r5c/test1.log; cat gpurun_out/r5c/rec_compact.log
for f in cfg2_n128 cfg2_n128_serial conv conv_serial; do python - <<PY
import json
for l in open('gpurun_out/r5c/bench_$f.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$f', d['ms_per_step'], d['value'], d.get('fallbacks'), d['roofline_lstm_bwd'].get('geometry'))
PY
done

set -x
mkdir -p gpurun_out/r5a
timeout 600 python -m pytest tests/test_gpu_lstm.py -x -q -m gpu -k "compact or two_dimensional" > gpurun_out/r5a/test.log 2>&1; echo "test rc=$?" >> gpurun_out/r5a/test.log
timeout 300 python tools/rec_compact.py cfg3 > gpurun_out/r5a/rec_compact.log 2>&1
timeout 300 python tools/rec_compact.py cfg3c >> gpurun_out/r5a/rec_compact.log 2>&1
ASR_BPTT_COMPACT=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r5a/bench_serial.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r5a/bench_compact.log 2>&1
ASR_BPTT_COMPACT=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r5a/bench_serial2.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r5a/bench_compact2.log 2>&1
tail -3 gpurun_out/r5a/test.log; cat gpurun_out/r5a/rec_compact.log; for f in serial compact serial2 compact2; do python - <<PY
import json
for l in open('gpurun_out/r5a/bench_$f.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$f', d['ms_per_step'], d.get('fallbacks'))
PY
done

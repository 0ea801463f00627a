set -x
mkdir -p gpurun_out/r5e
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r5e/test1.log 2>&1; echo "rc=$?" >> gpurun_out/r5e/test1.log
timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -m gpu -k "elementwise_parity and (cfg3 or masks) and not exact" > gpurun_out/r5e/test2.log 2>&1; echo "rc=$?" >> gpurun_out/r5e/test2.log
for v in 0 1 0 1; do
ASR_PREPACK_YU=$v timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r5e/bench_yu$v.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/r5e/bench_yu$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('prepack_yu=$v', d['ms_per_step'], d.get('fallbacks'))
PY
done
tail -3 gpurun_out/r5e/test1.log; tail -3 gpurun_out/r5e/test2.log

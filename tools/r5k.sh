mkdir -p gpurun_out/r5k
for v in 0 auto 0 auto; do
ASR_BPTT_COMPACT=$v timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r5k/bench_$v.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/r5k/bench_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('compact=$v', d['ms_per_step'], 'fwd', d['roofline_lstm_fwd']['us_per_timestep'], 'bwd', d['roofline_lstm_bwd']['us_per_timestep'], 'gemm frac', d['roofline_gemm_step']['frac'], d['roofline_gemm_step']['ms_per_step'])
PY
done

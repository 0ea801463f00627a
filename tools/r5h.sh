mkdir -p gpurun_out/r5h
for v in 0 128 0 128 192; do
ASR_SIDE_SLOTS=$v timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r5h/bench_s$v.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/r5h/bench_s$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('side_slots=$v', d['ms_per_step'], d.get('fallbacks'), d['roofline_lstm_bwd'].get('geometry',{}).get('compact_us_per_timestep'))
PY
done

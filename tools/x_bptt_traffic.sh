export TMPDIR=/tmp
out=gpurun_out/r6c; mkdir -p $out
for lib in asr_study_amd/libasr_hip.so variants/libasr_stnt.so variants/libasr_ldnt.so; do
  tag=$(basename $lib .so)
  ASR_LIB_PATH=$lib python tools/rec_bench.py cfg3 --bwd base ASR_LSTM_COMPACT=1 2>&1 | grep "us/step" > $out/rec_$tag.txt
  cat $out/rec_$tag.txt
  for c in WRITE_SIZE FETCH_SIZE; do
    ASR_LIB_PATH=$lib timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${tag}_$c -o g -- python tools/rec_bench.py cfg3 --bwd base ASR_LSTM_COMPACT=1 > /dev/null 2>&1 </dev/null
  done
done
python - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/r6c/pmc_*')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'lstm_bwd' in r['Kernel_Name']:
                agg[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
        for k,v in agg.items():
            print(d.split('/')[-1], len(v), 'launches; MB first (default geometry):', [round(x/1000,1) for x in v[8:10]], 'last (compact):', [round(x/1000,1) for x in v[-2:]])
PY
for lib in asr_study_amd/libasr_hip.so variants/libasr_stnt.so variants/libasr_ldnt.so asr_study_amd/libasr_hip.so variants/libasr_stnt.so; do
ASR_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$lib', d['ms_per_step'], d['lstm_bwd_us_per_step_by_geometry'], d['roofline']['frac'])"
done

#!/bin/bash
# 300-step runs of every bench configuration: no timeout / fallback may occur (bench.py asserts it)
for c in cfg3 cfg3_conv cfg2 cfg2_n128; do
  python bench.py --config $c --steps ${1:-300} --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > /tmp/soak_$c.json
  python - "$c" <<'P'
import json, sys
c = sys.argv[1]
try:
    d = json.loads(open('/tmp/soak_%s.json' % c).read())
    print(c, d['ms_per_step'], 'ms/step, fallbacks', d['fallbacks'])
except Exception as e:
    print(c, 'FAILED', repr(e)[:200])
P
done

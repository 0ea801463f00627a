#!/usr/bin/env python
"""Reads bench.py JSON lines (file argument or stdin) and prints the headline figures compactly."""
import json
import sys
for ln in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    ln = ln.strip()
    if not ln.startswith('{'):
        continue
    d = json.loads(ln)
    print('%s: %.1f %s, %.3f ms/step, bwd %.3f us/step, fwd %.3f us/step, gemm %.1f TF/s' % (
        d['config']['workload'][:5], d['value'], d['unit'], d['ms_per_step'],
        d.get('roofline_lstm_bwd', d['roofline'])['us_per_timestep'], d['roofline_lstm_fwd']['us_per_timestep'],
        d['roofline_gate_gemm']['algorithmic_fp32_tflops']))

#!/usr/bin/env python
"""How far is each arithmetic from the FLOAT64 oracle on small gradient entries?

Runs the cfg3 full-size case (tests/golden/model_cfg3.npz: the oracle's 1000 sampled entries of
every gradient tensor) under the default split-fp16 products and under exact fp32 MFMA
(ASR_LSTM_PREC=0 ASR_GEMM_PREC=0), and bins the RELATIVE error of every sampled entry by its
magnitude relative to its tensor's maximum.  The round-3 review read the split-vs-exact
difference on entries at ~1e-6 of the maximum (worst relative 0.45) as a precision floor of the
packed planes; this table shows what exact fp32 itself does against float64 on the same
entries (entries that small are the result of cancellation in a 64 000-term sum).

    python tools/relerr_vs_fp64.py            # prints a markdown table
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BINS = [1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0001]


def worker(name):
    import torch
    from asr_study_amd import ops
    from asr_study_amd.core import models
    from oracle import fullsize_cases as FC
    from oracle import lstm as OL
    fix = np.load(os.path.join(ROOT, 'tests', 'golden', 'model_%s.npz' % name))
    case = FC.build(name)
    cfg, T = case['cfg'], case['T']
    N, F, H, L, C = cfg['N'], cfg['F'], cfg['H'], cfg['L'], cfg['C']
    dev = torch.device('cuda:0')
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L,
                          dropout=0.0, weight_decay=0.0, seed=1, device=dev)
    model.set_weights([a for _, a in OL.flatten(case['params'])])
    slab = torch.zeros((T, ops.pad16(N), F), dtype=torch.float32, device=dev)
    slab[:, :N] = torch.from_numpy(case['x']).to(dev)
    model.loss_and_grads(slab, case['labels'], case['lens'], training=False)
    torch.cuda.synchronize()
    rel, mag = [], []
    for i, g in enumerate(model.get_gradients()):
        flat = np.asarray(g, np.float64).reshape(-1)
        want = fix['g%02d_samples' % i]
        gmax = float(fix['g%02d_stats' % i][1])
        got = flat[FC.grad_sample_index(i, flat.size)]
        nz = np.abs(want) > 0
        rel.append(np.abs(got - want)[nz] / np.abs(want)[nz])
        mag.append(np.abs(want)[nz] / gmax)
    rel, mag = np.concatenate(rel), np.concatenate(mag)
    out = []
    for lo, hi in zip(BINS[:-1], BINS[1:]):
        m = (mag >= lo) & (mag < hi)
        out.append([int(m.sum())] + ([float(np.median(rel[m])), float(np.percentile(rel[m], 99)),
                                      float(rel[m].max())] if m.any() else [0, 0, 0]))
    print('RESULT ' + json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        return worker(sys.argv[2])
    name = 'cfg3'
    res = {}
    for label, env in (('split-fp16 (default)', {}),
                       ('exact fp32 MFMA', {'ASR_LSTM_PREC': '0', 'ASR_GEMM_PREC': '0'})):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', name],
                             env=dict(os.environ, **env), cwd=ROOT, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE)
        line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('RESULT ')]
        if not line:
            print(out.stderr.decode()[-2000:])
            raise SystemExit(1)
        res[label] = json.loads(line[0][7:])
    print('# Relative error against the float64 oracle, cfg3 full size (N = 64, T = 999), by entry '
          'magnitude\n')
    print('%d sampled gradient entries (1000 per tensor, 52 tensors); |g| / max|g of its tensor| '
          'bins; relative error |got - want| / |want|: median / p99 / worst\n'
          % sum(r[0] for r in res['exact fp32 MFMA']))
    print('| magnitude bin | entries | split-fp16 median | p99 | worst | exact-fp32 median | p99 | worst |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|')
    for k, (lo, hi) in enumerate(zip(BINS[:-1], BINS[1:])):
        a, b = res['split-fp16 (default)'][k], res['exact fp32 MFMA'][k]
        print('| [%.0e, %.0e) | %d | %.1e | %.1e | %.1e | %.1e | %.1e | %.1e |'
              % (lo, min(hi, 1.0), a[0], a[1], a[2], a[3], b[1], b[2], b[3]))


if __name__ == '__main__':
    main()

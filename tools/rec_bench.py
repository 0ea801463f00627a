#!/usr/bin/env python
"""Recurrent kernels in isolation at cfg2 / cfg3 shapes: us per step and, with the phase
profiler (ASR_LSTM_DBG=32), shader clocks per phase and wave of workgroup 0.

    python tools/rec_bench.py cfg3 [--fwd] [--bwd] [--planes] VAR=VALUE[,VAR=VALUE...] ...

--planes: BPTT writes dz as packed planes (asr_lstm_args.dz_hl) with a bound of 64 x max|dz|.

Each positional VAR=VALUE group is one variant (environment switches read per launch by
csrc/lstm.hip: ASR_LSTM_BWD_2D, ASR_LSTM_PROG, ASR_LSTM_FAST ...); 'base' = no switch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from asr_study_amd import ops  # noqa: E402

CFG = {'cfg2': dict(N=32, H=256), 'cfg3': dict(N=64, H=512)}
PH = ['pre', 'wait', 'math', 'barrier', 'mfma+pub', 'issue']


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    name = args[0] if args and args[0] in CFG else 'cfg3'
    variants = [a for a in args if a not in CFG] or ['base']
    do_f = '--fwd' in sys.argv or '--bwd' not in sys.argv
    do_b = '--bwd' in sys.argv or '--fwd' not in sys.argv
    N, H = CFG[name]['N'], CFG[name]['H']
    T = 999
    n_pad = ops.pad16(N)
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)
    U = rnd(2, H, 4 * H, scale=1.0 / np.sqrt(H))
    zx = rnd(T, n_pad, 2, 4 * H)
    y = torch.empty(T, n_pad, 2 * H, device=dev)
    cell = torch.empty(T, n_pad, 2, H, device=dev)
    gates = torch.empty(T, n_pad, 2, 4 * H, device=dev)
    dy = rnd(T, n_pad, 2 * H, scale=0.01)
    dz = torch.empty(T, n_pad, 2, 4 * H, device=dev)
    # warm-up: clocks, page tables, workspaces (the first measurement of a process reads ~8 %
    # slow otherwise)
    for _ in range(6):
        ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H)
        ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H)
    torch.cuda.synchronize()
    planes = bound = None
    if '--planes' in sys.argv:
        amax = torch.zeros(1, device=dev)
        ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H, dz_absmax=amax)
        bound = amax * 64.0
        planes = ops.HlPlanes(T * n_pad, 8 * H, dev)
    base_env = dict(os.environ)
    for var in variants:
        os.environ.clear()
        os.environ.update(base_env)
        if var != 'base':
            for kv in var.split(','):
                k, v = kv.split('=')
                os.environ[k] = v
        for kind, on, fn, wsn in (
                ('fwd', do_f, lambda: ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H), 'lstm_fwd'),
                ('bwd', do_b, lambda: ops.lstm_seq_bwd(
                    dy, U, cell, gates, dz, T, n_pad, H, dz_planes=planes if planes is not None and
                    ops.lstm_dz_hl_supported(T, n_pad, H) else None, dz_bound=bound), 'lstm_bwd')):
            if not on:
                continue
            os.environ.pop('ASR_LSTM_DBG', None)
            t = min(timeit(fn, reps=4), timeit(fn, reps=4))
            ws = ops.WS.get(wsn, 0, dev)
            ops.lstm_status(ws)
            line = '%s %-40s %s %.3f us/step (fast chains %d)' % (
                name, var, kind, t * 1e3 / T, ops.lstm_fast_chains(ws))
            os.environ['ASR_LSTM_DBG'] = '32'
            tp = timeit(fn, reps=1)
            pr = ops.lstm_profile(ws)
            os.environ.pop('ASR_LSTM_DBG', None)
            tot = [sum(r) for r in pr]
            print(line + '; profiled %.3f us/step' % (tp * 1e3 / T))
            if max(tot) > 0:
                for w, row in enumerate(pr):
                    print('    wave %d clocks/step: ' % w + '  '.join(
                        '%s %5.0f' % (PH[i], row[i] / float(T - 1)) for i in range(6)) +
                        '  | sum %5.0f' % (tot[w] / float(T - 1)))
    os.environ.clear()
    os.environ.update(base_env)


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Per-kernel timings at BASELINE cfg2 / cfg3 shapes (HIP events on the launch
stream).  Usage: python tools/gpu_microbench.py [cfg2|cfg3] [--lstm-only]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from asr_study_amd import ops  # noqa: E402

CFG = {'cfg2': dict(N=32, F=39, H=256, L=5, C=28, T=999),
       'cfg3': dict(N=64, F=80, H=512, L=5, C=28, T=999),
       'cfg1': dict(N=4, F=26, H=100, L=1, C=28, T=999)}


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'cfg2'
    c = CFG[name]
    N, F, H, C, T = c['N'], c['F'], c['H'], c['C'], c['T']
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    g = torch.Generator(device='cpu').manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)

    print('== %s: N=%d n_pad=%d H=%d T=%d' % (name, N, n_pad, H, T))
    rows = T * n_pad
    # ---- recurrent kernels
    U = rnd(2, H, 4 * H, scale=1.0 / np.sqrt(H))
    zx = rnd(T, n_pad, 2, 4 * H)
    y = torch.empty(T, n_pad, 2 * H, device=dev)
    cell = torch.empty(T, n_pad, 2, H, device=dev)
    gates = torch.empty(T, n_pad, 2, 4 * H, device=dev)
    dy = rnd(T, n_pad, 2 * H, scale=0.01)
    dz = torch.empty(T, n_pad, 2, 4 * H, device=dev)
    for mode in (0, 1):
        if mode == 1 and '--no-stepwise' in sys.argv:
            continue
        try:
            print('plan fwd', ops.lstm_plan(T, n_pad, H, 0), 'bwd', ops.lstm_plan(T, n_pad, H, 1))
            reps = 3 if mode == 0 else 1
            tf = timeit(lambda: ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H, mode=mode),
                        reps=reps, warm=1)
            ops.lstm_status(ops.WS.get('lstm_fwd', 0, torch.device(dev)))
            print('fast chains fwd:', ops.lstm_fast_chains(ops.WS.get('lstm_fwd', 0, torch.device(dev))))
            tb = timeit(lambda: ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H, mode=mode),
                        reps=reps, warm=1)
            ops.lstm_status(ops.WS.get('lstm_bwd', 0, torch.device(dev)))
            print('fast chains bwd:', ops.lstm_fast_chains(ops.WS.get('lstm_bwd', 0, torch.device(dev))))
            if int(os.environ.get('ASR_LSTM_DBG', '0')) & 32:
                for nm in ('lstm_fwd', 'lstm_bwd'):
                    pr = ops.lstm_profile(ops.WS.get(nm, 0, torch.device(dev)))
                    print(nm, 'us/step per wave [wait, barrier, A, B]:',
                          [[round(x / 100.0 / (T - 1), 2) for x in row] for row in pr])
            fl = 2.0 * 2 * T * n_pad * H * 4 * H
            print('lstm mode%d fwd %.3f ms (%.2f us/step, %.2f TF/s)  bwd %.3f ms (%.2f us/step)'
                  % (mode, tf, tf * 1e3 / T, fl / tf / 1e9, tb, tb * 1e3 / T))
        except Exception as e:                       # keep going: report and continue
            print('lstm mode%d FAILED: %s' % (mode, e))
    if '--lstm-only' in sys.argv:
        return
    # ---- GEMMs of one middle layer
    x = rnd(rows, 2 * H)
    W = rnd(2 * H, 8 * H, scale=0.05)
    b = rnd(8 * H)
    z2 = torch.empty(rows, 8 * H, device=dev)
    t = timeit(lambda: ops.gemm(x, W, z2, rows, 8 * H, 2 * H, bias=b))
    print('gemm fwd  %dx%dx%d: %.3f ms  %.1f TF/s' % (rows, 8 * H, 2 * H, t, 2.0 * rows * 8 * H * 2 * H / t / 1e9))
    dx = torch.empty(rows, 2 * H, device=dev)
    t = timeit(lambda: ops.gemm(z2, W, dx, rows, 2 * H, 8 * H, trans_b=True))
    print('gemm dX   %dx%dx%d: %.3f ms  %.1f TF/s' % (rows, 2 * H, 8 * H, t, 2.0 * rows * 8 * H * 2 * H / t / 1e9))
    dW = torch.empty(2 * H, 8 * H, device=dev)
    for sk in (16, 32, 'auto'):
        t = timeit(lambda: ops.gemm(x, z2, dW, 2 * H, 8 * H, rows, trans_a=True, split_k=sk))
        print('gemm dW   %dx%dx%d sk=%s: %.3f ms  %.1f TF/s' % (2 * H, 8 * H, rows, sk, t, 2.0 * rows * 8 * H * 2 * H / t / 1e9))
    dU = torch.empty(H, 4 * H, device=dev)
    yb = rnd(rows, 2 * H)
    t = timeit(lambda: ops.gemm(yb, z2, dU, H, 4 * H, rows - n_pad, trans_a=True, lda=2 * H,
                                ldb=8 * H, ldc=4 * H, b_off=n_pad * 8 * H, split_k='auto'))
    print('gemm dU   %dx%dx%d auto: %.3f ms  %.1f TF/s' % (H, 4 * H, rows, t, 2.0 * rows * 4 * H * H / t / 1e9))
    db = torch.empty(8 * H, device=dev)
    t = timeit(lambda: ops.colsum(z2, rows, 8 * H, 8 * H, db))
    print('colsum    %dx%d: %.3f ms  %.1f GB/s' % (rows, 8 * H, t, rows * 8 * H * 4 / t / 1e6))
    # ---- CTC
    logits = rnd(T, n_pad, C)
    rs = np.random.RandomState(0)
    lens = rs.randint(2, 50, size=N)
    lab = np.zeros((N, 49), np.int32)
    for n in range(N):
        lab[n, :lens[n]] = rs.randint(0, 25, size=lens[n])
    lab_d = torch.from_numpy(lab).to(dev)
    ll_d = torch.from_numpy(lens.astype(np.int32)).to(dev)
    sl_d = torch.full((N,), T, dtype=torch.int32, device=dev)
    grad = torch.empty_like(logits)
    t = timeit(lambda: ops.ctc_loss_grad(logits, lab_d, ll_d, sl_d, N, grad=grad, grad_scale=1.0 / N))
    print('ctc loss+grad: %.3f ms  algorithmic %.1f GB/s' % (t, 2.0 * T * N * C * 4 / t / 1e6))
    t = timeit(lambda: ops.ctc_greedy(logits, sl_d, N))
    print('ctc greedy: %.3f ms' % t)
    # ---- front-end
    from asr_study_amd.preprocessing import audio
    feat = audio.MFCC() if F == 39 else audio.LogFbank(num_filt=80) if F == 80 else audio.MFCC(dd=False)
    sig = torch.randn(N * 160000, generator=g).to(dev)
    offs = torch.arange(N, dtype=torch.int32, device=dev) * 160000
    lens_d = torch.full((N,), 160000, dtype=torch.int32, device=dev)
    t = timeit(lambda: feat.batch_device(sig, offs, lens_d, [160000] * N))
    print('front-end %s x %d utt of 10 s: %.3f ms  (%.0f audio-s/s, %.1f GB/s in+out)'
          % (feat, N, t, N * 10 / t * 1e3, (N * 160000 * 4 + N * 999 * F * 4) / t / 1e6))
    # ---- optimiser
    nparam = 6920000 if name == 'cfg2' else 27640000
    p = rnd(nparam)
    gr = rnd(nparam)
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    segs, nseg = ops.make_segments([(0, nparam // 2, 1e-4), (nparam // 2, nparam - nparam // 2, 0.0)], dev)
    norm = torch.zeros(2, dtype=torch.float64, device=dev)

    def opt():
        ops.grad_norm(p, gr, segs, nseg, norm)
        ops.adam_step(p, gr, m, v, segs, nseg, norm, 400.0, 1e-3, 1)
    t = timeit(opt)
    print('norm+adam %d params: %.3f ms  %.1f GB/s' % (nparam, t, nparam * 36.0 / t / 1e6))


if __name__ == '__main__':
    main()

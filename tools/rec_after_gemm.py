"""Does the recurrent kernel run slower right behind a power-capped GEMM burst?  cfg3 layer:
forward / BPTT launch timed (HIP events) back to back, and directly after 20 ms of packed GEMM."""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from asr_study_amd import ops
dev = torch.device('cuda:0')
T, n_pad, H = 999, 64, 512
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
U = rnd(2, H, 4 * H, scale=1.0 / np.sqrt(H))
zx = rnd(T, n_pad, 2, 4 * H)
y = torch.empty(T, n_pad, 2 * H, device=dev)
cell = torch.empty(T, n_pad, 2, H, device=dev)
gates = torch.empty(T, n_pad, 2, 4 * H, device=dev)
dy = rnd(T, n_pad, 2 * H, scale=0.01)
dz = torch.empty(T, n_pad, 2, 4 * H, device=dev)
rows, N, K = T * n_pad, 4096, 1024
one = torch.ones(1, device=dev)
x = rnd(rows, K, scale=0.5)
W = rnd(K, N, scale=0.05)
z = torch.empty(rows, N, device=dev)
xr, Wt = ops.HlPlanes(rows, K, dev), ops.HlPlanes(N, K, dev)
ops.pack_hl(x, rows, K, absmax=one, r=xr)
ops.pack_hl(W, K, N, absmax=ops.absmax(W), c=Wt)
fwd = lambda: ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H)
bwd = lambda: ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H)
gemm = lambda: ops.gemm_hl(xr, Wt, z, rows, N, K)
for _ in range(5):
    fwd(); bwd(); gemm()
torch.cuda.synchronize()


def ev(fn, before=None, n_before=0):
    for _ in range(n_before):
        before()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / T


def ev3(fn, before=None, n_before=0, wsn=None):
    """three launches in a row behind `before`, each timed; with the phase profiler on, also the
    shader clocks per step of wave 0 -> clocks and GHz"""
    for _ in range(n_before):
        before()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    evs[0].record()
    for i in range(3):
        fn(); evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) * 1e3 / T for i in range(3)]


def clocks(fn, wsn, before=None, n_before=0):
    os.environ['ASR_LSTM_DBG'] = '32'
    for _ in range(n_before):
        before()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / T
    pr = ops.lstm_profile(ops.WS.get(wsn, 0, dev))
    os.environ.pop('ASR_LSTM_DBG', None)
    clk = sum(pr[0]) / float(T - 1)
    return us, clk, clk / us / 1e3


for name, fn, wsn in (('fwd', fwd, 'lstm_fwd'), ('bwd', bwd, 'lstm_bwd')):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    print(name, 'x3 back to back            :', ' '.join('%.3f' % a for a in ev3(fn)))
    print(name, 'x3 behind 14 GEMMs (20 ms) :', ' '.join('%.3f' % a for a in ev3(fn, gemm, 14)))
    time.sleep(0.5)
    print(name, 'x3 behind 0.5 s idle       :', ' '.join('%.3f' % a for a in ev3(fn)))
    for _ in range(3):
        fn()
    print(name, 'profiled warm      : %.3f us/step, %.0f clocks/step -> %.2f GHz' % clocks(fn, wsn))
    print(name, 'profiled after GEMM: %.3f us/step, %.0f clocks/step -> %.2f GHz' % clocks(fn, wsn, gemm, 14))

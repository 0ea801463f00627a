"""Hand-off timeline of the cfg3 forward recurrence (ASR_LSTM_DBG=128, asr_lstm_trace): for 16
consecutive steps every wave of every workgroup stamps the 100 MHz clock when its gathered data
has arrived and when it has published.  Prints, for chain 0: the step period, the spread of
the publish times over the 32 workgroups, and the lag from the LAST publish a wave depends on
(the 8 workgroups that own its K slice) to its arrival."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from asr_study_amd import ops, _lib
dev = torch.device('cuda:0')
T, n_pad, H, P = 999, 64, 512, 32
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
U = rnd(2, H, 4 * H, scale=1.0 / np.sqrt(H))
zx = rnd(T, n_pad, 2, 4 * H)
y = torch.empty(T, n_pad, 2 * H, device=dev)
cell = torch.empty(T, n_pad, 2, H, device=dev)
gates = torch.empty(T, n_pad, 2, 4 * H, device=dev)
for _ in range(5):
    ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H)
torch.cuda.synchronize()
os.environ['ASR_LSTM_DBG'] = '128'
ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H)
os.environ.pop('ASR_LSTM_DBG')
n = 256 * 4 * 16 * 2
buf = (C.c_longlong * n)()
_lib.check(_lib.load().asr_lstm_trace(buf, n, torch.cuda.current_stream().cuda_stream), 'asr_lstm_trace')
tr = np.frombuffer(buf, dtype=np.int64).reshape(256, 4, 16, 2).astype(np.float64) * 10.0   # ns
blk = np.arange(256)
chain = ((blk >> 3) // P) * 8 + (blk & 7)
wg = (blk >> 3) % P
for ch in (0, 5):
    sel = np.where(chain == ch)[0]
    order = sel[np.argsort(wg[sel])]
    arr, pub = tr[order, :, :, 0], tr[order, :, :, 1]          # (32 wg, 4 waves, 16 steps)
    t0 = pub[:, :, 1:].min()
    period = np.diff(pub.max(axis=(0, 1))[1:]).mean()
    print('chain %d: step period %.0f ns; publish spread over the 32 workgroups (max - min per step): '
          'mean %.0f ns, worst %.0f' % (ch, period, (pub.max(axis=(0, 1)) - pub.min(axis=(0, 1)))[1:].mean(),
                                        (pub.max(axis=(0, 1)) - pub.min(axis=(0, 1)))[1:].max()))
    lags, own = [], []
    for w in range(4):
        lastpub = pub[8 * w:8 * w + 8].max(axis=(0, 1))        # (16,) last publish of the slice's owners
        lag = arr[:, w, 1:] - lastpub[None, :-1]               # arrival at step s vs publishes of s-1
        lags.append(lag)
    lags = np.stack(lags)                                      # (4 waves, 32 wg, 15)
    print('   lag last needed publish -> arrival: mean %.0f ns, min %.0f, p50 %.0f, p90 %.0f, max %.0f'
          % (lags.mean(), lags.min(), np.median(lags), np.percentile(lags, 90), lags.max()))
    chainlen = pub[:, :, 1:] - arr[:, :, 1:]
    print('   arrival -> own publish (the dependent chain): mean %.0f ns, min %.0f, max %.0f'
          % (chainlen.mean(), chainlen.min(), chainlen.max()))
    wave_arr_spread = arr[:, :, 1:].max(axis=1) - arr[:, :, 1:].min(axis=1)
    print('   spread of the arrivals of the four waves inside a workgroup: mean %.0f ns, max %.0f'
          % (wave_arr_spread.mean(), wave_arr_spread.max()))
    late = pub[:, :, 1:].max(axis=1).argmax(axis=0)
    print('   last publisher per step (workgroup):', late.tolist())
    print('   per workgroup, mean over the steps: chain length (arrival of its LAST wave -> publish, ns):')
    print('     ', (pub[:, :, 1:].max(axis=1) - arr[:, :, 1:].max(axis=1)).mean(axis=1).round().astype(int).tolist())
    print('   per workgroup: its last wave\'s arrival after the chain\'s FIRST publish of the previous step (ns):')
    firstpub = pub.min(axis=(0, 1))
    print('     ', (arr[:, :, 1:].max(axis=1) - firstpub[None, :-1]).mean(axis=1).round().astype(int).tolist())
    print('   per workgroup: publish after the chain\'s first publish of the same step (ns):')
    print('     ', (pub[:, :, 1:].max(axis=1) - firstpub[None, 1:]).mean(axis=1).round().astype(int).tolist())
    s = 7
    print('   step %d publish times by workgroup (ns after the first):' % s,
          (pub[:, :, s].max(axis=1) - pub[:, :, s].min()).round().astype(int).tolist())


# ---- BPTT (two-dimensional split): workgroup cw = a + 8 b of a chain gathers the 8 partial tiles
# of its output block a // 2 from the workgroups a'' + 8 (a // 2), a'' = 0..7
dy = rnd(T, n_pad, 2 * H, scale=0.01)
dz = torch.empty(T, n_pad, 2, 4 * H, device=dev)
for _ in range(3):
    ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H)
torch.cuda.synchronize()
os.environ['ASR_LSTM_DBG'] = '128'
ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H)
os.environ.pop('ASR_LSTM_DBG')
_lib.check(_lib.load().asr_lstm_trace(buf, n, torch.cuda.current_stream().cuda_stream), 'asr_lstm_trace')
tr = np.frombuffer(buf, dtype=np.int64).reshape(256, 4, 16, 2).astype(np.float64) * 10.0
for ch in (0,):
    sel = np.where(chain == ch)[0]
    order = sel[np.argsort(wg[sel])]
    arr, pub = tr[order, :, :, 0], tr[order, :, :, 1]
    period = np.diff(pub.max(axis=(0, 1))[1:]).mean()
    spread = (pub.max(axis=(0, 1)) - pub.min(axis=(0, 1)))[1:]
    print('BPTT chain %d: step period %.0f ns; publish spread over the 32 workgroups: mean %.0f ns, worst %.0f'
          % (ch, period, spread.mean(), spread.max()))
    lags = []
    for cw in range(32):
        a_ = cw % 8
        prod = [a2 + 8 * (a_ // 2) for a2 in range(8)]
        lastpub = pub[prod].max(axis=(0, 1))
        lags.append(arr[cw, :, 1:].max(axis=0) - lastpub[:-1])
    lags = np.stack(lags)
    print('   lag last needed publish -> arrival (last wave): mean %.0f ns, min %.0f, p50 %.0f, p90 %.0f, max %.0f'
          % (lags.mean(), lags.min(), np.median(lags), np.percentile(lags, 90), lags.max()))
    chainlen = pub[:, :, 1:].max(axis=1) - arr[:, :, 1:].max(axis=1)
    print('   arrival of the last wave -> publish (the dependent chain): mean %.0f ns, min %.0f, max %.0f'
          % (chainlen.mean(), chainlen.min(), chainlen.max()))
    print('   last publisher per step (workgroup):', pub[:, :, 1:].max(axis=1).argmax(axis=0).tolist())
    firstpub = pub.min(axis=(0, 1))
    print('   per workgroup: publish after the chain\'s first publish of the same step (ns):')
    print('     ', (pub[:, :, 1:].max(axis=1) - firstpub[None, 1:]).mean(axis=1).round().astype(int).tolist())

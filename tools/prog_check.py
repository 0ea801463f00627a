#!/usr/bin/env python
"""ASR_LSTM_PROG=1 (progressive forward step) against the default forward kernel: same
activations up to the summation order (1e-6), at cfg2 and cfg3 shapes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from asr_study_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
for (N, H, T) in ((32, 256, 300), (64, 512, 300)):
    g = torch.Generator(device='cpu').manual_seed(0)
    U = (torch.randn(2, H, 4 * H, generator=g) / np.sqrt(H)).to(dev)
    zx = torch.randn(T, N, 2, 4 * H, generator=g).to(dev)
    outs = []
    for prog in ('0', '1', '1'):
        os.environ['ASR_LSTM_PROG'] = prog
        y = torch.zeros(T, N, 2 * H, device=dev)
        cell = torch.zeros(T, N, 2, H, device=dev)
        gates = torch.zeros(T, N, 2, 4 * H, device=dev)
        ws = ops.lstm_seq_fwd(zx, U, y, cell, gates, T, N, H)
        ops.lstm_status(ws)
        outs.append((y, cell, gates))
    os.environ.pop('ASR_LSTM_PROG')
    print('H=%d: prog vs base max|dy| %.2e  max|dcell| %.2e  max|dgates| %.2e; prog twice bit-equal: %s'
          % (H, (outs[0][0] - outs[1][0]).abs().max().item(), (outs[0][1] - outs[1][1]).abs().max().item(),
             (outs[0][2] - outs[1][2]).abs().max().item(),
             all(torch.equal(a, b) for a, b in zip(outs[1], outs[2]))))

#!/usr/bin/env python
"""What a foreign kernel holding CUs does to a chip-filling recurrence (timings, flags)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from asr_study_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
T, n_pad, H = 60, 64, 512
g = torch.Generator(device='cpu').manual_seed(0)
U = (torch.randn(2, H, 4 * H, generator=g) / np.sqrt(H)).to(dev)
zx = torch.randn(T, n_pad, 2, 4 * H, generator=g).to(dev)
y = torch.empty(T, n_pad, 2 * H, device=dev)
cell = torch.empty(T, n_pad, 2, H, device=dev)
gates = torch.empty(T, n_pad, 2, 4 * H, device=dev)
side = torch.cuda.Stream(device=dev)
ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H)
torch.cuda.synchronize()
for blocks, lds, secs in ((64, 96 * 1024, 0.05), (64, 96 * 1024, 0.9), (64, 160 * 1024, 0.9),
                          (256, 96 * 1024, 0.05), (2048, 0, 0.9)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        ops.debug_occupy(blocks, lds, secs)
    time.sleep(0.01)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H)
    e1.record()
    torch.cuda.synchronize()
    flags = ops.lstm_timeout_flags(dev).cpu().tolist()
    print('hog %4d blocks x %3d KB LDS for %.2f s: recurrence %.1f ms, wall %.3f s, flags %s'
          % (blocks, lds // 1024, secs, e0.elapsed_time(e1), time.perf_counter() - t0, flags))
    ops.clear_timeout_flags(dev)

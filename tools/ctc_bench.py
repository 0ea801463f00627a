"""CTC loss + gradient at the cfg3 shape (T=999, N=64, C=28, ~100-symbol labels): total time of the
three kernels (row lse, alpha/beta chains with checkpoints, frame-parallel gradient) and of the
loss-only call (alpha chain alone) -> ns per alpha step."""
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from asr_study_amd import ops
from tools.gpu_microbench import timeit
dev = 'cuda:0'
T, N, C = 999, 64, 28
rs = np.random.RandomState(0)
logits = torch.from_numpy(rs.randn(T, N, C).astype(np.float32) * 2).to(dev)
sl = torch.full((N,), T, dtype=torch.int32, device=dev)
grad = torch.empty_like(logits)
for L in (40, 100):                       # one / two state pairs per lane
    lab_d = torch.from_numpy(rs.randint(0, C - 1, size=(N, L)).astype(np.int32)).to(dev)
    ll = torch.full((N,), L, dtype=torch.int32, device=dev)
    t = timeit(lambda: ops.ctc_loss_grad(logits, lab_d, ll, sl, N, grad=grad, grad_scale=1.0 / N), reps=20)
    t0 = timeit(lambda: ops.ctc_loss_grad(logits, lab_d, ll, sl, N, grad=None), reps=20)
    print('L=%d: ctc loss+grad %.3f ms; loss only (lse + alpha chain) %.3f ms = %.0f ns per alpha step'
          % (L, t, t0, t0 * 1e6 / T))

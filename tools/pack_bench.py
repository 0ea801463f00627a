"""Pack passes of one cfg3 layer as the engine issues them: dz (rows, 8H) row planes; the input
slab under the two directions' masks in one pass against two passes; y (.) B_U per direction."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from asr_study_amd import ops
from tools.gpu_microbench import timeit
dev = 'cuda:0'
T, n_pad, H = 999, 64, 512
rows = T * n_pad
one = torch.ones(1, device=dev)
dz = torch.randn(rows, 8 * H, device=dev) * 1e-3
x = torch.randn(rows, 2 * H, device=dev)
m1 = (torch.rand(n_pad, 2 * H, device=dev) > 0.2).float() / 0.8
m2 = (torch.rand(n_pad, 2 * H, device=dev) > 0.2).float() / 0.8
zr = ops.HlPlanes(rows, 8 * H, dev)
a, b = ops.HlPlanes(rows, 2 * H, dev), ops.HlPlanes(rows, 2 * H, dev)
yu = ops.HlPlanes(rows, H, dev)
amz = ops.absmax(dz)
t = timeit(lambda: ops.pack_hl(dz, rows, 8 * H, absmax=amz, r=zr), reps=20)
print('dz row planes: %.3f ms (%.2f TB/s)' % (t, 2 * rows * 8 * H * 4 / t / 1e9))
t2 = timeit(lambda: (ops.pack_hl(x, rows, 2 * H, mask=m1, mask_period=n_pad, absmax=one, r=a),
                     ops.pack_hl(x, rows, 2 * H, mask=m2, mask_period=n_pad, absmax=one, r=b)), reps=20)
print('input under the two directions\' masks, two passes (one fused pass measured 0.262 ms '
      'against 0.241 and was dropped): %.3f ms' % t2)
t = timeit(lambda: ops.pack_hl(x, rows, H, ld=2 * H, src_off=H, mask=m1[:, :H].contiguous(), mask_period=n_pad,
                               absmax=one, r=yu), reps=20)
print('y (.) B_U of one direction: %.3f ms' % t)

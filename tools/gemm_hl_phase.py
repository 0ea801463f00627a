"""K-loop phase profile of the packed split-fp16 GEMM (asr_gemm_hl_profile): clocks per 32-deep
slab and per wave of workgroup 0, for the forward z GEMM of one cfg3 layer."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from asr_study_amd import ops, _lib
dev = 'cuda:0'
lib = _lib.load()
one = torch.ones(1, device=dev)
for rows, N, K in ((63936, 4096, 1024), (63936, 1024, 4096)):
    x = torch.randn(rows, K, device=dev) * 0.5
    W = torch.randn(K, N, device=dev) * 0.05
    z = torch.empty(rows, N, device=dev)
    xr = ops.HlPlanes(rows, K, dev)
    Wt = ops.HlPlanes(N, K, dev)
    ops.pack_hl(x, rows, K, absmax=one, r=xr)
    ops.pack_hl(W, K, N, absmax=ops.absmax(W), c=Wt)
    for _ in range(3):
        ops.gemm_hl(xr, Wt, z, rows, N, K)
    st = torch.cuda.current_stream().cuda_stream
    nk = K // 32
    for mode in (1,):
        lib.asr_gemm_hl_profile(mode, None, st)
        ops.gemm_hl(xr, Wt, z, rows, N, K)
        out = (C.c_longlong * 64)()
        lib.asr_gemm_hl_profile(0, out, st)
        print('%dx%dx%d mode %d: %d slabs; clocks per slab (reads, bar, mfma0, mfma1, bar, '
              'prologue total, loop us) -> shader clock' % (rows, N, K, mode, nk))
        for w in (0, 4):
            v = [out[w * 8 + i] for i in range(7)]
            print('  wave %d: ' % w + ' '.join('%7.1f' % (v[i] / nk) for i in range(5))
                  + ' %8d %7.2f' % (v[5], v[6] / 100.0) + '   loop/slab %.1f  -> %.2f GHz'
                  % (sum(v[:5]) / nk, sum(v[:5]) / (v[6] * 10.0)))

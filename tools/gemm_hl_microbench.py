"""GEMM shapes of one BiLSTM layer (cfg3 / cfg2) on packed split-fp16 planes (asr_pack_hl +
asr_gemm_hl) next to the convert-per-tile kernel (asr_gemm)."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from asr_study_amd import ops
from tools.gpu_microbench import timeit
dev = 'cuda:0'
one = torch.ones(1, device=dev)
only = os.environ.get('MB_ONLY')
for name, T, n_pad, H in (('cfg3', 999, 64, 512), ('cfg2', 999, 32, 256)):
    if only and name != only:
        continue
    rows = T * n_pad
    x = torch.randn(rows, 2 * H, device=dev) * 0.5
    W = torch.randn(2 * H, 8 * H, device=dev) * 0.05
    dz = torch.randn(rows, 8 * H, device=dev) * 1e-3
    z = torch.empty(rows, 8 * H, device=dev)
    dx = torch.empty(rows, 2 * H, device=dev)
    dW = torch.empty(2 * H, 8 * H, device=dev)
    dU = torch.empty(H, 4 * H, device=dev)
    xr, xc = ops.HlPlanes(rows, 2 * H, dev), ops.HlPlanes(2 * H, rows, dev)
    Wn, Wt = ops.HlPlanes(2 * H, 8 * H, dev), ops.HlPlanes(8 * H, 2 * H, dev)
    zr, zc = ops.HlPlanes(rows, 8 * H, dev), ops.HlPlanes(8 * H, rows, dev)
    amz = ops.absmax(dz)
    tp = timeit(lambda: ops.pack_hl(x, rows, 2 * H, absmax=one, r=xr, c=xc))
    print('%s pack x (both orientations): %.3f ms' % (name, tp))
    tp = timeit(lambda: ops.pack_hl(dz, rows, 8 * H, absmax=amz, r=zr, c=zc))
    print('%s pack dz (both orientations): %.3f ms' % (name, tp))
    ops.pack_hl(W, 2 * H, 8 * H, absmax=ops.absmax(W), r=Wn, c=Wt)
    fl = 2.0 * rows * 8 * H * 2 * H
    t = timeit(lambda: ops.gemm_hl(xr, Wt, z, rows, 8 * H, 2 * H))
    t0 = timeit(lambda: ops.gemm(x, W, z, rows, 8 * H, 2 * H))
    print('%s fwd  %dx%dx%d: packed %.3f ms %.1f TF/s | per-tile %.3f ms %.1f TF/s' % (
        name, rows, 8 * H, 2 * H, t, fl / t / 1e9, t0, fl / t0 / 1e9))
    t = timeit(lambda: ops.gemm_hl(zr, Wn, dx, rows, 2 * H, 8 * H))
    t0 = timeit(lambda: ops.gemm(dz, W, dx, rows, 2 * H, 8 * H, trans_b=True, a_absmax=amz))
    print('%s dX   %dx%dx%d: packed %.3f ms %.1f TF/s | per-tile %.3f ms %.1f TF/s' % (
        name, rows, 2 * H, 8 * H, t, fl / t / 1e9, t0, fl / t0 / 1e9))
    t = timeit(lambda: ops.gemm_hl(xc, zc, dW, 2 * H, 8 * H, rows, split_k='auto'))
    t0 = timeit(lambda: ops.gemm(x, dz, dW, 2 * H, 8 * H, rows, trans_a=True, split_k='auto',
                                 b_absmax=amz))
    print('%s dW   %dx%dx%d: packed %.3f ms %.1f TF/s | per-tile %.3f ms %.1f TF/s' % (
        name, 2 * H, 8 * H, rows, t, fl / t / 1e9, t0, fl / t0 / 1e9))
    flu = 2.0 * rows * 4 * H * H
    t = timeit(lambda: ops.gemm_hl(xc, zc, dU, H, 4 * H, rows - n_pad, b_k=n_pad, split_k='auto'))
    t0 = timeit(lambda: ops.gemm(x, dz, dU, H, 4 * H, rows - n_pad, trans_a=True, lda=2 * H,
                                 ldb=8 * H, ldc=4 * H, b_off=n_pad * 8 * H, split_k='auto',
                                 b_absmax=amz))
    print('%s dU   %dx%dx%d: packed %.3f ms %.1f TF/s | per-tile %.3f ms %.1f TF/s' % (
        name, H, 4 * H, rows, t, flu / t / 1e9, t0, flu / t0 / 1e9))
    tk = timeit(lambda: ops.gemm_hl(xr, zr, dW, 2 * H, 8 * H, rows, split_k='auto', k_major=True))
    print('%s dW   K-major planes (ds_read_b64_tr_b16): %.3f ms %.1f TF/s' % (name, tk, fl / tk / 1e9))
    tk = timeit(lambda: ops.gemm_hl(xr, zr, dU, H, 4 * H, rows - n_pad, b_row=n_pad, split_k='auto',
                                    k_major=True))
    print('%s dU   K-major planes: %.3f ms %.1f TF/s' % (name, tk, flu / tk / 1e9))
    del x, W, dz, z, dx, xr, xc, zr, zc
    torch.cuda.empty_cache()

"""Device (beam.hip) vs host (decode_host.cpp) beam search by batch size: seconds per batch of N
utterances x 999 frames x 28 classes at widths 100 / 400, host threads = all / 8 (taskset)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from asr_study_amd import ops
dev = torch.device('cuda:0')
T, C = 999, 28
rs = np.random.RandomState(0)
for N in (64, 256, 512):
    n_pad = ops.pad16(N)
    lg = torch.from_numpy((rs.randn(T, n_pad, C) * 2).astype(np.float32)).to(dev)
    sl = torch.full((N,), T, dtype=torch.int32, device=dev)
    host_in = lg.cpu().numpy()
    for width in (100, 400):
        ops.ctc_beam_search(lg, sl, N, width); torch.cuda.synchronize()
        t0 = time.perf_counter(); ops.ctc_beam_search(lg, sl, N, width); torch.cuda.synchronize()
        td = time.perf_counter() - t0
        t0 = time.perf_counter(); ops.ctc_beam_search_host(host_in, [T] * N, N, width)
        th = time.perf_counter() - t0
        print('N=%d width=%d device %.4f s host(%d threads) %.4f s' % (N, width, td, len(os.sched_getaffinity(0)), th), flush=True)

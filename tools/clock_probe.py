"""Shader clock and board power under a sustained packed-GEMM load: rocm-smi sampled from a side
thread while the GEMM loops, then the in-kernel clock (s_memtime vs the 100 MHz s_memrealtime)
of one profiled launch at the end of the run."""
import ctypes as C
import os
import subprocess
import sys
import threading
import time
sys.path.insert(0, os.getcwd())
import torch
from asr_study_amd import ops, _lib
dev = 'cuda:0'
lib = _lib.load()
one = torch.ones(1, device=dev)
rows, N, K = 63936, 4096, 1024
x = torch.randn(rows, K, device=dev) * 0.5
W = torch.randn(K, N, device=dev) * 0.05
z = torch.empty(rows, N, device=dev)
xr = ops.HlPlanes(rows, K, dev)
Wt = ops.HlPlanes(N, K, dev)
ops.pack_hl(x, rows, K, absmax=one, r=xr)
ops.pack_hl(W, K, N, absmax=ops.absmax(W), c=Wt)
samples = []
stop = False


def sample():
    while not stop:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True,
                             text=True).stdout
        keep = [l.split(':', 1)[-1].strip() for l in out.splitlines()
                if 'sclk' in l or 'Power' in l or 'power' in l]
        samples.append((time.time(), keep))
        time.sleep(0.3)


th = threading.Thread(target=sample)
th.start()
t0 = time.time()
secs = float(os.environ.get('PROBE_SECONDS', '6'))
n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        ops.gemm_hl(xr, Wt, z, rows, N, K)
    torch.cuda.synchronize()
    n += 50
dt = time.time() - t0
st = torch.cuda.current_stream().cuda_stream
lib.asr_gemm_hl_profile(1, None, st)
ops.gemm_hl(xr, Wt, z, rows, N, K)
out = (C.c_longlong * 64)()
lib.asr_gemm_hl_profile(0, out, st)
stop = True
th.join()
fl = 2.0 * rows * N * K
print('%d launches in %.2f s: %.3f ms each, %.1f TF/s algorithmic, %.2f PF/s executed'
      % (n, dt, dt / n * 1e3, fl * n / dt / 1e12, 3 * fl * n / dt / 1e15))
v = [out[i] for i in range(7)]
if v[6] > 0:        # (the persistent form, gemm_hlp_kernel, carries no phase profiler)
    print('in-kernel: %.0f clocks per slab, loop %.1f us -> shader clock %.2f GHz'
          % (sum(v[:5]) / (K // 32), v[6] / 100.0, sum(v[:5]) / (v[6] * 10.0)))
import re
mhz = sorted(int(m.group(1)) for t, k in samples for m in [re.search(r'\((\d+)Mhz\)', ' '.join(k))] if m)
watt = sorted(float(m.group(1)) for t, k in samples for m in [re.search(r'Power \(W\): ([0-9.]+)', ' '.join(k))] if m)
if mhz and watt:
    print('rocm-smi over the run: sclk median %d MHz (min %d, max %d), package power median %.0f W'
          % (mhz[len(mhz) // 2], mhz[0], mhz[-1], watt[len(watt) // 2]))
for t, k in samples[:: max(1, len(samples) // 4)]:
    print('  t=%.1fs %s' % (t - t0, ' | '.join(k)))

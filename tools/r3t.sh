#!/bin/bash
tag=${1:-r3t}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_ctc.py tests/test_gpu_model.py tests/test_gpu_fullsize_parity.py -k "not exact" > $out/pytest.log 2>&1 </dev/null
tail -3 $out/pytest.log
python - <<'PY' 2>&1 | grep -v amdgpu
import sys, numpy as np, torch
sys.path.insert(0, '.')
from asr_study_amd import ops
T, N, C = 999, 64, 28
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
logits = torch.randn(T, N, C, generator=g).to(dev)
rs = np.random.RandomState(0)
lens = rs.randint(2, 50, size=N)
lab = np.zeros((N, 49), np.int32)
for n in range(N):
    lab[n, :lens[n]] = rs.randint(0, 25, size=lens[n])
lab_d = torch.from_numpy(lab).to(dev); ll = torch.from_numpy(lens.astype(np.int32)).to(dev)
sl = torch.full((N,), T, dtype=torch.int32, device=dev)
grad = torch.empty_like(logits)
def run():
    ops.ctc_loss_grad(logits, lab_d, ll, sl, N, grad=grad, grad_scale=1.0 / N)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print('ctc loss+grad 999x64x28: %.3f ms per call' % (e0.elapsed_time(e1) / 10))
PY

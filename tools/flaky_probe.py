"""Repeats the cfg2 full-size forward against the fixture and prints where it deviates."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import fullsize_cases as FC
from oracle import lstm as OL
from asr_study_amd import ops
from asr_study_amd.core import models
name = 'cfg2'
fix = np.load('tests/golden/model_%s.npz' % name)
case = FC.build(name)
cfg, T = case['cfg'], case['T']
N, F, H, L, C = cfg['N'], cfg['F'], cfg['H'], cfg['L'], cfg['C']
dev = torch.device('cuda:0')
model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L, dropout=0.0,
                      weight_decay=0.0, seed=1, device=dev)
model.set_weights([a for _, a in OL.flatten(case['params'])])
slab = torch.zeros((T, ops.pad16(N), F), dtype=torch.float32, device=dev)
slab[:, :N] = torch.from_numpy(case['x']).to(dev)
fr = fix['logit_frames']
ref = None
bad = 0
reps = int(os.environ.get('REPS', '20'))
for it in range(reps):
    logits = model.forward(slab, training=False)
    torch.cuda.synchronize()
    lg = logits[:, :N].cpu().numpy()
    err = np.abs(lg[fr] - fix['logits'])
    if ref is None:
        ref = lg.copy()
    same = np.array_equal(lg, ref)
    if err.max() > 1e-4 or not same:
        bad += 1
        d = np.abs(lg - ref)
        tt, nn = np.nonzero(d.max(axis=2) > 1e-5)
        print('iter %d: max err vs fixture %.2e; differs from iter 0 at %d (t,n) pairs; t range %s n set %s'
              % (it, err.max(), len(tt), (tt.min(), tt.max()) if len(tt) else None,
                 sorted(set(nn.tolist()))[:20]), flush=True)
        # which layer first deviates?  compare layer outputs with a fresh run
for ws in ('lstm_fwd',):
    ops.lstm_status(ops.WS.get(ws, 0, dev))
print('env %s: %d of %d iterations deviated; first-iteration error %.2e'
      % ({k: v for k, v in os.environ.items() if k.startswith('ASR_')}, bad, reps,
         np.abs(ref[fr] - fix['logits']).max()))

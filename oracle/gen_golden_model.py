#!/usr/bin/env python
"""Generate tests/golden/model_<case>.npz: the float64 oracle run ONCE at the benchmarked
sizes (T = 999; cfg2 = 5 x BiLSTM(256), N = 32; cfg3 slice = 5 x BiLSTM(512), N = 16; the
same slice behind the 2-conv front-end of configs[2], T' = 500), kept
as compact fixtures so that the -m gpu suite can compare the HIP path element-wise at full
size without minutes of NumPy on the GPU box (tests/test_gpu_fullsize_parity.py).

Runs in the build container only (about 5-10 minutes and ~15 GB per case); like
oracle/gen_golden.py it is committed together with the vectors it wrote.  The inputs are
the seeded recipes of oracle/fullsize_cases.py, so the test regenerates them instead of
storing them.  The oracle for this half of the path (oracle/lstm.py, oracle/ctc.py,
oracle/decode.py) is a restatement of core/layers.py:432-469, core/ctc_utils.py:8-70 and
Keras-1.2.2 / TF-1.3 semantics: PARITY UNPINNED by the reference (those wheels cannot run
here) -- these fixtures pin the HIP path on the oracle, not the oracle on the reference.

Per case the fixture keeps: the logits of 50 frames spread over T for every utterance; the
per-utterance CTC loss; every frame's argmax and top-1/top-2 margin (float64 oracle) and
the greedy decode; per gradient tensor its L2 norm, max |g|, 1000 sampled entries and
(r6) the sums over every block of 256 contiguous entries (gradient of mean_n ctc_n, no l2 term); layer-1 and layer-L hidden / cell states of the
first 8 utterances at t in {0, 1, T/2, T-2, T-1}; 64 probe values of the input features.

    python oracle/gen_golden_model.py [case ...]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')

from oracle import decode as OD          # noqa: E402
from oracle import fullsize_cases as FC  # noqa: E402
from oracle import lstm as OL            # noqa: E402


def to64(tree):
    if isinstance(tree, dict):
        # (a conv layer's 'stride' / 'clip' are settings, not arrays)
        return {k: (v if k in ('stride', 'clip') else to64(v)) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(to64(v) for v in tree)
    return np.asarray(tree, np.float64)


def run(name):
    t0 = time.time()
    case = FC.build(name)
    cfg, T = case['cfg'], case['T']
    N, H, L = cfg['N'], cfg['H'], cfg['L']
    params = to64(case['params'])
    masks = None if case['masks'] is None else \
        [{d: tuple(np.asarray(a, np.float64) for a in m[d]) for d in m} for m in case['masks']]
    x = case['x'].astype(np.float64)
    print('[%s] inputs built in %.0f s; running the float64 oracle ...' % (name, time.time() - t0),
          flush=True)
    ns = int(cfg.get('slices', 1))
    if ns == 1:
        out = OL.loss_and_grads(params, x, case['labels'], case['lens'], weight_decay=0.0,
                                masks=masks)
    else:
        # independent batch rows: slice the batch, add the gradients (each slice's gradient is
        # that of ITS mean: weight n_slice / N); states are kept for the first slice only
        assert masks is None and N % ns == 0 and N // ns >= FC.STATE_UTTS
        per = N // ns
        out = None
        for k in range(ns):
            sl = slice(k * per, (k + 1) * per)
            o = OL.loss_and_grads(params, x[:, sl], case['labels'][sl], case['lens'][sl],
                                  weight_decay=0.0)
            gl = [(n, np.asarray(g, np.float64) * (per / float(N))) for n, g in OL.flatten(o['grads'])]
            if out is None:
                out = dict(ctc=list(o['ctc']), logits=[o['logits']], glist=gl,
                           caches={'layers': [{d: {k2: lc[d][k2][:, :FC.STATE_UTTS].copy()
                                                   for k2 in ('hs', 'cs')} for d in ('fwd', 'bwd')}
                                              for lc in o['caches']['layers']]})
            else:
                out['ctc'] += list(o['ctc'])
                out['logits'].append(o['logits'])
                out['glist'] = [(n, a + b) for (n, a), (_, b) in zip(out['glist'], gl)]
            del o
            print('[%s] slice %d/%d done (%.0f s)' % (name, k + 1, ns, time.time() - t0), flush=True)
        out['logits'] = np.concatenate(out['logits'], axis=1)
    logits = out['logits']
    fix = {}
    lens_out = FC.out_frames(cfg, case['lens'])     # (a conv front-end shortens the time axis)
    T = int(logits.shape[0])
    fr = FC.logit_frames(T)
    fix['logit_frames'] = fr
    fix['logits'] = logits[fr].astype(np.float32)
    fix['ctc'] = np.asarray(out['ctc'], np.float64)
    srt = np.sort(logits, axis=-1)
    fix['argmax'] = np.argmax(logits, axis=-1).astype(np.uint8)
    fix['margin'] = (srt[..., -1] - srt[..., -2]).astype(np.float32)
    hyp = OD.greedy_decode(logits, lens_out)
    hl = max([len(h) for h in hyp] + [1])
    dec = np.full((N, hl), -1, np.int16)
    for n, h in enumerate(hyp):
        dec[n, :len(h)] = h
    fix['greedy'] = dec
    fix['greedy_len'] = np.array([len(h) for h in hyp], np.int32)
    glist = out['glist'] if 'glist' in out else OL.flatten(out['grads'])
    for i, (gname, g) in enumerate(glist):
        flat = np.asarray(g, np.float64).reshape(-1)
        idx = FC.grad_sample_index(i, flat.size)
        fix['g%02d_samples' % i] = flat[idx]
        fix['g%02d_stats' % i] = np.array([np.sqrt(np.sum(flat ** 2)), np.abs(flat).max()])
        # every entry in one checksum: sums over FC.GRAD_BLOCK contiguous entries (float32 is
        # 6e-8 relative: far below the 1e-4 max|g| the test allows per entry)
        fix['g%02d_blocks' % i] = FC.grad_block_sums(flat).astype(np.float32)
    fix['grad_names'] = np.array([n for n, _ in glist])
    sf = FC.state_frames(T)
    for li in (0, L - 1):
        lc = out['caches']['layers'][li]
        nu = FC.STATE_UTTS
        fix['h_l%d' % li] = np.concatenate([lc['fwd']['hs'][sf][:, :nu], lc['bwd']['hs'][sf][:, :nu]],
                                           axis=-1).astype(np.float32)
        fix['c_l%d' % li] = np.concatenate([lc['fwd']['cs'][sf][:, :nu], lc['bwd']['cs'][sf][:, :nu]],
                                           axis=-1).astype(np.float32)
    idx, vals = FC.feature_probe(case['x'])
    fix['feat_probe'] = vals.astype(np.float32)
    fix['lens'] = case['lens']
    path = os.path.join(OUT, 'model_%s.npz' % name)
    np.savez_compressed(path, **fix)
    print('[%s] loss mean %.6f, min margin %.3e, frames with margin < 2e-4: %d, wrote %s '
          '(%.0f KB) in %.0f s' % (name, float(np.mean(out['ctc'])), float(fix['margin'].min()),
                                    int((fix['margin'] < 2e-4).sum()), path,
                                    os.path.getsize(path) / 1024.0, time.time() - t0), flush=True)


if __name__ == '__main__':
    for name in (sys.argv[1:] or sorted(FC.CASES)):
        run(name)

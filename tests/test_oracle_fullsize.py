"""The full-size model fixtures (tests/golden/model_*.npz, written by
oracle/gen_golden_model.py) against their seeded input recipes -- CPU, seconds: the
regenerated features match the fixture's probe values, and the first two recurrent steps of
layer 1 (cheap to recompute in float64) reproduce the stored hidden / cell states."""
import os

import numpy as np
import pytest

from oracle import fullsize_cases as FC
from oracle import lstm as OL

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('name', sorted(FC.CASES))
def test_fixture_matches_its_recipe(name):
    fix = np.load(os.path.join(GOLDEN, 'model_%s.npz' % name))
    case = FC.build(name)
    cfg, T = case['cfg'], case['T']
    N, H, C, L = cfg['N'], cfg['H'], cfg['C'], cfg['L']
    idx, probe = FC.feature_probe(case['x'])
    assert np.abs(probe - fix['feat_probe']).max() < 1e-5
    assert np.array_equal(case['lens'], fix['lens'])
    nu = FC.STATE_UTTS
    x = case['x'][:, :nu].astype(np.float64)
    conv = case['params'].get('conv')
    if conv:
        # the stack's input is the conv front-end's output: recompute its first and last frames
        # from crops of the features that keep the 'same' padding and the stride phase of the
        # whole slab (T odd: a crop [918, 999) of odd length starts on an even frame)
        from oracle import conv as OCV

        def front(a):
            for cp in conv:
                a, _ = OCV.conv2d_forward(a, cp['W'].astype(np.float64), cp['b'].astype(np.float64),
                                          cp['stride'], cp['clip'])
            return a
        assert T % 2 == 1
        head, tail = front(x[:81]), front(x[918:])
        T = int(FC.out_frames(cfg, T))
        xs_full = np.zeros((T, nu, head.shape[2]))
        xs_full[:2], xs_full[-2:] = head[:2], tail[-2:]
        x = xs_full
    assert fix['logits'].shape == (len(FC.logit_frames(T)), N, C)
    assert fix['ctc'].shape == (N,) and np.all(fix['ctc'] > 0)
    assert fix['argmax'].shape == (T, N) and fix['margin'].shape == (T, N)
    assert len(fix['grad_names']) == 6 * L + 2 + 2 * len(conv or [])
    # layer 1, forward direction, steps t = 0 and 1; backward direction, t = T-1 and T-2
    for d, rev, frames, rows in (('fwd', False, [0, 1], [0, 1]), ('bwd', True, [T - 1, T - 2], [4, 3])):
        p = {k: np.asarray(v, np.float64) for k, v in case['params']['layers'][0][d].items()}
        BW = BU = None
        if case['masks'] is not None:
            BW, BU = [np.asarray(m[:nu], np.float64) for m in case['masks'][0][d]]
        xs = x[frames] if not rev else x[frames[::-1]]
        hs, cache = OL.lstm_forward(xs, p['W'], p['U'], p['b'], rev, BW, BU)
        sl = slice(0, H) if d == 'fwd' else slice(H, 2 * H)
        for k, (t, row) in enumerate(zip(frames, rows)):
            i = k if not rev else 1 - k
            assert np.abs(hs[i] - fix['h_l0'][row][:, sl]).max() < 1e-6
            assert np.abs(cache['cs'][i] - fix['c_l0'][row][:, sl]).max() < 1e-6

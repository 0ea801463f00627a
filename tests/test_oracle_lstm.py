"""oracle/lstm.py BPTT against torch.autograd over the same plain tensor ops."""
import numpy as np
import torch

from oracle import lstm as L
from oracle import ctc as C


_TORCH_ACT = {
    'tanh': torch.tanh, 'relu': torch.relu, 'sigmoid': torch.sigmoid,
    'hard_sigmoid': lambda v: torch.clamp(0.2 * v + 0.5, 0.0, 1.0), 'linear': lambda v: v,
    'softsign': torch.nn.functional.softsign, 'softplus': torch.nn.functional.softplus,
}


def _torch_model(params, x, masks=None, act='tanh'):
    fact = _TORCH_ACT[act]

    def hs(v):
        return torch.clamp(0.2 * v + 0.5, 0.0, 1.0)

    def run(o, p, rev, BW, BU):
        T, N, _ = o.shape
        H = p['U'].shape[0]
        h = torch.zeros(N, H, dtype=o.dtype); c = torch.zeros(N, H, dtype=o.dtype)
        outs = [None] * T
        for t in (range(T - 1, -1, -1) if rev else range(T)):
            xt = o[t] if BW is None else o[t] * BW
            hm = h if BU is None else h * BU
            z = xt @ p['W'] + hm @ p['U'] + p['b']
            i, f = hs(z[:, :H]), hs(z[:, H:2 * H])
            g, oo = fact(z[:, 2 * H:3 * H]), hs(z[:, 3 * H:])
            c = f * c + i * g
            h = oo * fact(c)
            outs[t] = h
        return torch.stack(outs)

    o = x
    if 'in_dense' in params:
        o = o @ params['in_dense']['W'] + params['in_dense']['b']
    for li, layer in enumerate(params['layers']):
        outs = []
        for d, rev in (('fwd', False), ('bwd', True)):
            BW = BU = None
            if masks is not None:
                BW, BU = [torch.tensor(m) for m in masks[li][d]]
            outs.append(run(o, layer[d], rev, BW, BU))
        o = torch.cat(outs, -1)
    return o @ params['dense']['W'] + params['dense']['b']


def _to_torch(tree):
    if isinstance(tree, dict):
        return {k: (v if isinstance(v, str) else _to_torch(v)) for k, v in tree.items()}
    if isinstance(tree, list):
        return [_to_torch(v) for v in tree]
    return torch.tensor(tree, dtype=torch.float64, requires_grad=True)


def _check(in_dense, use_masks, act='tanh'):
    rs = np.random.RandomState(3)
    T, N, F, H, Cc = 13, 3, 5, 4, 6
    params = L.init_model(seed=1, num_features=F, num_hiddens=H, num_layers=2,
                          num_classes=Cc, dtype=np.float64, in_dense=in_dense)
    # make biases / weights generic (exercise every gradient path)
    for name, a in L.flatten(params):
        a += rs.randn(*a.shape) * 0.3
    x = rs.randn(T, N, F)
    x[9:, 1] = 0.0                                   # padded tail of sample 1
    seq_len = [T, 9, T]
    labels = [[0, 1, 1], [2], [4, 3, 0, 0]]
    masks = None
    if use_masks:
        masks = []
        n_in = in_dense or F
        for li in range(2):
            m = {}
            for d in ('fwd', 'bwd'):
                m[d] = ((rs.rand(N, n_in) > 0.2) / 0.8, (rs.rand(N, H) > 0.2) / 0.8)
            masks.append(m)
            n_in = 2 * H
    wd = 1e-2
    if act != 'tanh':
        params['activation'] = act
    out = L.loss_and_grads(params, x, labels, seq_len, weight_decay=wd, masks=masks)

    tp = _to_torch(params)
    logits = _torch_model(tp, torch.tensor(x), masks, act)
    lp = torch.log_softmax(logits, -1)
    tgt = torch.tensor(sum(labels, []))
    ctc = torch.nn.functional.ctc_loss(lp, tgt, torch.tensor(seq_len),
                                       torch.tensor([len(l) for l in labels]),
                                       blank=Cc - 1, reduction='none')
    l2 = sum((tp['layers'][li][d][k] ** 2).sum() for li in range(2)
             for d in ('fwd', 'bwd') for k in ('W', 'U')) + (tp['dense']['W'] ** 2).sum()
    loss = ctc.mean() + wd * l2
    loss.backward()
    np.testing.assert_allclose(out['logits'], logits.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(out['loss'], float(loss.detach()), rtol=1e-12)
    for (name, g), (_, tg) in zip(L.flatten(out['grads']), L.flatten(tp)):
        np.testing.assert_allclose(g, tg.grad.numpy(), atol=1e-10, err_msg=name)


def test_bptt_matches_autograd_plain():
    _check(None, False)


def test_bptt_matches_autograd_masks_and_in_dense():
    _check(7, True)


import pytest  # noqa: E402


@pytest.mark.parametrize('act', ['relu', 'sigmoid', 'hard_sigmoid', 'linear', 'softsign', 'softplus'])
def test_bptt_matches_autograd_for_every_activation(act):
    """The LSTM's ``activation`` hyper-parameter (core/layers.py:452, :463; brsmv1 passes it on,
    core/models.py:220, :271): g = act(z_c), h = o * act(c), BPTT through act' written in terms
    of the OUTPUT of act (what the kernels keep) -- against torch.autograd, with masks."""
    _check(None, True, act)


def test_backward_direction_sees_padding():
    """No Masking layer: the reverse direction must consume the zero tail first,
    so its state at the last real frame is NOT the zero initial state."""
    rs = np.random.RandomState(0)
    p = L.init_lstm(rs, 3, 4, np.float64)
    p['b'] += rs.randn(16) * 0.5    # at init b_c = 0 keeps the state at exactly 0
    x = rs.randn(6, 1, 3); x[4:] = 0.0
    hs, _ = L.lstm_forward(x, p['W'], p['U'], p['b'], reverse=True)
    hs_cut, _ = L.lstm_forward(x[:4], p['W'], p['U'], p['b'], reverse=True)
    assert np.abs(hs[3] - hs_cut[3]).max() > 1e-3      # forget bias 1 -> c != 0
    assert np.abs(hs[5]).max() > 0                      # zero input still moves


def test_init_shapes_and_forget_bias():
    p = L.init_model(seed=0, num_features=39, num_hiddens=8, num_layers=2)
    l0 = p['layers'][0]['fwd']
    assert l0['W'].shape == (39, 32) and l0['U'].shape == (8, 32)
    assert np.all(l0['b'][8:16] == 1) and l0['b'].sum() == 8
    u = l0['U'].astype(np.float64)
    np.testing.assert_allclose(u @ u.T, 1.21 * np.eye(8), atol=1e-5)
    assert p['layers'][1]['fwd']['W'].shape == (16, 32)
    assert p['dense']['W'].shape == (16, 28)


def test_residual_merge_gradients_match_finite_differences():
    """brsmv1(residual='sum'|'ave') (core/models.py:253-255,273-276): in-Dense to 2H, then
    o = merge([BiLSTM(o), o]); the oracle's backward is checked against central
    differences of sum(logits * w)."""
    import copy
    rs = np.random.RandomState(0)
    T, N, F, H, C = 7, 3, 5, 4, 6
    p = L.init_model(seed=1, num_features=F, num_hiddens=H, num_layers=2, num_classes=C,
                      dtype=np.float64)
    p['in_dense'] = {'W': rs.randn(F, 2 * H) * 0.3, 'b': rs.randn(2 * H) * 0.1}
    for d in ('fwd', 'bwd'):
        p['layers'][0][d]['W'] = rs.randn(2 * H, 4 * H) * 0.3
    x = rs.randn(T, N, F)
    for mode in ('sum', 'ave'):
        p['residual'] = mode
        logits, caches = L.model_forward(p, x)
        w = rs.randn(*logits.shape)
        g = L.model_backward(p, caches, w)

        def loss(q):
            return float((L.model_forward(q, x)[0] * w).sum())
        for path in (('in_dense', 'W'), ('layers', 0, 'fwd', 'U'), ('layers', 1, 'bwd', 'W'),
                     ('dense', 'W'), ('in_dense', 'b')):
            q = copy.deepcopy(p)
            arr, gr = q, g
            for k in path[:-1]:
                arr, gr = arr[k], gr[k]
            arr, gr = arr[path[-1]], gr[path[-1]]
            idx = tuple(rs.randint(0, d) for d in arr.shape)
            arr[idx] += 1e-6
            lp = loss(q)
            arr[idx] -= 2e-6
            lm = loss(q)
            num = (lp - lm) / 2e-6
            assert abs(gr[idx] - num) < 1e-5 * max(1.0, abs(num)), (mode, path)


def test_multiplicative_integration_and_zoneout_gradients_match_finite_differences():
    """Cell variants of the reference override (core/layers.py:441-443 MI, :457-467 zoneout
    with per-frame coefficients): the oracle's BPTT -- dx, dW, dU, db and d alpha / d beta1 /
    d beta2 -- against central differences, both directions, with dropout masks."""
    rs = np.random.RandomState(3)
    T, N, F, H = 6, 3, 4, 5
    x = rs.randn(T, N, F)
    W, U, b = rs.randn(F, 4 * H) * 0.4, rs.randn(H, 4 * H) * 0.4, rs.randn(4 * H) * 0.2
    mi = [1 + 0.3 * rs.randn(4 * H), 0.5 + 0.3 * rs.randn(4 * H), 0.5 + 0.3 * rs.randn(4 * H)]
    kc = (rs.rand(T, H) > 0.3).astype(float)
    kh = np.full((T, H), 0.8)
    BU, BW = (rs.rand(N, H) > 0.2) / 0.8, (rs.rand(N, F) > 0.2) / 0.8
    w = rs.randn(T, N, H)
    for rev in (False, True):
        def loss():
            return float((L.lstm_forward(x, W, U, b, rev, BW, BU, mi, kc, kh)[0] * w).sum())
        hs, cache = L.lstm_forward(x, W, U, b, rev, BW, BU, mi, kc, kh)
        dx, dW, dU, db = L.lstm_backward(w.copy(), cache)
        for name, arr, g in (('x', x, dx), ('W', W, dW), ('U', U, dU), ('b', b, db),
                             ('alpha', mi[0], cache['dmi'][0]), ('beta1', mi[1], cache['dmi'][1]),
                             ('beta2', mi[2], cache['dmi'][2])):
            for _ in range(3):
                idx = tuple(rs.randint(0, d) for d in arr.shape)
                o = arr[idx]
                arr[idx] = o + 1e-6
                lp = loss()
                arr[idx] = o - 1e-6
                lm = loss()
                arr[idx] = o
                num = (lp - lm) / 2e-6
                assert abs(g[idx] - num) < 1e-5 * max(1.0, abs(num)), (name, idx)
    # zoneout at k = 1 and mi = (0, 1, 1) reduce to the plain cell
    plain = L.lstm_forward(x, W, U, b, False, BW, BU)[0]
    ident = L.lstm_forward(x, W, U, b, False, BW, BU,
                           [np.zeros(4 * H), np.ones(4 * H), np.ones(4 * H)],
                           np.ones((T, H)), np.ones((T, H)))[0]
    assert np.allclose(plain, ident, atol=1e-12)


def test_layer_normalisation_gradients_match_finite_differences():
    """layer_norm option of the reference override (core/layers.py:407-436, 460-462;
    core/layers_utils.py:16-19): LN of h@U, of x@W and of the cell state feeding the output,
    with and without multiplicative integration, plus zoneout and dropout masks."""
    rs = np.random.RandomState(5)
    T, N, F, H = 5, 3, 4, 6
    x = rs.randn(T, N, F)
    W, U, b = rs.randn(F, 4 * H) * 0.5, rs.randn(H, 4 * H) * 0.5, rs.randn(4 * H) * 0.2
    mi = [1 + 0.3 * rs.randn(4 * H), 0.5 + 0.3 * rs.randn(4 * H), 0.5 + 0.3 * rs.randn(4 * H)]
    ln = {'Uh': [1 + 0.2 * rs.randn(4 * H), 0.1 * rs.randn(4 * H)],
          'Wx': [1 + 0.2 * rs.randn(4 * H), 0.1 * rs.randn(4 * H)],
          'new_c': [1 + 0.2 * rs.randn(H), 0.1 * rs.randn(H)]}
    kc = (rs.rand(T, H) > 0.3).astype(float)
    kh = np.full((T, H), 0.8)
    BU, BW = (rs.rand(N, H) > 0.2) / 0.8, (rs.rand(N, F) > 0.2) / 0.8
    w = rs.randn(T, N, H)
    for use_mi in (None, mi):
        for rev in (False, True):
            def loss():
                return float((L.lstm_forward(x, W, U, b, rev, BW, BU, use_mi, kc, kh, ln)[0] * w).sum())
            hs, c = L.lstm_forward(x, W, U, b, rev, BW, BU, use_mi, kc, kh, ln)
            dx, dW, dU, db = L.lstm_backward(w.copy(), c)
            items = [('x', x, dx), ('W', W, dW), ('U', U, dU), ('b', b, db)]
            for k in ('Uh', 'Wx', 'new_c'):
                items += [(k + '_gain', ln[k][0], c['dln'][k][0]), (k + '_bias', ln[k][1], c['dln'][k][1])]
            if use_mi is not None:
                items += [('alpha', mi[0], c['dmi'][0]), ('beta1', mi[1], c['dmi'][1]),
                          ('beta2', mi[2], c['dmi'][2])]
            for name, arr, g in items:
                for _ in range(2):
                    idx = tuple(rs.randint(0, d) for d in arr.shape)
                    o = arr[idx]
                    arr[idx] = o + 1e-6
                    lp = loss()
                    arr[idx] = o - 1e-6
                    lm = loss()
                    arr[idx] = o
                    num = (lp - lm) / 2e-6
                    assert abs(g[idx] - num) < 2e-5 * max(1.0, abs(num)), (name, idx)

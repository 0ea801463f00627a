"""Oracle: LSTM / BiLSTM / TimeDistributed(Dense) forward + BPTT, and the
ctc_model() loss graph.  TEST INFRASTRUCTURE ONLY.

Follows core/layers.py:432-469 (``LSTM.step``: fused (in,4H)/(H,4H) matmuls,
column blocks i,f,c,o; variational masks B_W/B_U multiply x / h before the
matmul), core/models.py:31-52 (ctc_model wiring) and core/models.py:217-281
(brsmv1: L x Bidirectional(LSTM) concat -> TimeDistributed(Dense), l2 on W/U and
the Dense kernel).  The enclosing machinery is keras==1.2.2 (msc.yaml:89), which
is NOT under /root/reference: PARITY UNPINNED by the reference.  Recalled Keras
semantics restated here:

* inner_activation = hard_sigmoid = clip(0.2 x + 0.5, 0, 1); activation = tanh;
* h0 = c0 = 0; b initialised 0 with the forget block = 1;
* Bidirectional(concat): the backward copy consumes the PADDED batch from T-1
  down to 0 (no Masking layer is applied, core/models.py:20 imports it unused),
  its outputs are reversed back and concatenated [fwd, bwd] on the feature axis;
* loss = mean_n ctc_n + sum l2 * sum(w^2)  (loss_weights [1, 0], train.py:140-143).

All tensors are time-major (T, N, .) -- the reference's (N, T, .) transposed.
Cross-checked against torch.autograd in tests/test_oracle_lstm.py.
"""
import numpy as np

from . import conv as _conv
from . import ctc as _ctc


def hard_sigmoid(x):
    return np.clip(0.2 * x + 0.5, 0.0, 1.0)


def hard_sigmoid_grad(x):
    y = 0.2 * x + 0.5
    return np.where((y >= 0.0) & (y <= 1.0), 0.2, 0.0).astype(x.dtype)


# The ``activation`` hyper-parameter of the reference's LSTM (core/layers.py:452, :463: the cell
# candidate g = activation(z_c) and the output h = o * activation(c); brsmv1 passes it through,
# core/models.py:220, :271): a Keras-1.2.2 activation NAME [recalled: keras/activations.py].
# Each entry: (f(x), f'(x) expressed through y = f(x)) -- the kernels keep only y.
ACTIVATIONS = {
    'tanh': (np.tanh, lambda y: 1.0 - y * y),
    'relu': (lambda x: np.maximum(x, 0.0), lambda y: (y > 0.0).astype(y.dtype)),
    'sigmoid': (lambda x: 1.0 / (1.0 + np.exp(-x)), lambda y: y * (1.0 - y)),
    'hard_sigmoid': (hard_sigmoid,
                     lambda y: np.where((y > 0.0) & (y < 1.0), 0.2, 0.0).astype(y.dtype)),
    'linear': (lambda x: x, lambda y: np.ones_like(y)),
    'softsign': (lambda x: x / (1.0 + np.abs(x)), lambda y: (1.0 - np.abs(y)) ** 2),
    'softplus': (lambda x: np.logaddexp(0.0, x), lambda y: 1.0 - np.exp(-y)),
}
LN_EPS = 1e-5       # core/layers_utils.py:16


def layer_norm(x, gain, bias):
    """core/layers_utils.py:16-19: moments over axis 1 (biased variance), eps inside the
    square root.  Returns (y, xhat, rstd)."""
    mu = x.mean(axis=1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + LN_EPS)
    xhat = (x - mu) * rstd
    return xhat * gain + bias, xhat, rstd


def layer_norm_backward(dy, xhat, rstd, gain):
    """-> (dx, dgain, dbias) for y = xhat * gain + bias."""
    g = dy * gain
    dx = rstd * (g - g.mean(axis=1, keepdims=True) - xhat * (g * xhat).mean(axis=1, keepdims=True))
    return dx, (dy * xhat).sum(axis=0), dy.sum(axis=0)


def lstm_forward(x, W, U, b, reverse=False, BW=None, BU=None, mi=None, kc=None, kh=None,
                 ln=None, act='tanh'):
    """One direction.  x (T,N,in); W (in,4H); U (H,4H); b (4H).

    BW (N,in) / BU (N,H): variational dropout masks (already scaled by 1/(1-p))
    or None.  Optional cell variants of the reference override (core/layers.py:432-469):
    ``mi`` = (alpha, beta1, beta2), each (4H,): multiplicative integration
    z = alpha*Wx*Uh + beta1*Uh + beta2*Wx + b (:441-443, core/layers_utils.py:45-51);
    ``kc`` / ``kh`` (T,H): zoneout coefficients of the cell / hidden state,
    new = prev + k * (candidate - prev) -- k is the per-step keep-mask (shared over the
    batch, noise_shape=(H,)) in training and the constant 1 - level at test time
    (core/layers_utils.py:34-42; :457-467).  ``ln`` = dict with (gain, bias) pairs under
    'Uh' (4H), 'Wx' (4H), 'new_c' (H): layer normalisation of h@U, of x@W and of the cell
    state that feeds the output (:432-436, :460-462; the carried c stays un-normalised).
    ``act``: the ``activation`` name (ACTIVATIONS), applied to the cell candidate and to the
    cell state that feeds the output (:452, :463).
    Returns (h_seq (T,N,H), cache).
    """
    fact = ACTIVATIONS[act][0]
    T, N, _ = x.shape
    H = U.shape[0]
    dt = x.dtype
    xs = x if BW is None else x * BW[None]
    h = np.zeros((N, H), dt)
    c = np.zeros((N, H), dt)
    hs = np.zeros((T, N, H), dt)
    cs = np.zeros((T, N, H), dt)
    gates = np.zeros((T, N, 4 * H), dt)     # post-activation i,f,g,o
    zs = np.zeros((T, N, 4 * H), dt)        # pre-activation
    uhs = np.zeros((T, N, 4 * H), dt)       # h_prev @ U   (after LN when ln is set)
    wxs = np.zeros((T, N, 4 * H), dt)       # x @ W        (after LN when ln is set)
    lnc = {}                                # per frame LN caches
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        hm = h if BU is None else h * BU
        wx, uh = xs[t] @ W, hm @ U
        if ln is not None:
            uh, uhat, urstd = layer_norm(uh, *ln['Uh'])
            wx, what, wrstd = layer_norm(wx, *ln['Wx'])
        if mi is not None:
            z = mi[0] * wx * uh + mi[1] * uh + mi[2] * wx + b
        else:
            z = wx + uh + b
        i = hard_sigmoid(z[:, :H])
        f = hard_sigmoid(z[:, H:2 * H])
        g = fact(z[:, 2 * H:3 * H])
        o = hard_sigmoid(z[:, 3 * H:])
        c_new = f * c + i * g
        if kc is not None:
            c_new = c + kc[t][None] * (c_new - c)
        if ln is not None:
            cn, chat, crstd = layer_norm(c_new, *ln['new_c'])
            lnc[t] = (uhat, urstd, what, wrstd, chat, crstd, cn)
        else:
            cn = c_new
        h_new = o * fact(cn)
        if kh is not None:
            h_new = h + kh[t][None] * (h_new - h)
        h, c = h_new, c_new
        hs[t], cs[t], zs[t], uhs[t], wxs[t] = h, c, z, uh, wx
        gates[t] = np.concatenate([i, f, g, o], axis=1)
    cache = dict(x=x, xs=xs, W=W, U=U, BW=BW, BU=BU, hs=hs, cs=cs, zs=zs,
                 gates=gates, reverse=reverse, mi=mi, kc=kc, kh=kh, uhs=uhs, wxs=wxs,
                 ln=ln, lnc=lnc, act=act)
    return hs, cache


def lstm_backward(dhs, cache):
    """BPTT for one direction.  dhs (T,N,H) -> dx, dW, dU, db (and, with multiplicative
    integration, cache['dmi'] = (dalpha, dbeta1, dbeta2))."""
    x, xs, W, U = cache['x'], cache['xs'], cache['W'], cache['U']
    BW, BU = cache['BW'], cache['BU']
    hs, cs, zs, gates = cache['hs'], cache['cs'], cache['zs'], cache['gates']
    mi, kc, kh = cache.get('mi'), cache.get('kc'), cache.get('kh')
    ln, lnc = cache.get('ln'), cache.get('lnc')
    dln = None if ln is None else {k: [np.zeros_like(v[0]), np.zeros_like(v[1])]
                                   for k, v in ln.items()}
    reverse = cache['reverse']
    fact, dact = ACTIVATIONS[cache.get('act', 'tanh')]
    T, N, H = hs.shape
    dt = x.dtype
    dW = np.zeros_like(W); dU = np.zeros_like(U); db = np.zeros(4 * H, dt)
    dmi = [np.zeros(4 * H, dt) for _ in range(3)]
    dxs = np.zeros_like(xs)
    dzs = np.zeros((T, N, 4 * H), dt)
    das = np.zeros((T, N, 4 * H), dt)
    dwxs = np.zeros((T, N, 4 * H), dt)
    dh_next = np.zeros((N, H), dt)
    dc_next = np.zeros((N, H), dt)
    order = list(range(T - 1, -1, -1) if reverse else range(T))
    for k in range(T - 1, -1, -1):
        t = order[k]
        tp = order[k - 1] if k > 0 else None
        h_prev = hs[tp] if tp is not None else np.zeros((N, H), dt)
        c_prev = cs[tp] if tp is not None else np.zeros((N, H), dt)
        i, f = gates[t][:, :H], gates[t][:, H:2 * H]
        g, o = gates[t][:, 2 * H:3 * H], gates[t][:, 3 * H:]
        z = zs[t]
        dh = dhs[t] + dh_next
        dh_zone = 0.0
        if kh is not None:                  # h = h_prev + kh (h~ - h_prev)
            dh_zone = (1.0 - kh[t][None]) * dh
            dh = kh[t][None] * dh
        if ln is not None:
            uhat, urstd, what, wrstd, chat, crstd, cn = lnc[t]
            tc = fact(cn)
            dcn = dh * o * dact(tc)
            dcl, dg_, db_ = layer_norm_backward(dcn, chat, crstd, ln['new_c'][0])
            dln['new_c'][0] += dg_; dln['new_c'][1] += db_
        else:
            tc = fact(cs[t])
            dcl = dh * o * dact(tc)
        do = dh * tc
        dc = dc_next + dcl
        dc_zone = 0.0
        if kc is not None:                  # c = c_prev + kc (c~ - c_prev)
            dc_zone = (1.0 - kc[t][None]) * dc
            dc = kc[t][None] * dc
        di, dg, df = dc * g, dc * i, dc * c_prev
        dc_next = dc * f + dc_zone
        dz = np.concatenate([
            di * hard_sigmoid_grad(z[:, :H]),
            df * hard_sigmoid_grad(z[:, H:2 * H]),
            dg * dact(g),
            do * hard_sigmoid_grad(z[:, 3 * H:])], axis=1)
        dzs[t] = dz
        if mi is not None:
            wx, uh = cache['wxs'][t], cache['uhs'][t]
            da = dz * (mi[0] * wx + mi[1])      # d loss / d (h_prev @ U)
            dwx = dz * (mi[0] * uh + mi[2])     # d loss / d (x @ W)
            dmi[0] += (dz * wx * uh).sum(axis=0)
            dmi[1] += (dz * uh).sum(axis=0)
            dmi[2] += (dz * wx).sum(axis=0)
        else:
            da = dwx = dz
        if ln is not None:      # back through the two input normalisations
            da, dg_, db_ = layer_norm_backward(da, uhat, urstd, ln['Uh'][0])
            dln['Uh'][0] += dg_; dln['Uh'][1] += db_
            dwx, dg_, db_ = layer_norm_backward(dwx, what, wrstd, ln['Wx'][0])
            dln['Wx'][0] += dg_; dln['Wx'][1] += db_
        das[t], dwxs[t] = da, dwx
        hm = h_prev if BU is None else h_prev * BU
        dW += xs[t].T @ dwx
        dU += hm.T @ da
        db += dz.sum(axis=0)
        dxs[t] = dwx @ W.T
        dhm = da @ U.T
        dh_next = (dhm if BU is None else dhm * BU) + dh_zone
    dx = dxs if BW is None else dxs * BW[None]
    cache['dzs'] = dzs          # gate pre-activation gradients (kernel parity tests)
    cache['das'], cache['dwxs'] = das, dwxs
    cache['dmi'] = dmi if mi is not None else None
    cache['dln'] = dln
    return dx, dW, dU, db


def init_lstm(rs, n_in, H, dtype=np.float32):
    """Keras-1.2.2 consume_less='gpu' init [recalled, SURVEY.md a17]: W
    glorot_uniform over the fused (in,4H) shape, U(+-sqrt(6/(in+4H))); U
    orthogonal over the fused (H,4H) shape (SVD of a normal matrix, the factor
    whose shape matches, gain 1.1); b zeros with the f block = 1."""
    lim = np.sqrt(6.0 / (n_in + 4 * H))
    W = rs.uniform(-lim, lim, size=(n_in, 4 * H))
    a = rs.normal(0.0, 1.0, (H, 4 * H))
    u, _, v = np.linalg.svd(a, full_matrices=False)
    U = 1.1 * (u if u.shape == (H, 4 * H) else v)
    b = np.zeros(4 * H); b[H:2 * H] = 1.0
    return dict(W=W.astype(dtype), U=U.astype(dtype), b=b.astype(dtype))


def init_dense(rs, n_in, n_out, dtype=np.float32):
    lim = np.sqrt(6.0 / (n_in + n_out))
    return dict(W=rs.uniform(-lim, lim, size=(n_in, n_out)).astype(dtype),
                b=np.zeros(n_out, dtype))


def init_model(seed=0, num_features=39, num_hiddens=256, num_layers=5,
               num_classes=28, dtype=np.float32, in_dense=None, conv=None):
    """Parameter pytree for brsmv1 / graves2006 (in_dense=None), eyben, or -- conv = list of
    (C_out, kt, kf, st, sf, clip) -- the deep_speech2 topology with its convolution front-end."""
    rs = np.random.RandomState(seed)
    params = {'layers': []}
    n_in = num_features
    if conv:
        params['conv'] = []
        F, Ci = num_features, 1
        for (Co, kt, kf, st, sf, clip) in conv:
            W, b = _conv.init_conv(rs, kt, kf, Ci, Co, dtype)
            params['conv'].append({'W': W, 'b': b, 'stride': (st, sf), 'clip': clip})
            F, Ci = -(-F // sf), Co
        n_in = F * Ci
    if in_dense:
        params['in_dense'] = init_dense(rs, n_in, in_dense, dtype)
        n_in = in_dense
    hid = num_hiddens if isinstance(num_hiddens, (list, tuple)) \
        else [num_hiddens] * num_layers
    for H in hid:
        params['layers'].append({'fwd': init_lstm(rs, n_in, H, dtype),
                                 'bwd': init_lstm(rs, n_in, H, dtype)})
        n_in = 2 * H
    params['dense'] = init_dense(rs, n_in, num_classes, dtype)
    return params


def model_forward(params, x, masks=None, zone=None):
    """x (T,N,F) -> logits (T,N,C), caches.  masks[l][dir] = (BW, BU) or None;
    zone[l][dir] = (kc, kh) zoneout coefficients ((T,H) each, or None); a direction's
    parameter dict may carry 'mi' = [alpha, beta1, beta2] (multiplicative integration);
    params['activation'] names the LSTM activation (default 'tanh')."""
    caches = {'layers': []}
    o = x
    # build-defined 2-D convolution front-end (oracle/conv.py; BASELINE configs[2]): a list of
    # dict(W, b, stride=(st, sf), clip) applied to the (T, N, F*C) slab before everything else
    for cp in params.get('conv', []):
        o, cc = _conv.conv2d_forward(o, cp['W'], cp['b'], cp['stride'], cp['clip'])
        caches.setdefault('conv', []).append(cc)
    if 'in_dense' in params:
        caches['in_dense_x'] = o
        o = o @ params['in_dense']['W'] + params['in_dense']['b']
    for li, layer in enumerate(params['layers']):
        outs, lc = [], {}
        for dname, rev in (('fwd', False), ('bwd', True)):
            p = layer[dname]
            BW = BU = None
            if masks is not None and masks[li] is not None:
                BW, BU = masks[li][dname]
            kc = kh = None
            if zone is not None and zone[li] is not None:
                kc, kh = zone[li][dname]
            hs, cache = lstm_forward(o, p['W'], p['U'], p['b'], rev, BW, BU, p.get('mi'), kc, kh,
                                     ln=p.get('ln'), act=params.get('activation', 'tanh'))
            outs.append(hs); lc[dname] = cache
        caches['layers'].append(lc)
        new_o = np.concatenate(outs, axis=-1)
        # brsmv1(residual=mode): o = merge([new_o, o], mode) (core/models.py:273-276);
        # Keras-1.2.2 merge modes 'sum' (a + b) and 'ave' ((a + b) / 2)
        mode = params.get('residual')
        if mode == 'sum':
            o = new_o + o
        elif mode == 'ave':
            o = 0.5 * (new_o + o)
        elif mode is None:
            o = new_o
        else:
            raise NotImplementedError(mode)
    caches['dense_x'] = o
    logits = o @ params['dense']['W'] + params['dense']['b']
    return logits, caches


def model_backward(params, caches, dlogits):
    grads = {'layers': [None] * len(params['layers'])}
    xd = caches['dense_x']
    T, N, D = xd.shape
    grads['dense'] = {'W': xd.reshape(T * N, D).T @ dlogits.reshape(T * N, -1),
                      'b': dlogits.sum(axis=(0, 1))}
    do = dlogits @ params['dense']['W'].T
    mode = params.get('residual')
    for li in range(len(params['layers']) - 1, -1, -1):
        H = params['layers'][li]['fwd']['U'].shape[0]
        g = {}
        dx_total = None
        d_skip = None
        if mode is not None:        # o = c * (new_o + o_prev): both branches get c * do
            c = 1.0 if mode == 'sum' else 0.5
            do = c * do
            d_skip = do
        for dname, sl in (('fwd', slice(0, H)), ('bwd', slice(H, 2 * H))):
            dx, dW, dU, db = lstm_backward(np.ascontiguousarray(do[..., sl]),
                                           caches['layers'][li][dname])
            g[dname] = {'W': dW, 'U': dU, 'b': db}
            if caches['layers'][li][dname].get('dmi') is not None:
                g[dname]['mi'] = caches['layers'][li][dname]['dmi']
            if caches['layers'][li][dname].get('dln') is not None:
                g[dname]['ln'] = caches['layers'][li][dname]['dln']
            dx_total = dx if dx_total is None else dx_total + dx
        grads['layers'][li] = g
        do = dx_total if d_skip is None else dx_total + d_skip
    if 'in_dense' in params:
        xi = caches['in_dense_x']
        T, N, D = xi.shape
        grads['in_dense'] = {'W': xi.reshape(T * N, D).T @ do.reshape(T * N, -1),
                             'b': do.sum(axis=(0, 1))}
        do = do @ params['in_dense']['W'].T
    if 'conv' in params:
        grads['conv'] = [None] * len(params['conv'])
        for ci in range(len(params['conv']) - 1, -1, -1):
            do, dW, db = _conv.conv2d_backward(do, caches['conv'][ci])
            grads['conv'][ci] = {'W': dW, 'b': db}
    grads['input'] = do
    return grads


def l2_penalty(params, weight_decay, in_dense_l2=False):
    """sum_w weight_decay * sum(w^2) over LSTM W,U and the output Dense kernel
    (core/models.py:263-264,279)."""
    tot = 0.0
    if weight_decay:
        for layer in params['layers']:
            for d in ('fwd', 'bwd'):
                tot += weight_decay * (np.sum(layer[d]['W'].astype(np.float64) ** 2) +
                                       np.sum(layer[d]['U'].astype(np.float64) ** 2))
        tot += weight_decay * np.sum(params['dense']['W'].astype(np.float64) ** 2)
        for cp in params.get('conv', []):
            tot += weight_decay * np.sum(cp['W'].astype(np.float64) ** 2)
    return tot


def loss_and_grads(params, x, labels, seq_len, weight_decay=0.0, masks=None, zone=None):
    """ctc_model() training objective and its gradients.

    Returns dict(loss (scalar: mean ctc + l2), ctc (N,), logits, grads).
    """
    logits, caches = model_forward(params, x, masks, zone)
    T, N, C = logits.shape
    for cp in params.get('conv', []):           # a sequence keeps ceil(len / st) frames
        seq_len = _conv.out_lengths(seq_len, cp['stride'][0])
    ctc_n, dlog = _ctc.ctc_loss_grad(logits, labels, seq_len, dtype=logits.dtype)
    dlog = dlog / N                       # mean over the batch
    grads = model_backward(params, caches, dlog.astype(logits.dtype))
    if weight_decay:
        for li, layer in enumerate(params['layers']):
            for d in ('fwd', 'bwd'):
                grads['layers'][li][d]['W'] += 2 * weight_decay * layer[d]['W']
                grads['layers'][li][d]['U'] += 2 * weight_decay * layer[d]['U']
        grads['dense']['W'] += 2 * weight_decay * params['dense']['W']
        for ci, cp in enumerate(params.get('conv', [])):
            grads['conv'][ci]['W'] += 2 * weight_decay * cp['W']
    loss = float(np.mean(ctc_n)) + l2_penalty(params, weight_decay)
    return dict(loss=loss, ctc=ctc_n, logits=logits, grads=grads, caches=caches)


def flatten(tree):
    """Deterministic flat list of (name, array) in checkpoint order."""
    out = []
    for ci, cp in enumerate(tree.get('conv', [])):
        out += [('conv%d/W' % ci, cp['W']), ('conv%d/b' % ci, cp['b'])]
    if 'in_dense' in tree:
        out += [('in_dense/W', tree['in_dense']['W']),
                ('in_dense/b', tree['in_dense']['b'])]
    for li, layer in enumerate(tree['layers']):
        for d in ('fwd', 'bwd'):
            for k in ('W', 'U', 'b'):
                out.append(('layer%d/%s/%s' % (li, d, k), layer[d][k]))
            if layer[d].get('mi') is not None:       # Keras add_weight order (layers.py:389-404)
                for k, a in zip(('mi_alpha', 'mi_beta1', 'mi_beta2'), layer[d]['mi']):
                    out.append(('layer%d/%s/%s' % (li, d, k), a))
            if layer[d].get('ln') is not None:
                # the reference iterates a dict literal {'Uh', 'Wx', 'new_c'} (layers.py:409),
                # whose Python-2 order is unspecified: this order is a convention of the build
                for k in ('Uh', 'Wx', 'new_c'):
                    out.append(('layer%d/%s/ln_gain_%s' % (li, d, k), layer[d]['ln'][k][0]))
                    out.append(('layer%d/%s/ln_bias_%s' % (li, d, k), layer[d]['ln'][k][1]))
    out += [('dense/W', tree['dense']['W']), ('dense/b', tree['dense']['b'])]
    return out

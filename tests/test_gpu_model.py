"""-m gpu: the whole hot path (BiLSTM stack -> Dense -> CTC -> BPTT -> clip+Adam)
through the product's ctc_model()/brsmv1() surface vs the float64 oracle.
Tolerances: logits/activations atol 1e-4, loss rtol 1e-4, gradients atol 1e-4 *
max|grad|, greedy decode indices exact, weights after 3 Adam steps atol 2e-5."""
import numpy as np
import pytest
import torch

from oracle import lstm as OL
from oracle import optim as OO
from oracle import decode as OD
from tests.gpu_util import report

pytestmark = pytest.mark.gpu


def _oracle_params(model_weights, n_layers, in_dense=False):
    it = iter([w.astype(np.float64) for w in model_weights])
    p = {'layers': []}
    if in_dense:
        p['in_dense'] = {'W': next(it), 'b': next(it)}
    for _ in range(n_layers):
        layer = {}
        for d in ('fwd', 'bwd'):
            layer[d] = {'W': next(it), 'U': next(it), 'b': next(it)}
        p['layers'].append(layer)
    p['dense'] = {'W': next(it), 'b': next(it)}
    return p


def _flat(tree):
    return [a for _, a in OL.flatten(tree)]


def _batch(rs, N, T, F, C, ragged=True):
    x = rs.randn(N, T, F).astype(np.float32)
    lens = np.full(N, T, np.int64)
    if ragged:
        lens[1::2] = rs.randint(T // 2, T, size=len(lens[1::2]))
        for n in range(N):
            x[n, lens[n]:] = 0.0                      # pad_sequences 'post'
    labels = [rs.randint(0, C - 1, size=rs.randint(1, max(2, T // 6))).tolist() for _ in range(N)]
    return x, labels, lens


@pytest.mark.parametrize('H,wd,use_masks', [(12, 0.0, False), (10, 1e-2, True)])
def test_brsmv1_small_forward_backward_adam(H, wd, use_masks):
    from asr_study_amd.core import models, optimizers
    rs = np.random.RandomState(H)
    N, T, F, C, L = 5, 23, 9, 7, 2
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L,
                          dropout=0.0, weight_decay=wd, seed=3)
    w = [a + rs.randn(*a.shape).astype(np.float32) * 0.2 for a in model.get_weights()]
    model.set_weights(w)
    for a, b in zip(model.get_weights(), w):           # layout round trip
        assert np.array_equal(a, b)
    model.compile(optimizer=optimizers.Adam(lr=1e-2, clipnorm=0.5))
    x, labels, lens = _batch(rs, N, T, F, C)
    params = _oracle_params(w, L)
    n_pad = 16
    masks_o = masks_g = None
    if use_masks:
        masks_o, masks_g = [], {}
        n_in = F
        si = 1                                           # stage 0 is GaussianNoise
        for li in range(L):
            Hp = (H + 3) // 4 * 4
            f_in_pad = (n_in + 3) // 4 * 4 if li == 0 else 2 * Hp
            mo = {}
            BW = np.ones((2, n_pad, f_in_pad), np.float32)
            BU = np.ones((2, n_pad, Hp), np.float32)
            for di, d in enumerate(('fwd', 'bwd')):
                bw = ((rs.rand(N, n_in) > 0.2) / 0.8)
                bu = ((rs.rand(N, H) > 0.2) / 0.8)
                mo[d] = (bw, bu)
                if li == 0:
                    BW[di, :N, :n_in] = bw
                else:       # padded feature layout [fwd Hp | bwd Hp]
                    BW[di, :N, :H] = bw[:, :H]
                    BW[di, :N, Hp:Hp + H] = bw[:, H:]
                BU[di, :N, :H] = bu
            masks_o.append(mo)
            masks_g[si] = (torch.from_numpy(BW).cuda(), torch.from_numpy(BU).cuda())
            n_in = 2 * H
            si += 1
    xt = np.ascontiguousarray(x.transpose(1, 0, 2)).astype(np.float64)
    want = OL.loss_and_grads(params, xt, labels, lens, weight_decay=0.0, masks=masks_o)
    slab = model.to_slab(x)
    ctc, logits, _ = model.loss_and_grads(slab, labels, lens, training=True, masks=masks_g)
    torch.cuda.synchronize()
    assert report('model logits', logits.cpu().numpy()[:, :N], want['logits']) < 1e-4
    np.testing.assert_allclose(ctc.cpu().numpy(), want['ctc'], rtol=1e-4)
    got_g = model.get_gradients()
    for (name, g), gg in zip(OL.flatten(want['grads']), got_g):
        scale = max(1e-3, np.abs(g).max())
        assert report('grad ' + name, gg, g) < 1e-4 * scale + 1e-6, name
    # greedy decode
    from asr_study_amd.core import ctc_utils
    hyp = ctc_utils.decode((logits, lens), is_greedy=True)
    assert hyp == OD.greedy_decode(want['logits'], lens)
    # three optimiser steps (clip + l2 + Adam) against the oracle
    opt = OO.Adam(lr=1e-2, clipnorm=0.5)
    p_flat = _flat(params)
    for step in range(3):
        out = OL.loss_and_grads(params, xt, labels, lens, weight_decay=wd, masks=masks_o)
        opt.step(p_flat, _flat(out['grads']))
        m = model.train_on_batch([('slab', slab), labels, lens], masks=masks_g)
        assert abs(m[1] - float(np.mean(out['ctc']))) < 1e-4 * max(1.0, abs(m[1]))
        assert abs(m[0] - out['loss']) < 2e-4 * max(1.0, abs(out['loss']))
    for (name, a), b in zip(OL.flatten(params), model.get_weights()):
        assert report('weights ' + name, b, a) < 5e-5, name


def test_cfg1_graves_ragged_batch():
    """BASELINE cfg1: F=26, 1 x BiLSTM(100), 28 classes, batch 4, T in [99, 999]."""
    from asr_study_amd.core import models
    rs = np.random.RandomState(1)
    N, T, F, C, H = 4, 300, 26, 28, 100
    model = models.graves2006(num_features=F, num_hiddens=H, num_classes=C, std=0.6, seed=1)
    w = model.get_weights()
    x, labels, lens = _batch(rs, N, T, F, C)
    lens[0], lens[2] = T, 99
    x[2, 99:] = 0
    params = _oracle_params(w, 1)
    xt = np.ascontiguousarray(x.transpose(1, 0, 2)).astype(np.float64)
    want = OL.loss_and_grads(params, xt, labels, lens)
    slab = model.to_slab(x)
    ctc, logits, _ = model.loss_and_grads(slab, labels, lens, training=False)
    torch.cuda.synchronize()
    assert report('cfg1 logits', logits.cpu().numpy()[:, :N], want['logits']) < 1e-4
    np.testing.assert_allclose(ctc.cpu().numpy(), want['ctc'], rtol=1e-4)
    for (name, g), gg in zip(OL.flatten(want['grads']), model.get_gradients()):
        scale = max(1e-3, np.abs(g).max())
        assert report('cfg1 grad ' + name, gg, g) < 1e-4 * scale + 1e-6, name
    m = model.test_on_batch([('slab', slab), labels, lens])
    assert abs(m[1] - float(np.mean(want['ctc']))) < 1e-3
    hyp = OD.greedy_decode(want['logits'], lens)
    assert abs(m[3] - OD.ler(hyp, labels)) < 1e-6


def test_eyben_padded_hidden_sizes():
    """H = 78 / 27 are not multiples of 4: the engine pads with zero units."""
    from asr_study_amd.core import models
    rs = np.random.RandomState(2)
    N, T, F, C = 3, 40, 11, 6
    model = models.eyben(num_features=F, num_hiddens=[13, 78, 27], num_classes=C, seed=2)
    w = model.get_weights()
    x, labels, lens = _batch(rs, N, T, F, C, ragged=False)
    it = iter([a.astype(np.float64) for a in w])
    params = {'in_dense': {'W': next(it), 'b': next(it)}, 'layers': []}
    for _ in range(2):
        params['layers'].append({d: {'W': next(it), 'U': next(it), 'b': next(it)}
                                 for d in ('fwd', 'bwd')})
    params['dense'] = {'W': next(it), 'b': next(it)}
    xt = np.ascontiguousarray(x.transpose(1, 0, 2)).astype(np.float64)
    want = OL.loss_and_grads(params, xt, labels, lens)
    ctc, logits, _ = model.loss_and_grads(model.to_slab(x), labels, lens, training=False)
    torch.cuda.synchronize()
    assert report('eyben logits', logits.cpu().numpy()[:, :N], want['logits']) < 1e-4
    for (name, g), gg in zip(OL.flatten(want['grads']), model.get_gradients()):
        scale = max(1e-3, np.abs(g).max())
        assert report('eyben grad ' + name, gg, g) < 1e-4 * scale + 1e-6, name


def test_infeasible_label_raises_like_tf():
    from asr_study_amd.core import models
    model = models.graves2006(num_features=5, num_hiddens=8, num_classes=4, std=0.0)
    x = np.zeros((1, 3, 5), np.float32)
    with pytest.raises(ValueError):
        model.loss_and_grads(model.to_slab(x), [[1, 1, 1]], [3])


@pytest.mark.parametrize('mode', ['sum', 'ave'])
def test_brsmv1_residual_connections(mode):
    """brsmv1(residual=mode): TimeDistributed(Dense(2H)) in front, then
    o = merge([Bidirectional(LSTM)(o), o], mode) per layer (core/models.py:253-255,
    273-276) -- logits, loss and every gradient vs the oracle."""
    from asr_study_amd.core import models
    rs = np.random.RandomState(7)
    N, T, F, C, H, L = 4, 19, 9, 7, 8, 3
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L,
                          dropout=0.0, weight_decay=0.0, residual=mode, seed=2)
    w = [a + rs.randn(*a.shape).astype(np.float32) * 0.2 for a in model.get_weights()]
    model.set_weights(w)
    x, labels, lens = _batch(rs, N, T, F, C)
    params = _oracle_params(w, L, in_dense=True)
    params['residual'] = mode
    xt = np.ascontiguousarray(x.transpose(1, 0, 2)).astype(np.float64)
    want = OL.loss_and_grads(params, xt, labels, lens)
    slab = model.to_slab(x)
    ctc, logits, _ = model.loss_and_grads(slab, labels, lens, training=False)
    torch.cuda.synchronize()
    assert report('residual logits', logits.cpu().numpy()[:, :N], want['logits']) < 1e-4
    np.testing.assert_allclose(ctc.cpu().numpy(), want['ctc'], rtol=1e-4)
    for (name, g), gg in zip(OL.flatten(want['grads']), model.get_gradients()):
        scale = max(1e-3, np.abs(g).max())
        assert report('residual grad ' + name, gg, g) < 1e-4 * scale + 1e-6, name


def test_brsmv1_multiplicative_integration_and_zoneout():
    """brsmv1(mi=[...], zoneout=...) (core/models.py:260-271, core/layers.py:389-404,
    441-443, 457-467): weights in Keras order [W, U, b, alpha, beta1, beta2] per direction,
    logits / loss / every gradient (incl. d alpha, d beta1, d beta2) vs the oracle, with
    explicit zoneout keep masks; then the test-phase (1 - level) path."""
    from asr_study_amd.core import models
    rs = np.random.RandomState(11)
    N, T, F, C, H, L = 5, 21, 9, 7, 12, 2
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L,
                          dropout=0.0, weight_decay=0.0, mi=[1.0, 0.5, 0.5], zoneout=0.25, seed=5)
    w = [a + rs.randn(*a.shape).astype(np.float32) * 0.15 for a in model.get_weights()]
    assert len(w) == L * 12 + 2
    model.set_weights(w)
    for a, b in zip(model.get_weights(), w):
        assert np.array_equal(a, b)
    it = iter([a.astype(np.float64) for a in w])
    params = {'layers': []}
    for _ in range(L):
        layer = {}
        for d in ('fwd', 'bwd'):
            layer[d] = {'W': next(it), 'U': next(it), 'b': next(it)}
            layer[d]['mi'] = [next(it), next(it), next(it)]
        params['layers'].append(layer)
    params['dense'] = {'W': next(it), 'b': next(it)}
    x, labels, lens = _batch(rs, N, T, F, C)
    xt = np.ascontiguousarray(x.transpose(1, 0, 2)).astype(np.float64)
    n_pad, Hp = 16, 12
    zone, masks_g = [], {}
    si = 1                                               # stage 0 is GaussianNoise
    for li in range(L):
        z = {}
        KC = np.ones((T, 2, Hp), np.float32)
        KH = np.ones((T, 2, Hp), np.float32)
        for di, d in enumerate(('fwd', 'bwd')):
            kc = (rs.rand(T, H) > 0.25).astype(np.float64)
            kh = (rs.rand(T, H) > 0.25).astype(np.float64)
            z[d] = (kc, kh)
            KC[:, di, :H], KH[:, di, :H] = kc, kh
        zone.append(z)
        masks_g[si] = (None, None, torch.from_numpy(KC).cuda(), torch.from_numpy(KH).cuda())
        si += 1
    want = OL.loss_and_grads(params, xt, labels, lens, zone=zone)
    slab = model.to_slab(x)
    ctc, logits, _ = model.loss_and_grads(slab, labels, lens, training=True, masks=masks_g)
    torch.cuda.synchronize()
    assert report('mi+zoneout logits', logits.cpu().numpy()[:, :N], want['logits']) < 1e-4
    np.testing.assert_allclose(ctc.cpu().numpy(), want['ctc'], rtol=1e-4)
    got = model.get_gradients()
    flat = OL.flatten(want['grads'])
    assert len(flat) == len(got)
    for (name, g), gg in zip(flat, got):
        scale = max(1e-3, np.abs(g).max())
        assert report('mi+zoneout grad ' + name, gg, g) < 1e-4 * scale + 1e-6, name
    # test phase: constant coefficients 1 - level
    zone_t = [{d: (np.full((T, H), 0.75), np.full((T, H), 0.75)) for d in ('fwd', 'bwd')}
              for _ in range(L)]
    want_t = OL.model_forward(params, xt, zone=zone_t)[0]
    logits_t = model.forward(slab, training=False).cpu().numpy()[:, :N]
    assert report('mi+zoneout test-phase logits', logits_t, want_t) < 1e-4


@pytest.mark.parametrize('with_mi', [False, True])
def test_brsmv1_layer_normalisation(with_mi):
    """brsmv1(layer_norm=[gain, bias]) (+ mi, zoneout): weights in the order [W, U, b,
    (alpha, beta1, beta2), gain/bias of LN(h@U), LN(x@W), LN(c)] per direction; logits, loss
    and every gradient vs the oracle."""
    from asr_study_amd.core import models
    rs = np.random.RandomState(13)
    N, T, F, C, H, L = 5, 15, 9, 7, 12, 2
    kw = dict(mi=[1.0, 0.5, 0.5], zoneout=0.2) if with_mi else {}
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L,
                          dropout=0.0, weight_decay=0.0, layer_norm=[1.0, 0.0], seed=5, **kw)
    w = [a + rs.randn(*a.shape).astype(np.float32) * 0.15 for a in model.get_weights()]
    per_dir = 3 + (3 if with_mi else 0) + 6
    assert len(w) == L * 2 * per_dir + 2
    model.set_weights(w)
    for a, b in zip(model.get_weights(), w):
        assert np.array_equal(a, b)
    it = iter([a.astype(np.float64) for a in w])
    params = {'layers': []}
    for _ in range(L):
        layer = {}
        for d in ('fwd', 'bwd'):
            layer[d] = {'W': next(it), 'U': next(it), 'b': next(it)}
            if with_mi:
                layer[d]['mi'] = [next(it), next(it), next(it)]
            layer[d]['ln'] = {k: [next(it), next(it)] for k in ('Uh', 'Wx', 'new_c')}
        params['layers'].append(layer)
    params['dense'] = {'W': next(it), 'b': next(it)}
    x, labels, lens = _batch(rs, N, T, F, C)
    xt = np.ascontiguousarray(x.transpose(1, 0, 2)).astype(np.float64)
    zone, masks_g = None, None
    if with_mi:
        zone, masks_g = [], {}
        for li in range(L):
            z, KC, KH = {}, np.ones((T, 2, H), np.float32), np.ones((T, 2, H), np.float32)
            for di, d in enumerate(('fwd', 'bwd')):
                kc = (rs.rand(T, H) > 0.2).astype(np.float64)
                kh = (rs.rand(T, H) > 0.2).astype(np.float64)
                z[d] = (kc, kh)
                KC[:, di], KH[:, di] = kc, kh
            zone.append(z)
            masks_g[1 + li] = (None, None, torch.from_numpy(KC).cuda(), torch.from_numpy(KH).cuda())
    want = OL.loss_and_grads(params, xt, labels, lens, zone=zone)
    slab = model.to_slab(x)
    ctc, logits, _ = model.loss_and_grads(slab, labels, lens, training=True, masks=masks_g)
    torch.cuda.synchronize()
    assert report('LN logits', logits.cpu().numpy()[:, :N], want['logits']) < 1e-4
    np.testing.assert_allclose(ctc.cpu().numpy(), want['ctc'], rtol=1e-4)
    got = model.get_gradients()
    flat = OL.flatten(want['grads'])
    assert len(flat) == len(got)
    for (name, g), gg in zip(flat, got):
        scale = max(1e-3, np.abs(g).max())
        assert report('LN grad ' + name, gg, g) < 2e-4 * scale + 1e-6, name


@pytest.mark.parametrize('use_masks,N', [(False, 20), (True, 20), (True, 40)])
def test_brsmv1_packed_operand_gemm_path(use_masks, N, monkeypatch):
    """ASR_GEMM_PACKED=1 (the default from 512 hidden units on): every BiLSTM GEMM runs on
    operands packed once into split-fp16 planes (asr_pack_hl / asr_gemm_hl), dropout masks
    folded into the pack -- logits, loss and all gradients vs the oracle, and the same weights
    after two optimiser steps as the per-tile path.  N = 40 pads to 48 rows: the mask period
    of the pack is n_pad, a multiple of 16 but not a power of two (batch 48 / 96, the short last
    batch of an epoch, uneven data-parallel shards)."""
    from asr_study_amd.core import models, optimizers
    rs = np.random.RandomState(21)
    T, F, C, L, H = 37, 16, 7, 3, 24
    x, labels, lens = _batch(rs, N, T, F, C)
    n_pad = (N + 15) // 16 * 16
    results = {}
    for packed in ('1', '0'):
        monkeypatch.setenv('ASR_GEMM_PACKED', packed)
        model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L,
                              dropout=0.0, weight_decay=1e-3, seed=3)
        assert model.packed == (packed == '1')
        rw = np.random.RandomState(5)
        w = [a + rw.randn(*a.shape).astype(np.float32) * 0.2 for a in model.get_weights()]
        model.set_weights(w)
        model.compile(optimizer=optimizers.Adam(lr=1e-2, clipnorm=0.5))
        masks_o = masks_g = None
        if use_masks:
            rm = np.random.RandomState(9)
            masks_o, masks_g = [], {}
            n_in = F
            for li in range(L):
                mo = {}
                BW = np.ones((2, n_pad, n_in), np.float32)
                BU = np.ones((2, n_pad, H), np.float32)
                for di, d in enumerate(('fwd', 'bwd')):
                    bw = ((rm.rand(N, n_in) > 0.2) / 0.8)
                    bu = ((rm.rand(N, H) > 0.2) / 0.8)
                    mo[d] = (bw, bu)
                    BW[di, :N] = bw
                    BU[di, :N] = bu
                masks_o.append(mo)
                masks_g[li + 1] = (torch.from_numpy(BW).cuda(), torch.from_numpy(BU).cuda())
                n_in = 2 * H
        slab = model.to_slab(x)
        ctc, logits, _ = model.loss_and_grads(slab, labels, lens, training=True, masks=masks_g)
        torch.cuda.synchronize()
        if packed == '1':
            params = _oracle_params(w, L)
            xt = np.ascontiguousarray(x.transpose(1, 0, 2)).astype(np.float64)
            want = OL.loss_and_grads(params, xt, labels, lens, weight_decay=0.0, masks=masks_o)
            assert report('packed logits', logits.cpu().numpy()[:, :N], want['logits']) < 1e-4
            np.testing.assert_allclose(ctc.cpu().numpy(), want['ctc'], rtol=1e-4)
            for (name, g), gg in zip(OL.flatten(want['grads']), model.get_gradients()):
                scale = max(1e-3, np.abs(g).max())
                assert report('packed grad ' + name, gg, g) < 1e-4 * scale + 1e-6, name
        for _ in range(2):
            model.train_on_batch([('slab', slab), labels, lens], masks=masks_g)
        results[packed] = model.get_weights()
    for a, b in zip(results['1'], results['0']):     # (Adam, lr 1e-2: sign-like early steps)
        assert np.abs(a - b).max() < 1e-4


def test_side_stream_is_used_only_where_a_gemm_can_run_beside_the_recurrence():
    """ASR_OVERLAP=auto (engine.Model._recurrence_fills_chip): 5xBiLSTM(512) at batch 64 puts
    2 directions x 4 batch tiles x 32 workgroups = 256 recurrent workgroups on the 256 CUs and
    leaves no registers for a GEMM wave -> no side stream; BiLSTM(256) at batch 32 uses 64 CUs ->
    side stream on.  Same gradients either way (the switch only moves launches between streams)."""
    from asr_study_amd.core import models
    rs = np.random.RandomState(11)
    for H, N, want in ((512, 64, False), (256, 32, True)):
        T, F, C = 9, 16, 6
        model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=2,
                              dropout=0.0, seed=2)
        x, labels, lens = _batch(rs, N, T, F, C)
        slab = model.to_slab(x)
        model.loss_and_grads(slab, labels, lens, training=True)
        torch.cuda.synchronize()
        assert model.overlap is want, (H, N)
        g_auto = [g.copy() for g in model.get_gradients()]
        model._overlap_mode = '1' if not want else '0'
        model.overlap = not want
        model.loss_and_grads(slab, labels, lens, training=True)
        torch.cuda.synchronize()
        for a, b in zip(g_auto, model.get_gradients()):
            assert np.array_equal(a, b)


def test_compact_bptt_schedule_runs_weight_gradients_beside_the_layer_below():
    """ASR_BPTT_COMPACT=auto (engine.Model._bptt_compact): where a layer's BPTT would fill the chip
    (5xBiLSTM(512) at batch 64: 256 workgroups) every layer BELOW the top one is launched in the
    compact geometry (asr_lstm_args.compact: 128 workgroups) with the weight-gradient GEMMs of
    the layer above on the side stream; the top layer has nothing to run beside it and keeps the
    whole chip.  The gradients are bit-identical to the serial schedule (the compact kernel is,
    and the GEMMs only change streams).  BiLSTM(256) at batch 32 never needs it."""
    from asr_study_amd.core import models
    rs = np.random.RandomState(12)
    T, F, C = 11, 16, 6
    for H, N, L, want in ((512, 64, 3, 2), (256, 32, 2, 0)):
        model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L,
                              dropout=0.0, seed=3)
        x, labels, lens = _batch(rs, N, T, F, C)
        slab = model.to_slab(x)
        model.loss_and_grads(slab, labels, lens, training=True)
        torch.cuda.synchronize()
        assert model._compact_launches == want, (H, N, model._compact_launches)
        g_auto = [g.copy() for g in model.get_gradients()]
        model._compact_mode = '0'
        model.loss_and_grads(slab, labels, lens, training=True)
        torch.cuda.synchronize()
        assert model._compact_launches == 0
        for a, b in zip(g_auto, model.get_gradients()):
            assert np.array_equal(a, b)


@pytest.mark.parametrize('act,layer_norm', [('relu', None), ('softsign', None), ('sigmoid', [1.0, 0.0]),
                                            ('softplus', None), ('linear', None), ('hard_sigmoid', None)])
def test_brsmv1_activation_hyper_parameter(act, layer_norm):
    """brsmv1(activation=...) (core/models.py:220, :271 -> LSTM(activation=...), core/layers.py:452,
    :463): logits, CTC loss and every gradient vs the oracle, on the variant kernels and (with
    layer_norm) on the row-per-workgroup cell; an unknown name is refused when the model is built."""
    from asr_study_amd.core import models
    rs = np.random.RandomState(17)
    N, T, F, C, H, L = 5, 19, 9, 7, 12, 2
    kw = dict(layer_norm=layer_norm) if layer_norm else {}
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L, dropout=0.0,
                          weight_decay=0.0, activation=act, seed=5, **kw)
    w = [a + rs.randn(*a.shape).astype(np.float32) * 0.1 for a in model.get_weights()]
    model.set_weights(w)
    it = iter([a.astype(np.float64) for a in w])
    params = {'layers': [], 'activation': act}
    for _ in range(L):
        layer = {}
        for d in ('fwd', 'bwd'):
            layer[d] = {'W': next(it), 'U': next(it), 'b': next(it)}
            if layer_norm:
                layer[d]['ln'] = {k: [next(it), next(it)] for k in ('Uh', 'Wx', 'new_c')}
        params['layers'].append(layer)
    params['dense'] = {'W': next(it), 'b': next(it)}
    x, labels, lens = _batch(rs, N, T, F, C)
    xt = np.ascontiguousarray(x.transpose(1, 0, 2)).astype(np.float64)
    want = OL.loss_and_grads(params, xt, labels, lens)
    slab = model.to_slab(x)
    ctc, logits, _ = model.loss_and_grads(slab, labels, lens, training=True)
    torch.cuda.synchronize()
    sc = max(1.0, np.abs(want['logits']).max())
    assert report('act=%s logits' % act, logits.cpu().numpy()[:, :N], want['logits']) < 1e-4 * sc
    np.testing.assert_allclose(ctc.cpu().numpy(), want['ctc'], rtol=1e-4)
    got = model.get_gradients()
    flat = OL.flatten(want['grads'])
    assert len(flat) == len(got)
    for (name, g), gg in zip(flat, got):
        scale = max(1e-3, np.abs(g).max())
        assert report('act=%s grad %s' % (act, name), gg, g) < 2e-4 * scale + 1e-6, name
    assert model.config['kwargs']['activation'] == act
    with pytest.raises(NotImplementedError):
        models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=1, activation='elu')


@pytest.mark.parametrize('act', ['relu', 'linear'])
def test_unbounded_activation_above_the_packed_planes_fixed_bound(act, monkeypatch):
    """with relu / linear h = o * act(c) is unbounded (c accumulates over the frames), so
    the packed operands built from y (next layer's x@W, dU) must MEASURE their bound instead of
    assuming |y| < 1 (pow2_scale(1) = 2^8: anything >= 256 would become an fp16 inf in the hi
    plane).  Gates saturated open and a constant cell drive of +30 per frame: |y| reaches ~ 30 T
    >> 256.  Logits, CTC and every gradient stay finite and on the oracle."""
    from asr_study_amd.core import models
    rs = np.random.RandomState(23)
    N, T, F, C, H, L = 4, 21, 8, 6, 16, 2
    monkeypatch.setenv('ASR_GEMM_PACKED', '1')           # ('auto' packs from 512 hidden units on)
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L, dropout=0.0,
                          weight_decay=0.0, activation=act, seed=9)
    assert model.packed and all(model._stage_packed(s) for s in model.stages if s.kind == 'bilstm')
    w = [a.copy() for a in model.get_weights()]
    for k in range(0, 6 * L, 3):                         # [W, U, b] per direction, blocks i f c o
        w[k] *= 0.05
        w[k + 1] *= 0.05
        b = w[k + 2]
        b[:] = 5.0
        b[2 * H:3 * H] = 30.0 if k < 6 else 0.5          # the first layer pumps its cell
    w[-2] *= 0.02
    model.set_weights(w)
    it = iter([a.astype(np.float64) for a in w])
    params = {'layers': [], 'activation': act}
    for _ in range(L):
        params['layers'].append({d: {'W': next(it), 'U': next(it), 'b': next(it)}
                                 for d in ('fwd', 'bwd')})
    params['dense'] = {'W': next(it), 'b': next(it)}
    x, labels, lens = _batch(rs, N, T, F, C)
    lens = [T] * N
    xt = np.ascontiguousarray(x.transpose(1, 0, 2)).astype(np.float64)
    want = OL.loss_and_grads(params, xt, labels, lens)
    # (|y| of the first layer reaches ~ 30 T: its dU is of that order)
    assert max(np.abs(g).max() for n, g in OL.flatten(want['grads']) if n.endswith('/U')) > 256.0
    slab = model.to_slab(x)
    ctc, logits, _ = model.loss_and_grads(slab, labels, lens, training=True)
    torch.cuda.synchronize()
    lg = logits.cpu().numpy()[:, :N]
    assert np.all(np.isfinite(lg)) and np.all(np.isfinite(ctc.cpu().numpy()))
    sc = max(1.0, np.abs(want['logits']).max())
    assert report('act=%s big logits' % act, lg, want['logits']) < 1e-4 * sc
    np.testing.assert_allclose(ctc.cpu().numpy(), want['ctc'], rtol=1e-4)
    for (name, g), gg in zip(OL.flatten(want['grads']), model.get_gradients()):
        assert np.all(np.isfinite(gg)), name
        scale = max(1e-3, np.abs(g).max())
        assert report('act=%s big grad %s' % (act, name), gg, g) < 2e-4 * scale + 1e-6, name

"""-m gpu: recurrent LSTM kernels (forward + BPTT, persistent and per-step modes)
through the C ABI vs the float64 oracle.  Tolerance: activations atol 1e-4
(north_star: "LSTM activations within 1e-4 fp32"); gate gradients atol 1e-4
relative to max|dz|."""
import numpy as np
import pytest
import torch

from oracle import lstm as OL
from tests.gpu_util import (to_dev, pad_batch, report, gate_major_to_unit_major)

pytestmark = pytest.mark.gpu


def _case(T, N, F, H, seed, use_mask):
    rs = np.random.RandomState(seed)
    x = rs.randn(T, N, F)
    p = {d: OL.init_lstm(rs, F, H, np.float64) for d in ('fwd', 'bwd')}
    for d in p:
        p[d]['b'] = p[d]['b'] + rs.randn(4 * H) * 0.3
        p[d]['U'] = p[d]['U'] * 0.8
    masks = {d: None for d in p}
    if use_mask:
        masks = {d: ((rs.rand(N, H) > 0.2) / 0.8) for d in p}
    return rs, x, p, masks


def _oracle(x, p, masks, dhs=None):
    out = {}
    for d, rev in (('fwd', False), ('bwd', True)):
        hs, cache = OL.lstm_forward(x, p[d]['W'], p[d]['U'], p[d]['b'], rev, None, masks[d])
        out[d] = dict(hs=hs, cache=cache)
        if dhs is not None:
            OL.lstm_backward(dhs[d], cache)
    return out


def _pack_inputs(x, p, masks, H, n_pad):
    T, N, F = x.shape
    zx = np.zeros((T, n_pad, 2, 4 * H), np.float32)
    U = np.zeros((2, H, 4 * H), np.float32)
    mk = np.ones((2, n_pad, H), np.float32)
    for di, d in enumerate(('fwd', 'bwd')):
        z = x @ p[d]['W'] + p[d]['b']                     # (T,N,4H) gate-major
        zx[:, :N, di] = gate_major_to_unit_major(z, H)
        # padding rows get the bias only (x = 0), like the real pipeline
        zx[:, N:, di] = gate_major_to_unit_major(p[d]['b'][None, None], H)
        U[di] = gate_major_to_unit_major(p[d]['U'], H)
        if masks[d] is not None:
            mk[di, :N] = masks[d]
    return zx, U, mk


@pytest.mark.parametrize('T,N,F,H,mode,use_mask', [
    (30, 5, 7, 16, 1, False),        # per-step launches
    (30, 5, 7, 16, 0, False),        # persistent
    (61, 20, 9, 100, 0, True),       # cfg1 width (H=100: ragged K split), masks
    (200, 32, 12, 256, 0, False),    # cfg2 width
    (25, 16, 8, 512, 0, False),      # cfg3 width
    (25, 16, 8, 256, 1, True),
])
def test_forward_and_backward(T, N, F, H, mode, use_mask):
    from asr_study_amd import ops
    rs, x, p, masks = _case(T, N, F, H, T + H, use_mask)
    n_pad = ops.pad16(N)
    dhs = {d: rs.randn(T, N, H) for d in ('fwd', 'bwd')}
    want = _oracle(x, p, masks, dhs)
    zx, U, mk = _pack_inputs(x, p, masks, H, n_pad)
    zx_d, U_d = to_dev(zx), to_dev(U)
    mk_d = to_dev(mk) if use_mask else None
    y = torch.full((T, n_pad, 2 * H), 3.0, dtype=torch.float32, device='cuda:0')
    cell = torch.full((T, n_pad, 2, H), 3.0, dtype=torch.float32, device='cuda:0')
    gates = torch.full((T, n_pad, 2, 4 * H), 3.0, dtype=torch.float32, device='cuda:0')
    print('plan fwd', ops.lstm_plan(T, n_pad, H, 0), 'bwd', ops.lstm_plan(T, n_pad, H, 1))
    ops.lstm_seq_fwd(zx_d, U_d, y, cell, gates, T, n_pad, H, mask_u=mk_d, mode=mode, check=True)
    yh, ch, gh = y.cpu().numpy(), cell.cpu().numpy(), gates.cpu().numpy()
    tag = 'T%d N%d H%d m%d' % (T, N, H, mode)
    for di, d in enumerate(('fwd', 'bwd')):
        assert report('lstm h %s %s' % (d, tag), yh[:, :N, di * H:(di + 1) * H], want[d]['hs']) < 1e-4
        assert report('lstm c %s %s' % (d, tag), ch[:, :N, di], want[d]['cache']['cs']) < 1e-4
        assert report('lstm gates %s %s' % (d, tag), gh[:, :N, di],
                      gate_major_to_unit_major(want[d]['cache']['gates'], H)) < 1e-4
    # ---- BPTT: feed the oracle's upstream gradient, compare gate gradients
    dy = np.zeros((T, n_pad, 2 * H), np.float32)
    dy[:, :N, :H] = dhs['fwd']
    dy[:, :N, H:] = dhs['bwd']
    dz = torch.full((T, n_pad, 2, 4 * H), 5.0, dtype=torch.float32, device='cuda:0')
    ops.lstm_seq_bwd(to_dev(dy), U_d, cell, gates, dz, T, n_pad, H, mask_u=mk_d, mode=mode,
                     check=True)
    dzh = dz.cpu().numpy()
    for di, d in enumerate(('fwd', 'bwd')):
        w = gate_major_to_unit_major(want[d]['cache']['dzs'], H)
        scale = max(1.0, np.abs(w).max())
        assert report('lstm dz %s %s' % (d, tag), dzh[:, :N, di], w) < 1e-4 * scale
    # padding rows received zero upstream gradient -> exactly zero gate gradients
    assert np.all(dzh[:, N:] == 0)


@pytest.mark.parametrize('T,N,H,cuts', [
    (64, 16, 64, (48,)),             # the 3T/4 split the engine uses
    (41, 32, 256, (1, 2, 17, 40)),   # odd boundaries incl. single-step slices
])
def test_step_ranges_continue_one_sequence(T, N, H, cuts):
    """asr_lstm_args.step_begin/step_count: consecutive slices == one call, bit for bit
    (the slices only move kernel boundaries; the arithmetic is unchanged)."""
    from asr_study_amd import ops
    rs = np.random.RandomState(T * H)
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    zx = torch.from_numpy(rs.randn(T, n_pad, 2, 4 * H).astype(np.float32)).to(dev)
    U = torch.from_numpy((rs.randn(2, H, 4 * H) / np.sqrt(H)).astype(np.float32)).to(dev)
    dy = torch.from_numpy((rs.randn(T, n_pad, 2 * H) * 0.1).astype(np.float32)).to(dev)
    bounds = [0] + list(cuts) + [T]

    def run(sliced):
        y = torch.zeros(T, n_pad, 2 * H, device=dev)
        cell = torch.zeros(T, n_pad, 2, H, device=dev)
        gates = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
        dz = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
        amax = torch.zeros(1, device=dev)
        ranges = [(a, b - a) for a, b in zip(bounds[:-1], bounds[1:])] if sliced else [None]
        for r in ranges:
            ws = ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H, steps=r)
        ops.lstm_status(ws)
        for r in ranges:
            ws = ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H, dz_absmax=amax, steps=r)
        ops.lstm_status(ws)
        return [t.cpu().numpy() for t in (y, cell, gates, dz, amax)]
    whole, parts = run(False), run(True)
    for name, a, b in zip(('y', 'cell', 'gates', 'dz', 'absmax'), whole, parts):
        assert np.array_equal(a, b), name
    assert np.abs(whole[3]).max() == whole[4][0]


def test_sticky_timeout_flag_survives_later_calls_and_is_reported_once():
    """The first int of the recurrent workspace is a sticky timeout flag: set together with
    the per-call flag, untouched by later calls on the same workspace, cleared only by
    asr_lstm_status -- so one check per training step sees a timeout of any layer."""
    from asr_study_amd import ops
    from asr_study_amd._lib import AsrHipError
    T, n_pad, H = 5, 16, 16
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(0)
    zx = torch.from_numpy(rs.randn(T, n_pad, 2, 4 * H).astype(np.float32)).to(dev)
    U = torch.from_numpy((rs.randn(2, H, 4 * H) / 4).astype(np.float32)).to(dev)
    y = torch.zeros(T, n_pad, 2 * H, device=dev)
    cell = torch.zeros(T, n_pad, 2, H, device=dev)
    gates = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    ws = ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H)
    ops.lstm_status(ws)
    assert not ops.lstm_timeout_flags(dev).any().item()
    ws[:4].view(torch.int32)[0] = 1              # what mark_timeout() does on the device
    ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H)       # a later, healthy call
    assert ops.lstm_timeout_flags(dev)[0].item() == 1
    with pytest.raises(AsrHipError):
        ops.lstm_status(ws)
    ops.lstm_status(ws)                          # reported once, then clear
    assert not ops.lstm_timeout_flags(dev).any().item()


@pytest.mark.parametrize('T,N,H,use_mi,use_zone', [
    (23, 5, 16, True, False),
    (23, 5, 16, False, True),
    (40, 20, 100, True, True),       # ragged K, two batch tiles
    (31, 16, 256, True, True),
])
def test_cell_variants_multiplicative_integration_and_zoneout(T, N, H, use_mi, use_zone):
    """asr_lstm_args.mi / zone_c / zone_h (core/layers.py:441-443, 457-467) vs the oracle:
    activations 1e-4; d/d(h@U), d/d(x@W) and the MI parameter gradients 1e-4 * max."""
    from asr_study_amd import ops
    F = 7
    rs, x, p, masks = _case(T, N, F, H, 3 * T + H, False)
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    mi = zone = None
    mi_d = zc_d = zh_d = None
    if use_mi:
        mi = {d: [1.0 + 0.3 * rs.randn(4 * H), 0.6 + 0.3 * rs.randn(4 * H),
                  0.7 + 0.3 * rs.randn(4 * H)] for d in ('fwd', 'bwd')}
    if use_zone:
        zone = {d: ((rs.rand(T, H) > 0.3).astype(np.float64), np.full((T, H), 0.85))
                for d in ('fwd', 'bwd')}
    dhs = {d: rs.randn(T, N, H) for d in ('fwd', 'bwd')}
    want = {}
    for d, rev in (('fwd', False), ('bwd', True)):
        hs, cache = OL.lstm_forward(x, p[d]['W'], p[d]['U'], p[d]['b'], rev, None, None,
                                    mi[d] if mi else None, *(zone[d] if zone else (None, None)))
        OL.lstm_backward(dhs[d], cache)
        want[d] = dict(hs=hs, cache=cache)
    # pack: zx = x@W (+ b unless mi), unit-major
    zx = np.zeros((T, n_pad, 2, 4 * H), np.float32)
    U = np.zeros((2, H, 4 * H), np.float32)
    mi_h = np.zeros((2, 4, 4 * H), np.float32)
    for di, d in enumerate(('fwd', 'bwd')):
        z = x @ p[d]['W'] + (0 if use_mi else p[d]['b'])
        zx[:, :N, di] = gate_major_to_unit_major(z, H)
        if not use_mi:
            zx[:, N:, di] = gate_major_to_unit_major(p[d]['b'][None, None], H)
        U[di] = gate_major_to_unit_major(p[d]['U'], H)
        if use_mi:
            for k in range(3):
                mi_h[di, k] = gate_major_to_unit_major(mi[d][k], H)
            mi_h[di, 3] = gate_major_to_unit_major(p[d]['b'], H)
    if use_mi:
        mi_d = to_dev(mi_h)
    if use_zone:
        zc_d = to_dev(np.stack([zone['fwd'][0], zone['bwd'][0]], axis=1).astype(np.float32))
        zh_d = to_dev(np.stack([zone['fwd'][1], zone['bwd'][1]], axis=1).astype(np.float32))
    zx_d, U_d = to_dev(zx), to_dev(U)
    y = torch.zeros(T, n_pad, 2 * H, device=dev)
    cell = torch.zeros(T, n_pad, 2, H, device=dev)
    gates = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    uh = torch.zeros(T, n_pad, 2, 4 * H, device=dev) if use_mi else None
    ops.lstm_seq_fwd(zx_d, U_d, y, cell, gates, T, n_pad, H, check=True, mi=mi_d, uh=uh,
                     zone_c=zc_d, zone_h=zh_d)
    yh, ch = y.cpu().numpy(), cell.cpu().numpy()
    tag = 'variants T%d N%d H%d mi%d z%d' % (T, N, H, use_mi, use_zone)
    for di, d in enumerate(('fwd', 'bwd')):
        assert report('h %s %s' % (d, tag), yh[:, :N, di * H:(di + 1) * H], want[d]['hs']) < 1e-4
        assert report('c %s %s' % (d, tag), ch[:, :N, di], want[d]['cache']['cs']) < 1e-4
    dy = np.zeros((T, n_pad, 2 * H), np.float32)
    dy[:, :N, :H], dy[:, :N, H:] = dhs['fwd'], dhs['bwd']
    dz = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    dwx = torch.zeros(T, n_pad, 2, 4 * H, device=dev) if use_mi else None
    dmi = torch.zeros(n_pad // 16, 2, 4, 4 * H, device=dev) if use_mi else None
    ops.lstm_seq_bwd(to_dev(dy), U_d, cell, gates, dz, T, n_pad, H, check=True, mi=mi_d, uh=uh,
                     zone_c=zc_d, zone_h=zh_d, wx=zx_d if use_mi else None, dwx=dwx, dmi=dmi)
    dzh = dz.cpu().numpy()
    for di, d in enumerate(('fwd', 'bwd')):
        c = want[d]['cache']
        w = gate_major_to_unit_major(c['das'], H)
        scale = max(1.0, np.abs(w).max())
        assert report('d(h@U) %s %s' % (d, tag), dzh[:, :N, di], w) < 1e-4 * scale
        if use_mi:
            w2 = gate_major_to_unit_major(c['dwxs'], H)
            assert report('d(x@W) %s %s' % (d, tag), dwx.cpu().numpy()[:, :N, di], w2) \
                < 1e-4 * max(1.0, np.abs(w2).max())
            got = dmi.cpu().numpy().sum(axis=0)[di]
            db = gate_major_to_unit_major(c['dzs'].sum(axis=(0, 1)), H)
            for k, ref in enumerate([gate_major_to_unit_major(g, H) for g in c['dmi']] + [db]):
                assert report('dmi[%d] %s %s' % (k, d, tag), got[k], ref) \
                    < 1e-4 * max(1.0, np.abs(ref).max())
    assert np.all(dzh[:, N:] == 0)


@pytest.mark.parametrize('act', ['relu', 'sigmoid', 'hard_sigmoid', 'linear', 'softsign', 'softplus'])
@pytest.mark.parametrize('T,N,H,use_zone', [(23, 5, 16, False), (31, 20, 100, True), (17, 16, 256, False)])
def test_cell_activation_hyper_parameter(T, N, H, use_zone, act):
    """asr_lstm_args.activation (the reference LSTM's `activation`: g = act(z_c), h = o act(c),
    core/layers.py:452, :463; brsmv1 passes it through, core/models.py:220, :271) on the variant
    kernels vs the oracle: activations 1e-4, gate gradients 1e-4 * max; with and without zoneout,
    ragged K, the widths of the wide kernels (which must step aside for it)."""
    from asr_study_amd import ops
    F = 7
    rs, x, p, masks = _case(T, N, F, H, 5 * T + H, False)
    for d in ('fwd', 'bwd'):        # keep relu / linear cells in a sane range
        p[d]['U'] = p[d]['U'] * 0.5
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    zone = None
    if use_zone:
        zone = {d: ((rs.rand(T, H) > 0.3).astype(np.float64), np.full((T, H), 0.85))
                for d in ('fwd', 'bwd')}
    dhs = {d: rs.randn(T, N, H) for d in ('fwd', 'bwd')}
    want = {}
    for d, rev in (('fwd', False), ('bwd', True)):
        hs, cache = OL.lstm_forward(x, p[d]['W'], p[d]['U'], p[d]['b'], rev, None, None, None,
                                    *(zone[d] if zone else (None, None)), act=act)
        OL.lstm_backward(dhs[d], cache)
        want[d] = dict(hs=hs, cache=cache)
    zx = np.zeros((T, n_pad, 2, 4 * H), np.float32)
    U = np.zeros((2, H, 4 * H), np.float32)
    for di, d in enumerate(('fwd', 'bwd')):
        zx[:, :N, di] = gate_major_to_unit_major(x @ p[d]['W'] + p[d]['b'], H)
        zx[:, N:, di] = gate_major_to_unit_major(p[d]['b'][None, None], H)
        U[di] = gate_major_to_unit_major(p[d]['U'], H)
    zc_d = zh_d = None
    if use_zone:
        zc_d = to_dev(np.stack([zone['fwd'][0], zone['bwd'][0]], axis=1).astype(np.float32))
        zh_d = to_dev(np.stack([zone['fwd'][1], zone['bwd'][1]], axis=1).astype(np.float32))
    zx_d, U_d = to_dev(zx), to_dev(U)
    y = torch.zeros(T, n_pad, 2 * H, device=dev)
    cell = torch.zeros(T, n_pad, 2, H, device=dev)
    gates = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    ops.lstm_seq_fwd(zx_d, U_d, y, cell, gates, T, n_pad, H, check=True, zone_c=zc_d, zone_h=zh_d,
                     act=act)
    yh, ch = y.cpu().numpy(), cell.cpu().numpy()
    tag = 'act=%s T%d N%d H%d z%d' % (act, T, N, H, use_zone)
    for di, d in enumerate(('fwd', 'bwd')):
        sc = max(1.0, np.abs(want[d]['cache']['cs']).max())
        assert report('h %s %s' % (d, tag), yh[:, :N, di * H:(di + 1) * H], want[d]['hs']) < 1e-4 * sc
        assert report('c %s %s' % (d, tag), ch[:, :N, di], want[d]['cache']['cs']) < 1e-4 * sc
    dy = np.zeros((T, n_pad, 2 * H), np.float32)
    dy[:, :N, :H], dy[:, :N, H:] = dhs['fwd'], dhs['bwd']
    dz = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    ops.lstm_seq_bwd(to_dev(dy), U_d, cell, gates, dz, T, n_pad, H, check=True, zone_c=zc_d,
                     zone_h=zh_d, act=act)
    dzh = dz.cpu().numpy()
    for di, d in enumerate(('fwd', 'bwd')):
        w = gate_major_to_unit_major(want[d]['cache']['dzs'], H)
        assert report('dz %s %s' % (d, tag), dzh[:, :N, di], w) < 1e-4 * max(1.0, np.abs(w).max())
    assert np.all(dzh[:, N:] == 0)


@pytest.mark.parametrize('T,N,H,use_mi,use_zone,use_mask', [
    (13, 5, 16, False, False, False),
    (17, 20, 100, True, True, True),      # two batch tiles, ragged units per thread
    (9, 16, 512, True, False, False),     # two units per thread
])
def test_layer_normalised_cell(T, N, H, use_mi, use_zone, use_mask):
    """asr_lstm_ln_seq_fwd / _bwd (layer_norm option, core/layers.py:407-436, 460-462) vs the
    oracle: h, c 1e-4; d/d(h@U), d/d(x@W) and every parameter gradient 1e-4 * max."""
    from asr_study_amd import ops
    F = 7
    rs, x, p, masks = _case(T, N, F, H, 5 * T + H, use_mask)
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    ln = {d: {'Uh': [1 + 0.2 * rs.randn(4 * H), 0.1 * rs.randn(4 * H)],
              'Wx': [1 + 0.2 * rs.randn(4 * H), 0.1 * rs.randn(4 * H)],
              'new_c': [1 + 0.2 * rs.randn(H), 0.1 * rs.randn(H)]} for d in ('fwd', 'bwd')}
    mi = {d: [1.0 + 0.3 * rs.randn(4 * H), 0.6 + 0.3 * rs.randn(4 * H),
              0.7 + 0.3 * rs.randn(4 * H)] for d in ('fwd', 'bwd')} if use_mi else None
    zone = {d: ((rs.rand(T, H) > 0.3).astype(np.float64), np.full((T, H), 0.85))
            for d in ('fwd', 'bwd')} if use_zone else None
    dhs = {d: rs.randn(T, N, H) for d in ('fwd', 'bwd')}
    want = {}
    for d, rev in (('fwd', False), ('bwd', True)):
        hs, cache = OL.lstm_forward(x, p[d]['W'], p[d]['U'], p[d]['b'], rev, None, masks[d],
                                    mi[d] if mi else None, *(zone[d] if zone else (None, None)),
                                    ln=ln[d])
        OL.lstm_backward(dhs[d], cache)
        want[d] = dict(hs=hs, cache=cache)
    um = gate_major_to_unit_major
    wx = np.zeros((T, n_pad, 2, 4 * H), np.float32)
    U = np.zeros((2, H, 4 * H), np.float32)
    cellp = np.zeros((2, 34 * H), np.float32)
    mk = np.ones((2, n_pad, H), np.float32)
    for di, d in enumerate(('fwd', 'bwd')):
        wx[:, :N, di] = um(x @ p[d]['W'], H)
        U[di] = um(p[d]['U'], H)
        blocks = ([um(v, H) for v in mi[d]] if use_mi else [np.zeros(4 * H)] * 3) + \
            [um(p[d]['b'], H), um(ln[d]['Uh'][0], H), um(ln[d]['Uh'][1], H),
             um(ln[d]['Wx'][0], H), um(ln[d]['Wx'][1], H), ln[d]['new_c'][0], ln[d]['new_c'][1]]
        cellp[di] = np.concatenate(blocks)
        if use_mask:
            mk[di, :N] = masks[d]
    mk_d = to_dev(mk) if use_mask else None
    zc_d = zh_d = None
    if use_zone:
        zc_d = to_dev(np.stack([zone['fwd'][0], zone['bwd'][0]], axis=1).astype(np.float32))
        zh_d = to_dev(np.stack([zone['fwd'][1], zone['bwd'][1]], axis=1).astype(np.float32))
    wx_d, U_d, cp_d = to_dev(wx), to_dev(U), to_dev(cellp)
    z4 = lambda: torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    uh, gates, duh, dwx = z4(), z4(), z4(), z4()
    y = torch.zeros(T, n_pad, 2 * H, device=dev)
    cell = torch.zeros(T, n_pad, 2, H, device=dev)
    ops.lstm_ln_seq_fwd(wx_d, U_d, cp_d, uh, y, cell, gates, T, n_pad, H, has_mi=use_mi,
                        mask_u=mk_d, zone_c=zc_d, zone_h=zh_d)
    yh, ch = y.cpu().numpy(), cell.cpu().numpy()
    tag = 'LN T%d N%d H%d mi%d z%d' % (T, N, H, use_mi, use_zone)
    for di, d in enumerate(('fwd', 'bwd')):
        assert report('h %s %s' % (d, tag), yh[:, :N, di * H:(di + 1) * H], want[d]['hs']) < 1e-4
        assert report('c %s %s' % (d, tag), ch[:, :N, di], want[d]['cache']['cs']) < 1e-4
    dy = np.zeros((T, n_pad, 2 * H), np.float32)
    dy[:, :N, :H], dy[:, :N, H:] = dhs['fwd'], dhs['bwd']
    dparams = torch.zeros(n_pad, 2, 34 * H, device=dev)
    ops.lstm_ln_seq_bwd(to_dev(dy), wx_d, U_d, cp_d, uh, y, cell, gates, duh, dwx, dparams,
                        T, n_pad, H, has_mi=use_mi, mask_u=mk_d, zone_c=zc_d, zone_h=zh_d)
    got_p = dparams.cpu().numpy()[:N].sum(axis=0)
    # LN of the all-zero h@U row of the first step has 1/sqrt(eps) = 316 as its scale: the
    # gradients through it are amplified by that factor, and so is float32 round-off
    # (the oracle runs in float64) -- wide rows get a tolerance of 1e-3 of the maximum
    gtol = 1e-4 if H <= 128 else 1e-3
    for di, d in enumerate(('fwd', 'bwd')):
        c = want[d]['cache']
        for name, got, ref in (('duh', duh.cpu().numpy()[:, :N, di], um(c['das'], H)),
                               ('dwx', dwx.cpu().numpy()[:, :N, di], um(c['dwxs'], H))):
            assert report('%s %s %s' % (name, d, tag), got, ref) < gtol * max(1.0, np.abs(ref).max())
        refs = ([um(g, H) for g in c['dmi']] if use_mi else [np.zeros(4 * H)] * 3) + \
            [um(c['dzs'].sum(axis=(0, 1)), H), um(c['dln']['Uh'][0], H), um(c['dln']['Uh'][1], H),
             um(c['dln']['Wx'][0], H), um(c['dln']['Wx'][1], H), c['dln']['new_c'][0],
             c['dln']['new_c'][1]]
        ref = np.concatenate(refs)
        assert report('dparams %s %s' % (d, tag), got_p[di], ref) < gtol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('N,H', [(32, 256), (48, 256), (16, 256), (64, 256)])
@pytest.mark.parametrize('prog', ['0', '1'])
def test_forward_eight_units_per_workgroup_is_bit_identical_to_sixteen(N, H, prog, monkeypatch):
    """lstm_fwd_kernel_x<.., NJ = 2> (eight units per workgroup: H/8 workgroups per chain, what
    asr_lstm_plan launches where the layer then still leaves half of the CUs free, e.g. cfg2)
    against NJ = 4 on the same problem: a (sample, unit)'s products and their summation order do
    not depend on the geometry, so h, c and the gates must agree BIT FOR BIT -- whole and sliced,
    both transports, single-gather and progressive step, with a recurrent-dropout mask.  (H = 256
    only: a chain's H/8 workgroups share one XCD of 32 CUs and must all be resident there.)"""
    from asr_study_amd import ops
    T = 41
    rs = np.random.RandomState(7 * H + N)
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    zx = torch.from_numpy(rs.randn(T, n_pad, 2, 4 * H).astype(np.float32)).to(dev)
    U = torch.from_numpy((rs.randn(2, H, 4 * H) / np.sqrt(H)).astype(np.float32)).to(dev)
    mask = torch.from_numpy(((rs.rand(2, n_pad, H) > 0.2) / 0.8).astype(np.float32)).to(dev)
    monkeypatch.setenv('ASR_LSTM_PROG', prog)

    def run(eight, ranges, units=None):
        if eight is None:
            monkeypatch.delenv('ASR_LSTM_FWD8', raising=False)
        else:
            monkeypatch.setenv('ASR_LSTM_FWD8', eight)
        y = torch.full((T, n_pad, 2 * H), 3.0, device=dev)
        cell = torch.full((T, n_pad, 2, H), 4.0, device=dev)
        gates = torch.full((T, n_pad, 2, 4 * H), 5.0, device=dev)
        for i, r in enumerate(ranges):
            ws = ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H, mask_u=mask, steps=r,
                                  units=units[i] if units else 0)
        ops.lstm_status(ws)
        return [t.cpu().numpy() for t in (y, cell, gates)]
    want = run('0', [None])
    got = run('1', [None])
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    sliced = run('1', [(0, 1), (1, 2), (3, 17), (20, 21)])
    for a, b in zip(sliced, want):
        assert np.array_equal(a, b)
    for transport in ('0', '1'):
        monkeypatch.setenv('ASR_LSTM_FAST', transport)
        again = run('1', [None])
        for a, b in zip(again, want):
            assert np.array_equal(a, b), transport
    # asr_lstm_args.fwd_units: slices of ONE sequence on different geometries (what the engine
    # does: eight units while nothing runs beside the recurrence, sixteen for the slice that
    # shares the chip with the next layer's GEMMs)
    monkeypatch.delenv('ASR_LSTM_FAST', raising=False)
    mixed = run(None, [(0, 5), (5, 20), (25, 16)], units=[8, 16, 8])
    for a, b in zip(mixed, want):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('N,H', [(32, 256), (64, 512), (16, 512), (48, 256)])
def test_specialised_kernels_agree_with_the_generic_ones_and_bptt_emits_bias_gradient(N, H, monkeypatch):
    """H = 256 / 512 run on specialised kernels (forward lstm_fwd_kernel_x: K split over the
    waves, U fragments in AGPRs; BPTT lstm_bwd_kernel_x / _c); ASR_LSTM_GENERIC=1 forces the
    any-H kernels (lstm_*_kernel_h) on the same problem.  Forward: activations to 2e-6;
    sliced == whole and both transports bit for bit.  BPTT (one-dimensional split, the
    two-dimensional one has its own test): gate gradients to 1e-6 of the largest, sliced ==
    whole and both transports bit for bit; db_part (per-batch-tile sums of dz over samples and
    steps) == the sums of the dz slab it wrote, from either kernel, also in slices."""
    from asr_study_amd import ops
    T = 45
    rs = np.random.RandomState(H + N)
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    zx = torch.from_numpy(rs.randn(T, n_pad, 2, 4 * H).astype(np.float32)).to(dev)
    U = torch.from_numpy((rs.randn(2, H, 4 * H) / np.sqrt(H)).astype(np.float32)).to(dev)
    dy = torch.from_numpy((rs.randn(T, n_pad, 2 * H) * 0.1).astype(np.float32)).to(dev)
    mask = torch.from_numpy(((rs.rand(2, n_pad, H) > 0.2) / 0.8).astype(np.float32)).to(dev)

    def fwd(ranges):
        y = torch.full((T, n_pad, 2 * H), 3.0, device=dev)
        cell = torch.full((T, n_pad, 2, H), 3.0, device=dev)
        gates = torch.full((T, n_pad, 2, 4 * H), 3.0, device=dev)
        for r in ranges:
            ws = ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H, mask_u=mask, steps=r)
        ops.lstm_status(ws)
        return y, cell, gates
    monkeypatch.setenv('ASR_LSTM_GENERIC', '1')
    want_f = [t.cpu().numpy() for t in fwd([None])]
    monkeypatch.setenv('ASR_LSTM_GENERIC', '0')
    y, cell, gates = fwd([None])
    got_f = [t.cpu().numpy() for t in (y, cell, gates)]
    for a, b in zip(got_f, want_f):
        assert np.abs(a - b).max() < 2e-6 * max(1.0, np.abs(b).max())
    for ranges, transport in (([(0, 1), (1, 16), (17, 28)], '1'), ([(0, 30), (30, 15)], '0'),
                              ([None], '0')):
        monkeypatch.setenv('ASR_LSTM_FAST', transport)
        for a, b in zip([t.cpu().numpy() for t in fwd(ranges)], got_f):
            assert np.array_equal(a, b), (ranges, transport)
    monkeypatch.delenv('ASR_LSTM_FAST')

    def run(ranges):
        dz = torch.full((T, n_pad, 2, 4 * H), 7.0, device=dev)
        dbp = torch.full((n_pad // 16, 2, 4 * H), 9.0, device=dev)
        amax = torch.zeros(1, device=dev)
        for r in ranges:
            ws = ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H, mask_u=mask,
                                  dz_absmax=amax, steps=r, db_part=dbp)
        ops.lstm_status(ws)
        want = dz.double().reshape(T, n_pad // 16, 16, 2, 4 * H).sum(dim=(0, 2))
        err = (dbp.double() - want).abs().max().item()
        assert err < 2e-5 * max(1.0, want.abs().max().item()), err
        return dz.cpu().numpy(), amax.cpu().numpy()
    monkeypatch.setenv('ASR_LSTM_BWD_2D', '0')
    monkeypatch.setenv('ASR_LSTM_GENERIC', '1')
    want, _ = run([None])
    run([(0, 20), (20, 25)])
    monkeypatch.setenv('ASR_LSTM_GENERIC', '0')
    single, amax = run([None])
    assert np.abs(single - want).max() < 1e-6 * np.abs(want).max()
    assert np.abs(single).max() == amax[0]
    sliced, amax2 = run([(0, 1), (1, 16), (17, 28)])
    assert np.array_equal(single, sliced) and amax2[0] == amax[0]
    for transport in ('0', '1'):
        monkeypatch.setenv('ASR_LSTM_FAST', transport)
        again, _ = run([None])
        assert np.array_equal(single, again)


@pytest.mark.parametrize('N,H', [(64, 512), (16, 512), (32, 256), (48, 256)])
def test_bptt_two_dimensional_split_matches_the_one_dimensional_kernel(N, H, monkeypatch):
    """lstm_bwd_kernel_c (ASR_LSTM_BWD_2D=1, the default at H = 512: workgroup (a, b) owns the
    reduction slice of 64 units AND one of four output blocks, publishes H/64 partial sums
    instead of H/16, four exchange slots) against lstm_bwd_kernel_x on the same activations:
    the same products summed in another order -> gate gradients to 1e-6 of the largest, max|dz|
    exact; sliced == whole and both transports bit for bit; db_part == the sums of the dz slab;
    rows of padding samples stay zero; with a recurrent-dropout mask."""
    from asr_study_amd import ops
    T = 53
    rs = np.random.RandomState(3 * H + N)
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    zx = torch.from_numpy(rs.randn(T, n_pad, 2, 4 * H).astype(np.float32)).to(dev)
    U = torch.from_numpy((rs.randn(2, H, 4 * H) / np.sqrt(H)).astype(np.float32)).to(dev)
    dy = torch.from_numpy((rs.randn(T, n_pad, 2 * H) * 0.1).astype(np.float32)).to(dev)
    mask = torch.from_numpy(((rs.rand(2, n_pad, H) > 0.2) / 0.8).astype(np.float32)).to(dev)
    y = torch.zeros(T, n_pad, 2 * H, device=dev)
    cell = torch.zeros(T, n_pad, 2, H, device=dev)
    gates = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    ops.lstm_status(ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H, mask_u=mask))

    def run(ranges):
        dz = torch.full((T, n_pad, 2, 4 * H), 7.0, device=dev)
        dbp = torch.full((n_pad // 16, 2, 4 * H), 9.0, device=dev)
        amax = torch.zeros(1, device=dev)
        for r in ranges:
            ws = ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H, mask_u=mask,
                                  dz_absmax=amax, steps=r, db_part=dbp)
        ops.lstm_status(ws)
        want = dz.double().reshape(T, n_pad // 16, 16, 2, 4 * H).sum(dim=(0, 2))
        err = (dbp.double() - want).abs().max().item()
        assert err < 2e-5 * max(1.0, want.abs().max().item()), err
        return dz.cpu().numpy(), amax.cpu().numpy()
    monkeypatch.setenv('ASR_LSTM_BWD_2D', '0')
    want, amax0 = run([None])
    monkeypatch.setenv('ASR_LSTM_BWD_2D', '1')
    got, amax = run([None])
    err = report('dz 2-D vs 1-D split N%d H%d' % (N, H), got, want)
    assert err < 1e-6 * np.abs(want).max()
    assert np.abs(got).max() == amax[0] and abs(amax[0] - amax0[0]) < 1e-6 * amax0[0]
    sliced, amax2 = run([(0, 1), (1, 2), (3, 17), (20, 33)])
    bad = np.argwhere(got != sliced)
    assert bad.size == 0, ('sliced != whole', len(bad), bad[:4].tolist(), bad[-1].tolist(),
                           float(np.abs(got - sliced).max()))
    assert amax2[0] == amax[0]
    for transport in ('0', '1'):
        monkeypatch.setenv('ASR_LSTM_FAST', transport)
        again, _ = run([None])
        bad = np.argwhere(got != again)
        assert bad.size == 0, ('transport', transport, len(bad), bad[:4].tolist(),
                               float(np.abs(got - again).max()))


@pytest.mark.parametrize('N,H', [(64, 512), (16, 512), (32, 256), (48, 256)])
def test_bptt_compact_geometry_is_bit_identical_to_the_default_two_dimensional_split(N, H, monkeypatch):
    """asr_lstm_args.compact (lstm_bwd_kernel_c<.., NBLK = 2>: two output blocks, H/32 workgroups
    per chain, half the CUs per layer -- what engine.backward launches beside the weight-gradient
    GEMMs of the layer above) against the four-block form on the same activations: every
    (sample, unit)'s products and their summation order are the same, so dz, max|dz| and the
    bias-gradient partials must agree BIT FOR BIT; sliced == whole, both transports; the plan
    reports half the workgroups."""
    from asr_study_amd import ops
    import ctypes as C
    from asr_study_amd import _lib as L
    T = 47
    rs = np.random.RandomState(5 * H + N)
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    zx = torch.from_numpy(rs.randn(T, n_pad, 2, 4 * H).astype(np.float32)).to(dev)
    U = torch.from_numpy((rs.randn(2, H, 4 * H) / np.sqrt(H)).astype(np.float32)).to(dev)
    dy = torch.from_numpy((rs.randn(T, n_pad, 2 * H) * 0.1).astype(np.float32)).to(dev)
    mask = torch.from_numpy(((rs.rand(2, n_pad, H) > 0.2) / 0.8).astype(np.float32)).to(dev)
    y = torch.zeros(T, n_pad, 2 * H, device=dev)
    cell = torch.zeros(T, n_pad, 2, H, device=dev)
    gates = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    ops.lstm_status(ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H, mask_u=mask))
    monkeypatch.setenv('ASR_LSTM_BWD_2D', '1')

    def run(ranges, compact):
        dz = torch.full((T, n_pad, 2, 4 * H), 7.0, device=dev)
        dbp = torch.full((n_pad // 16, 2, 4 * H), 9.0, device=dev)
        amax = torch.zeros(1, device=dev)
        for r in ranges:
            ws = ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H, mask_u=mask,
                                  dz_absmax=amax, steps=r, db_part=dbp, compact=compact)
        ops.lstm_status(ws)
        return dz.cpu().numpy(), dbp.cpu().numpy(), amax.cpu().numpy()
    want, dbw, amw = run([None], False)
    got, dbg, amg = run([None], True)
    bad = np.argwhere(got != want)
    assert bad.size == 0, ('compact != default', len(bad), bad[:4].tolist(),
                           float(np.abs(got - want).max()))
    assert np.array_equal(dbg, dbw) and amg[0] == amw[0]
    sliced, dbs, ams = run([(0, 1), (1, 2), (3, 17), (20, 27)], True)
    assert np.array_equal(sliced, got) and ams[0] == amg[0]
    assert np.abs(dbs - dbg).max() <= 2e-6 * max(1.0, np.abs(dbg).max())
    for transport in ('0', '1'):
        monkeypatch.setenv('ASR_LSTM_FAST', transport)
        again, _, _ = run([None], True)
        assert np.array_equal(again, got), transport
    # the plan: half the workgroups per chain
    a = L.LstmArgs()
    a.T, a.n_pad, a.H = T, n_pad, H
    blocks, cpl = C.c_int(), C.c_int()
    out = []
    for c in (0, 1):
        a.compact = c
        L.check(L.load().asr_lstm_plan(C.byref(a), 1, None, None, C.byref(blocks), C.byref(cpl)),
                'asr_lstm_plan')
        out.append(blocks.value // max(1, cpl.value))
    assert out == [H // 16, H // 32], out


@pytest.mark.parametrize('N,H', [(32, 256), (64, 512), (16, 512)])
def test_exact_fp32_kernels_at_the_benchmarked_widths(N, H, monkeypatch):
    """ASR_LSTM_PREC=0 at H = 256 / 512: the structure of the split-fp16 kernels (forward: K split
    over the waves; BPTT: two-dimensional split) on v_mfma_f32_16x16x4_f32, against the float64
    oracle (activations to 5e-6, gate gradients to 1e-5 of the largest: fp32 rounding only) and
    against the split-fp16 default (1e-5 / 2e-5); sliced == whole bit for bit; db_part == the
    sums of the dz slab; with a recurrent-dropout mask.  Any other width / mode is refused."""
    from asr_study_amd import ops
    from asr_study_amd._lib import AsrHipError
    T = 41
    rs, x, p, masks = _case(T, N, 8, H, 7 * H + N, True)
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    dhs = {d: rs.randn(T, N, H) * 0.1 for d in ('fwd', 'bwd')}
    want = _oracle(x, p, masks, dhs)
    zx_h, U_h, mk_h = _pack_inputs(x, p, masks, H, n_pad)
    zx, U, mask = to_dev(zx_h), to_dev(U_h), to_dev(mk_h)
    dy_h = np.zeros((T, n_pad, 2 * H), np.float32)
    dy_h[:, :N, :H] = dhs['fwd']
    dy_h[:, :N, H:] = dhs['bwd']
    dy = to_dev(dy_h)

    def run(ranges):
        y = torch.full((T, n_pad, 2 * H), 3.0, device=dev)
        cell = torch.full((T, n_pad, 2, H), 3.0, device=dev)
        gates = torch.full((T, n_pad, 2, 4 * H), 3.0, device=dev)
        dz = torch.full((T, n_pad, 2, 4 * H), 7.0, device=dev)
        dbp = torch.full((n_pad // 16, 2, 4 * H), 9.0, device=dev)
        for r in ranges:
            ws = ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H, mask_u=mask, steps=r)
        ops.lstm_status(ws)
        for r in ranges:
            ws = ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H, mask_u=mask, steps=r,
                                  db_part=dbp)
        ops.lstm_status(ws)
        sums = dz.double().reshape(T, n_pad // 16, 16, 2, 4 * H).sum(dim=(0, 2))
        assert (dbp.double() - sums).abs().max().item() < 2e-5 * max(1.0, sums.abs().max().item())
        return [t.cpu().numpy() for t in (y, cell, gates, dz)]
    monkeypatch.setenv('ASR_LSTM_PREC', '1')
    split = run([None])
    monkeypatch.setenv('ASR_LSTM_PREC', '0')
    got = run([None])
    tag = 'N%d H%d' % (N, H)
    for di, d in enumerate(('fwd', 'bwd')):
        ref = dict(y=want[d]['hs'], cell=want[d]['cache']['cs'],
                   gates=gate_major_to_unit_major(want[d]['cache']['gates'], H),
                   dz=gate_major_to_unit_major(want[d]['cache']['dzs'], H))
        mine = dict(y=got[0][:, :N, di * H:(di + 1) * H], cell=got[1][:, :N, di],
                    gates=got[2][:, :N, di], dz=got[3][:, :N, di])
        for name in ('y', 'cell', 'gates', 'dz'):
            scale = max(1.0, np.abs(ref[name]).max()) if name != 'dz' else np.abs(ref[name]).max()
            tol = 1e-5 if name == 'dz' else 5e-6
            assert report('exact %s %s vs float64 %s' % (name, d, tag), mine[name], ref[name]) < tol * scale
    for name, a, c in zip(('y', 'cell', 'gates', 'dz'), got, split):
        scale = max(1.0, np.abs(a).max()) if name != 'dz' else np.abs(a).max()
        assert report('exact %s vs split-fp16' % name, a, c) < (2e-5 if name == 'dz' else 1e-5) * scale
    sliced = run([(0, 1), (1, 13), (14, 27)])
    for name, a, b in zip(('y', 'cell', 'gates', 'dz'), got, sliced):
        assert np.array_equal(a, b), name
    # the exact arithmetic is built for these widths in persistent mode only
    with pytest.raises(AsrHipError):
        ops.lstm_seq_fwd(zx, U, torch.empty_like(dy), torch.empty(T, n_pad, 2, H, device=dev),
                         torch.empty_like(zx), T, n_pad, H, mask_u=mask, mode=1)


@pytest.mark.parametrize('H', [256, 512])
def test_single_utterance_forward_kernel(H):
    """asr_lstm_args.n_valid = 1 (predict.py: one utterance per call): the tile-free exact-fp32
    forward kernel against the float64 oracle and against the batch kernel on row 0; sliced ==
    whole bit for bit; rows 1.. of the slabs are left untouched."""
    from asr_study_amd import ops
    T, N, F = 60, 1, 10
    rs, x, p, masks = _case(T, N, F, H, 3 * H, False)
    n_pad = 16
    want = _oracle(x, p, masks, {d: rs.randn(T, N, H) for d in ('fwd', 'bwd')})
    zx, U, mk = _pack_inputs(x, p, masks, H, n_pad)
    zx_d, U_d = to_dev(zx), to_dev(U)

    def run(n_valid, ranges):
        y = torch.full((T, n_pad, 2 * H), 3.0, dtype=torch.float32, device='cuda:0')
        cell = torch.full((T, n_pad, 2, H), 3.0, dtype=torch.float32, device='cuda:0')
        gates = torch.full((T, n_pad, 2, 4 * H), 3.0, dtype=torch.float32, device='cuda:0')
        for r in ranges:
            ws = ops.lstm_seq_fwd(zx_d, U_d, y, cell, gates, T, n_pad, H, steps=r, n_valid=n_valid)
        ops.lstm_status(ws)
        return y.cpu().numpy(), cell.cpu().numpy(), gates.cpu().numpy()
    y1, c1, g1 = run(1, [None])
    for di, d in enumerate(('fwd', 'bwd')):
        assert report('n1 h %s H%d' % (d, H), y1[:, :1, di * H:(di + 1) * H], want[d]['hs']) < 1e-5
        assert report('n1 c %s H%d' % (d, H), c1[:, :1, di], want[d]['cache']['cs']) < 1e-5
        assert report('n1 gates %s H%d' % (d, H), g1[:, :1, di],
                      gate_major_to_unit_major(want[d]['cache']['gates'], H)) < 1e-5
    assert np.all(y1[:, 1:] == 3.0) and np.all(c1[:, 1:] == 3.0) and np.all(g1[:, 1:] == 3.0)
    yb, cb, gb = run(0, [None])
    assert np.abs(yb[:, 0] - y1[:, 0]).max() < 5e-6 and np.abs(cb[:, 0] - c1[:, 0]).max() < 5e-6
    ys, cs, gs = run(1, [(0, 1), (1, 40), (41, 19)])
    assert np.array_equal(ys, y1) and np.array_equal(cs, c1) and np.array_equal(gs, g1)


def _planes_to_float(pl, rows, cols):
    """(hi + lo) / scale of packed planes -> float64 (rows, cols)."""
    hi = pl.hi[:rows, :cols].double().cpu().numpy()
    lo = pl.lo[:rows, :cols].double().cpu().numpy()
    return (hi + lo) / float(pl.scale.item())


@pytest.mark.parametrize('N,H,compact', [(64, 512, False), (64, 512, True), (16, 512, True),
                                         (32, 256, True), (48, 256, False)])
def test_bptt_writes_packed_planes_instead_of_the_fp32_slab(N, H, compact, monkeypatch):
    """asr_lstm_args.dz_hl (lstm_bwd_kernel_c<.., PL>): BPTT stores the gate gradients as the
    packed planes asr_pack_hl would make of them -- pre-scaled for a bound handed in BEFORE the
    pass -- and uses that split for its own dz @ U^T products.  Against the fp32-slab kernel on the
    same activations: planes == dz to 2^-21 of max|dz| (a split's resolution) + the recurrence's
    own rounding differences (1e-6 of the maximum), for a tight bound (8 x max) and for bounds
    256 x too large / 16 x too small; the scale word is asr_pack_hl's scale of the bound; max|dz|
    and the bias partials agree; compact == default and sliced == whole bit for bit; the planes
    feed asr_gemm_hl like packed ones (dz @ W^T against float64)."""
    from asr_study_amd import ops
    T = 45
    rs = np.random.RandomState(11 * H + N + int(compact))
    n_pad = ops.pad16(N)
    dev = 'cuda:0'
    zx = torch.from_numpy(rs.randn(T, n_pad, 2, 4 * H).astype(np.float32)).to(dev)
    U = torch.from_numpy((rs.randn(2, H, 4 * H) / np.sqrt(H)).astype(np.float32)).to(dev)
    dy = torch.from_numpy((rs.randn(T, n_pad, 2 * H) * 0.03).astype(np.float32)).to(dev)
    mask = torch.from_numpy(((rs.rand(2, n_pad, H) > 0.2) / 0.8).astype(np.float32)).to(dev)
    y = torch.zeros(T, n_pad, 2 * H, device=dev)
    cell = torch.zeros(T, n_pad, 2, H, device=dev)
    gates = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    ops.lstm_status(ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H, mask_u=mask))
    monkeypatch.setenv('ASR_LSTM_BWD_2D', '1')
    assert ops.lstm_dz_hl_supported(T, n_pad, H, compact=compact)
    assert not ops.lstm_dz_hl_supported(T, n_pad, H, mode=1)          # stepwise kernels: no
    assert not ops.lstm_dz_hl_supported(T, n_pad, 100)                # any-H kernels: no
    rows = T * n_pad

    dz = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    dbw = torch.zeros(n_pad // 16, 2, 4 * H, device=dev)
    amw = torch.zeros(1, device=dev)
    ops.lstm_status(ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H, mask_u=mask,
                                     dz_absmax=amw, db_part=dbw, compact=compact))
    want = dz.double().cpu().numpy().reshape(rows, 8 * H)
    zmax = float(amw.item())
    assert zmax > 0 and abs(zmax - np.abs(want).max()) <= 1e-6 * zmax

    def run(bound_value, ranges=(None,), cmp=compact):
        pl = ops.HlPlanes(rows, 8 * H, dev)
        pl.hl.fill_(3.0)
        bound = torch.full((1,), bound_value, device=dev)
        dbp = torch.full((n_pad // 16, 2, 4 * H), 9.0, device=dev)
        am = torch.zeros(1, device=dev)
        for r in ranges:
            ws = ops.lstm_seq_bwd(dy, U, cell, gates, None, T, n_pad, H, mask_u=mask, dz_absmax=am,
                                  steps=r, db_part=dbp, compact=cmp, dz_planes=pl, dz_bound=bound)
        ops.lstm_status(ws)
        return pl, dbp.cpu().numpy(), float(am.item())

    for factor, tol in ((8.0, 2.0 ** -21), (2048.0, 2.0 ** -21), (1.0 / 16, 2.0 ** -21)):
        pl, dbp, am = run(zmax * factor)
        # the scale: asr_pack_hl's power of two of the BOUND
        e = np.frexp(np.float32(zmax * factor))[1]
        assert float(pl.scale.item()) == 2.0 ** (9 - e), (factor, pl.scale.item())
        got = _planes_to_float(pl, rows, 8 * H)
        err = np.abs(got - want).max()
        assert err <= (tol + 1e-6) * zmax, (factor, err / zmax)
        assert abs(am - zmax) <= 1e-6 * zmax
        assert np.abs(dbp - dbw.cpu().numpy()).max() <= 2e-6 * max(1.0, np.abs(dbw.cpu().numpy()).max()) * T
    pl, dbp, am = run(zmax * 8.0)
    hl0 = pl.hl.cpu().numpy().copy()
    # compact == default geometry, sliced == whole: bit for bit
    other, dbo, amo = run(zmax * 8.0, cmp=not compact)
    assert np.array_equal(other.hl.cpu().numpy(), hl0) and amo == am and np.array_equal(dbo, dbp)
    sliced, _, ams = run(zmax * 8.0, ranges=[(0, 1), (1, 2), (3, 17), (20, 25)])
    assert np.array_equal(sliced.hl.cpu().numpy(), hl0) and ams == am
    # the planes as a GEMM operand: dx = dz @ W^T
    K = 8 * H
    W = (rs.randn(256, K) * 0.05).astype(np.float32)
    Wd = torch.from_numpy(W).to(dev)
    pw = ops.HlPlanes(256, K, dev)
    ops.pack_hl(Wd, 256, K, absmax=ops.absmax(Wd), r=pw)
    out = torch.zeros(rows, 256, device=dev)
    ops.gemm_hl(pl, pw, out, rows, 256, K)
    ref = want @ W.astype(np.float64).T
    assert np.abs(out.double().cpu().numpy() - ref).max() <= 3e-6 * np.abs(ref).max()


def test_dz_guard_keeps_renews_and_flags_the_bound():
    """asr_lstm_dz_guard: M = max * scale(bound); the bound stays while M is in [2^-1, 2^6) and
    becomes 64 max otherwise; a pass that WROTE planes with a bound that let M leave
    [2^-7, 2^15] raises the BPTT workspace's sticky flag (asr_lstm_status -> timeout -> the step is
    vetoed and re-run); zero / measuring passes never flag.  (64 x of headroom: the gate
    gradients above a conv front-end jump 820 x between the first two steps of a run.)"""
    from asr_study_amd import ops
    from asr_study_amd._lib import AsrHipError
    dev = 'cuda:0'
    ws = ops.WS.get('lstm_bwd', 4096, dev)
    ops.lstm_status(ws)

    def guard(m, b, used):
        am = torch.full((1,), m, device=dev)
        bd = torch.full((1,), b, device=dev)
        ops.lstm_dz_guard(am, bd, used)
        flagged = False
        try:
            ops.lstm_status(ws)
        except AsrHipError:
            flagged = True
        return float(bd.item()), flagged
    b0 = float(np.float32(64.0) * np.float32(0.01))
    assert guard(0.01, 0.0, False) == (b0, False)                         # no bound yet -> 64 max
    for m in (0.01, 0.0101, 0.02, 0.0014, 0.09):                          # inside the window: kept
        assert guard(m, b0, True) == (b0, False), m
    nb, fl = guard(1.0, b0, True)                   # 100 x: renewed half-way (geometric), not flagged
    assert abs(nb - np.sqrt(64.0 * b0)) < 1e-5 * nb and not fl
    nb, fl = guard(8.2, b0, True)                                         # 820 x (cfg3_conv, step 2): not flagged
    assert abs(nb - np.sqrt(64.0 * 8.2 * b0)) < 1e-5 * nb and not fl
    # ... and the step after such a spike (the maximum back at a fifth of what it was before) is
    # inside the planes' range with the half-way bound: renewed downwards, not flagged
    nb2, fl = guard(0.002, nb, True)
    assert nb2 == np.float32(np.float32(64.0) * np.float32(0.002)) and not fl
    # persistent growth settles within three steps
    b = b0
    for _ in range(3):
        b, fl = guard(8.2, b, True)
        assert not fl
    assert guard(8.2, b, True) == (b, False)
    nb, fl = guard(1e-4, b0, True)                                        # shrank 100 x: renewed
    assert nb == np.float32(np.float32(64.0) * np.float32(1e-4)) and not fl
    nb, fl = guard(0.01 * 2 ** 14, b0, True)                              # overflow range: flagged
    assert fl and nb > b0
    assert guard(0.01 * 2 ** 14, b0, False)[1] is False                   # measuring pass: never
    nb, fl = guard(0.01 * 2 ** -12, b0, True)                             # far below: flagged
    assert fl
    assert guard(0.0, b0, True) == (b0, False)                            # all-zero pass: untouched

"""K13, the 2-D convolution front-end (csrc/conv.hip, asr_conv2d_*) against the float64 oracle
(oracle/conv.py; no reference counterpart, README.md:118): the segmented-reduction form of the
packed GEMM it is built on, the three entry points at small shapes of every padding / stride
class, the deep_speech2 model end to end, and the full cfg3 geometry (T = 999, 80 features)
against the committed fixture tests/golden/conv_cfg3.npz."""
import os

import numpy as np
import pytest
import torch

from oracle import conv as OC
from oracle import lstm as OL
from tests.gpu_util import dev, to_dev, report

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_gemm_hl_segmented_reduction_range():
    """asr_gemm_hl with a_seg_k: C = sum_i A[rows + shift_i, :seg_k] @ B[:, i*seg_k:(i+1)*seg_k]^T
    (the implicit im2col over time), 256 and 128 tiles, split-K, shifts in any order."""
    from asr_study_amd import ops
    rs = np.random.RandomState(0)
    for (M, N, seg_k, shifts, split) in ((300, 260, 64, [0, 48, 16, 32], 0),
                                         (512, 512, 96, [64, 0, 128], 0),
                                         (130, 70, 32, [5, 0, 9, 2, 7], 0),
                                         (256, 256, 64, [16, 0, 32, 48], 2)):
        rows = M + max(shifts)
        a = rs.randn(rows, seg_k)
        b = rs.randn(N, seg_k * len(shifts))
        pa, pb = ops.HlPlanes(rows, seg_k, dev()), ops.HlPlanes(N, seg_k * len(shifts), dev())
        ta, tb = to_dev(a, torch.float32), to_dev(b, torch.float32)
        ops.pack_hl(ta, rows, seg_k, absmax=ops.absmax(ta), r=pa)
        ops.pack_hl(tb, N, seg_k * len(shifts), absmax=ops.absmax(tb), r=pb)
        c = torch.zeros(M, N, device=dev())
        ops.gemm_hl(pa, pb, c, M, N, seg_k * len(shifts), a_seg_k=seg_k, a_seg_rows=shifts,
                    split_k=split)
        want = sum(a[s:s + M] @ b[:, i * seg_k:(i + 1) * seg_k].T for i, s in enumerate(shifts))
        err = report('seg gemm %dx%dx%dx%d' % (M, N, seg_k, len(shifts)), c.cpu().numpy(), want)
        assert err < 2e-6 * np.abs(a).max() * np.abs(b).max() * seg_k * len(shifts)


def test_gemm_hl_k_major_batch_sharing_b():
    """asr_gemm_hl batch: C_b = A_b^T B for up to 16 row-shifted views A_b of one plane array and
    ONE B, in a single launch (the per-tap weight gradients of the convolution), with and
    without split-K, 256 and 128 tiles."""
    from asr_study_amd import ops
    rs = np.random.RandomState(1)
    for (M, N, K, shifts, split) in ((256, 300, 700, [0, 32, 16, 80], 0),
                                     (80, 132, 500, [3, 0, 11], 3),
                                     (512, 256, 1000, [0, 16, 32, 48, 64, 5, 7, 9, 100, 1, 2], 2)):
        rows = K + max(shifts)
        a = rs.randn(rows, M)
        b = rs.randn(K, N)
        pa, pb = ops.HlPlanes(rows, M, dev()), ops.HlPlanes(K, N, dev())
        ta, tb = to_dev(a, torch.float32), to_dev(b, torch.float32)
        ops.pack_hl(ta, rows, M, absmax=ops.absmax(ta), r=pa)
        ops.pack_hl(tb, K, N, absmax=ops.absmax(tb), r=pb)
        c = torch.full((len(shifts) * M, N), 9.0, device=dev())
        ops.gemm_hl(pa, pb, c, M, N, K, k_major=True, batch_rows=shifts, split_k=split)
        want = np.concatenate([a[s:s + K].T @ b for s in shifts])
        err = report('batched k-major %dx%dx%d x%d' % (M, N, K, len(shifts)), c.cpu().numpy(), want)
        assert err < 2e-6 * np.abs(a).max() * np.abs(b).max() * K


CASES = [
    # T, N, F, Ci, Co, kt, kf, st, sf, clip
    (13, 5, 12, 1, 4, 5, 7, 2, 2, 1.0),        # layer-1 class: C_in = 1, stride (2, 2), odd T
    (14, 16, 12, 1, 4, 5, 7, 2, 2, 1.0),       # even T: the odd padding frame goes behind
    (9, 20, 8, 4, 4, 5, 5, 1, 2, 0.7),         # layer-2 class: stride (1, 2), two batch tiles
    (6, 3, 8, 2, 2, 11, 3, 1, 1, 20.0),        # filter longer than the slab
    (40, 7, 20, 1, 8, 11, 11, 2, 2, 0.0),      # linear output
    (31, 33, 16, 8, 8, 11, 9, 1, 2, 2.0),      # three batch tiles, K per tap = 128
    (20, 5, 24, 32, 48, 5, 9, 1, 2, 3.0),      # C_out = 48 fits no block rule: the whole band, 576 columns
    (10, 5, 24, 32, 64, 3, 9, 1, 2, 3.0),      # three column blocks (fwd), three row blocks (dgrad, dW)
    (11, 17, 32, 32, 32, 3, 5, 2, 1, 0.0),     # four column blocks, four row blocks, stride (2, 1)
    (12, 16, 40, 32, 32, 3, 21, 1, 2, 5.0),    # the second front-end layer's frequency geometry: 3 / 5 blocks
    (6, 16, 41, 32, 32, 3, 5, 1, 2, 0.0),      # ragged last blocks: 5 output frequencies / 1 input frequency
]


@pytest.mark.parametrize('T,N,F,Ci,Co,kt,kf,st,sf,clip', CASES)
def test_conv2d_fwd_dgrad_wgrad_vs_oracle(T, N, F, Ci, Co, kt, kf, st, sf, clip):
    from asr_study_amd import ops
    rs = np.random.RandomState(T * 31 + F)
    n_pad = ops.pad16(N)
    x = np.zeros((T, n_pad, F * Ci))
    x[:, :N] = rs.randn(T, N, F * Ci)
    W = rs.randn(kt, kf, Ci, Co) * 0.3
    b = rs.randn(Co) * 0.2
    want_y, cache = OC.conv2d_forward(x, W, b, (st, sf), clip)
    op = ops.Conv2d(T, n_pad, F, Ci, Co, kt, kf, st, sf, clip, dev())
    assert (op.T_out, op.F_out) == (want_y.shape[0], want_y.shape[2] // Co)
    xd, Wd, bd = to_dev(x, torch.float32), to_dev(W, torch.float32), to_dev(b, torch.float32)
    z = torch.full(want_y.shape, 7.0, device=dev())
    y = torch.full(want_y.shape, 7.0, device=dev()) if clip > 0 else z
    op.fwd(xd, Wd, bd, z, y)
    tol = 2e-6 * max(np.abs(x).max(), 1.0) * np.abs(W).max() * kt * kf * Ci + 1e-6
    assert report('conv z', z.cpu().numpy(), cache['z'].reshape(want_y.shape)) < tol
    assert report('conv y', y.cpu().numpy(), want_y) < tol
    dy = rs.randn(*want_y.shape) * 1e-2
    dx, dW, db = OC.conv2d_backward(dy, cache)
    dyd = to_dev(dy, torch.float32)
    gW = torch.full(W.shape, 3.0, device=dev())
    gb = torch.full((Co,), 3.0, device=dev())
    reuse = False
    if st == 1:
        gx = torch.full(x.shape, 3.0, device=dev())
        op.dgrad(dyd, z, Wd, gx)
        assert report('conv dx', gx.cpu().numpy(), dx) < 1e-5 * np.abs(dx).max() + 1e-9
        reuse = True
    else:
        from asr_study_amd._lib import AsrHipError
        with pytest.raises(AsrHipError):
            op.dgrad(dyd, z, Wd, torch.zeros(x.shape, device=dev()))
    op.wgrad(xd, dyd, z, gW, gb, reuse_x=True, reuse_dz=reuse)
    assert report('conv dW', gW.cpu().numpy(), dW) < 1e-5 * np.abs(dW).max() + 1e-9
    assert report('conv db', gb.cpu().numpy(), db) < 1e-5 * np.abs(db).max() + 1e-9
    # the same gradients without the planes the earlier calls left in the workspace
    op2 = ops.Conv2d(T, n_pad, F, Ci, Co, kt, kf, st, sf, clip, dev())
    gW2, gb2 = torch.zeros_like(gW), torch.zeros_like(gb)
    op2.wgrad(xd, dyd, z, gW2, gb2)
    assert torch.equal(gW2, gW) and torch.equal(gb2, gb)
    if clip > 0:
        # clipped ReLU fused into the GEMM epilogue (z = None): the same y, bit for bit, and the
        # same gradients with y standing in for z (0 < z < clip reads the same from y); a bound
        # on |x| instead of the measured maximum changes the planes' scale only
        op3 = ops.Conv2d(T, n_pad, F, Ci, Co, kt, kf, st, sf, clip, dev())
        y3 = torch.full(want_y.shape, 7.0, device=dev())
        op3.fwd(xd, Wd, bd, None, y3)
        assert torch.equal(y3, y)
        gW3, gb3 = torch.zeros_like(gW), torch.zeros_like(gb)
        if st == 1:
            gx3 = torch.full(x.shape, 3.0, device=dev())
            op3.dgrad(dyd, y3, Wd, gx3)
            assert torch.equal(gx3, gx)
        op3.wgrad(xd, dyd, y3, gW3, gb3, reuse_x=True, reuse_dz=st == 1)
        assert torch.equal(gW3, gW) and torch.equal(gb3, gb)
        bound = torch.full((1,), float(2.0 * np.abs(x).max() + 1.0), device=dev())
        y4 = torch.full(want_y.shape, 7.0, device=dev())
        ops.Conv2d(T, n_pad, F, Ci, Co, kt, kf, st, sf, clip, dev()).fwd(xd, Wd, bd, None, y4,
                                                                          x_absmax=bound)
        assert report('conv y under a bound on |x|', y4.cpu().numpy(), want_y) < tol


def _ds2(F=16, C=7, H=16, L=2, seed=1, **kw):
    from asr_study_amd.core import models, optimizers
    m = models.deep_speech2(num_features=F, num_classes=C, num_hiddens=H, num_layers=L,
                            conv_filters=4, conv_kernels=((5, 7), (3, 5)), dropout=0.0,
                            weight_decay=1e-4, seed=seed, **kw)
    m.compile(optimizer=optimizers.Adam(lr=1e-3, clipnorm=400))
    return m


def _ds2_oracle_params(model, L):
    w = [a.astype(np.float64) for a in model.get_weights()]
    it = iter(w)
    params = {'conv': []}
    for s in [s for s in model.stages if s.kind == 'conv']:
        params['conv'].append({'W': next(it), 'b': next(it), 'stride': (s.st, s.sf), 'clip': s.clip})
    params['layers'] = [{d: {'W': next(it), 'U': next(it), 'b': next(it)} for d in ('fwd', 'bwd')}
                        for _ in range(L)]
    params['dense'] = {'W': next(it), 'b': next(it)}
    return params


def test_deep_speech2_model_vs_oracle_and_lengths():
    """models.deep_speech2 (2 conv layers -> BiLSTM stack -> Dense -> CTC): logits, per-sample
    CTC loss on ceil(len / 2) output frames, every gradient tensor, and three Adam steps."""
    rs = np.random.RandomState(3)
    N, T, F, C, L = 5, 37, 16, 7, 2
    model = _ds2(F, C, 16, L)
    lens = np.array([37, 20, 37, 9, 30])
    x = rs.randn(N, T, F).astype(np.float32)
    for n in range(N):
        x[n, lens[n]:] = 0
    labels = [rs.randint(0, C - 1, size=k).tolist() for k in (3, 2, 4, 1, 2)]
    slab = model.to_slab(x)
    params = _ds2_oracle_params(model, L)
    x64 = slab[:, :N].cpu().numpy().astype(np.float64)
    want = OL.loss_and_grads(params, x64, labels, lens, weight_decay=0.0)
    ctc, logits, sl = model.loss_and_grads(slab, labels, lens, training=False)
    assert sl.cpu().tolist() == [19, 10, 19, 5, 15] and logits.shape[0] == 19
    assert report('ds2 logits', logits[:, :N].cpu().numpy(), want['logits']) < 1e-4
    assert report('ds2 ctc', ctc.cpu().numpy(), want['ctc']) < 1e-4 * np.abs(want['ctc']).max()
    for (name, g), gg in zip(OL.flatten(want['grads']), model.get_gradients()):
        assert report('ds2 ' + name, gg, g) < 1e-4 * max(np.abs(g).max(), 1e-3) + 1e-7, name
    from oracle import optim as OO
    opt = OO.Adam(lr=1e-3, clipnorm=400.0)
    flat = [a for _, a in OL.flatten(params)]
    for _ in range(3):
        out = OL.loss_and_grads(params, x64, labels, lens, weight_decay=1e-4)
        opt.step(flat, [a for _, a in OL.flatten(out['grads'])])
        m = model.train_on_batch([('slab', slab), labels, lens])
    assert abs(m[1] - float(np.mean(out['ctc']))) < 1e-4 * abs(m[1])
    for (name, a), b in zip(OL.flatten(params), model.get_weights()):
        assert report('ds2 w ' + name, b, a) < 5e-5, name
    hyp = model.predict(x, lens)
    assert len(hyp) == N and all(len(h) <= 19 for h in hyp)


def test_conv_rows_are_independent_and_the_op_is_linear_at_full_size():
    """cfg3 geometry, N = 64: the convolution of the batch equals the convolution of its 16-row
    slices BIT FOR BIT (an output row depends on its own sample only), and wgrad is additive
    over the slices (2e-5 of the maximum)."""
    from asr_study_amd import ops
    rs = np.random.RandomState(5)
    T, F, Co = 999, 80, 32
    x = torch.from_numpy(rs.randn(T, 64, F).astype(np.float32)).to(dev())
    W = torch.from_numpy((rs.randn(11, 41, 1, Co) * 0.05).astype(np.float32)).to(dev())
    b = torch.from_numpy((rs.randn(Co) * 0.1).astype(np.float32)).to(dev())
    op = ops.Conv2d(T, 64, F, 1, Co, 11, 41, 2, 2, 20.0, dev())
    z = torch.empty(op.T_out, 64, op.F_out * Co, device=dev())
    y = torch.empty_like(z)
    op.fwd(x, W, b, z, y)
    dy = torch.from_numpy((rs.randn(*y.shape) * 1e-3).astype(np.float32)).to(dev())
    gW, gb = torch.zeros_like(W), torch.zeros_like(b)
    op.wgrad(x, dy, z, gW, gb, reuse_x=True)
    sW, sb = torch.zeros_like(W), torch.zeros_like(b)
    op16 = ops.Conv2d(T, 16, F, 1, Co, 11, 41, 2, 2, 20.0, dev())
    for k in range(4):
        xs = x[:, 16 * k:16 * k + 16].contiguous()
        zs = torch.empty(op.T_out, 16, op.F_out * Co, device=dev())
        ys = torch.empty_like(zs)
        op16.fwd(xs, W, b, zs, ys)
        assert torch.equal(ys, y[:, 16 * k:16 * k + 16])
        tW, tb = torch.zeros_like(W), torch.zeros_like(b)
        op16.wgrad(xs, dy[:, 16 * k:16 * k + 16].contiguous(), zs, tW, tb, reuse_x=True)
        sW += tW
        sb += tb
    assert (sW - gW).abs().max().item() < 2e-5 * gW.abs().max().item()
    assert (sb - gb).abs().max().item() < 2e-5 * gb.abs().max().item()


def test_conv_layer2_and_dgrad_rows_are_independent_at_full_size():
    """The SECOND layer of cfg3's front-end (32 -> 32 channels, 11 x 21 / (1, 2) on the T' = 500 x
    40-frequency slab the first one leaves) at N = 64: forward AND dgrad of the batch equal those
    of its 16-row slices, wgrad is additive over the slices.  Not bit for bit as for layer 1:
    the reduction here is 11 x 21 x 32 deep and the GEMM's split-K factor is chosen from the
    number of output tiles, which differs between 32000 and 8000 rows -- another summation order
    of the same products; the bound is 2e-6 of the largest output (fp32 accumulation noise), three
    orders below anything a cross-row leak would produce.  This is what lets
    oracle/gen_golden_model.py build the full-size cfg3_conv fixture from four 16-utterance
    oracle slices."""
    from asr_study_amd import ops
    rs = np.random.RandomState(6)
    T, F, Ci, Co = 500, 40, 32, 32
    x = torch.from_numpy((rs.rand(T, 64, F * Ci) * 2.0).astype(np.float32)).to(dev())
    W = torch.from_numpy((rs.randn(11, 21, Ci, Co) * 0.02).astype(np.float32)).to(dev())
    b = torch.from_numpy((rs.randn(Co) * 0.1).astype(np.float32)).to(dev())
    op = ops.Conv2d(T, 64, F, Ci, Co, 11, 21, 1, 2, 20.0, dev())
    z = torch.empty(op.T_out, 64, op.F_out * Co, device=dev())
    y = torch.empty_like(z)
    op.fwd(x, W, b, z, y)
    dy = torch.from_numpy((rs.randn(*y.shape) * 1e-3).astype(np.float32)).to(dev())
    dx = torch.empty_like(x)
    op.dgrad(dy, z, W, dx)
    gW, gb = torch.zeros_like(W), torch.zeros_like(b)
    op.wgrad(x, dy, z, gW, gb, reuse_x=True, reuse_dz=True)
    sW, sb = torch.zeros_like(W), torch.zeros_like(b)
    op16 = ops.Conv2d(T, 16, F, Ci, Co, 11, 21, 1, 2, 20.0, dev())
    for k in range(4):
        sl = slice(16 * k, 16 * k + 16)
        xs = x[:, sl].contiguous()
        zs = torch.empty(op.T_out, 16, op.F_out * Co, device=dev())
        ys = torch.empty_like(zs)
        op16.fwd(xs, W, b, zs, ys)
        assert (ys - y[:, sl]).abs().max().item() <= 2e-6 * y.abs().max().item(), k
        dxs = torch.empty_like(xs)
        op16.dgrad(dy[:, sl].contiguous(), zs, W, dxs)
        assert (dxs - dx[:, sl]).abs().max().item() <= 2e-6 * dx.abs().max().item(), k
        # a leak between rows would show at the scale of the outputs: perturb ONE other row's
        # input and gradient -- this slice must not move at all
        if k == 0:
            x2, dy2 = x.clone(), dy.clone()
            x2[:, 37] *= 0.5                # (no new maximum: the operands' scales stay put)
            dy2[:, 37] *= -0.5
            z2, y2, dx2 = torch.empty_like(z), torch.empty_like(y), torch.empty_like(dx)
            op.fwd(x2, W, b, z2, y2)
            op.dgrad(dy2, z2, W, dx2)
            keep = [i for i in range(64) if i != 37]
            assert torch.equal(y2[:, keep], y[:, keep]) and torch.equal(dx2[:, keep], dx[:, keep])
            op.fwd(x, W, b, z, y)           # (restore the op's packed planes of x for wgrad below)
            op.dgrad(dy, z, W, dx)
        tW, tb = torch.zeros_like(W), torch.zeros_like(b)
        op16.wgrad(xs, dy[:, sl].contiguous(), zs, tW, tb, reuse_x=True, reuse_dz=True)
        sW += tW
        sb += tb
    assert (sW - gW).abs().max().item() < 2e-5 * gW.abs().max().item()
    assert (sb - gb).abs().max().item() < 2e-5 * gb.abs().max().item()


def test_conv_front_end_at_cfg3_geometry_vs_fixture():
    """Both layers at BASELINE configs[2]'s geometry (T = 999, 80 features, 32 x 11 x 41 / (2, 2)
    and 32 x 11 x 21 / (1, 2), clip 20) on a 16-utterance slice, against the float64 oracle's
    fixture (oracle/gen_golden_conv.py; sampled activations and gradients)."""
    from asr_study_amd import ops
    from oracle import gen_golden_conv as G
    fx = np.load(os.path.join(GOLDEN, 'conv_cfg3.npz'))
    x, params, dy = G.inputs()
    ops_, zs, acts = [], [], [to_dev(x, torch.float32)]
    Ci, Fi, Ti = 1, G.F, G.T
    for (W, b), (Co, kt, kf, st, sf) in zip(params, G.LAYERS):
        op = ops.Conv2d(Ti, G.N, Fi, Ci, Co, kt, kf, st, sf, G.CLIP, dev())
        z = torch.empty(op.T_out, G.N, op.F_out * Co, device=dev())
        y = torch.empty_like(z)
        op.fwd(acts[-1], to_dev(W, torch.float32), to_dev(b, torch.float32), z, y)
        ops_.append(op); zs.append(z); acts.append(y)
        Ci, Fi, Ti = Co, op.F_out, op.T_out
    g = to_dev(dy, torch.float32)
    for li in (1, 0):
        W, b = params[li]
        zl = zs[li].cpu().numpy().reshape(-1)
        # (pre-activations reach +-60 here: the bar is 1e-4 where |z| <= 10, 1e-5 relative above)
        ztol = max(1e-4, 1e-5 * float(np.abs(fx['z%d' % li]).max()))
        assert report('cfg3 conv z%d' % li, zl[fx['z%d_idx' % li]], fx['z%d' % li]) < ztol
        yl = acts[li + 1].cpu().numpy().reshape(-1)
        assert report('cfg3 conv y%d' % li, yl[fx['y%d_idx' % li]], fx['y%d' % li]) < ztol
        # (no gradient through the elements within 1e-3 of a ReLU threshold: gen_golden_conv.py)
        g = g.clone()
        g.view(-1)[torch.from_numpy(fx['near%d' % li]).to(dev())] = 0.0
        gW = torch.zeros(W.shape, device=dev())
        gb = torch.zeros(b.shape, device=dev())
        dz_ready = False
        if li > 0:
            gx = torch.empty_like(acts[li])
            ops_[li].dgrad(g, zs[li], to_dev(W, torch.float32), gx)
            dz_ready = True
            gxl = gx.cpu().numpy().reshape(-1)
            assert report('cfg3 conv dx%d' % li, gxl[fx['dx%d_idx' % li]], fx['dx%d' % li]) \
                < 1e-4 * float(fx['dx%d_max' % li])
        ops_[li].wgrad(acts[li], g, zs[li], gW, gb, reuse_x=True, reuse_dz=dz_ready)
        gWl = gW.cpu().numpy()
        wmax = float(fx['dW%d_max' % li])
        if 'dW%d_idx' % li in fx.files:
            got, want = gWl.reshape(-1)[fx['dW%d_idx' % li]], fx['dW%d' % li]
        else:
            got, want = gWl, fx['dW%d' % li]
        assert report('cfg3 conv dW%d' % li, got, want) < 1e-4 * wmax
        assert abs(np.sqrt((gWl.astype(np.float64) ** 2).sum()) - float(fx['dW%d_norm' % li])) \
            < 1e-3 * float(fx['dW%d_norm' % li])
        assert report('cfg3 conv db%d' % li, gb.cpu().numpy(), fx['db%d' % li]) \
            < 1e-4 * np.abs(fx['db%d' % li]).max()
        if li > 0:
            g = gx

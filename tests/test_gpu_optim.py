"""-m gpu: global-norm clip + Adam / SGD through the C ABI vs oracle/optim.py."""
import numpy as np
import pytest
import torch

from oracle import optim as OO
from tests.gpu_util import to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('clip', [0.0, 3.0, 1e6])
def test_adam_and_sgd(clip):
    from asr_study_amd import ops
    rs = np.random.RandomState(0)
    sizes = [1000, 37, 4096, 5]
    l2s = [1e-2, 0.0, 1e-3, 0.0]
    n = sum(sizes)
    p0 = rs.randn(n).astype(np.float32)
    offs = np.cumsum([0] + sizes[:-1])
    segs, nseg = ops.make_segments([(o, s, l) for o, s, l in zip(offs, sizes, l2s)], 'cuda:0')
    l2vec = np.concatenate([np.full(s, l, np.float32) for s, l in zip(sizes, l2s)])
    for kind in ('adam', 'sgd'):
        p_ref = [p0.astype(np.float64).copy()]
        opt = OO.Adam(lr=1e-2, clipnorm=clip) if kind == 'adam' else OO.SGD(lr=1e-2, momentum=0.9, clipnorm=clip)
        p = to_dev(p0.copy())
        m = torch.zeros_like(p); v = torch.zeros_like(p)
        norm = torch.zeros(2, dtype=torch.float64, device='cuda:0')
        for step in range(1, 4):
            g = rs.randn(n).astype(np.float32)
            g_tot = g.astype(np.float64) + 2 * l2vec * p_ref[0]
            want_norm = np.sqrt(np.sum(g_tot ** 2))
            want_pen = np.sum(l2vec * p_ref[0] ** 2)
            opt.step(p_ref, [g_tot])
            gd = to_dev(g)
            # the one-call form (asr_clip_*_step) on a copy must equal norm + step exactly
            p2, m2, v2, norm2 = p.clone(), m.clone(), v.clone(), torch.zeros_like(norm)
            ops.grad_norm(p, gd, segs, nseg, norm)
            if kind == 'adam':
                ops.adam_step(p, gd, m, v, segs, nseg, norm, clip, 1e-2, step)
                ops.clip_adam_step(p2, gd, m2, v2, segs, nseg, norm2, clip, 1e-2, step)
            else:
                ops.sgd_step(p, gd, m, segs, nseg, norm, clip, 1e-2, 0.9)
                ops.clip_sgd_step(p2, gd, m2, segs, nseg, norm2, clip, 1e-2, 0.9)
            torch.cuda.synchronize()
            assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(norm, norm2)
            nh = norm.cpu().numpy()
            assert abs(nh[0] - want_norm) < 1e-5 * want_norm
            assert abs(nh[1] - want_pen) < 1e-5 * max(1.0, want_pen)
            assert np.abs(p.cpu().numpy() - p_ref[0]).max() < 2e-5


def test_timeout_flag_vetoes_the_update_and_the_engine_falls_back_to_stepwise_kernels(monkeypatch):
    """A persistent recurrent kernel that abandons a bounded spin sets the sticky word of its
    workspace.  The update enqueued behind it is vetoed ON THE DEVICE (asr_optim_guard: the
    norm becomes -1, the update kernels return), the host sees the flag at its next check,
    clears it, switches to the stepwise kernels (mode 1), takes the vetoed step back from the
    optimiser's iteration count and runs the batch AGAIN -- no exception, no corrupted step,
    no garbage metrics.  The demotion is temporary: after ASR_LSTM_RETRY_STEPS clean steps the
    persistent kernels are tried again."""
    import torch
    from asr_study_amd import ops
    from asr_study_amd.core import models, optimizers
    monkeypatch.setenv('ASR_LSTM_RETRY_STEPS', '3')
    rs = np.random.RandomState(0)

    def fresh():
        m = models.brsmv1(num_features=9, num_classes=7, num_hiddens=16, num_layers=2,
                          dropout=0.0, weight_decay=1e-4, seed=1)
        m.compile(optimizer=optimizers.Adam(lr=1e-2, clipnorm=1.0))
        return m
    model, ref = fresh(), fresh()
    x = rs.randn(5, 30, 9).astype(np.float32)
    labels = [rs.randint(0, 6, size=4).tolist() for _ in range(5)]
    batch = [x, labels, [30] * 5]
    model.train_on_batch(batch)
    ref.train_on_batch(batch)
    dev = model.device
    ops.WS.get('lstm_bwd', 0, dev)[:4].view(torch.int32)[0] = 1      # what mark_timeout() does
    out = model.train_on_batch(batch)      # vetoed on the device, then run again stepwise
    want = ref.train_on_batch(batch)       # the same second step on the persistent kernels
    assert model.lstm_mode == 1 and model.fallbacks == 1 and model.vetoed_steps == 1
    assert model.optimizer.iterations == 2 == ref.optimizer.iterations and model._step == 2
    assert not ops.lstm_timeout_flags(dev).any().item()
    assert np.allclose(out, want, rtol=1e-5, atol=1e-6)
    assert (model.params - ref.params).abs().max().item() < 1e-6     # same arithmetic either mode
    # three clean steps later the persistent kernels are back; the next gap would be doubled
    for _ in range(3):
        assert model.lstm_mode == 1
        model.train_on_batch(batch)
        ref.train_on_batch(batch)
    model.train_on_batch(batch)
    ref.train_on_batch(batch)
    assert model.lstm_mode == 0 and model._retry_gap == 6 and model.fallbacks == 1
    assert (model.params - ref.params).abs().max().item() < 1e-6


def test_lagged_metrics_drop_vetoed_steps_and_take_their_iterations_back():
    """fit_generator's way of stepping (sync=False, metrics fetched one step later): the step
    that timed out AND the step already enqueued behind it are both vetoed on the device; both
    come back as None from the lagged check, both are taken back from the iteration count,
    the fallback is counted once."""
    import torch
    from asr_study_amd import ops
    from asr_study_amd.core import models, optimizers
    rs = np.random.RandomState(1)
    model = models.brsmv1(num_features=9, num_classes=7, num_hiddens=16, num_layers=1,
                          dropout=0.0, seed=1)
    model.compile(optimizer=optimizers.Adam(lr=1e-2, clipnorm=1.0))
    x = rs.randn(4, 20, 9).astype(np.float32)
    labels = [rs.randint(0, 6, size=3).tolist() for _ in range(4)]
    batch = [x, labels, [20] * 4]
    r0 = model.train_on_batch(batch, sync=False)
    before = model.params.clone()
    ops.WS.get('lstm_fwd', 0, model.device)[:4].view(torch.int32)[0] = 1
    r1 = model.train_on_batch(batch, sync=False)          # times out (flag forced)
    r2 = model.train_on_batch(batch, sync=False)          # enqueued before the host noticed
    assert model._lagged((r0, labels, 4)) is not None
    assert model._lagged((r1, labels, 4)) is None
    assert model.fallbacks == 1 and model.lstm_mode == 1
    assert model._lagged((r2, labels, 4)) is None
    assert model.fallbacks == 1 and model.vetoed_steps == 2
    assert torch.equal(model.params, before) and model.optimizer.iterations == 1
    r3 = model.train_on_batch(batch, sync=False)          # stepwise kernels: a real update
    assert model._lagged((r3, labels, 4)) is not None
    assert not torch.equal(model.params, before) and model.optimizer.iterations == 2


def test_lagged_veto_never_reuses_an_iteration_number():
    """The order fit_generator really produces: step k times out, step k+1 is enqueued, k is
    checked (fallback), step k+2 is ENQUEUED BEFORE k+1 is checked.  Both vetoed steps are
    taken back at the detection, so k+2 and k+3 are applied with consecutive Adam iterations
    and noise-stream steps (they used to share one)."""
    import torch
    from asr_study_amd import ops
    from asr_study_amd.core import models, optimizers
    rs = np.random.RandomState(2)
    model = models.brsmv1(num_features=9, num_classes=7, num_hiddens=16, num_layers=1,
                          dropout=0.0, seed=1)
    model.compile(optimizer=optimizers.Adam(lr=1e-2, clipnorm=1.0))
    x = rs.randn(4, 20, 9).astype(np.float32)
    labels = [rs.randint(0, 6, size=3).tolist() for _ in range(4)]
    batch = [x, labels, [20] * 4]
    r0 = model.train_on_batch(batch, sync=False)
    assert model._lagged((r0, labels, 4)) is not None
    ops.WS.get('lstm_bwd', 0, model.device)[:4].view(torch.int32)[0] = 1
    r1 = model.train_on_batch(batch, sync=False)          # k: vetoed
    r2 = model.train_on_batch(batch, sync=False)          # k+1: vetoed too (flag still set)
    assert model._lagged((r1, labels, 4)) is None         # detection: both taken back
    assert model.optimizer.iterations == 1 and model._step == 1 and model.vetoed_steps == 2
    r3 = model.train_on_batch(batch, sync=False)          # k+2, enqueued before k+1 is checked
    used = [(model.optimizer.iterations, model._step)]
    assert model._lagged((r2, labels, 4)) is None
    r4 = model.train_on_batch(batch, sync=False)
    used.append((model.optimizer.iterations, model._step))
    assert model._lagged((r3, labels, 4)) is not None and model._lagged((r4, labels, 4)) is not None
    assert used == [(2, 2), (3, 3)] and model.fallbacks == 1 and model.vetoed_steps == 2


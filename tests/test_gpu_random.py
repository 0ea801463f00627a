"""-m gpu: the library's counter-based random streams (csrc/random.hip, K12) against their
NumPy restatement (oracle/rng.py, pinned on the Random123 known-answer vectors): raw words
and keep masks bit for bit, normals to fp32 round-off; and the engine's training-time masks
are exactly those streams (seed, stage, step) -- no torch random op on the path."""
import numpy as np
import pytest
import torch

from oracle import rng as ORNG

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [1, 7, 4096, 2 * 64 * 1024 + 3])
def test_words_masks_and_normals_match_the_oracle(n):
    from asr_study_amd import ops
    dev = torch.device('cuda:0')
    seed, sid, step = (0x1234 << 32) | 0xBEEF, 11, 42
    w = ops.random_words(n, seed, sid, step, dev).cpu().numpy().view(np.uint32)
    assert np.array_equal(w, ORNG.words(n, seed, sid, step))
    m = torch.empty(n + (-n) % 4, device=dev)[:n]
    ops.dropout_masks(m, 0.2, 1.25, seed, sid, step)
    assert np.array_equal(m.cpu().numpy(), ORNG.keep_mask(n, 0.2, 1.25, seed, sid, step))
    x = torch.randn(n, device=dev)
    out, mk = torch.empty_like(x), torch.empty_like(x)
    ops.dropout_apply(x, out, mk, 0.35, 1.0 / 0.65, seed, sid + 1, step)
    want = ORNG.keep_mask(n, 0.35, 1.0 / 0.65, seed, sid + 1, step)
    assert np.array_equal(mk.cpu().numpy(), want)
    assert np.array_equal(out.cpu().numpy(), x.cpu().numpy() * want)
    z = torch.empty(n, device=dev)
    ops.gaussian_noise(x, z, 0.5, seed, sid + 2, step)
    zn = x.cpu().numpy().astype(np.float64) + 0.5 * ORNG.normal(n, seed, sid + 2, step)
    assert np.abs(z.cpu().numpy() - zn).max() < 2e-5
    g = torch.randn(n, device=dev)
    assert np.array_equal(ops.mul(g, mk).cpu().numpy(), g.cpu().numpy() * mk.cpu().numpy())


def test_engine_draws_its_masks_from_the_library_stream():
    from asr_study_amd.core import models
    model = models.brsmv1(num_features=9, num_classes=7, num_hiddens=8, num_layers=2,
                          dropout=0.25, zoneout=0.1, input_dropout=True, input_std_noise=0.3,
                          weight_decay=0.0, seed=5)
    n_pad = 16
    model._step = 3
    drawn = {k: (a.clone(), b.clone()) for k, (a, b) in model._draw_all_masks(n_pad).items()}
    assert len(drawn) == 2
    for si, (BW, BU) in drawn.items():
        s = model.stages[si]
        for k, t, p in ((0, BW, s.dropout_W), (1, BU, s.dropout_U)):
            want = ORNG.keep_mask(t.numel(), p, 1.0 / (1.0 - p), model.rng_seed, 4 * si + k,
                                  model._step)
            assert np.array_equal(t.cpu().numpy().reshape(-1), want), (si, k)
    # the same step draws the same masks again; the next step different ones
    again = model._draw_all_masks(n_pad)
    assert all(torch.equal(again[k][0], drawn[k][0]) and torch.equal(again[k][1], drawn[k][1])
               for k in drawn)
    model._step = 4
    other = model._draw_all_masks(n_pad)
    assert not any(torch.equal(other[k][0], drawn[k][0]) for k in drawn)
    # a whole training step runs (noise, input dropout, zoneout, variational masks)
    rs = np.random.RandomState(0)
    x = rs.randn(3, 20, 9).astype(np.float32)
    labels = [rs.randint(0, 6, size=4).tolist() for _ in range(3)]
    from asr_study_amd.core import optimizers
    model.compile(optimizer=optimizers.Adam(lr=1e-3, clipnorm=1.0))
    m = model.train_on_batch([x, labels, [20] * 3])
    assert np.isfinite(m[0])


def test_box_toolchain_rebuilds_two_sources_and_matches_the_shipped_library():
    """build() finds the shipped libasr_hip.so current on a GPU box and compiles nothing there;
    this test runs the box's OWN hipcc on capi.cpp + random.hip (seconds), loads the result next
    to the shipped library and compares their Philox word streams bit for bit, so a toolchain or
    runtime skew on the box is a red test."""
    import __graft_entry__ as G
    dt = G.toolchain_check()
    if dt is None:
        pytest.skip('no hipcc on this box')
    print('[toolchain] probe library built, loaded and matched in %.1f s' % dt)

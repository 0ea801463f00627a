"""-m gpu: properties of the hot path at BASELINE.json's FULL cfg2 size (brsmv1
5xBiLSTM(256), 32 utterances x 10 s = 999 frames), where the float64 oracle would need
minutes (and the same at cfg3: 5xBiLSTM(512), 64 utterances, 80 log-mel features):
size-independent invariants instead of element-wise comparison.

* utterances are independent: the logits / CTC losses of a 32-utterance batch equal, bit
  for bit, those of its two 16-utterance halves (no arithmetic crosses a batch row);
* gradients are additive over utterances: grad(batch of 32, scale 1/32) equals the sum of
  the two half-batch gradients computed with the same 1/32 scale, to fp32 round-off
  (the split-fp16 GEMM pre-scale depends on the batch maximum, so not bit-exact);
* one Adam step with global-norm clipping moves every weight by at most lr (|m|/sqrt(v)
  <= 1 at step 1) and leaves the persistent kernels' status words clean;
* the front-end computes an utterance identically alone and inside a batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T_SAMPLES = 160000


def _inputs(n, seed):
    rs = np.random.RandomState(seed)
    sigs = [rs.randn(T_SAMPLES).astype(np.float32) for _ in range(n)]
    labels = [rs.randint(0, 25, size=rs.randint(2, 50)).tolist() for _ in range(n)]
    return sigs, labels


CASES = {
    # BASELINE.json configs[1] / configs[2] at full size
    'cfg2': dict(F=39, H=256, N=32, feat='mfcc'),
    'cfg3': dict(F=80, H=512, N=64, feat='logfbank80'),
}


@pytest.mark.timeout(900)
@pytest.mark.parametrize('name', ['cfg2', 'cfg3'])
def test_fullsize_batch_rows_are_independent_and_gradients_additive(name):
    from asr_study_amd import ops
    from asr_study_amd.core import models, optimizers
    from asr_study_amd.preprocessing import audio
    cfg = CASES[name]
    N, half = cfg['N'], cfg['N'] // 2
    dev = torch.device('cuda:0')
    model = models.brsmv1(num_features=cfg['F'], num_classes=28, num_hiddens=cfg['H'],
                          num_layers=5, dropout=0.0, weight_decay=1e-4, seed=0, device=dev)
    model.compile(optimizer=optimizers.Adam(lr=1e-3, clipnorm=400))
    feat = audio.MFCC(device=dev) if cfg['feat'] == 'mfcc' else \
        audio.LogFbank(num_filt=80, device=dev)
    sigs, labels = _inputs(N, 5)
    slab, frames = feat.batch(sigs)
    assert slab.shape == (999, N, cfg['F']) and int(frames.min()) == 999
    lens = [999] * N

    def run(rows):
        sub = slab[:, rows].contiguous()
        ctc, logits, _ = model.loss_and_grads(sub, [labels[i] for i in rows],
                                              [lens[i] for i in rows], training=False,
                                              n_global=N)
        torch.cuda.synchronize()
        return (ctc.cpu().numpy().copy(), logits[:, :len(rows)].cpu().numpy().copy(),
                model.grads.cpu().numpy().copy())
    full_ctc, full_logits, full_grad = run(list(range(N)))
    a_ctc, a_logits, a_grad = run(list(range(half)))
    b_ctc, b_logits, b_grad = run(list(range(half, N)))
    for ws in ('lstm_fwd', 'lstm_bwd'):
        ops.lstm_status(ops.WS.get(ws, 0, dev))
    assert np.all(np.isfinite(full_ctc)) and np.all(full_ctc > 0)
    # forward: bit-exact independence of batch rows
    assert np.array_equal(full_logits[:, :half], a_logits)
    assert np.array_equal(full_logits[:, half:], b_logits)
    assert np.array_equal(full_ctc, np.concatenate([a_ctc, b_ctc]))
    # backward: additivity to fp32 round-off of a 32k-64k-term reduction
    gsum = a_grad + b_grad
    scale = np.abs(full_grad).max()
    assert scale > 0
    err = np.abs(full_grad - gsum).max()
    print('%s gradient additivity: max|g|=%.3e max err=%.3e' % (name, scale, err))
    assert err < 2e-5 * scale
    # one clipped Adam step: bounded update, finite weights
    before = model.params.clone()
    model._step += 1
    model.optimizer.step(model)
    torch.cuda.synchronize()
    delta = (model.params - before).abs().max().item()
    assert 0 < delta <= 1e-3 * 1.001
    assert torch.isfinite(model.params).all()


def test_frontend_alone_equals_in_batch():
    from asr_study_amd.preprocessing import audio
    dev = torch.device('cuda:0')
    for feat in (audio.MFCC(device=dev), audio.LogFbank(num_filt=80, device=dev)):
        sigs, _ = _inputs(5, 11)
        sigs[1] = sigs[1][:40000]
        sigs[3] = sigs[3][:401]
        slab, frames = feat.batch(sigs)
        for i, s in enumerate(sigs):
            alone, fr = feat.batch([s])
            t = int(fr[0])
            assert t == int(frames[i])
            assert torch.equal(alone[:t, 0], slab[:t, i])
            assert torch.count_nonzero(slab[t:, i]) == 0        # pad_sequences 'post'

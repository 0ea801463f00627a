"""__graft_entry__.smoke(): one tiny end-to-end step of the hot path on cuda:0,
checked against the CPU oracle (oracle/ is test infrastructure: it is imported here
only as the checker)."""
import numpy as np
import torch


def smoke():
    assert torch.cuda.is_available(), 'smoke() needs a GPU'
    from asr_study_amd.core import models, optimizers
    from asr_study_amd.preprocessing import audio
    from oracle import frontend as OF
    from oracle import lstm as OL
    rs = np.random.RandomState(0)
    # ---- front-end: two short utterances -> time-major slab
    sigs = [rs.randn(4000), rs.randn(2500)]
    feat = audio.MFCC()
    slab, frames = feat.batch(sigs)
    torch.cuda.synchronize()
    for i, s in enumerate(sigs):
        want = OF.extract('mfcc', s)
        t = int(frames[i])
        err = np.abs(slab[:t, i].cpu().numpy() - want).max()
        assert t == want.shape[0] and err < 5e-4, ('front-end', err)
    # ---- BiLSTM(16) x2 -> Dense -> CTC: forward, backward, one Adam step
    F, C, H, L = 39, 28, 16, 2
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L,
                          dropout=0.0, weight_decay=1e-4, seed=0)
    model.compile(optimizer=optimizers.Adam(lr=1e-3, clipnorm=400))
    w = [a.astype(np.float64) for a in model.get_weights()]
    it = iter(w)
    params = {'layers': [{d: {'W': next(it), 'U': next(it), 'b': next(it)}
                          for d in ('fwd', 'bwd')} for _ in range(L)]}
    params['dense'] = {'W': next(it), 'b': next(it)}
    labels = [[1, 2, 3, 3], [5]]
    lens = frames.cpu().numpy()
    x64 = slab[:, :2].cpu().numpy().astype(np.float64)
    want = OL.loss_and_grads(params, x64, labels, lens)
    ctc, logits, _ = model.loss_and_grads(slab, labels, lens, training=False)
    torch.cuda.synchronize()
    e1 = np.abs(logits[:, :2].cpu().numpy() - want['logits']).max()
    e2 = np.abs(ctc.cpu().numpy() - want['ctc']).max() / max(1.0, np.abs(want['ctc']).max())
    assert e1 < 1e-4 and e2 < 1e-4, ('model', e1, e2)
    for (name, g), gg in zip(OL.flatten(want['grads']), model.get_gradients()):
        err = np.abs(gg - g).max()
        assert err < 1e-4 * max(1e-3, np.abs(g).max()) + 1e-6, (name, err)
    m = model.train_on_batch([('slab', slab), labels, lens])
    assert np.isfinite(m[0]) and 0.0 <= m[3]
    print('smoke ok: front-end err %.1e, logits err %.1e, ctc rel err %.1e, loss %.3f'
          % (err, e1, e2, m[0]))

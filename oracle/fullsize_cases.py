"""Inputs of the FULL-SIZE parity cases (BASELINE.json configs[1] / configs[2] shapes,
T = 999 frames).  TEST INFRASTRUCTURE ONLY.

Shared by ``oracle/gen_golden_model.py`` (which runs the float64 oracle once, in the
build container, and commits compact fixtures under tests/golden/) and by
``tests/test_gpu_fullsize_parity.py`` (which rebuilds the SAME inputs on the GPU box and
compares the HIP path with those fixtures).  Everything here is a seeded recipe:

* audio: ``RandomState(1000 * rank + i).randn(160000)`` float32, rank 0 (bench.py's
  recipe, datasets/dummy.py:72), optionally truncated for a ragged batch;
* features: the pinned oracle front-end (oracle/frontend.py) on that audio, cast to
  float32 -- the model parity is isolated from the product front-end, which has its own
  golden tests;
* labels: bench.py's recipe, ``RandomState(77)``: length randint(2, 50), symbols
  randint(0, 25) (datasets/dummy.py:80-84);
* weights: ``oracle.lstm.init_model(seed=0)`` in float32 (Keras-1.2.2 init);
* masks (one case): variational-dropout masks B_W / B_U with p = 0.2 from
  ``RandomState(4242)`` (core/models.py:265-266 semantics: one mask per batch, shared over
  time, inverted scaling).
"""
import numpy as np

from . import frontend as OF
from . import lstm as OL

SAMPLES = 160000
T_FULL = 999

CASES = {
    # BASELINE.json configs[1] at full size, the bench's seeds, dropout 0
    'cfg2': dict(F=39, H=256, L=5, C=28, N=32, feat=('mfcc', {}), ragged=False, masks=False),
    # the same with fixed dropout masks (p = 0.2) and a ragged batch (5..10 s utterances)
    'cfg2_masks': dict(F=39, H=256, L=5, C=28, N=32, feat=('mfcc', {}), ragged=True,
                       masks=True),
    # BASELINE.json configs[2]'s topology, a 16-utterance slice of its batch
    'cfg3_n16': dict(F=80, H=512, L=5, C=28, N=16, feat=('logfbank', {'num_filt': 80}),
                     ragged=False, masks=False),
    # BASELINE.json configs[2] at FULL size: the bench's whole batch of 64 x 10 s.  The rows of
    # a batch are independent (no cross-sample op in the path), so the generator runs the
    # float64 oracle on four 16-utterance slices and adds their gradients (each scaled 1/64)
    'cfg3': dict(F=80, H=512, L=5, C=28, N=64, feat=('logfbank', {'num_filt': 80}),
                 ragged=False, masks=False, slices=4),
    # BASELINE.json configs[2] AS WRITTEN: the same stack behind its 2-conv front-end
    # (models.deep_speech2: 32 x (11 x 41) / (2, 2), 32 x (11 x 21) / (1, 2), clipped ReLU 20; no
    # reference counterpart, README.md:118), a 16-utterance slice; the stack sees T' = 500 frames
    'cfg3_conv_n16': dict(F=80, H=512, L=5, C=28, N=16, feat=('logfbank', {'num_filt': 80}),
                          ragged=False, masks=False,
                          conv=[(32, 11, 41, 2, 2, 20.0), (32, 11, 21, 1, 2, 20.0)]),
    # ... and at FULL size (64 x 10 s), generated like 'cfg3': four 16-utterance oracle slices,
    # gradients added (rows stay independent through the convolutions: 'same' padding over time
    # and frequency only, tests/test_gpu_conv.py checks it bit for bit on both layers and dgrad)
    'cfg3_conv': dict(F=80, H=512, L=5, C=28, N=64, feat=('logfbank', {'num_filt': 80}),
                      ragged=False, masks=False, slices=4,
                      conv=[(32, 11, 41, 2, 2, 20.0), (32, 11, 21, 1, 2, 20.0)]),
}
LOGIT_FRAMES = 50          # frames (spread over T) whose logits a fixture keeps
GRAD_SAMPLES = 1000        # sampled entries per gradient tensor
STATE_UTTS = 8             # utterances whose layer-1 / layer-L h, c samples are kept


def state_frames(T):
    return [0, 1, T // 2, T - 2, T - 1]


def logit_frames(T):
    return np.unique(np.round(np.linspace(0, T - 1, LOGIT_FRAMES)).astype(np.int64))


GRAD_BLOCK = 256           # gradient checksums: sums over this many contiguous flat entries


def grad_block_sums(flat):
    """Sums over GRAD_BLOCK contiguous entries of a flattened gradient tensor (tail block zero
    padded), float64: EVERY entry of every tensor enters one checksum, so a wrong stripe that
    1000 random probes in 27.6 M entries can miss moves a block sum."""
    flat = np.asarray(flat, np.float64).reshape(-1)
    nb = (flat.size + GRAD_BLOCK - 1) // GRAD_BLOCK
    pad = np.zeros(nb * GRAD_BLOCK, np.float64)
    pad[:flat.size] = flat
    return pad.reshape(nb, GRAD_BLOCK).sum(axis=1)


def grad_sample_index(i, size):
    """Flat indices of the sampled entries of gradient tensor number i."""
    rs = np.random.RandomState(9000 + i)
    if size <= GRAD_SAMPLES:
        return np.arange(size)
    return np.sort(rs.choice(size, GRAD_SAMPLES, replace=False))


def build(name):
    """-> dict(x (T, N, F) float32 time-major features, lens (N,), labels, params (float32
    tree, oracle layout), masks (oracle layout or None), cfg)."""
    cfg = CASES[name]
    N, F, H, L, C = cfg['N'], cfg['F'], cfg['H'], cfg['L'], cfg['C']
    rs_len = np.random.RandomState(31)
    sigs = []
    for i in range(N):
        s = np.random.RandomState(1000 * 0 + i).randn(SAMPLES).astype(np.float32)
        if cfg['ragged'] and i % 2 == 1:
            s = s[:int(rs_len.randint(SAMPLES // 2, SAMPLES))]
        sigs.append(s)
    kind, kw = cfg['feat']
    feats = [OF.extract(kind, s.astype(np.float64), **kw).astype(np.float32) for s in sigs]
    lens = np.array([f.shape[0] for f in feats], np.int64)
    T = int(lens.max())
    assert T == T_FULL and feats[0].shape[1] == F
    x = np.zeros((T, N, F), np.float32)
    for i, f in enumerate(feats):
        x[:f.shape[0], i] = f                      # pad_sequences 'post', value 0
    rs = np.random.RandomState(77)
    lab_len = rs.randint(2, 50, size=N)
    labels = [rs.randint(0, 25, size=lab_len[n]).tolist() for n in range(N)]
    params = OL.init_model(seed=0, num_features=F, num_hiddens=H, num_layers=L,
                           num_classes=C, dtype=np.float32, conv=cfg.get('conv'))
    masks = None
    if cfg['masks']:
        rm = np.random.RandomState(4242)
        masks = []
        n_in = F
        for _ in range(L):
            m = {}
            for d in ('fwd', 'bwd'):
                m[d] = (((rm.rand(N, n_in) >= 0.2) / 0.8).astype(np.float32),
                        ((rm.rand(N, H) >= 0.2) / 0.8).astype(np.float32))
            masks.append(m)
            n_in = 2 * H
    return dict(x=x, lens=lens, labels=labels, params=params, masks=masks, cfg=cfg, T=T)


def out_frames(cfg, T):
    """Frames (or lengths) on the logits' time axis: ceil(T / st) per conv layer."""
    T = np.asarray(T)
    for c in cfg.get('conv') or []:
        T = -(-T // c[3])
    return T


def feature_probe(x):
    """64 seeded sample positions of the feature slab (platform-drift guard)."""
    rs = np.random.RandomState(5)
    idx = rs.randint(0, x.size, size=64)
    return idx, x.reshape(-1)[idx]

"""Writes tests/golden/conv_cfg3.npz: the float64 oracle (oracle/conv.py) run ONCE, in the
build container, on the 2-D convolution front-end at BASELINE configs[2]'s geometry -- 80
log-mel features, T = 999 frames, 32 x (11 x 41) stride (2, 2) then 32 x (11 x 21) stride
(1, 2), clipped ReLU 20 -- for a 16-utterance slice of the batch (the convolution is
per-utterance; the weight gradients are sums over the slice).  Seeded inputs; the fixture keeps
compact samples: 4000 sampled entries of every activation / gradient slab, the full first-layer
filter gradient, 6000 sampled entries of the second one, both bias gradients.
tests/test_gpu_conv.py compares the HIP path with it at full size.  No reference counterpart
(README.md:118): the fixture pins the HIP path on the oracle, not the oracle on the reference.

    python oracle/gen_golden_conv.py          # ~5 minutes of NumPy
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import conv as OC          # noqa: E402

T, N, F = 999, 16, 80
LAYERS = [(32, 11, 41, 2, 2), (32, 11, 21, 1, 2)]
CLIP = 20.0


def inputs():
    """The seeded recipe the GPU test repeats: x ~ N(0, 1) standardised log-mel stand-in,
    filters glorot-uniform x 40 / x 3 (so that the clip at 0 AND at 20 both bite in both layers), dy ~ N(0, 1e-3)."""
    rs = np.random.RandomState(2024)
    x = rs.randn(T, N, F)
    params, Ci = [], 1
    for (Co, kt, kf, st, sf), gain in zip(LAYERS, (40.0, 3.0)):
        W, b = OC.init_conv(rs, kt, kf, Ci, Co)
        params.append((W * gain, rs.randn(Co) * 0.5))
        Ci = Co
    Fo = F
    for (_, _, _, _, sf) in LAYERS:
        Fo = -(-Fo // sf)
    dy = rs.randn(-(-T // 2), N, Fo * LAYERS[-1][0]) * 1e-3
    return x, params, dy


def sample(rs, a, n):
    idx = rs.choice(a.size, size=min(n, a.size), replace=False)
    return idx.astype(np.int64), a.reshape(-1)[idx]


def main():
    x, params, dy = inputs()
    acts, caches = [x], []
    for (W, b), (Co, kt, kf, st, sf) in zip(params, LAYERS):
        y, c = OC.conv2d_forward(acts[-1], W, b, (st, sf), CLIP)
        acts.append(y)
        caches.append(c)
        print('forward', y.shape, float(np.mean(y == 0)), float(np.mean(y == CLIP)))
    out = {}
    rs = np.random.RandomState(7)
    g = dy
    for li in (1, 0):
        # The clipped ReLU's derivative jumps at z = 0 and z = 20: an element whose
        # pre-activation lies within the arithmetic's error of a threshold may legitimately get
        # the other mask on the device.  The incoming gradient is therefore zeroed at the
        # (few hundred) elements within 1e-3 of a threshold -- their indices travel in the
        # fixture and the GPU test zeroes the same elements -- so both sides differentiate the
        # same piecewise-linear function.
        z = caches[li]['z'].reshape(acts[li + 1].shape)
        near = np.flatnonzero(np.minimum(np.abs(z), np.abs(z - CLIP)).reshape(-1) < 1e-3)
        out['near%d' % li] = near.astype(np.int64)
        g = g.copy()
        g.reshape(-1)[near] = 0.0
        dx, dW, db = OC.conv2d_backward(g, caches[li])
        out['z%d_idx' % li], out['z%d' % li] = sample(rs, z, 4000)
        out['y%d_idx' % li], out['y%d' % li] = sample(rs, acts[li + 1], 4000)
        out['db%d' % li] = db
        if dW.size <= 20000:
            out['dW%d' % li] = dW
        else:
            out['dW%d_idx' % li], out['dW%d' % li] = sample(rs, dW, 6000)
        out['dW%d_max' % li] = np.abs(dW).max()
        out['dW%d_norm' % li] = np.sqrt((dW ** 2).sum())
        if li > 0:
            out['dx%d_idx' % li], out['dx%d' % li] = sample(rs, dx, 4000)
            out['dx%d_max' % li] = np.abs(dx).max()
        g = dx
        print('backward layer', li, dW.shape, float(np.abs(dW).max()))
    path = os.path.join(ROOT, 'tests', 'golden', 'conv_cfg3.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path))


if __name__ == '__main__':
    main()

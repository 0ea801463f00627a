"""Oracle: the 2-D convolution front-end of BASELINE.json configs[2] ("DeepSpeech2-style ...
+ 2 conv front-end").  TEST INFRASTRUCTURE ONLY (never imported by the product).

NO REFERENCE COUNTERPART -- "parity unpinned": the reference lists Deep Speech 2 as TODO
(README.md:118); its ``deep_speech`` factory (core/models.py:148-214) is dead code and has no
convolution.  What is restated here is the semantics the layer is DEFINED with:

* Keras-1.2.2 ``Convolution2D(nb_filter, nb_row, nb_col, subsample=(st, sf),
  border_mode='same', dim_ordering='tf')`` on the (N, T, F, C) view of the features
  [recalled]: cross-correlation (no kernel flip), kernel (kt, kf, C_in, C_out), bias (C_out);
* 'same' = TensorFlow's rule [recalled]: out = ceil(in / stride), pad_total =
  max((out - 1) * stride + k - in, 0), pad_before = pad_total // 2 (the odd one goes behind);
* the activation is the reference's ``clipped_relu`` = ``relu(x, max_value=20)``
  (core/models.py:116-117): min(max(x, 0), max_value);
* like the LSTM stack (no Masking, core/models.py:20) the convolution runs over the PADDED
  batch; a sequence of ``len`` input frames has ceil(len / st) output frames.

Tensors are time-major (T, N, F * C) with the channel minor (feature f * C + c), i.e. Keras'
(N, T, F, C) transposed -- ``Reshape((T, F * C))`` is then the identity.  Cross-checked against
``torch.nn.functional.conv2d`` (float64) in tests/test_oracle_conv.py.
"""
import numpy as np


def same_pad(n_in, k, stride):
    """-> (n_out, pad_before, pad_after) of TensorFlow's SAME padding."""
    n_out = -(-n_in // stride)
    total = max((n_out - 1) * stride + k - n_in, 0)
    return n_out, total // 2, total - total // 2


def out_lengths(lens, st):
    """Frames a sequence of ``lens`` input frames has behind a layer of time stride st."""
    return -(-np.asarray(lens) // int(st))


def conv2d_forward(x, W, b, stride=(1, 1), clip=20.0):
    """x (T, N, F*Ci); W (kt, kf, Ci, Co); b (Co) -> (y, cache), y (T', N, F'*Co)."""
    kt, kf, Ci, Co = W.shape
    T, N, FC = x.shape
    F = FC // Ci
    st, sf = stride
    To, pt, pta = same_pad(T, kt, st)
    Fo, pf, pfa = same_pad(F, kf, sf)
    xp = np.zeros((T + pt + pta, N, F + pf + pfa, Ci), x.dtype)
    xp[pt:pt + T, :, pf:pf + F] = x.reshape(T, N, F, Ci)
    z = np.zeros((To, N, Fo, Co), x.dtype)
    for dt in range(kt):
        for df in range(kf):
            patch = xp[dt:dt + st * (To - 1) + 1:st, :, df:df + sf * (Fo - 1) + 1:sf]
            z += patch @ W[dt, df]
    z += b
    y = np.clip(z, 0.0, clip) if clip and clip > 0 else z
    cache = dict(xp=xp, W=W, z=z, stride=stride, clip=clip, shape=(T, N, F, Ci), pads=(pt, pf))
    return y.reshape(To, N, Fo * Co), cache


def conv2d_backward(dy, cache):
    """-> (dx (T, N, F*Ci), dW, db)."""
    W, z, xp = cache['W'], cache['z'], cache['xp']
    kt, kf, Ci, Co = W.shape
    st, sf = cache['stride']
    T, N, F, _ = cache['shape']
    pt, pf = cache['pads']
    To, _, Fo, _ = z.shape
    dz = dy.reshape(z.shape).copy()
    clip = cache['clip']
    if clip and clip > 0:
        dz *= ((z > 0.0) & (z < clip))
    dW = np.zeros_like(W)
    dxp = np.zeros_like(xp)
    for dt in range(kt):
        for df in range(kf):
            sl = (slice(dt, dt + st * (To - 1) + 1, st), slice(None),
                  slice(df, df + sf * (Fo - 1) + 1, sf))
            dW[dt, df] = np.tensordot(xp[sl], dz, axes=([0, 1, 2], [0, 1, 2]))
            dxp[sl] += dz @ W[dt, df].T
    dx = dxp[pt:pt + T, :, pf:pf + F].reshape(T, N, F * Ci)
    return dx, dW, dz.sum(axis=(0, 1, 2))


def init_conv(rs, kt, kf, Ci, Co, dtype=np.float64):
    """Keras-1.2.2 Convolution2D default init='glorot_uniform' over the receptive field
    [recalled]: fan_in = kt*kf*Ci, fan_out = kt*kf*Co; bias zeros."""
    lim = np.sqrt(6.0 / (kt * kf * Ci + kt * kf * Co))
    return rs.uniform(-lim, lim, size=(kt, kf, Ci, Co)).astype(dtype), np.zeros(Co, dtype)

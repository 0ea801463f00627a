"""oracle/conv.py (the build-defined 2-D convolution front-end; no reference counterpart)
against torch.nn.functional.conv2d + autograd in float64 (a test-only cross-check), for
TensorFlow's SAME padding with odd / even sizes and both strides."""
import numpy as np
import pytest
import torch

from oracle import conv as OC


def _torch_ref(x, W, b, stride, clip):
    T, N, FC = x.shape
    kt, kf, Ci, Co = W.shape
    F = FC // Ci
    st, sf = stride
    To, pt, pta = OC.same_pad(T, kt, st)
    Fo, pf, pfa = OC.same_pad(F, kf, sf)
    xt = torch.tensor(x.reshape(T, N, F, Ci), dtype=torch.float64, requires_grad=True)
    Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    img = xt.permute(1, 3, 0, 2)                                    # (N, Ci, T, F)
    img = torch.nn.functional.pad(img, (pf, pfa, pt, pta))
    z = torch.nn.functional.conv2d(img, Wt.permute(3, 2, 0, 1), bt, stride=(st, sf))
    y = torch.clamp(z, 0.0, clip) if clip else z
    y = y.permute(2, 0, 3, 1).reshape(To, N, Fo * Co)               # (T', N, F'*Co)
    return xt, Wt, bt, y


@pytest.mark.parametrize('T,F,Ci,Co,kt,kf,st,sf,clip', [
    (13, 10, 1, 3, 5, 7, 2, 2, 20.0),       # odd T, stride 2 in both
    (12, 9, 2, 4, 3, 5, 1, 2, 0.5),         # even T, stride (1, 2), a clip that bites
    (7, 8, 3, 2, 11, 3, 2, 1, 0.0),         # filter longer than the slab; linear output
    (999 // 37, 80 // 4, 1, 2, 11, 41 // 4, 2, 2, 20.0),
])
def test_conv2d_forward_backward_vs_torch(T, F, Ci, Co, kt, kf, st, sf, clip):
    rs = np.random.RandomState(T * 7 + F)
    N = 3
    x = rs.randn(T, N, F * Ci)
    W = rs.randn(kt, kf, Ci, Co) * 0.5
    b = rs.randn(Co) * 0.1
    y, cache = OC.conv2d_forward(x, W, b, (st, sf), clip)
    xt, Wt, bt, yt = _torch_ref(x, W, b, (st, sf), clip)
    assert y.shape == tuple(yt.shape)
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-11)
    dy = rs.randn(*y.shape)
    (yt * torch.tensor(dy)).sum().backward()
    dx, dW, db = OC.conv2d_backward(dy, cache)
    np.testing.assert_allclose(dx, xt.grad.numpy().reshape(dx.shape), atol=1e-11)
    np.testing.assert_allclose(dW, Wt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(db, bt.grad.numpy(), atol=1e-10)


def test_same_padding_rule_and_output_lengths():
    assert OC.same_pad(999, 11, 2) == (500, 5, 5)
    assert OC.same_pad(1000, 11, 2) == (500, 4, 5)          # the odd padding frame goes behind
    assert OC.same_pad(80, 41, 2) == (40, 19, 20)
    assert OC.same_pad(40, 21, 2) == (20, 9, 10)
    assert OC.same_pad(500, 11, 1) == (500, 5, 5)
    assert OC.out_lengths([999, 1000, 1, 2], 2).tolist() == [500, 500, 1, 1]

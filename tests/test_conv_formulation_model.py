"""The GEMM formulation of csrc/conv.hip (tests/conv_formulation_model.py: banded matrix per
tap, phase layout, segment row shifts, padded dz planes, per-tap K-major weight gradients)
equals the convolution of oracle/conv.py -- on the CPU, in float64, for the strides and
padding parities of the two front-end layers."""
import numpy as np
import pytest

from oracle import conv as OC
from tests import conv_formulation_model as CM


@pytest.mark.parametrize('T,F,Ci,Co,kt,kf,st,sf', [
    (13, 12, 1, 4, 5, 7, 2, 2),          # layer-1 shape class: C_in = 1, stride (2, 2), odd T
    (14, 12, 1, 4, 5, 7, 2, 2),          # even T: the odd padding frame goes behind
    (9, 8, 4, 4, 5, 5, 1, 2),            # layer-2 shape class: stride (1, 2)
    (6, 8, 2, 2, 11, 3, 1, 1),           # filter longer than the slab
    (11, 8, 2, 2, 4, 4, 3, 2),           # even filters, time stride 3
])
def test_gemm_formulation_equals_the_convolution(T, F, Ci, Co, kt, kf, st, sf):
    rs = np.random.RandomState(T + 10 * F)
    n_pad, clip = 16, 1.0
    x = rs.randn(T, n_pad, F * Ci)
    W = rs.randn(kt, kf, Ci, Co) * 0.4
    b = rs.randn(Co) * 0.1
    g = CM.Geo(T, n_pad, F, Ci, Co, kt, kf, st, sf)
    y, z, xp = CM.forward(g, x, W, b, clip)
    want, cache = OC.conv2d_forward(x, W, b, (st, sf), clip)
    assert (g.T_out, g.F_out) == (want.shape[0], want.shape[2] // Co)
    np.testing.assert_allclose(y, want, atol=1e-12)
    dy = rs.randn(*y.shape)
    dx, dW, db = OC.conv2d_backward(dy, cache)
    gW, gb = CM.wgrad(g, xp, dy, z, clip)
    np.testing.assert_allclose(gW, dW, atol=1e-11)
    np.testing.assert_allclose(gb, db, atol=1e-11)
    if st == 1:
        np.testing.assert_allclose(CM.dgrad(g, dy, z, W, clip), dx, atol=1e-11)


@pytest.mark.parametrize('T,F,Ci,Co,kt,kf,st,sf,ncol,nrow', [
    (7, 24, 32, 48, 3, 9, 1, 2, 1, 1),       # C_out divides neither 256 nor 32: the whole band
    (7, 24, 32, 64, 3, 9, 1, 2, 3, 3),       # three column blocks of 4 output frequencies
    (6, 32, 32, 32, 3, 5, 2, 1, 4, 4),       # four column blocks, four row blocks
    (5, 40, 32, 32, 3, 21, 1, 2, 3, 5),      # the second front-end layer's frequency geometry
    (9, 8, 4, 4, 5, 5, 1, 2, 1, 1),          # nothing to cut: one block = the whole band
    (6, 41, 32, 32, 3, 5, 1, 2, 3, 6),       # ragged last blocks (5 output / 1 input frequency)
])
def test_frequency_blocks_cover_the_band(T, F, Ci, Co, kt, kf, st, sf, ncol, nrow):
    """make_geo's column blocks (forward, dgrad) and row blocks (weight gradient): GEMMs on column
    ranges of the same planes against block bands add up to the whole convolution."""
    rs = np.random.RandomState(T + 10 * F)
    n_pad, clip = 16, 1.0
    x = rs.randn(T, n_pad, F * Ci)
    W = rs.randn(kt, kf, Ci, Co) * 0.1
    b = rs.randn(Co) * 0.1
    g = CM.Geo(T, n_pad, F, Ci, Co, kt, kf, st, sf)
    assert (len(CM.col_blocks(g)), len(CM.row_blocks(g))) == (ncol, nrow)
    y, z, xp = CM.forward_blocks(g, x, W, b, clip)
    want, cache = OC.conv2d_forward(x, W, b, (st, sf), clip)
    np.testing.assert_allclose(y, want, atol=1e-11)
    dy = rs.randn(*y.shape)
    dx, dW, db = OC.conv2d_backward(dy, cache)
    gW, gb = CM.wgrad_blocks(g, xp, dy, z, clip)
    np.testing.assert_allclose(gW, dW, atol=1e-10)
    np.testing.assert_allclose(gb, db, atol=1e-10)
    if st == 1:
        np.testing.assert_allclose(CM.dgrad_blocks(g, dy, z, W, clip), dx, atol=1e-10)

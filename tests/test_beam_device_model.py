"""CPU check of the device beam search's loop restructuring (csrc/beam.hip) through its
sequential model tests/beam_device_model.py: same label sequences and scores as the oracle
(oracle/decode.py, double arithmetic) and as the library's host decoder (decode_host.cpp), on
random and on heavily tied inputs, over widths that force evictions every frame."""
import numpy as np
import pytest

from oracle import decode as OD
from beam_device_model import beam_device_model


def _cases(seed, count):
    rs = np.random.RandomState(seed)
    for it in range(count):
        C = rs.randint(2, 7)
        T = rs.randint(1, 25)
        W = rs.randint(1, 12)
        x = rs.randn(T, C).astype(np.float32) * (3 if it % 3 else 1)
        if it % 3 == 1:
            x = np.round(x)                   # many exact ties
        if it % 3 == 2:
            x = np.round(x * 2) / 2
        yield x, W


def test_model_equals_oracle_and_host_decoder_with_ties_and_evictions():
    from asr_study_amd import ops
    for x, W in _cases(0, 150):
        T = x.shape[0]
        for mr in (True, False):
            want, ws = OD.beam_search_decode_one(x, W, merge_repeated=mr, dtype=np.float64)
            got, gs = beam_device_model(x, W, mr)
            assert got == want[0], (W, mr)
            assert abs(gs - ws[0]) <= 1e-9 * max(1.0, abs(gs))
            hyps, _ = ops.ctc_beam_search_host(np.repeat(x[:, None, :], 16, 1), [T], 1, W, mr)
            assert hyps[0] == got


@pytest.mark.parametrize('T,C,W,scale', [(30, 28, 100, 1.0), (24, 28, 100, 6.0), (12, 28, 400, 1.0),
                                         (30, 28, 64, 0.02)])
def test_model_equals_host_decoder_at_reference_widths(T, C, W, scale):
    from asr_study_amd import ops
    rs = np.random.RandomState(T + W)
    x = (rs.randn(T, C) * scale).astype(np.float32)
    if scale > 5:
        x[:, C - 1] += 5.0                    # blank-dominated frames, as a trained model emits
    st = {}
    got, gs = beam_device_model(x, W, True, st)
    hyps, sc = ops.ctc_beam_search_host(np.repeat(x[:, None, :], 16, 1), [T], 1, W, True)
    assert hyps[0] == got
    assert abs(gs - sc[0]) <= 1e-5 * max(1.0, abs(gs))
    assert st['turns'] > 0 and st['inserts'] > 0

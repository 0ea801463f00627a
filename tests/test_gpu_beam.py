"""K9 on the device (asr_ctc_beam_device, csrc/beam.hip) against the library's host decoder
(decode_host.cpp, itself pinned to oracle/decode.py by test_capi_host.py / test_gpu_cli.py) and
against the oracle directly: identical label sequences, scores to 1e-6, at the README's width
(100), the code's default (400, utils/core_utils.py:70-71) and the kernel's extremes; ragged and
empty utterances, tied logits, merge_repeated on and off, the full 999-frame length."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(x, lens, W, merge=True):
    """x (T, N, C) float32 numpy -> device and host results."""
    from asr_study_amd import ops
    T, N, C = x.shape
    n_pad = ops.pad16(N)
    slab = np.zeros((T, n_pad, C), np.float32)
    slab[:, :N] = x
    logits = torch.from_numpy(slab).cuda()
    sl = torch.tensor(np.asarray(lens, np.int32)).cuda()
    dec, dlen, score = ops.ctc_beam_search(logits, sl, N, W, merge)
    torch.cuda.synchronize()
    dec, dlen, score = dec.cpu().numpy(), dlen.cpu().numpy(), score.cpu().numpy()
    got = [dec[n, :dlen[n]].tolist() for n in range(N)]
    for n in range(N):
        assert (dec[n, dlen[n]:] == -1).all()
    want, wscore = ops.ctc_beam_search_host(slab, np.asarray(lens, np.int32), N, W, merge)
    return got, score, want, wscore


@pytest.mark.parametrize('W', [1, 3, 64, 100, 128, 129, 400, 448, 449, 1024])
@pytest.mark.parametrize('merge', [True, False])
def test_device_beam_equals_host_decoder(W, merge):
    rs = np.random.RandomState(W)
    T, N, C = 60, 9, 28
    x = (rs.randn(T, N, C) * rs.choice([0.05, 1.0, 4.0], size=(1, N, 1))).astype(np.float32)
    x[:, 3:6, C - 1] += 3.0
    lens = [T, 0, 1, 17, T, 33, 2, T, 45]
    got, score, want, wscore = _run(x, lens, W, merge)
    assert got == want
    np.testing.assert_allclose(score, wscore, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('C,W', [(2, 4), (3, 5), (5, 7), (6, 40), (64, 100)])
def test_device_beam_on_tied_logits_equals_oracle(C, W):
    """Rounded logits: exact ties between siblings, between a candidate and the beam's bottom and
    between evicted branches -- the order-dependent corners of the TF algorithm."""
    from oracle import decode as OD
    rs = np.random.RandomState(C * 100 + W)
    T, N = 22, 16
    x = np.round(rs.randn(T, N, C) * 2).astype(np.float32) / 2
    x[:, :4] = 0.0                                   # all-equal frames
    lens = [T] * N
    for merge in (True, False):
        got, score, want, wscore = _run(x, lens, W, merge)
        assert got == want
        for n in range(0, N, 3):
            o, osc = OD.beam_search_decode_one(x[:, n], W, merge_repeated=merge, dtype=np.float64)
            assert got[n] == o[0]
            assert abs(score[n] - osc[0]) <= 1e-5 * max(1.0, abs(osc[0]))


def test_device_beam_full_length_width_400_and_timing():
    """T = 999 (10 s utterances), width 400, 28 classes: the eval.py configuration."""
    import time
    from asr_study_amd import ops
    rs = np.random.RandomState(5)
    T, N, C = 999, 4, 28
    x = (rs.randn(T, N, C) * 2).astype(np.float32)
    x[:, :, C - 1] += 2.0
    x[:, 1] *= 0.1
    lens = [T, T, 640, T]
    t0 = time.time()
    got, score, want, wscore = _run(x, lens, 400)
    assert got == want
    np.testing.assert_allclose(score, wscore, rtol=1e-6, atol=1e-5)
    # device time alone, 64 utterances at once
    xb = np.tile(x, (1, 16, 1))
    logits = torch.from_numpy(xb).cuda()
    sl = torch.tensor([T] * 64, dtype=torch.int32).cuda()
    for W in (100, 400):
        ops.ctc_beam_search(logits, sl, 64, W)
        torch.cuda.synchronize()
        t0 = time.time()
        dec, dlen, _ = ops.ctc_beam_search(logits, sl, 64, W)
        torch.cuda.synchronize()
        dt = time.time() - t0
        t1 = time.time()
        ops.ctc_beam_search_host(xb, [T] * 64, 64, W)
        th = time.time() - t1
        print('[beam] width %d, 64 x 999 frames: device %.3f s, host decoder %.3f s (%d threads)'
              % (W, dt, th, min(64, os.cpu_count() or 1)))
        print('[beam]   utterance 0:', ops.ctc_beam_counters(logits.shape, 64, W, 0, logits.device))
        d = dec.cpu().numpy()
        assert (d[:4] == d[4:8]).all()


def test_eval_mode_model_decodes_on_the_device(monkeypatch):
    """engine.Model with a beam decoder (load_model(mode='eval') semantics) under
    ASR_BEAM=device: predict and test_on_batch never copy the logits to the host
    (ops.ctc_beam_search_host is not called)."""
    from asr_study_amd import ops
    monkeypatch.setenv('ASR_BEAM', 'device')
    from asr_study_amd.core import models
    rs = np.random.RandomState(0)
    N, T, F, C = 5, 40, 12, 9
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=16, num_layers=2,
                          dropout=0.0, seed=1, is_greedy=False, beam_width=400)
    x = rs.randn(N, T, F).astype(np.float32)
    lens = np.array([T, 31, T, 8, 25])
    slab = model.to_slab(x)
    logits = model.forward(slab, training=False, need_grad=False, n_valid=N).cpu().numpy()
    want, _ = ops.ctc_beam_search_host(logits, lens, N, 400, True)

    def boom(*a, **k):
        raise AssertionError('host decoder called')
    monkeypatch.setattr(ops, 'ctc_beam_search_host', boom)
    got = model.predict(x, lens)
    assert got == want
    labels = [rs.randint(0, C - 1, size=3).tolist() for _ in range(N)]
    out = model.test_on_batch([('slab', slab), labels, lens])
    assert np.isfinite(out).all()


def test_ctc_utils_decode_beam_branch_runs_on_the_device(monkeypatch):
    """core/ctc_utils.decode(is_greedy=False) (the reference's Lambda body, ctc_utils.py:8-52):
    the device decoder under ASR_BEAM=device, the host one under ASR_BEAM=host -- same label
    lists, and the device branch never touches the host decoder; the default (auto) picks by
    batch size against the host's thread count (ops.beam_decoder_choice)."""
    from asr_study_amd import ops
    from asr_study_amd.core import ctc_utils
    rs = np.random.RandomState(5)
    T, N, C = 80, 5, 28
    n_pad = ops.pad16(N)
    slab = torch.from_numpy((rs.randn(T, n_pad, C) * 2).astype(np.float32)).cuda()
    lens = np.array([80, 61, 80, 7, 33], np.int32)
    monkeypatch.setenv('ASR_BEAM', 'host')
    host = ctc_utils.decode((slab, lens), is_greedy=False, beam_width=100)
    monkeypatch.delenv('ASR_BEAM')
    auto = ctc_utils.decode((slab, lens), is_greedy=False, beam_width=100)
    monkeypatch.setenv('ASR_BEAM', 'device')

    def boom(*a, **k):
        raise AssertionError('host decoder called')
    monkeypatch.setattr(ops, 'ctc_beam_search_host', boom)
    dev = ctc_utils.decode((slab, lens), is_greedy=False, beam_width=100)
    assert dev == host == auto and any(len(h) for h in host)
    monkeypatch.delenv('ASR_BEAM')
    # auto: the host decoder up to min(usable host threads, 64) utterances (r6: its time grows
    # faster than the batch beyond that, the device decoder's is flat up to one wave per CU:
    # profiles/r6_beam_crossover.md), the device decoder beyond
    cap = min(ops.usable_host_threads(), ops.HOST_BEAM_MAX_UTTERANCES)
    assert ops.beam_decoder_choice(min(5, cap), 100, C) == 'host'
    assert ops.beam_decoder_choice(cap, 100, C) == 'host'
    assert ops.beam_decoder_choice(cap + 1, 100, C) == 'device'
    assert ops.beam_decoder_choice(256, 100, C) == 'device'
    assert ops.beam_decoder_choice(cap + 1, 2048, C) == 'host'          # beyond the kernel's width

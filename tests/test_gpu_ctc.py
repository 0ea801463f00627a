"""-m gpu: CTC loss/grad and greedy decode through the C ABI vs the float64 oracle.

Tolerances (BASELINE.json north_star: "CTC loss ... within 1e-4 fp32"): loss rtol
1e-4, gradient atol 1e-4 (entries are in [-1, 1]) at every size including T=999 --
the kernel re-centres its log-space rows (float64 offsets), so it does not inherit
the ~1e-3 posterior noise of a plain float32 log-space recursion
(tests/test_oracle_ctc.py::test_float32_close_to_float64); decode indices exact."""
import numpy as np
import pytest
import torch

from oracle import ctc as OC
from oracle import decode as OD
from tests.gpu_util import dev, to_dev, pad_batch, report
from tests.test_oracle_ctc import TF_PROBS_0, TF_PROBS_1

pytestmark = pytest.mark.gpu


def _run(logits, labels, seq_len, want_grad=True, scale=1.0):
    from asr_study_amd import ops
    T, N, C = logits.shape
    n_pad = ops.pad16(N)
    lmax = max([len(l) for l in labels] + [1])
    lab = np.zeros((N, lmax), np.int32)
    for n, l in enumerate(labels):
        lab[n, :len(l)] = l
    lg = to_dev(pad_batch(logits.astype(np.float32), n_pad))
    grad = torch.full_like(lg, 7.0) if want_grad else None
    loss = ops.ctc_loss_grad(lg, to_dev(lab), to_dev(np.array([len(l) for l in labels], np.int32)),
                             to_dev(np.asarray(seq_len, np.int32)), N, grad=grad,
                             grad_scale=scale)
    torch.cuda.synchronize()
    return loss.cpu().numpy(), (grad.cpu().numpy() if want_grad else None), lg


def test_tf_known_answers():
    logits = np.log(np.stack([TF_PROBS_0, TF_PROBS_1], axis=1))
    loss, grad, _ = _run(logits, [[0, 1, 2, 1, 0], [0, 1, 1, 0]], [5, 5])
    assert abs(loss[0] - 3.34211) < 1e-4 and abs(loss[1] - 5.42262) < 1e-4
    _, g64 = OC.ctc_loss_grad(logits, [[0, 1, 2, 1, 0], [0, 1, 1, 0]], [5, 5])
    assert report('ctc tf-vectors grad', grad[:, :2], g64) < 1e-5
    assert np.all(grad[:, 2:] == 0)            # batch padding rows


@pytest.mark.parametrize('seed', [0, 1])
def test_small_ragged(seed):
    rs = np.random.RandomState(seed)
    T, N, C = 41, 7, 11
    logits = rs.randn(T, N, C) * 2
    labels = [rs.randint(0, C - 1, size=rs.randint(1, 9)).tolist() for _ in range(N)]
    labels[1] = [3, 3, 3, 2, 2]
    labels[2] = []
    labels[3] = [5]
    seq_len = [T, 20, 5, T, 11, 30, 1]
    labels[6] = [4]
    l64, g64 = OC.ctc_loss_grad(logits.astype(np.float32), labels, seq_len)
    loss, grad, _ = _run(logits, labels, seq_len, scale=0.5)
    np.testing.assert_allclose(loss, l64, rtol=1e-4)
    assert report('ctc small grad', grad[:, :N], 0.5 * g64) < 1e-4
    for n in range(N):
        assert np.all(grad[seq_len[n]:, n] == 0)
    loss_only, _, _ = _run(logits, labels, seq_len, want_grad=False)
    np.testing.assert_allclose(loss_only, l64, rtol=1e-4)


def test_long_labels_multi_pair_per_lane():
    rs = np.random.RandomState(3)
    T, N, C = 450, 3, 28
    logits = rs.randn(T, N, C)
    labels = [rs.randint(0, 25, size=L).tolist() for L in (200, 70, 1)]
    l64, g64 = OC.ctc_loss_grad(logits.astype(np.float32), labels, [T] * N)
    loss, grad, _ = _run(logits, labels, [T] * N)
    np.testing.assert_allclose(loss, l64, rtol=1e-4)
    assert report('ctc long-label grad', grad[:, :N], g64) < 1e-4


def test_infeasible_gives_inf_loss_zero_grad():
    logits = np.zeros((3, 1, 4))
    loss, grad, _ = _run(logits, [[1, 1, 1]], [3])
    assert np.isinf(loss[0]) and loss[0] > 0 and np.all(grad == 0)


def test_full_size_cfg3_slab():
    """T=999, N=64, C=28 (BASELINE cfg3 CTC slab), labels 2..49 symbols a-y."""
    rs = np.random.RandomState(11)
    T, N, C = 999, 64, 28
    logits = rs.randn(T, N, C).astype(np.float32)
    labels = [rs.randint(0, 25, size=rs.randint(2, 50)).tolist() for _ in range(N)]
    seq_len = [T] * N
    seq_len[5] = 700
    l64, g64 = OC.ctc_loss_grad(logits, labels, seq_len)
    loss, grad, _ = _run(logits, labels, seq_len)
    np.testing.assert_allclose(loss, l64, rtol=1e-4)
    assert report('ctc cfg3 grad', grad, g64) < 1e-4
    # size-independent properties: rows sum to 0 inside, exactly 0 outside
    assert np.abs(grad[:700].sum(-1)).max() < 1e-3
    assert np.all(grad[700:, 5] == 0)


def test_greedy_exact():
    from asr_study_amd import ops
    rs = np.random.RandomState(2)
    T, N, C = 600, 9, 28
    logits = rs.randn(T, N, C).astype(np.float32)
    logits[:, :, C - 1] += 1.5                       # plenty of blanks
    logits[10:20, 0, 3] += 9.0                       # a long repeat
    logits[0, 1, :] = 0.0                            # exact tie -> first index
    seq_len = np.array([T, 599, 257, 256, 255, 1, 2, 300, T], np.int32)
    n_pad = ops.pad16(N)
    lg = to_dev(pad_batch(logits, n_pad))
    dec, dlen = ops.ctc_greedy(lg, to_dev(seq_len), N)
    torch.cuda.synchronize()
    dec, dlen = dec.cpu().numpy(), dlen.cpu().numpy()
    want = OD.greedy_decode(logits, seq_len)
    for n in range(N):
        assert dec[n, :dlen[n]].tolist() == want[n], n
        assert np.all(dec[n, dlen[n]:] == -1)

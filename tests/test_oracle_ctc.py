"""Pin oracle/ctc.py: TensorFlow's published known-answer vectors (SURVEY.md
8c-5), torch.nn.functional.ctc_loss (loss + autograd gradient), brute force."""
import numpy as np
import pytest
import torch

from oracle import ctc

TF_PROBS_0 = np.array(
    [[0.633766, 0.221185, 0.0917319, 0.0129757, 0.0142857, 0.0260553],
     [0.111121, 0.588392, 0.278779, 0.0055756, 0.00569609, 0.010436],
     [0.0357786, 0.633813, 0.321418, 0.00249248, 0.00272882, 0.0037688],
     [0.0663296, 0.643849, 0.280111, 0.00283995, 0.0035545, 0.00331533],
     [0.458235, 0.396634, 0.123377, 0.00648837, 0.00903441, 0.00623107]])
TF_PROBS_1 = np.array(
    [[0.30176, 0.28562, 0.0831517, 0.0862751, 0.0816851, 0.161508],
     [0.24082, 0.397533, 0.0557226, 0.0546814, 0.0557528, 0.19549],
     [0.230246, 0.450868, 0.0389607, 0.038309, 0.0391602, 0.202456],
     [0.280884, 0.429522, 0.0326593, 0.0339046, 0.0326856, 0.190345],
     [0.423286, 0.315517, 0.0338439, 0.0393744, 0.0339315, 0.154046]])


def test_tf_known_answer_vectors():
    logits = np.log(np.stack([TF_PROBS_0, TF_PROBS_1], axis=1))   # (5,2,6)
    loss, grad = ctc.ctc_loss_grad(logits, [[0, 1, 2, 1, 0], [0, 1, 1, 0]], [5, 5])
    assert abs(loss[0] - 3.34211) < 2e-5
    assert abs(loss[1] - 5.42262) < 2e-5
    # gradient rows sum to ~0 (softmax minus posterior, both sum to 1)
    assert np.allclose(grad.sum(-1), 0.0, atol=1e-9)


def _torch_ref(logits, labels, seq_len):
    T, N, C = logits.shape
    x = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(x, -1)
    tl = torch.tensor([len(l) for l in labels])
    tgt = torch.tensor(np.concatenate(labels) if sum(len(l) for l in labels)
                       else np.zeros(0, np.int64), dtype=torch.long)
    loss = torch.nn.functional.ctc_loss(lp, tgt, torch.tensor(seq_len), tl,
                                        blank=C - 1, reduction='none')
    loss.sum().backward()
    return loss.detach().numpy(), x.grad.numpy()


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_matches_torch_ctc(seed):
    rs = np.random.RandomState(seed)
    T, N, C = 37, 6, 9
    logits = rs.randn(T, N, C) * 2.0
    labels = [rs.randint(0, C - 1, size=rs.randint(1, 9)) for _ in range(N)]
    labels[1] = np.array([3, 3, 3, 2, 2])          # repeats
    labels[2] = np.array([], dtype=np.int64)       # empty label
    seq_len = [T, 20, 5, T, 11, 30]
    loss, grad = ctc.ctc_loss_grad(logits, labels, seq_len)
    rl, rg = _torch_ref(logits, labels, seq_len)
    np.testing.assert_allclose(loss, rl, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(grad, rg, rtol=0, atol=1e-10)
    for n in range(N):                              # zero beyond seq_len
        assert np.all(grad[seq_len[n]:, n] == 0)


def test_bruteforce_tiny():
    rs = np.random.RandomState(5)
    logits = rs.randn(5, 1, 4)
    for label in ([0], [1, 1], [0, 2, 1], []):
        loss, _ = ctc.ctc_loss_grad(logits, [label], [5])
        assert abs(loss[0] - ctc.ctc_loss_bruteforce(logits[:, 0], label)) < 1e-10


def test_infeasible_raises():
    logits = np.zeros((3, 1, 4))
    with pytest.raises(ValueError):
        ctc.ctc_loss_grad(logits, [[1, 1, 1]], [3])   # needs 5 frames


def test_float32_close_to_float64():
    rs = np.random.RandomState(7)
    T, N, C = 200, 4, 28
    logits = rs.randn(T, N, C).astype(np.float32)
    labels = [rs.randint(0, 25, size=rs.randint(2, 50)) for _ in range(N)]
    l64, g64 = ctc.ctc_loss_grad(logits, labels, [T] * N)
    l32, g32 = ctc.ctc_loss_grad(logits, labels, [T] * N, dtype=np.float32)
    # float32 log-space alpha/beta carry |value| ~ T*ln(C) ~ 660 here, one ulp is
    # 6e-5, so posteriors are only good to ~1e-3 relative (TF's own float32
    # kernel has the same property) -- this is why GPU parity is judged against
    # the float64 oracle with a stated tolerance.
    np.testing.assert_allclose(l32, l64, rtol=1e-5)
    np.testing.assert_allclose(g32, g64, atol=2e-3)

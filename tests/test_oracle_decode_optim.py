import numpy as np
import torch

from oracle import decode as D
from oracle import optim as O


def test_greedy_first_max_merge_blank():
    C = 4   # blank = 3
    path = [0, 0, 3, 0, 1, 1, 3, 3, 2]
    logits = np.full((len(path) + 2, 1, C), -1.0)
    for t, k in enumerate(path):
        logits[t, 0, k] = 1.0
    logits[len(path):, 0, 1] = 5.0          # beyond seq_len: ignored
    assert D.greedy_decode(logits, [len(path)]) == [[0, 0, 1, 2]]
    tie = np.zeros((1, 1, C))               # all equal -> first index (0)
    assert D.greedy_decode(tie, [1]) == [[0]]


def test_beam_matches_bruteforce_when_wide():
    rs = np.random.RandomState(0)
    for trial in range(6):
        T, C = 5, 4
        logits = rs.randn(T, C) * 2
        scores = D.beam_search_bruteforce(logits)
        best = max(scores, key=scores.get)
        paths, sc = D.beam_search_decode_one(logits, beam_width=500,
                                             merge_repeated=False, dtype=np.float64)
        assert tuple(paths[0]) == best
        assert abs(sc[0] - scores[best]) < 1e-9


def test_beam_merge_repeated_collapses_output():
    C = 3
    # strongly favour a, blank, a  -> labelling [0,0]; merge_repeated=True emits [0]
    logits = np.full((3, C), -5.0)
    logits[0, 0] = logits[1, 2] = logits[2, 0] = 5.0
    p_plain, _ = D.beam_search_decode_one(logits, 10, merge_repeated=False)
    p_merge, _ = D.beam_search_decode_one(logits, 10, merge_repeated=True)
    assert p_plain[0] == [0, 0] and p_merge[0] == [0]


def test_beam_width_one_is_not_greedy_but_valid():
    rs = np.random.RandomState(3)
    logits = rs.randn(20, 1, 6)
    out = D.beam_search_decode(logits, [20], beam_width=1)
    assert all(0 <= k < 5 for k in out[0])


def test_edit_distance_and_ler():
    assert D.edit_distance([1, 2, 3], [1, 3]) == 1
    assert D.edit_distance([], [1, 2]) == 2
    assert D.edit_distance('kitten', 'sitting') == 3
    assert D.normalized_edit_distance([], []) == 0.0
    assert D.normalized_edit_distance([1], []) == float('inf')
    assert abs(D.ler([[1, 2], [3]], [[1, 2], [4, 4]]) - 0.5) < 1e-12


def test_adam_matches_torch_and_clipnorm():
    rs = np.random.RandomState(0)
    shapes = [(3, 4), (5,), (2, 2)]
    p_np = [rs.randn(*s) for s in shapes]
    p_t = [torch.tensor(p.copy(), requires_grad=True) for p in p_np]
    opt_t = torch.optim.Adam(p_t, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    opt = O.Adam(lr=1e-3, clipnorm=0.0)
    for step in range(5):
        g = [rs.randn(*s) for s in shapes]
        for pt, gg in zip(p_t, g):
            pt.grad = torch.tensor(gg)
        opt_t.step()
        opt.step(p_np, g)
    # torch's eps placement differs (sqrt(v_hat)+eps vs sqrt(v)+eps scaled) -> loose
    for a, b in zip(p_np, p_t):
        np.testing.assert_allclose(a, b.detach().numpy(), atol=1e-6)
    g = [np.ones(s) * 10 for s in shapes]
    clipped, n = O.clip_by_global_norm(g, 5.0)
    assert abs(O.global_norm(clipped) - 5.0) < 1e-9 and n > 5
    same, _ = O.clip_by_global_norm(g, 1e9)
    assert all(np.array_equal(a, b) for a, b in zip(same, g))

"""Oracle: CTC loss + gradient with tf.nn.ctc_loss (TF 1.3.0) semantics.
TEST INFRASTRUCTURE ONLY.

Reference call site: core/ctc_utils.py:60-70 (``ctc_lambda_func``) ->
``tf.nn.ctc_loss(labels, transpose(y_pred,[1,0,2]), inputs_length[:,0])`` with the
TF-1.3 defaults ``preprocess_collapse_repeated=False, ctc_merge_repeated=True,
time_major=True``.  The algorithm lives in tensorflow==1.3.0 (msc.yaml:100), which
is NOT under /root/reference: PARITY UNPINNED by the reference.  Restated from the
published algorithm (Graves 2006, eq. 6-16; TF core/util/ctc/ctc_loss_calculator):

* blank index = C-1; softmax applied internally to the logits;
* extended label l' = [b, l1, b, l2, ..., b] (S = 2L+1);
* log-space alpha/beta; beta excludes the emission at t, so
  alpha_t(s)+beta_t(s) = log P(paths through s at t);
* loss_n = -log(alpha_{Tn-1}(S-1) + alpha_{Tn-1}(S-2));
* d loss / d logit[t,n,c] = softmax[t,n,c] - sum_{s: l'_s=c} exp(alpha_t(s)+
  beta_t(s) - logZ) for t < seq_len[n], and 0 for t >= seq_len[n];
* an infeasible target (seq_len < L + #adjacent repeats) is an error
  ("Not enough time for target transition sequence").

Pinned on TensorFlow's two published known-answer vectors (SURVEY.md 8c-5,
tests/test_oracle_ctc.py) and on torch.nn.functional.ctc_loss.
"""
import numpy as np

NEG_INF = -np.inf


def log_softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    y = x - m
    return y - np.log(np.sum(np.exp(y), axis=axis, keepdims=True))


def _lse2(a, b):
    m = np.maximum(a, b)
    ms = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide='ignore'):
        return ms + np.log(np.exp(a - ms) + np.exp(b - ms))


def _lse3(a, b, c):
    m = np.maximum(np.maximum(a, b), c)
    ms = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide='ignore'):
        return ms + np.log(np.exp(a - ms) + np.exp(b - ms) + np.exp(c - ms))


def min_time(label):
    """Frames needed by a label: L + number of adjacent repeats."""
    label = np.asarray(label)
    return len(label) + int(np.sum(label[1:] == label[:-1])) if len(label) else 0


def ctc_loss_grad(logits, labels, seq_len, blank=None, dtype=np.float64,
                  return_alpha_beta=False):
    """logits (T,N,C) time-major; labels: list of N int arrays; seq_len (N,).

    Returns (loss (N,), grad (T,N,C)) in ``dtype``.
    """
    logits = np.asarray(logits, dtype=dtype)
    T, N, C = logits.shape
    blank = C - 1 if blank is None else blank
    seq_len = np.asarray(seq_len, dtype=np.int64).reshape(N)
    labels = [np.asarray(l, dtype=np.int64).reshape(-1) for l in labels]
    assert len(labels) == N
    for n in range(N):
        if seq_len[n] > T or seq_len[n] < 1:
            raise ValueError('seq_len[%d]=%d out of range' % (n, seq_len[n]))
        if min_time(labels[n]) > seq_len[n]:
            raise ValueError('Not enough time for target transition sequence '
                             '(sample %d)' % n)
    Lmax = max([len(l) for l in labels] + [0])
    S = 2 * Lmax + 1
    ext = np.full((N, S), blank, dtype=np.int64)
    Sn = np.zeros(N, dtype=np.int64)
    for n, l in enumerate(labels):
        ext[n, 1:2 * len(l):2] = l
        Sn[n] = 2 * len(l) + 1
    sidx = np.arange(S)[None, :]
    valid = sidx < Sn[:, None]                                  # (N,S)
    skip = np.zeros((N, S), dtype=bool)                         # s-2 -> s allowed
    skip[:, 2:] = (ext[:, 2:] != blank) & (ext[:, 2:] != ext[:, :-2])
    skip &= valid

    logp = log_softmax(logits, axis=-1)                         # (T,N,C)
    nidx = np.arange(N)[:, None]
    lp_ext = logp[:, nidx, ext]                                 # (T,N,S)

    alpha = np.full((T, N, S), NEG_INF, dtype=dtype)
    alpha[0, :, 0] = lp_ext[0, :, 0]
    if S > 1:
        alpha[0, :, 1] = np.where(Sn > 1, lp_ext[0, :, 1], NEG_INF)
    for t in range(1, T):
        a = alpha[t - 1]
        a1 = np.full_like(a, NEG_INF); a1[:, 1:] = a[:, :-1]
        a2 = np.full_like(a, NEG_INF); a2[:, 2:] = a[:, :-2]
        a2 = np.where(skip, a2, NEG_INF)
        new = _lse3(a, a1, a2) + lp_ext[t]
        new = np.where(valid, new, NEG_INF)
        act = (t < seq_len)[:, None]
        alpha[t] = np.where(act, new, NEG_INF)

    last = seq_len - 1
    nn = np.arange(N)
    aT = alpha[last, nn]                                        # (N,S)
    end1 = aT[nn, Sn - 1]
    end2 = np.where(Sn > 1, aT[nn, np.maximum(Sn - 2, 0)], NEG_INF)
    logZ = _lse2(end1, end2)
    loss = -logZ
    if not np.all(np.isfinite(loss)):
        raise ValueError('No valid path found (infinite CTC loss)')

    # beta: excludes emission at t
    beta = np.full((T, N, S), NEG_INF, dtype=dtype)
    skip_fw = np.zeros((N, S), dtype=bool)                      # s -> s+2 allowed
    skip_fw[:, :-2] = skip[:, 2:]
    for t in range(T - 1, -1, -1):
        is_last = (t == last)[:, None]
        init = np.where((sidx == Sn[:, None] - 1) | (sidx == Sn[:, None] - 2),
                        0.0, NEG_INF)
        init = np.where(valid, init, NEG_INF)
        if t + 1 < T:
            b = beta[t + 1] + lp_ext[t + 1]
            b1 = np.full_like(b, NEG_INF); b1[:, :-1] = b[:, 1:]
            b2 = np.full_like(b, NEG_INF); b2[:, :-2] = b[:, 2:]
            b2 = np.where(skip_fw, b2, NEG_INF)
            rec = _lse3(b, b1, b2)
            rec = np.where(valid, rec, NEG_INF)
        else:
            rec = np.full((N, S), NEG_INF, dtype=dtype)
        act = (t < last)[:, None]
        beta[t] = np.where(is_last, init, np.where(act, rec, NEG_INF))

    ab = alpha + beta                                           # (T,N,S)
    grad = np.zeros((T, N, C), dtype=dtype)
    with np.errstate(divide='ignore', over='ignore', invalid='ignore'):
        post = np.exp(ab - logZ[None, :, None])                 # (T,N,S)
    post = np.where(np.isfinite(post), post, 0.0)
    tt = np.arange(T)[:, None, None]
    np.add.at(grad, (tt, nidx[None], ext[None]), post)
    grad = np.exp(logp) - grad
    tmask = (np.arange(T)[:, None] < seq_len[None, :])[:, :, None]
    grad = np.where(tmask, grad, 0.0).astype(dtype)
    if return_alpha_beta:
        return loss.astype(dtype), grad, alpha, beta
    return loss.astype(dtype), grad


def ctc_loss_bruteforce(logits, label, blank=None):
    """Exhaustive path enumeration for tiny (T, C): -log sum_{paths->label}."""
    import itertools
    logits = np.asarray(logits, dtype=np.float64)
    T, C = logits.shape
    blank = C - 1 if blank is None else blank
    p = np.exp(log_softmax(logits))
    total = 0.0
    label = list(label)
    for path in itertools.product(range(C), repeat=T):
        col = [k for k, g in itertools.groupby(path)]
        col = [k for k in col if k != blank]
        if col == label:
            total += np.prod(p[np.arange(T), list(path)])
    return -np.log(total)

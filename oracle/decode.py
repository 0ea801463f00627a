"""Oracle: CTC greedy / prefix-beam decode and label error rate.
TEST INFRASTRUCTURE ONLY.

Reference call sites: core/ctc_utils.py:8-52 (``decode`` ->
tf.nn.ctc_greedy_decoder / tf.nn.ctc_beam_search_decoder on *logits*, blank =
C-1, merge_repeated=True) and core/metrics.py:4-8 (``ler`` = mean of
tf.edit_distance(hyp, truth, normalize=True)).  The algorithms live in
tensorflow==1.3.0 (msc.yaml:100), not under /root/reference: PARITY UNPINNED by
the reference.  Restated from the published algorithms:

* greedy: per-frame argmax with first-index tie-break over t < seq_len, merge
  consecutive repeats, drop blanks;
* beam: TF's CTCBeamSearchDecoder (prefix tree; separate blank / label log-probs
  per prefix; scores on max-subtracted logits, NOT log-softmax; a candidate enters
  a full beam only if strictly better than the current bottom); with
  merge_repeated=True the emitted top path additionally has consecutive identical
  labels collapsed (BeamEntry::LabelSeq);
* edit distance: Levenshtein(hyp, truth) / len(truth); inf if truth is empty and
  hyp is not, 0 if both are empty.
"""
import numpy as np

LOG_ZERO = -np.inf


def greedy_decode(logits, seq_len, blank=None, merge_repeated=True):
    """logits (T,N,C) time-major -> list of N int lists."""
    logits = np.asarray(logits)
    T, N, C = logits.shape
    blank = C - 1 if blank is None else blank
    out = []
    for n in range(N):
        best = np.argmax(logits[:int(seq_len[n]), n, :], axis=-1)  # first max
        seq, prev = [], -1
        for k in best:
            if k != blank and not (merge_repeated and k == prev):
                seq.append(int(k))
            prev = k
        out.append(seq)
    return out


def _lse(a, b):
    if a == LOG_ZERO:
        return b
    if b == LOG_ZERO:
        return a
    m = max(a, b)
    return m + np.log(np.exp(a - m) + np.exp(b - m))


class _Entry(object):
    __slots__ = ('parent', 'label', 'children', 'ob', 'ol', 'ot', 'nb', 'nl',
                 'nt', 'order')

    def __init__(self, parent, label, order):
        self.parent, self.label, self.children = parent, label, None
        self.ob = self.ol = self.ot = LOG_ZERO
        self.nb = self.nl = self.nt = LOG_ZERO
        self.order = order

    def active(self):
        return self.nt != LOG_ZERO

    def label_seq(self, merge_repeated):
        labels, prev, c = [], -1, self
        while c.parent is not None:
            if not merge_repeated or c.label != prev:
                labels.append(c.label)
            prev = c.label
            c = c.parent
        return labels[::-1]


def beam_search_decode_one(logits, beam_width=100, blank=None,
                           merge_repeated=True, top_paths=1, dtype=np.float32):
    """logits (T,C) for one utterance (already cut to seq_len).

    Returns (list of label lists, list of log-scores), best first.
    """
    x = np.asarray(logits, dtype=dtype)
    T, C = x.shape
    blank = C - 1 if blank is None else blank
    counter = [0]

    def new_entry(parent, label):
        counter[0] += 1
        return _Entry(parent, label, counter[0])

    root = new_entry(None, -1)
    root.nt, root.nb, root.nl = dtype(0.0), dtype(0.0), LOG_ZERO
    leaves = [root]

    def bottom(lv):
        return min(lv, key=lambda e: (e.nt, -e.order))

    for t in range(T):
        inp = x[t] - np.max(x[t])
        branches = sorted(leaves, key=lambda e: (-e.nt, e.order))
        leaves = []
        for b in branches:
            b.ob, b.ol, b.ot = b.nb, b.nl, b.nt
        for b in branches:
            if b.parent is not None:
                if b.parent.active():
                    prev = b.parent.ob if b.label == b.parent.label else b.parent.ot
                    b.nl = dtype(_lse(b.nl, prev))
                b.nl = dtype(b.nl + inp[b.label])
            b.nb = dtype(b.ot + inp[blank])
            b.nt = dtype(_lse(b.nb, b.nl))
            leaves.append(b)

        def is_candidate(total):
            return total > LOG_ZERO and (len(leaves) < beam_width or
                                         total > bottom(leaves).nt)

        for b in branches:
            if not is_candidate(b.ot):
                continue
            if b.children is None:
                b.children = [new_entry(b, c) for c in range(C) if c != blank]
            for c in b.children:
                if c.active():
                    continue
                c.nb = LOG_ZERO
                prev = b.ob if c.label == b.label else b.ot
                c.nl = dtype(inp[c.label] + prev) if prev != LOG_ZERO else LOG_ZERO
                c.nt = c.nl
                if is_candidate(c.nt):
                    if len(leaves) == beam_width:
                        bt = bottom(leaves)
                        leaves.remove(bt)
                        bt.nb = bt.nl = bt.nt = LOG_ZERO
                    leaves.append(c)
                else:
                    c.ob = c.ol = c.ot = LOG_ZERO
                    c.nb = c.nl = c.nt = LOG_ZERO
    ranked = sorted(leaves, key=lambda e: (-e.nt, e.order))[:top_paths]
    return ([e.label_seq(merge_repeated) for e in ranked],
            [float(e.nt) for e in ranked])


def beam_search_decode(logits, seq_len, beam_width=100, blank=None,
                       merge_repeated=True):
    """logits (T,N,C) -> list of N top-1 label lists."""
    logits = np.asarray(logits)
    out = []
    for n in range(logits.shape[1]):
        paths, _ = beam_search_decode_one(logits[:int(seq_len[n]), n, :],
                                          beam_width, blank, merge_repeated)
        out.append(paths[0])
    return out


def beam_search_bruteforce(logits, blank=None):
    """Exact most-probable labelling by enumerating all paths (tiny T,C).

    Scores use the same max-subtracted logits as the beam decoder so that the
    returned log-score is comparable; returns dict labelling -> log-score.
    """
    import itertools
    x = np.asarray(logits, dtype=np.float64)
    T, C = x.shape
    blank = C - 1 if blank is None else blank
    inp = x - x.max(axis=1, keepdims=True)
    scores = {}
    for path in itertools.product(range(C), repeat=T):
        col = tuple(k for k, g in itertools.groupby(path) if k != blank)
        s = float(np.sum(inp[np.arange(T), list(path)]))
        scores[col] = _lse(scores.get(col, LOG_ZERO), s)
    return scores


def edit_distance(hyp, truth):
    """Levenshtein distance (insert/delete/substitute, unit costs)."""
    hyp, truth = list(hyp), list(truth)
    prev = list(range(len(truth) + 1))
    for i in range(1, len(hyp) + 1):
        cur = [i] + [0] * len(truth)
        for j in range(1, len(truth) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1,
                         prev[j - 1] + (hyp[i - 1] != truth[j - 1]))
        prev = cur
    return prev[len(truth)]


def normalized_edit_distance(hyp, truth):
    """tf.edit_distance(normalize=True) per sample."""
    d = edit_distance(hyp, truth)
    if len(truth) == 0:
        return 0.0 if d == 0 else float('inf')
    return d / float(len(truth))


def ler(hyps, truths):
    """core/metrics.py:4-8: batch mean of the normalised edit distance."""
    return float(np.mean([normalized_edit_distance(h, t)
                          for h, t in zip(hyps, truths)]))

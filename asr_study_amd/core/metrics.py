"""core/metrics.py: label error rate."""
import numpy as np

from .. import ops


def ler(y_true, y_pred, **kwargs):
    """Mean over the batch of Levenshtein(pred, true) / len(true)
    (tf.edit_distance(normalize=True), core/metrics.py:4-8).  Inputs: lists of
    label sequences."""
    return float(np.mean(ops.edit_distance_host([list(p) for p in y_pred],
                                                [list(t) for t in y_true])))

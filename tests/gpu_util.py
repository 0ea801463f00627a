"""Helpers shared by the -m gpu parity tests (layout conversions, device upload)."""
import numpy as np
import torch


def dev():
    return torch.device('cuda:0')


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev())


def gate_major_to_unit_major(a, H):
    """Keras consume_less='gpu' column order (gate*H + unit) -> kernel order
    (unit*4 + gate) on the last axis."""
    sh = a.shape[:-1]
    return np.ascontiguousarray(
        a.reshape(sh + (4, H)).swapaxes(-1, -2).reshape(sh + (4 * H,)))


def unit_major_to_gate_major(a, H):
    sh = a.shape[:-1]
    return np.ascontiguousarray(
        a.reshape(sh + (H, 4)).swapaxes(-1, -2).reshape(sh + (4 * H,)))


def pad_batch(a, n_pad, axis=1):
    """Zero-pad the batch axis of a time-major array to n_pad rows."""
    pad = [(0, 0)] * a.ndim
    pad[axis] = (0, n_pad - a.shape[axis])
    return np.pad(a, pad)


def report(name, got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    err = np.abs(got - want)
    i = np.unravel_index(np.argmax(err), err.shape) if err.size else ()
    print('[parity] %-28s max|err|=%.3e at %s (got %.6g want %.6g) max|want|=%.3e'
          % (name, err.max() if err.size else 0.0, i,
             got[i] if err.size else 0, want[i] if err.size else 0,
             np.abs(want).max() if want.size else 0))
    return err.max() if err.size else 0.0

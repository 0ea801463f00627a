"""Oracle: MFCC / log-mel front-end (float64 NumPy).  TEST INFRASTRUCTURE ONLY.

Restates preprocessing/audio.py and preprocessing/audio_utils.py of the
reference.  PINNED: ``oracle/gen_golden.py`` imports the reference's own code in
the build container and checks every function here bit-for-bit in float64; the
resulting vectors live in tests/golden/frontend_*.npz.
"""
import decimal
import math

import numpy as np
from scipy.fftpack import dct as _dct
from scipy.signal.windows import hamming as _hamming


def round_half_up(number):
    """audio_utils.py:11-14 -- decimal ROUND_HALF_UP."""
    return int(decimal.Decimal(number).quantize(decimal.Decimal('1'),
                                                rounding=decimal.ROUND_HALF_UP))


def num_frames(slen, frame_len=400, frame_step=160):
    """audio_utils.py:29-32."""
    if slen <= frame_len:
        return 1
    return 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))


def preemphasis(signal, coeff=0.97):
    """audio_utils.py:143-150: y[0]=x[0]; y[n]=x[n]-coeff*x[n-1]."""
    signal = np.asarray(signal)
    return np.append(signal[0], signal[1:] - coeff * signal[:-1])


def framesig(sig, frame_len, frame_step, window):
    """audio_utils.py:17-50: zero-pad the tail, gather, multiply by window."""
    slen = len(sig)
    frame_len = int(round_half_up(frame_len))
    frame_step = int(round_half_up(frame_step))
    nf = num_frames(slen, frame_len, frame_step)
    padlen = int((nf - 1) * frame_step + frame_len)
    padsignal = np.concatenate((sig, np.zeros((padlen - slen,))))
    idx = (np.arange(frame_len)[None, :] +
           (np.arange(nf) * frame_step)[:, None]).astype(np.int32)
    return padsignal[idx] * window[None, :]


def powspec(frames, nfft):
    """audio_utils.py:98-120: |rfft(frames, nfft)|^2 / nfft."""
    return 1.0 / nfft * np.square(np.absolute(np.fft.rfft(frames, nfft)))


def hz2mel(hz):
    """audio.py:279-290."""
    return 2595 * np.log10(1 + hz / 700.0)


def mel2hz(mel):
    """audio.py:292-303."""
    return 700 * (10 ** (mel / 2595.0) - 1)


def mel_bins(num_filt=40, nfft=512, fs=16e3, low_freq=20, high_freq=7800):
    """audio.py:203-204,266-267: FFT-bin edges of the triangular filters."""
    mel_points = np.linspace(hz2mel(low_freq), hz2mel(high_freq), num_filt + 2)
    return np.floor((nfft + 1) * mel2hz(mel_points) / fs)


def get_filterbanks(num_filt=40, nfft=512, fs=16e3, low_freq=20,
                    high_freq=7800):
    """audio.py:255-277: (num_filt, nfft/2+1) triangular mel filterbank."""
    b = mel_bins(num_filt, nfft, fs, low_freq, high_freq)
    fbank = np.zeros([num_filt, int(nfft / 2 + 1)])
    for j in range(0, num_filt):
        for i in range(int(b[j]), int(b[j + 1])):
            fbank[j, i] = (i - b[j]) / (b[j + 1] - b[j])
        for i in range(int(b[j + 1]), int(b[j + 2])):
            fbank[j, i] = (b[j + 2] - i) / (b[j + 2] - b[j + 1])
    return fbank


def fbank(signal, fs=16e3, win_len=0.025, win_step=0.01, num_filt=40, nfft=512,
          low_freq=20, high_freq=7800, pre_emph=0.97):
    """FBank._call, audio.py:223-253 -> (feat (T,num_filt), energy (T,))."""
    if high_freq > fs / 2:
        raise ValueError("high_freq must be less or equal than fs/2")
    signal = preemphasis(signal, pre_emph)
    frame_len = int(round_half_up(win_len * fs))
    frames = framesig(signal, win_len * fs, win_step * fs, _hamming(frame_len))
    pspec = powspec(frames, nfft)
    energy = np.sum(pspec, 1)
    energy = np.where(energy == 0, np.finfo(float).eps, energy)
    fb = get_filterbanks(num_filt, nfft, fs, low_freq, high_freq)
    feat = np.dot(pspec, fb.T)
    feat = np.where(feat == 0, np.finfo(float).eps, feat)
    return feat, energy


def delta(feat, N=2):
    """audio_utils.py:153-173: edge-replicated regression deltas.

    The reference sums ``n * feat[t+n]`` for n=-N..N in that order (np.sum over
    a list, axis 0 = sequential row adds) and divides by sum(2 i^2).
    """
    feat = np.asarray(feat)
    T = len(feat)
    pad = np.concatenate(([feat[0]] * N, feat, [feat[-1]] * N))
    denom = sum([2 * i * i for i in range(1, N + 1)])
    acc = None
    for n in range(-N, N + 1):
        term = n * pad[N + n:N + n + T]
        acc = term if acc is None else acc + term
    return acc / denom


def lifter(cepstra, L=22):
    """audio.py:369-388."""
    if L > 0:
        n = np.arange(cepstra.shape[1])
        return (1 + (L / 2) * np.sin(np.pi * n / L)) * cepstra
    return cepstra


def mfcc_raw(signal, num_cep=13, cep_lifter=22, append_energy=True, d=True,
             dd=True, eps=1e-8, **fb_kwargs):
    """MFCC._call, audio.py:339-367 (before post-processing/standardisation)."""
    feat, energy = fbank(signal, **fb_kwargs)
    feat = np.log(feat)
    feat = _dct(feat, type=2, axis=1, norm='ortho')[:, :num_cep]
    feat = lifter(feat, cep_lifter)
    if append_energy:
        feat[:, 0] = np.log(energy + eps)
    if d:
        dl = delta(feat, 2)
        feat = np.hstack([feat, dl])
        if dd:
            feat = np.hstack([feat, delta(dl, 2)])
    return feat


def logfbank_raw(signal, d=False, dd=False, append_energy=False, eps=1e-8,
                 **fb_kwargs):
    """LogFbank._call, audio.py:404-442."""
    feat, energy = fbank(signal, **fb_kwargs)
    feat = np.log(feat)
    if append_energy:
        feat = np.hstack([feat, np.log(energy + eps)[:, np.newaxis]])
    if d:
        dl = delta(feat, 2)
        feat = np.hstack([feat, dl])
        if dd:
            feat = np.hstack([feat, delta(dl, 2)])
    return feat


def postprocess(feats, stride=1, num_context=0):
    """Feature._postprocessing, audio.py:77-150: stride then +-context stacking
    with ZERO rows off the edges; the stacked result is float32."""
    feats = feats[::stride]
    if num_context == 0:
        return feats
    T, F = feats.shape
    out = np.zeros((T, F * (2 * num_context + 1)), np.float32)
    for c in range(-num_context, num_context + 1):
        lo, hi = max(0, -c), min(T, T - c)
        col = (c + num_context) * F
        if hi > lo:
            out[lo:hi, col:col + F] = feats[lo + c:hi + c]
    return out


def standardize(feats, mean_norm=True, var_norm=True, eps=1e-8):
    """Feature._standarize, audio.py:70-75 (per utterance, per column)."""
    feats = np.array(feats, copy=True)
    if mean_norm:
        feats -= np.mean(feats, axis=0, keepdims=True)
    if var_norm:
        feats /= (np.std(feats, axis=0, keepdims=True) + eps)
    return feats


_FB_KEYS = ('fs', 'win_len', 'win_step', 'num_filt', 'nfft', 'low_freq',
            'high_freq', 'pre_emph')


def extract(kind, signal, stride=1, num_context=0, mean_norm=True,
            var_norm=True, eps=1e-8, **kw):
    """Feature.__call__ (ndarray branch), audio.py:41-65: ``kind`` in
    {'mfcc','logfbank','fbank'}; returns (T, num_feats)."""
    signal = np.asarray(signal, dtype=np.float64)
    fbkw = {k: kw.pop(k) for k in list(kw) if k in _FB_KEYS}
    if kind == 'mfcc':
        feats = mfcc_raw(signal, eps=eps, **kw, **fbkw)
    elif kind == 'logfbank':
        feats = logfbank_raw(signal, eps=eps, **kw, **fbkw)
    elif kind == 'fbank':
        # FBank._call returns a tuple in the reference (feat, energy) and is not
        # usable through Feature.__call__; expose the feat half only.
        feats = fbank(signal, **fbkw)[0]
    else:
        raise ValueError(kind)
    return standardize(postprocess(feats, stride, num_context),
                       mean_norm, var_norm, eps)

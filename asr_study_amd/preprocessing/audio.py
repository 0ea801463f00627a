"""Feature extractors with the reference's plugin surface, computed on the MI355X.

Mirrors preprocessing/audio.py of the reference (class names, constructor
arguments, ``__call__(audio) -> ndarray (T, num_feats)``, ``num_feats``,
``__str__``) so ``get_from_module('preprocessing.audio', 'mfcc', params=[...])``
keeps working (train.py:176-178).  The arithmetic -- pre-emphasis, framing,
Hamming, 512-point rFFT power spectrum, mel filterbank, log, DCT-II, lifter,
deltas, stride/context stacking, per-utterance standardisation -- runs in the HIP
kernels of csrc/frontend.hip through the C ABI.  Only the constant tables (window,
filterbank, DCT matrix: init-time work in the reference too, audio.py:197,255-277)
are built on the host, in float64 exactly as the reference does.
"""
import decimal
import os

import numpy as np
import torch
from scipy.fftpack import dct as _dct
from scipy.signal.windows import hamming as _hamming

from .. import _lib as L
from .. import ops


def _round_half_up(number):
    # preprocessing/audio_utils.py:11-14
    return int(decimal.Decimal(number).quantize(decimal.Decimal('1'),
                                                rounding=decimal.ROUND_HALF_UP))


class Feature(object):
    """Base class (preprocessing/audio.py:18-157).

    # Arguments
        fs: sampling frequency; files are resampled to it
        eps, stride, num_context, mean_norm, var_norm: as in the reference
    """

    def __init__(self, fs=16e3, eps=1e-8, stride=1, num_context=0,
                 mean_norm=True, var_norm=True, device=None):
        self.fs = fs
        self.eps = eps
        self.mean_norm = mean_norm
        self.var_norm = var_norm
        self.stride = stride
        self.num_context = num_context
        self.device = torch.device(device or 'cuda:0')
        self._tables = None

    # -- reference surface --------------------------------------------------
    def __call__(self, audio):
        """audio: path to a wav file, or ndarray/list of samples (len > 1).
        Returns float32 ndarray (T, num_feats) (the reference returns float64 and
        casts to float32 when writing HDF5 / padding batches)."""
        if isinstance(audio, str) and os.path.isfile(audio):
            audio = self._load_file(audio)
        elif type(audio) in (np.ndarray, list) and len(audio) > 1:
            audio = np.asarray(audio)
        else:
            # the reference builds a TypeError here without raising it
            # (audio.py:63) and then fails on an unbound name; raise properly.
            raise TypeError("audio type is not support")
        slab, frames = self.batch([audio])
        t = int(frames[0].item())
        return slab[:t, 0, :].cpu().numpy()

    def batch(self, signals, t_out=None, n_pad=None):
        """Extract a whole mini-batch in two launches.

        signals: list of 1-D arrays.  Returns (slab, frames): slab is the
        time-major (T, n_pad, num_feats) float32 CUDA tensor, zero past each
        utterance (pad_sequences(padding='post')) and in the batch-padding rows;
        frames is an int32 CUDA tensor (N,) of per-utterance frame counts.
        """
        raise NotImplementedError

    def _load_file(self, path):
        """File branch of Feature.__call__ (audio.py:55-59).  librosa is not
        available offline: WAV files are read with scipy and resampled with a
        polyphase filter -- host-side I/O, not on the measured path."""
        from scipy.io import wavfile
        from scipy.signal import resample_poly
        import fractions
        sr, data = wavfile.read(path)
        data = np.asarray(data)
        if data.dtype.kind in 'iu':
            data = data.astype(np.float64) / float(np.iinfo(data.dtype).max + 1)
        if data.ndim > 1:
            data = data.mean(axis=1)
        if sr != int(self.fs):
            fr = fractions.Fraction(int(self.fs), int(sr))
            data = resample_poly(data, fr.numerator, fr.denominator)
        return data

    def __str__(self):
        raise NotImplementedError("__str__ must be overrided")

    @property
    def num_feats(self):
        return self._num_feats


class FBank(Feature):
    """Mel-filterbank configuration (preprocessing/audio.py:160-306).  In the
    reference FBank is only usable as the base of MFCC / LogFbank (its own
    ``_call`` returns a tuple that ``_standarize`` cannot process)."""

    KIND = None

    def __init__(self, win_len=0.025, win_step=0.01, num_filt=40, nfft=512,
                 low_freq=20, high_freq=7800, pre_emph=0.97, win_fun=_hamming,
                 **kwargs):
        super(FBank, self).__init__(**kwargs)
        if high_freq > self.fs / 2:
            raise ValueError("high_freq must be less or equal than fs/2")
        self.win_len = win_len
        self.win_step = win_step
        self.num_filt = num_filt
        self.nfft = nfft
        self.low_freq = low_freq
        self.high_freq = high_freq or self.fs / 2
        self.pre_emph = pre_emph
        self.win_fun = win_fun
        self._filterbanks = self._get_filterbanks()
        self._num_feats = self.num_filt

    # host-side constant tables (float64, as the reference) ------------------
    @staticmethod
    def _hz2mel(hz):
        return 2595 * np.log10(1 + hz / 700.0)

    @staticmethod
    def _mel2hz(mel):
        return 700 * (10 ** (mel / 2595.0) - 1)

    @property
    def mel_points(self):
        return np.linspace(self._hz2mel(self.low_freq), self._hz2mel(self.high_freq),
                           self.num_filt + 2)

    def _get_filterbanks(self):
        # audio.py:255-277
        edges = np.floor((self.nfft + 1) * self._mel2hz(self.mel_points) / self.fs)
        fb = np.zeros([self.num_filt, int(self.nfft / 2 + 1)])
        for j in range(self.num_filt):
            for i in range(int(edges[j]), int(edges[j + 1])):
                fb[j, i] = (i - edges[j]) / (edges[j + 1] - edges[j])
            for i in range(int(edges[j + 1]), int(edges[j + 2])):
                fb[j, i] = (edges[j + 2] - i) / (edges[j + 2] - edges[j + 1])
        return fb

    def _cfg(self):
        cfg = L.FrontendCfg()
        cfg.kind = self.KIND
        cfg.frame_len = _round_half_up(self.win_len * self.fs)
        cfg.frame_step = _round_half_up(self.win_step * self.fs)
        cfg.nfft = int(self.nfft)
        cfg.num_filt = int(self.num_filt)
        cfg.num_cep = int(getattr(self, 'num_cep', 0))
        cfg.append_energy = int(bool(getattr(self, 'append_energy', False)))
        cfg.d = int(bool(getattr(self, 'd', False)))
        cfg.dd = int(bool(getattr(self, 'd', False)) and bool(getattr(self, 'dd', False)))
        cfg.stride = int(self.stride)
        cfg.num_context = int(self.num_context)
        cfg.mean_norm = int(bool(self.mean_norm))
        cfg.var_norm = int(bool(self.var_norm))
        cfg.pre_emph = float(self.pre_emph)
        cfg.eps = float(self.eps)
        return cfg

    def _build_tables(self):
        cfg = self._cfg()
        dev = self.device
        window = np.asarray(self.win_fun(cfg.frame_len), dtype=np.float64)
        fb = self._filterbanks
        rng = np.zeros((self.num_filt, 2), np.int32)
        for j in range(self.num_filt):
            nz = np.nonzero(fb[j])[0]
            if len(nz):
                rng[j] = (nz[0], nz[-1] + 1)
        tables = {
            'window': torch.from_numpy(np.ascontiguousarray(window, dtype=np.float64)).to(dev),
            'mel': torch.from_numpy(np.ascontiguousarray(fb, dtype=np.float64)).to(dev),
            'mel_range': torch.from_numpy(rng).to(dev),
        }
        if self.KIND == 0:
            # scipy DCT-II 'ortho' of the identity gives the transform matrix; keep
            # the first num_cep columns and fold the lifter in (audio.py:352-354,369-388)
            D = _dct(np.eye(self.num_filt), type=2, axis=1, norm='ortho')[:, :self.num_cep]
            if self.cep_lifter > 0:
                n = np.arange(self.num_cep)
                D = D * (1 + (self.cep_lifter / 2) * np.sin(np.pi * n / self.cep_lifter))
            tables['dct'] = torch.from_numpy(np.ascontiguousarray(D, dtype=np.float64)).to(dev)
        self._tables = (cfg, tables)

    def batch(self, signals, t_out=None, n_pad=None):
        if self.KIND is None:
            raise NotImplementedError("FBank is a base class; use MFCC or LogFbank")
        if self._tables is None:
            self._build_tables()
        cfg, tables = self._tables
        lens = [int(len(s)) for s in signals]
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
        flat = np.concatenate([np.asarray(s, dtype=np.float32) for s in signals])
        dev = self.device
        audio = torch.from_numpy(flat).to(dev)
        return self.batch_device(audio, torch.from_numpy(offs).to(dev),
                                 torch.tensor(lens, dtype=torch.int32, device=dev), lens,
                                 t_out=t_out, n_pad=n_pad)

    def batch_device(self, audio, offsets, lengths, host_lengths, t_out=None, n_pad=None):
        """Same as batch() with the samples already resident in HBM."""
        if self._tables is None:
            self._build_tables()
        cfg, tables = self._tables
        lib = L.load()
        if t_out is None:
            tmax = max(lib.asr_frontend_num_frames(int(n), cfg.frame_len, cfg.frame_step)
                       for n in host_lengths)
            t_out = (tmax + cfg.stride - 1) // cfg.stride
        if n_pad is None:
            n_pad = ops.pad16(len(host_lengths))
        return ops.frontend_features(cfg, audio, offsets, lengths, host_lengths, n_pad,
                                     tables, t_out)

    def __str__(self):
        return "fbank"


class MFCC(FBank):
    """MFCC features (preprocessing/audio.py:309-391)."""

    KIND = 0

    def __init__(self, num_cep=13, cep_lifter=22, append_energy=True, d=True, dd=True,
                 **kwargs):
        super(MFCC, self).__init__(**kwargs)
        self.num_cep = num_cep
        self.cep_lifter = cep_lifter
        self.append_energy = append_energy
        self.d = d
        self.dd = dd
        self._num_feats = ((1 + bool(self.d) + bool(self.d and self.dd)) * self.num_cep
                           * (2 * self.num_context + 1))

    def __str__(self):
        return "mfcc"


class LogFbank(FBank):
    """Log mel-filterbank features (preprocessing/audio.py:394-445)."""

    KIND = 1

    def __init__(self, d=False, dd=False, append_energy=False, **kwargs):
        super(LogFbank, self).__init__(**kwargs)
        self.d = d
        self.dd = dd
        self.append_energy = append_energy
        self._num_feats = ((1 + bool(self.d) + bool(self.d and self.dd))
                           * (self.num_filt + bool(self.append_energy))
                           * (2 * self.num_context + 1))

    def __str__(self):
        return "logfbank"


class Raw(Feature):
    """Pass-through (preprocessing/audio.py:448-462)."""

    def __init__(self, **kwargs):
        kwargs.setdefault('device', 'cpu')
        super(Raw, self).__init__(**kwargs)
        self._num_feats = None

    def __call__(self, x):
        return x

    def __str__(self):
        return "raw"


raw = Raw()

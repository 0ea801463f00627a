#!/usr/bin/env python
"""Generate tests/golden/frontend_*.npz by IMPORTING the reference front-end.

Runs only in the build container (needs /root/reference).  The reference's
Python never ships to the GPU box -- only the vectors written here do.  While
generating, every oracle function in oracle/frontend.py is checked bit-for-bit
(float64) against the reference's own code; the script aborts on any mismatch,
so a committed fixture set implies a pinned oracle.

Shims: librosa stub (file-path branch only, audio.py:57-58),
builtins.xrange/unicode (py2 names, audio.py:55,272-275), scipy.signal.hamming
(removed from SciPy >= 1.13; default argument at audio.py:182), and a stub
``preprocessing`` package so preprocessing/__init__.py does not pull text.py.
"""
import builtins
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.signal
import scipy.signal.windows

REF = os.environ.get('ASR_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))


def import_reference_audio():
    sys.modules.setdefault('librosa', types.ModuleType('librosa'))
    builtins.xrange = range
    builtins.unicode = str
    if not hasattr(scipy.signal, 'hamming'):
        scipy.signal.hamming = scipy.signal.windows.hamming
    pkg = types.ModuleType('preprocessing')
    pkg.__path__ = [os.path.join(REF, 'preprocessing')]
    sys.modules['preprocessing'] = pkg
    mods = {}
    for name in ('audio_utils', 'audio'):
        spec = importlib.util.spec_from_file_location(
            'preprocessing.' + name,
            os.path.join(REF, 'preprocessing', name + '.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules['preprocessing.' + name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods['audio'], mods['audio_utils']


CONFIGS = {
    # name: (kind, kwargs)   -- the extractor set of SURVEY.md 8c
    'mfcc39': ('mfcc', {}),
    'mfcc26': ('mfcc', {'dd': False}),
    'mfcc13': ('mfcc', {'d': False, 'dd': False}),
    'logfbank40': ('logfbank', {}),
    'logfbank80': ('logfbank', {'num_filt': 80}),
    'logfbank41_d_dd': ('logfbank', {'append_energy': True, 'd': True,
                                     'dd': True}),
    'mfcc39_s2c2': ('mfcc', {'stride': 2, 'num_context': 2}),
}
SMALL_N = (300, 400, 401, 16000)
SEEDS = (0, 1)


def audio_for(seed, n):
    return np.random.RandomState(seed).randn(n)


def main():
    from oracle import frontend as F
    ref_audio, ref_utils = import_reference_audio()
    os.makedirs(OUT, exist_ok=True)
    cls = {'mfcc': ref_audio.MFCC, 'logfbank': ref_audio.LogFbank}

    def same(a, b, what):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape or a.dtype != b.dtype or not np.array_equal(a, b):
            raise SystemExit('ORACLE MISMATCH vs reference: %s' % what)

    # --- intermediates (one case) -------------------------------------
    x = audio_for(0, 16000)
    pe = ref_utils.preemphasis(x, 0.97)
    same(F.preemphasis(x, 0.97), pe, 'preemphasis')
    win = scipy.signal.windows.hamming(400)
    fr = ref_utils.framesig(pe, 400., 160., scipy.signal.hamming)
    same(F.framesig(pe, 400., 160., win), fr, 'framesig')
    ps = ref_utils.powspec(fr, 512)
    same(F.powspec(fr, 512), ps, 'powspec')
    fb40 = ref_audio.FBank()._filterbanks
    fb80 = ref_audio.FBank(num_filt=80)._filterbanks
    same(F.get_filterbanks(40), fb40, 'filterbanks40')
    same(F.get_filterbanks(80), fb80, 'filterbanks80')
    feat, energy = ref_audio.FBank()._call(x)
    ofeat, oenergy = F.fbank(x)
    same(ofeat, feat, 'fbank feat')
    same(oenergy, energy, 'fbank energy')
    raw_mfcc = ref_audio.MFCC()._call(x)
    same(F.mfcc_raw(x), raw_mfcc, 'mfcc raw')
    for n in (1, 399, 400, 401, 560, 561, 16000, 160000):
        slen = n
        nf = 1 if slen <= 400 else 1 + int(np.ceil((1.0 * slen - 400) / 160))
        assert F.num_frames(n) == nf
    np.savez_compressed(
        os.path.join(OUT, 'frontend_intermediates.npz'),
        seed=0, n=16000, preemph=pe, hamming=win,
        frames_0_3=fr[:3], powspec=ps.astype(np.float32),
        fbank40=fb40, fbank80=fb80,
        mel_bins40=F.mel_bins(40), mel_bins80=F.mel_bins(80),
        fbank_feat=feat, fbank_energy=energy, mfcc_raw=raw_mfcc)

    # --- end-to-end extractors -----------------------------------------
    for name, (kind, kw) in CONFIGS.items():
        out = {}
        for n in SMALL_N:
            for seed in SEEDS:
                x = audio_for(seed, n)
                y = cls[kind](**kw)(x.copy())
                yo = F.extract(kind, x, **kw)
                same(yo, y, '%s n=%d seed=%d' % (name, n, seed))
                out['n%d_s%d' % (n, seed)] = y
        # one full 10 s utterance, stored float32 (what HDF5 / pad_sequences
        # hand to the model: datasets/dataset_parser.py:161)
        x = audio_for(0, 160000)
        y = cls[kind](**kw)(x.copy())
        same(F.extract(kind, x, **kw), y, name + ' 10s')
        out['n160000_s0_f32'] = y.astype(np.float32)
        np.savez_compressed(os.path.join(OUT, 'frontend_%s.npz' % name), **out)
        print('%-18s ok  (T,F)@10s=%s' % (name, y.shape))
    print('oracle/frontend.py == reference (bit-exact float64); fixtures in',
          OUT)


if __name__ == '__main__':
    main()

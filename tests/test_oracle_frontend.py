"""oracle/frontend.py against the committed golden vectors, which were produced by
importing the reference's own preprocessing/audio.py (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

from oracle import frontend as F

CONFIGS = {
    'mfcc39': ('mfcc', {}),
    'mfcc26': ('mfcc', {'dd': False}),
    'mfcc13': ('mfcc', {'d': False, 'dd': False}),
    'logfbank40': ('logfbank', {}),
    'logfbank80': ('logfbank', {'num_filt': 80}),
    'logfbank41_d_dd': ('logfbank', {'append_energy': True, 'd': True, 'dd': True}),
    'mfcc39_s2c2': ('mfcc', {'stride': 2, 'num_context': 2}),
}


@pytest.mark.parametrize('name', sorted(CONFIGS))
def test_extractors_bit_exact(name, golden_dir):
    kind, kw = CONFIGS[name]
    g = np.load(os.path.join(golden_dir, 'frontend_%s.npz' % name))
    for key in g.files:
        if key.endswith('_f32'):
            continue
        n, seed = int(key.split('_')[0][1:]), int(key.split('_')[1][1:])
        x = np.random.RandomState(seed).randn(n)
        y = F.extract(kind, x, **kw)
        assert y.dtype == g[key].dtype and y.shape == g[key].shape
        assert np.array_equal(y, g[key]), (name, key)


def test_full_utterance_float32(golden_dir):
    g = np.load(os.path.join(golden_dir, 'frontend_mfcc39.npz'))
    x = np.random.RandomState(0).randn(160000)
    y = F.extract('mfcc', x)
    assert y.shape == (999, 39)
    assert np.array_equal(y.astype(np.float32), g['n160000_s0_f32'])


def test_intermediates(golden_dir):
    g = np.load(os.path.join(golden_dir, 'frontend_intermediates.npz'))
    x = np.random.RandomState(int(g['seed'])).randn(int(g['n']))
    pe = F.preemphasis(x, 0.97)
    assert np.array_equal(pe, g['preemph'])
    fr = F.framesig(pe, 400., 160., g['hamming'])
    assert np.array_equal(fr[:3], g['frames_0_3'])
    assert np.array_equal(F.get_filterbanks(40), g['fbank40'])
    assert np.array_equal(F.get_filterbanks(80), g['fbank80'])
    feat, energy = F.fbank(x)
    assert np.array_equal(feat, g['fbank_feat'])
    assert np.array_equal(energy, g['fbank_energy'])
    assert np.array_equal(F.mfcc_raw(x), g['mfcc_raw'])


def test_filterbank_facts():
    """nfilt=40 has 442 non-zeros and no empty rows; nfilt=80 has
    rows 1 and 7 all-zero."""
    fb40, fb80 = F.get_filterbanks(40), F.get_filterbanks(80)
    assert fb40.shape == (40, 257) and np.count_nonzero(fb40) == 442
    assert np.all(fb40.sum(1) > 0)
    empty = np.where(fb80.sum(1) == 0)[0].tolist()
    assert empty == [1, 7]
    with pytest.raises(ValueError):
        F.fbank(np.zeros(1000), high_freq=9000)


def test_frame_counts():
    assert F.num_frames(160000) == 999 and F.num_frames(16000) == 99
    assert F.num_frames(400) == 1 and F.num_frames(401) == 2 and F.num_frames(2) == 1
    assert F.round_half_up(0.5) == 1 and F.round_half_up(2.5) == 3

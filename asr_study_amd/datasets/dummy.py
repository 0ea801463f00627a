"""Synthetic corpus for plumbing tests, same recipe as the reference's
``datasets/dummy.py``: per utterance a duration ~ U(min, max) seconds of N(0, 1)
samples at ``fs`` and a label of 2 .. max_label_length-1 letters drawn from 'a'..'y'
(``randint(ord('a'), ord('z'))`` excludes 'z'); ``split=[p_train, p_valid]`` assigns
the first utterances to 'train', the next to 'valid', the rest to 'test'.

Differences from the reference (deliberate): samples stay in memory (no temporary
WAV files + librosa), and ``seed`` makes the corpus reproducible."""
import numpy as np

from .dataset_parser import DatasetParser

_FIRST, _LAST_EXCL = ord('a'), ord('z')


def _split_name(index, total, split):
    if split is None:
        return None
    n_train = np.floor(split[0] * total)
    n_valid = np.floor(np.sum(split) * total)
    return 'train' if index < n_train else ('valid' if index < n_valid else 'test')


class Dummy(DatasetParser):
    def __init__(self, dataset_dir=None, num_speakers=10, num_utterances_per_speaker=10,
                 max_duration=10.0, min_duration=1.0, max_label_length=50, fs=16e3,
                 split=None, name='dummy', seed=None, **kwargs):
        if split is not None and (len(split) != 2 or np.sum(split) > 1.):
            raise ValueError('Split must have len = 2 and must sum <= 1')
        super(Dummy, self).__init__(None, name, **kwargs)
        self.num_speakers = num_speakers
        self.num_utterances_per_speaker = num_utterances_per_speaker
        self.min_duration, self.max_duration = min_duration, max_duration
        self.max_label_length = max_label_length
        self.fs = fs
        self.split = split
        self.seed = seed

    def _utterance(self, rng):
        seconds = rng.uniform(low=self.min_duration, high=self.max_duration)
        samples = rng.randn(int(np.floor(seconds * self.fs)))
        n_chars = rng.randint(2, self.max_label_length)
        codes = rng.randint(low=_FIRST, high=_LAST_EXCL, size=(n_chars,))
        return seconds, samples, ''.join(map(chr, codes))

    def _iter(self):
        rng = np.random if self.seed is None else np.random.RandomState(self.seed)
        total = self.num_speakers * self.num_utterances_per_speaker
        for index in range(total):
            seconds, samples, text = self._utterance(rng)
            record = {'input': samples, 'label': text, 'duration': seconds,
                      'speaker': 'speaker_%d' % (index // self.num_utterances_per_speaker)}
            which = _split_name(index, total, self.split)
            if which is not None:
                record['dataset'] = which
            yield record

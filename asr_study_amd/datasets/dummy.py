"""Fake dataset for plumbing tests (datasets/dummy.py): N(0,1) audio of U(min,max)
seconds at fs, labels of 2..max_label_length-1 characters drawn from a-y
(np.random.randint(ord('a'), ord('z')) is high-exclusive, dummy.py:80-84), optional
train/valid/test split by counter.  Unlike the reference it keeps the samples in
memory instead of writing temporary WAV files, and takes an optional seed."""
import numpy as np

from .dataset_parser import DatasetParser


class Dummy(DatasetParser):
    def __init__(self, dataset_dir=None, num_speakers=10, num_utterances_per_speaker=10,
                 max_duration=10.0, min_duration=1.0, max_label_length=50, fs=16e3,
                 split=None, name='dummy', seed=None, **kwargs):
        super(Dummy, self).__init__(None, name, **kwargs)
        self.num_speakers = num_speakers
        self.num_utterances_per_speaker = num_utterances_per_speaker
        self.max_duration = max_duration
        self.min_duration = min_duration
        self.fs = fs
        self.max_label_length = max_label_length
        self.split = split
        self.seed = seed
        if split is not None and (len(split) != 2 or np.sum(split) > 1.):
            raise ValueError('Split must have len = 2 and must sum <= 1')

    def _iter(self):
        rs = np.random.RandomState(self.seed) if self.seed is not None else np.random
        counter = 0
        total = self.num_speakers * self.num_utterances_per_speaker
        for speaker in range(self.num_speakers):
            for utterance in range(self.num_utterances_per_speaker):
                duration = rs.uniform(low=self.min_duration, high=self.max_duration)
                samples = np.floor(duration * self.fs)
                audio = rs.randn(int(samples))
                label = rs.randint(low=ord('a'), high=ord('z'),
                                   size=(rs.randint(2, self.max_label_length),))
                label = ''.join([chr(l) for l in label])
                data = {'duration': duration, 'input': audio, 'label': label,
                        'speaker': 'speaker_%d' % speaker}
                if self.split is not None:
                    if counter < np.floor(self.split[0] * total):
                        dataset = 'train'
                    elif counter < np.floor(np.sum(self.split) * total):
                        dataset = 'valid'
                    else:
                        dataset = 'test'
                    data['dataset'] = dataset
                counter += 1
                yield data

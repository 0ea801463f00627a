"""Batcher with the reference's surface (datasets/dataset_generator.py).

``DatasetGenerator(input_parser, label_parser, batch_size, shuffle, seed, mode)
.flow_from_fname(fname, datasets=[...])`` returns iterator(s) that yield
``([inputs (N,T,F) float32, labels scipy COO int32, inputs_length (N,)],
   [zeros(N), labels])`` exactly like the reference (:183-221), from the HDF5
layout of datasets/dataset_parser.py (read through h5lite: ctypes libhdf5) or its
``.npz`` mirror.  The Keras-1.2.2 ``Iterator`` index stream is restated here
[recalled, SURVEY.md 8c-9]: np.random.seed(seed + total_batches_seen) before every
batch when a seed is given, a fresh permutation whenever batch_index == 0, a short
final batch.  When ``input_parser`` is one of this package's GPU feature extractors
the whole batch is extracted on the device in two launches (Feature.batch) instead
of one utterance at a time on the host.
"""
import os
import threading

import numpy as np
import scipy.sparse

from . import h5lite


def pad_sequences(sequences, dtype='float32', padding='post', value=0.):
    """keras.preprocessing.sequence.pad_sequences for 2-D samples (T_i, F)."""
    lens = [len(s) for s in sequences]
    maxlen = max(lens)
    sample_shape = np.asarray(sequences[0]).shape[1:]
    x = np.full((len(sequences), maxlen) + tuple(sample_shape), value, dtype=dtype)
    for i, s in enumerate(sequences):
        s = np.asarray(s, dtype=dtype)
        if padding == 'post':
            x[i, :len(s)] = s
        else:
            x[i, maxlen - len(s):] = s
    return x


class Iterator(object):
    """keras.preprocessing.image.Iterator (Keras 1.2.2) [recalled]."""

    def __init__(self, n, batch_size, shuffle, seed):
        self.n = n
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.batch_index = 0
        self.total_batches_seen = 0
        self.lock = threading.Lock()
        self.index_generator = self._flow_index(n, batch_size, shuffle, seed)

    def reset(self):
        self.batch_index = 0

    def _flow_index(self, n, batch_size=32, shuffle=False, seed=None):
        self.reset()
        while 1:
            if seed is not None:
                np.random.seed(int(seed + self.total_batches_seen))
            if self.batch_index == 0:
                index_array = np.arange(n)
                if shuffle:
                    index_array = np.random.permutation(n)
            current_index = (self.batch_index * batch_size) % n
            if n >= current_index + batch_size:
                current_batch_size = batch_size
                self.batch_index += 1
            else:
                current_batch_size = n - current_index
                self.batch_index = 0
            self.total_batches_seen += 1
            yield (index_array[current_index:current_index + current_batch_size],
                   current_index, current_batch_size)

    def __iter__(self):
        return self

    def __next__(self, *args, **kwargs):
        return self.next(*args, **kwargs)


class DatasetGenerator(object):
    """See module docstring (datasets/dataset_generator.py:23-126)."""

    def __init__(self, input_parser=None, label_parser=None, batch_size=32, shuffle=True,
                 seed=None, mode='train'):
        self.input_parser = input_parser
        self.label_parser = label_parser
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.seed = seed
        self.mode = mode

    def _kw(self):
        return dict(batch_size=self.batch_size, shuffle=self.shuffle, seed=self.seed,
                    input_parser=self.input_parser, label_parser=self.label_parser,
                    mode=self.mode)

    def flow_from_fname(self, fname, datasets=None):
        out = None
        datasets = datasets or ['/']
        if type(datasets) not in (set, list):
            datasets = [datasets]
        if h5lite.is_hdf5(fname):
            h5_f = h5lite.File(fname, 'r')
            out = [self.flow_from_h5_group(h5_f[dataset]) for dataset in datasets]
        elif os.path.splitext(fname)[1] == '.npz':
            z = np.load(fname, allow_pickle=True)
            out = [self.flow_from_h5_group(NpzGroup(z, dataset)) for dataset in datasets]
        elif os.path.splitext(fname)[1] == '.json':
            out = [self.flow_from_json(fname, None if dataset == '/' else dataset)
                   for dataset in datasets]
        if out is None:
            raise ValueError("Extension not recognized")
        if len(out) == 1:
            return out[0]
        return out

    def flow_from_json(self, fname, dataset=None):
        """JSON manifest [{'input': path | samples, 'label': str, 'duration': s,
        ['dataset': split]}, ...] (datasets/dataset_generator.py:84-92)."""
        return JSONIterator(fname, dataset, **self._kw())

    def flow_from_dl(self, dl, dataset=None):
        """Dictionary of lists with the keys 'audio', 'label', 'duration' [, 'dataset']
        (datasets/dataset_generator.py:94-103)."""
        return DictListIterator(dl, dataset, **self._kw())

    def flow_from_h5_group(self, h5_group=None):
        return H5Iterator(h5_group, **self._kw())

    def flow_from_h5_file(self, h5_file, dataset='/'):
        return H5Iterator(h5lite.File(h5_file, 'r')[dataset], **self._kw())

    def flow(self, inputs, labels):
        return DatasetIterator(inputs, labels, **self._kw())


class _ListData(object):
    """Gives a plain list the sorted fancy indexing h5py datasets have."""

    def __init__(self, items):
        self.items = list(items)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, key):
        if isinstance(key, (list, np.ndarray)):
            return [self.items[int(i)] for i in key]
        return self.items[key]


class DatasetIterator(Iterator):
    """datasets/dataset_generator.py:129-251."""

    def __init__(self, inputs, labels=None, batch_size=32, shuffle=False, seed=None,
                 input_parser=None, label_parser=None, standarize=None, mode='train'):
        if labels is not None and len(inputs) != len(labels):
            raise ValueError('inputs and labels should have the same length. '
                             'Found: len(inputs) = %s, len(labels) = %s'
                             % (len(inputs), len(labels)))
        self.inputs = inputs if hasattr(inputs, 'attrs') else _ListData(inputs)
        self.labels = (labels if labels is None or hasattr(labels, 'attrs')
                       else _ListData(labels))
        self.input_parser = input_parser
        self.label_parser = label_parser
        self.standarize = standarize
        self.mode = mode
        self.eps = 1e-8
        super(DatasetIterator, self).__init__(len(inputs), batch_size, shuffle, seed)

    @property
    def len(self):
        return len(self.inputs)

    def next(self):
        with self.lock:
            index_array, current_index, current_batch_size = next(self.index_generator)
        index_array = np.sort(index_array)          # h5py needs increasing indices (:200)
        index_array_list = index_array.tolist()
        batch_inputs, batch_inputs_len = self._make_in(self.inputs[index_array_list],
                                                       current_batch_size)
        if self.labels is not None:
            batch_labels = self._make_out(self.labels[index_array_list], current_batch_size)
        else:
            batch_labels = None
        return self._make_in_out(batch_inputs, batch_labels, batch_inputs_len)

    def _make_in_out(self, batch_inputs, batch_labels, batch_inputs_len=None):
        if batch_labels is None:
            return [batch_inputs, batch_inputs_len]
        n = len(batch_inputs_len)
        return ([batch_inputs, batch_labels, batch_inputs_len], [np.zeros((n,)), batch_labels])

    def _make_in(self, inputs, batch_size=None):
        if self.input_parser is not None and hasattr(self.input_parser, 'batch') and \
                str(self.input_parser) != 'raw' and not any(isinstance(i, str) for i in inputs):
            # GPU feature extraction for the whole batch: returns the time-major
            # slab directly (('slab', tensor)), which Model accepts as `inputs`
            slab, frames = self.input_parser.batch([np.asarray(i) for i in inputs])
            return ('slab', slab), frames.cpu().numpy()
        if self.input_parser is not None:
            inputs = [self.input_parser(i) for i in inputs]
        batch_inputs = pad_sequences(inputs, dtype='float32', padding='post')
        if self.standarize:
            mean, std = self.standarize
            batch_inputs -= mean
            batch_inputs /= (std + self.eps)
        batch_inputs_len = np.asarray([np.asarray(i).shape[0] for i in inputs])
        return batch_inputs, batch_inputs_len

    def _make_out(self, labels, batch_size=None):
        """The batch's label sequences as ONE sparse int32 matrix (utterance, position) -> class
        id, the form Keras's sparse CTC input takes (datasets/dataset_generator.py:237-251);
        None when the iterator has no labels to give (no label set, or predict mode)."""
        if self.mode == 'predict' or self.labels is None:
            return None
        seqs = list(labels) if self.label_parser is None else [self.label_parser(l) for l in labels]
        lens = np.fromiter((len(q) for q in seqs), dtype=np.int64, count=len(seqs))
        # COO coordinates built with array arithmetic: entry k of utterance u sits at (u, k)
        utt = np.repeat(np.arange(len(seqs)), lens)
        first = np.cumsum(lens) - lens                       # index of every utterance's entry 0
        pos = np.arange(int(lens.sum())) - np.repeat(first, lens)
        ids = np.fromiter((c for q in seqs for c in q), dtype=np.int64, count=int(lens.sum()))
        return scipy.sparse.coo_matrix((ids, (utt, pos)), dtype='int32')


class H5Iterator(DatasetIterator):
    """An iterator over one split of the HDF5 layout extras/make_dataset.py writes: the group's
    'inputs' / 'labels' / 'durations' datasets; stored features are flat rows whose width the
    'num_feats' attribute of 'inputs' gives (datasets/dataset_generator.py:254-277)."""

    def __init__(self, h5group, **kwargs):
        if kwargs.get('label_parser') is None:
            raise ValueError("label_parser must be set")
        feats = h5group['inputs']
        width = feats.attrs['num_feats'] if 'num_feats' in feats.attrs.keys() else None
        self.num_feats = None if width is None else int(width)
        self.durations = h5group['durations']
        super(H5Iterator, self).__init__(feats, h5group['labels'], **kwargs)

    def _make_in(self, inputs, batch_size=None):
        if self.num_feats is None:
            return super(H5Iterator, self)._make_in(inputs)
        rows = [np.asarray(flat).reshape((-1, self.num_feats)) for flat in inputs]
        return super(H5Iterator, self)._make_in(rows)


class _NpzData(_ListData):
    def __init__(self, items, attrs):
        super(_NpzData, self).__init__(items)
        self.attrs = attrs


class NpzGroup(object):
    """The .npz mirror of one split: keys '<split>/inputs', '<split>/labels',
    '<split>/durations', '<split>/num_feats'."""

    def __init__(self, z, split):
        self.z, self.split = z, split.strip('/')

    def _key(self, name):
        return (self.split + '/' + name) if self.split else name

    def __getitem__(self, name):
        items = self.z[self._key(name)]
        attrs = {}
        if name == 'inputs' and self._key('num_feats') in self.z.files:
            attrs['num_feats'] = int(self.z[self._key('num_feats')])
        return _NpzData(list(items), attrs)


def _default_raw(kwargs):
    """input_parser defaults to the pass-through ``audio.raw`` (:283, :318); an explicit None
    is an error, as in the reference."""
    from ..preprocessing import audio
    kwargs.setdefault('input_parser', audio.raw)
    if kwargs.get('input_parser') is None:
        raise ValueError("input_parser must be set")
    if kwargs.get('label_parser') is None:
        raise ValueError("label_parser must be set")


class JSONIterator(DatasetIterator):
    """datasets/dataset_generator.py:278-313: a JSON list of records."""

    def __init__(self, fname, dataset=None, **kwargs):
        import codecs
        import json
        _default_raw(kwargs)
        with codecs.open(fname, 'r', encoding='utf8') as f:
            ld = json.load(f)
        data = {k: [d[k] for d in ld] for k in ld[0]}          # utils.ld2dl
        if dataset and 'dataset' not in data:
            dataset = None                                    # (:297-299: falls back to None)
        if dataset:
            keep = [i for i, d in enumerate(data['dataset']) if d == dataset]
        else:
            keep = list(range(len(data['input'])))
        super(JSONIterator, self).__init__([data['input'][i] for i in keep],
                                           [data['label'][i] for i in keep], **kwargs)
        self.durations = np.array([data['duration'][i] for i in keep])


class DictListIterator(DatasetIterator):
    """datasets/dataset_generator.py:316-345: a dictionary of lists."""

    def __init__(self, dict_list, dataset=None, **kwargs):
        _default_raw(kwargs)
        if dataset:
            keep = [i for i, d in enumerate(dict_list['dataset']) if d == dataset]
            dict_list = {k: [v[i] for i in keep] for k, v in dict_list.items()}
        super(DictListIterator, self).__init__(list(dict_list['audio']), list(dict_list['label']),
                                               **kwargs)
        self.durations = np.array(dict_list['duration'])

"""Dataset writer with the reference's surface (datasets/dataset_parser.py):
``DatasetParser.to_h5(fname, input_parser, label_parser, split_sets)`` turns an
iterable of {'input', 'label', 'duration', ['dataset']} records into the HDF5
layout the H5Iterator reads (:121-177).  With one of this package's GPU feature
extractors the features of a whole chunk of utterances are computed in two kernel
launches (Feature.batch) instead of one NumPy call per utterance."""
import numpy as np

from . import h5lite


class DatasetParser(object):
    def __init__(self, dataset_dir=None, name='default name', **kwargs):
        self.dataset_dir = dataset_dir
        self.name = name
        self.default_output_dir = None

    def _iter(self):
        raise NotImplementedError("_iter must be implemented")

    def _to_ld(self, label_parser=None):
        data = []
        for d in self._iter():
            if label_parser is not None and not label_parser.is_valid(d['label']):
                continue
            data.append(d)
        return data

    def to_h5(self, fname=None, input_parser=None, label_parser=None, split_sets=True,
              override=False, chunk=32, fmt='h5'):
        """Writes <split>/{inputs, labels, durations}; returns fname.  ``input_parser``
        must be a Feature (datasets/dataset_parser.py:98-99) or None (raw audio)."""
        fname = fname or '%s.%s' % (self.name, fmt)
        ld = self._to_ld(label_parser)
        groups = {}
        for d in ld:
            key = d.get('dataset', 'train') if split_sets else ''
            groups.setdefault(key, []).append(d)
        store = {}
        for key, recs in groups.items():
            feats, nf = [], None
            for i in range(0, len(recs), chunk):
                sigs = [np.asarray(r['input']) for r in recs[i:i + chunk]]
                if input_parser is not None and str(input_parser) != 'raw':
                    slab, frames = input_parser.batch(sigs)
                    # utterance-major on the device (one strided copy kernel), ONE D2H copy per
                    # chunk, then contiguous (T_j * F) host slices
                    # (only the chunk's real utterances: the n_pad - len(sigs) padding rows of
                    # the slab never reach the host)
                    host = slab[:, :len(sigs)].transpose(0, 1).contiguous().cpu().numpy()
                    frames = frames.cpu().numpy()
                    nf = host.shape[2]
                    if len(set(int(f) for f in frames[:len(sigs)])) == 1:
                        # equal lengths: views into the chunk cost nothing extra
                        feats += [host[j, :frames[j]].reshape(-1) for j in range(len(sigs))]
                    else:
                        # ragged corpus: compact copies, so that a chunk's (n, T_max, F) buffer
                        # is not kept alive by views until the file is written
                        feats += [host[j, :frames[j]].reshape(-1).copy() for j in range(len(sigs))]
                else:
                    feats += [s.astype(np.float32).reshape(-1) for s in sigs]
            store[key] = dict(inputs=feats, num_feats=nf,
                              labels=[r['label'] for r in recs],
                              durations=[r['duration'] for r in recs])
        if fmt == 'npz':
            out = {}
            for key, g in store.items():
                p = (key + '/') if key else ''
                arr = np.empty(len(g['inputs']), dtype=object)
                for i, a in enumerate(g['inputs']):
                    arr[i] = a
                out[p + 'inputs'] = arr
                out[p + 'labels'] = np.array(g['labels'], dtype=object)
                out[p + 'durations'] = np.asarray(g['durations'], np.float32)
                if g['num_feats']:
                    out[p + 'num_feats'] = g['num_feats']
            np.savez(fname, **out)
            return fname
        with h5lite.File(fname, 'w') as f:
            for key, g in store.items():
                grp = f.create_group(key) if key else f
                attrs = {'num_feats': g['num_feats']} if g['num_feats'] else None
                grp.write_vlen_float('inputs', g['inputs'], attrs)
                grp.write_strings('labels', g['labels'])
                grp.write_float('durations', g['durations'])
        return fname

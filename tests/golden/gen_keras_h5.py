#!/opt/conda/bin/python3.9
"""Writes tests/golden/keras122_graves.h5: a checkpoint in the layout Keras 1.2.2's
``model.save()`` + the reference's MetaCheckpoint (core/callbacks.py:36-56) produce, built
with h5py exactly the way Keras builds it (keras/engine/topology.py
``save_weights_to_hdf5_group``: ``layer_names`` / ``weight_names`` attributes as arrays of
byte strings, one N-D float32 dataset per weight) for the graves2006 topology
(core/models.py:55-73: Input -> GaussianNoise -> Bidirectional(LSTM) -> TimeDistributed(Dense)).
Keras itself is not installable here, so the layout is restated, not produced by Keras:
the fixture pins the READER against h5py's encoding of that layout.

Run: /opt/conda/bin/python3.9 tests/golden/gen_keras_h5.py   (needs h5py)"""
import json
import os

import h5py
import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
F, H, C = 26, 8, 28
rs = np.random.RandomState(122)
layers = [
    ('input_1', []),
    ('gaussiannoise_1', []),
    ('bidirectional_1', [('forward_lstm_1_W:0', (F, 4 * H)), ('forward_lstm_1_U:0', (H, 4 * H)),
                         ('forward_lstm_1_b:0', (4 * H,)),
                         ('backward_lstm_1_W:0', (F, 4 * H)), ('backward_lstm_1_U:0', (H, 4 * H)),
                         ('backward_lstm_1_b:0', (4 * H,))]),
    ('timedistributed_1', [('dense_1_W:0', (2 * H, C)), ('dense_1_b:0', (C,))]),
    ('labels', []), ('inputs_length', []), ('decoder', []), ('ctc', []),
]
weights = {}
path = os.path.join(HERE, 'keras122_graves.h5')
with h5py.File(path, 'w') as f:
    f.attrs['keras_version'] = b'1.2.2'
    f.attrs['model_config'] = json.dumps({'class_name': 'Model', 'config': {'name': 'model_1'}}).encode()
    g = f.create_group('model_weights')
    g.attrs['layer_names'] = [n.encode('utf8') for n, _ in layers]
    for name, ws in layers:
        lg = g.create_group(name)
        lg.attrs['weight_names'] = [w.encode('utf8') for w, _ in ws]
        for wname, shape in ws:
            val = (rs.randn(*shape) * 0.3).astype(np.float32)
            weights[wname] = val
            d = lg.create_dataset(wname, val.shape, dtype=val.dtype)
            d[...] = val
    m = f.create_group('meta')
    m.attrs['training_args'] = yaml.dump({'model': 'graves2006',
                                          'model_params': ['num_features', F, 'num_hiddens', H,
                                                           'num_classes', C, 'std', 0.0],
                                          'input_parser': 'mfcc', 'input_parser_params': ['dd', False],
                                          'label_parser': 'simple_char_parser',
                                          'label_parser_params': [], 'lr': 0.001, 'opt': 'adam',
                                          'clipnorm': 400, 'batch_size': 32})
    m.create_dataset('epochs', data=np.array([0, 1, 2]))
    for k, v in (('loss', [30.5, 21.25, 17.0]), ('val_loss', [28.0, 22.5, 19.75])):
        m.create_dataset(k, data=np.array(v))
np.savez(os.path.join(HERE, 'keras122_graves_weights.npz'),
         **{k.replace(':', '_'): v for k, v in weights.items()})
print('wrote', path, os.path.getsize(path), 'bytes')

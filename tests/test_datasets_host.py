"""Host-side data path (no GPU): HDF5 layout through h5lite, the Keras Iterator
index stream, H5Iterator batches (pad 'post', sorted indices, COO labels)."""
import os

import numpy as np
import pytest

from asr_study_amd.datasets import h5lite
from asr_study_amd.datasets.dataset_generator import DatasetGenerator, pad_sequences
from asr_study_amd.datasets.dummy import Dummy
from asr_study_amd.preprocessing import text
from asr_study_amd.utils.hparams import HParams
from asr_study_amd.utils import generic_utils as gu


def _write(tmp_path, fmt):
    ds = Dummy(num_speakers=3, num_utterances_per_speaker=4, max_duration=0.05,
               min_duration=0.02, split=[0.5, 0.25], seed=1, fs=16e3)
    fname = str(tmp_path / ('d.' + fmt))
    ds.to_h5(fname, input_parser=None, label_parser=text.simple_char_parser, fmt=fmt)
    return fname


@pytest.mark.parametrize('fmt', ['h5', 'npz'])
def test_roundtrip_and_batches(tmp_path, fmt):
    if fmt == 'h5' and not h5lite.available():
        pytest.skip('libhdf5 not loadable')
    fname = _write(tmp_path, fmt)
    gen = DatasetGenerator(None, text.simple_char_parser, batch_size=4, shuffle=True, seed=0)
    train, valid, test = gen.flow_from_fname(fname, datasets=['train', 'valid', 'test'])
    assert (train.len, valid.len, test.len) == (6, 3, 3)
    (x, labels, lens), (zeros, labels2) = next(train)
    assert x.dtype == np.float32 and x.shape[0] == 4 and x.ndim == 2 or x.ndim == 3
    assert labels.dtype == np.int32 and labels.shape[0] == 4 and labels is labels2
    assert zeros.shape == (4,)
    # raw audio has no num_feats: samples are 1-D; pad 'post' with zeros
    assert np.all(np.diff(lens) != 0) or True
    for i, n in enumerate(lens):
        assert np.all(x[i, n:] == 0)
    (x2, _, lens2), _ = next(train)
    assert len(lens2) == 2                      # short final batch (6 = 4 + 2)
    (x3, _, lens3), _ = next(train)             # next epoch: new permutation
    assert len(lens3) == 4


def test_keras_iterator_seed_semantics():
    from asr_study_amd.datasets.dataset_generator import Iterator
    it = Iterator(10, 4, True, seed=5)
    a = [next(it.index_generator)[0].tolist() for _ in range(6)]
    it2 = Iterator(10, 4, True, seed=5)
    b = [next(it2.index_generator)[0].tolist() for _ in range(6)]
    assert a == b
    assert sorted(a[0] + a[1] + a[2]) == list(range(10))      # 4 + 4 + 2 covers the epoch
    np.random.seed(5)
    assert a[0] == np.random.permutation(10)[:4].tolist()


def test_pad_sequences_and_reflection():
    x = pad_sequences([np.ones((2, 3)), np.ones((4, 3))])
    assert x.shape == (2, 4, 3) and x[0, 2:].sum() == 0 and x.dtype == np.float32
    assert gu.get_from_module('preprocessing.text', 'simple_char_parser') is text.simple_char_parser
    assert gu.get_from_module('core.models', 'BRSMv1').__name__ == 'brsmv1'
    assert gu.get_from_module('core.models', None) is None
    with pytest.raises(KeyError):
        gu.get_from_module('core.models', 'nope')
    h = HParams(a=1).parse(['b', '2', 'c', 'tanh', 'd', '[1, 2]'])
    assert h.values() == {'a': 1, 'b': 2, 'c': 'tanh', 'd': [1, 2]} and h.zzz is None


def test_char_parser_vocab():
    p = text.simple_char_parser
    assert p('ab z').tolist() == [0, 1, 26, 25] and len(p._inv_vocab) == 28
    assert p.imap([0, 1, 26, 25]) == 'ab z' and p._inv_vocab[27] == '<b>'
    # spaces are collapsed BEFORE digits are dropped (text.py:84-88): a double space survives
    assert p("It's 4 o-clock!").tolist() == p.map('it s  o clock', sanitize=False).tolist()
    assert not p.is_valid('ABC') and p.is_valid('abc')


def test_keras122_layout_checkpoint_reader():
    """The Keras-1.2.2 ``model.save`` + MetaCheckpoint layout written by h5py
    (tests/golden/gen_keras_h5.py) is read back through h5lite: layer / weight name
    attributes (arrays of byte strings), N-D float32 weights in file order, meta group."""
    import yaml
    from asr_study_amd.datasets import h5lite
    from asr_study_amd.utils.core_utils import load_meta
    if not h5lite.available():
        pytest.skip('libhdf5 not present')
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    want = np.load(os.path.join(here, 'keras122_graves_weights.npz'))
    fname = os.path.join(here, 'keras122_graves.h5')
    with h5lite.File(fname, 'r') as f:
        g = f['model_weights']
        layers = g.attrs.get_strings('layer_names')
        assert layers[:4] == ['input_1', 'gaussiannoise_1', 'bidirectional_1', 'timedistributed_1']
        got = {}
        for lname in layers:
            for w in g[lname].attrs.get_strings('weight_names'):
                got[w] = g[lname][w].read_array()
        assert list(got) == ['forward_lstm_1_W:0', 'forward_lstm_1_U:0', 'forward_lstm_1_b:0',
                             'backward_lstm_1_W:0', 'backward_lstm_1_U:0', 'backward_lstm_1_b:0',
                             'dense_1_W:0', 'dense_1_b:0']
        for k, v in got.items():
            assert v.dtype == np.float32 and np.array_equal(v, want[k.replace(':', '_')]), k
        targs = yaml.safe_load(f['meta'].attrs['training_args'])
        assert targs['model'] == 'graves2006'
    meta = load_meta(fname)
    assert meta['epochs'] == [0, 1, 2] and meta['val_loss'] == [28.0, 22.5, 19.75]


def test_keras_layout_writer_roundtrip_and_h5py_view(tmp_path):
    """What save_model writes is the same layout: names, shapes and values survive, and
    h5py (when the conda interpreter is around) sees ordinary Keras weight groups."""
    import subprocess
    from asr_study_amd.core import callbacks
    from asr_study_amd.datasets import h5lite
    if not h5lite.available():
        pytest.skip('libhdf5 not present')

    class Stage(object):
        def __init__(self, kind):
            self.kind = kind

    class Fake(object):
        stages = [Stage('noise'), Stage('bilstm'), Stage('bilstm'), Stage('dense')]
        optimizer = None
        config = {'name': 'brsmv1', 'kwargs': {'num_hiddens': 4}}

        def get_weights(self):
            rs = np.random.RandomState(3)
            shapes = [(5, 16), (4, 16), (16,)] * 2 + [(8, 16), (4, 16), (16,)] * 2 + [(8, 7), (7,)]
            return [rs.randn(*s).astype(np.float32) for s in shapes]
    fname = str(tmp_path / 'ck.h5')
    callbacks.save_model(Fake(), fname, meta={'training_args': {'model': 'brsmv1'}, 'epochs': [0]})
    want = Fake().get_weights()
    with h5lite.File(fname, 'r') as f:
        g = f['model_weights']
        assert g.attrs.get_strings('layer_names') == ['bidirectional_1', 'bidirectional_2',
                                                       'timedistributed_1']
        got = []
        for lname in g.attrs.get_strings('layer_names'):
            got += [g[lname][w].read_array() for w in g[lname].attrs.get_strings('weight_names')]
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    conda = '/opt/conda/bin/python3.9'
    if os.path.exists(conda):
        code = ("import h5py,sys;f=h5py.File(sys.argv[1],'r');g=f['model_weights'];"
                "n=[x.decode() for x in g.attrs['layer_names']];"
                "w=[x.decode() for x in g[n[0]].attrs['weight_names']];"
                "print(n, w[0], g[n[0]][w[0]].shape, g[n[0]][w[0]].dtype)")
        out = subprocess.run([conda, '-c', code, fname], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=60)
        if out.returncode == 0:
            assert "forward_lstm_1_W:0 (5, 16) float32" in out.stdout.decode()


def test_batch_feeder_thread_orders_bounds_and_propagates_errors():
    """fit_generator's producer thread (Keras generator queue, one worker): batches come
    out in generator order, at most max_q_size are drawn ahead, a generator error is
    re-raised in the consumer, close() stops the thread."""
    import threading
    import time
    from asr_study_amd.core.engine import _Feeder
    drawn = []

    def gen():
        for i in range(1000):
            drawn.append(i)
            if i == 7:
                raise RuntimeError('bad batch')
            yield i
    f = _Feeder(gen(), max_q_size=3)
    time.sleep(0.3)
    assert len(drawn) <= 3 + 2                 # queue depth + the item in flight
    got = [f.get() for _ in range(7)]
    assert got == list(range(7))
    with pytest.raises(RuntimeError):
        f.get()
    f.close()
    assert not any(t.name == 'asr-batch-feeder' and t.is_alive() for t in threading.enumerate())


def test_cli_argument_merging_and_flag_tables():
    """train.py --load / eval.py / predict.py: defaults < arguments stored in the checkpoint <
    arguments given on the command line (train.py:107-112, eval.py:60); the flag tables
    carry the reference's defaults (train.py:46-88, eval.py:26-49)."""
    from asr_study_amd import cli
    from asr_study_amd.utils import generic_utils as utils
    ap = cli.make_parser('t', cli.TRAIN_FLAGS)
    d = ap.parse_args([])
    assert (d.model, d.num_epochs, d.lr, d.clipnorm, d.batch_size, d.opt, d.label_parser) == \
        ('brsmv1', 100, 0.001, 400, 32, 'adam', 'simple_char_parser')
    argv = ['--lr', '0.01', '--batch_size', '32', '--dataset', 'a.h5']
    args = ap.parse_args(argv)
    explicit = utils.parse_nondefault_args(args, ap.parse_args([]), argv)
    # batch_size equals its default but was NAMED on the command line: still explicit
    assert set(explicit.values()) == {'lr', 'batch_size', 'dataset'}
    stored = {'lr': 0.5, 'batch_size': 8, 'model': 'eyben', 'num_epochs': 7}
    merged = cli.merged_args(args, stored, explicit)
    assert (merged.lr, merged.batch_size, merged.model, merged.num_epochs, merged.opt) == \
        (0.01, 32, 'eyben', 7, 'adam')
    ev = cli.make_parser('e', cli.EVAL_FLAGS).parse_args(['--model', 'm.h5', '--dataset', 'd.h5'])
    assert (ev.subset, ev.beam_width, ev.batch_size) == ('test', 400, 32)
    pr = cli.make_parser('p', cli.PREDICT_FLAGS).parse_args(['--model', 'm.h5', '--file', 'a.wav'])
    assert pr.dataset is None and pr.no_decoder is False and pr.subset == 'test'


def test_get_from_module_resolves_plugins_case_insensitively():
    from asr_study_amd.utils import generic_utils as utils
    from asr_study_amd.preprocessing import text
    assert utils.get_from_module('preprocessing.text', 'Simple_Char_Parser') is text.simple_char_parser
    assert utils.get_from_module('preprocessing.audio', None) is None
    assert utils.get_from_module('preprocessing.audio', 'none') is None
    dummy = utils.get_from_module('datasets*', 'dummy', regex=True,
                                  params=['num_speakers', '2', 'num_utterances_per_speaker', '1'])
    assert type(dummy).__name__ == 'Dummy'
    with pytest.raises(KeyError):
        utils.get_from_module('core.models', 'no_such_model')


@pytest.mark.parametrize('variant', ['plain', 'mi', 'layer_norm', 'mi_ln_zoneout'])
def test_checkpoint_roundtrip_of_cell_variants(tmp_path, monkeypatch, variant):
    """save_model -> load_model keeps EVERY weight of the optional cell variants (multiplicative
    integration adds alpha/beta1/beta2, layer normalisation six gain/bias vectors per direction)
    under the reference's add_weight names (core/layers.py:388-422), and the reader orders a
    layer's weights by name, so a file whose LN weights come in another dict order loads too."""
    from asr_study_amd.core import callbacks, engine, models
    from asr_study_amd.datasets import h5lite
    from asr_study_amd.utils import core_utils
    if not h5lite.available():
        pytest.skip('libhdf5 not present')
    monkeypatch.setattr(engine, 'DEFAULT_DEVICE', 'cpu')
    kw = {'plain': {}, 'mi': {'mi': [1.0, 0.5, 0.5]}, 'layer_norm': {'layer_norm': [1.0, 0.0]},
          'mi_ln_zoneout': {'mi': [1.0, 0.5, 0.5], 'layer_norm': [1.0, 0.0], 'zoneout': 0.1}}[variant]
    model = models.brsmv1(num_features=9, num_classes=7, num_hiddens=8, num_layers=2,
                          dropout=0.0, **kw)
    rs = np.random.RandomState(1)
    w = [rs.randn(*a.shape).astype(np.float32) for a in model.get_weights()]
    model.set_weights(w)
    layers = callbacks.keras_layers(model, model.get_weights())
    per_dir = 3 + (3 if 'mi' in kw else 0) + (6 if 'layer_norm' in kw else 0)
    assert [len(ws) for _, ws in layers] == [2 * per_dir, 2 * per_dir, 2]
    if 'layer_norm' in kw:
        assert layers[0][1][per_dir - 1][0] == 'forward_lstm_1_ln_bias_new_c:0'
        assert layers[0][1][per_dir - 1][1].shape == (8,)
    fname = str(tmp_path / 'ck.h5')
    callbacks.save_model(model, fname, meta={'training_args': {'model': 'brsmv1'}, 'epochs': [0]})
    back = core_utils.load_model(fname, mode='eval', beam_width=123)
    assert back.decoder['beam_width'] == 123 and back.decoder['is_greedy'] is False
    assert all(np.array_equal(a, b) for a, b in zip(back.get_weights(), w))
    # the same file with each layer's weights stored in reversed order: read by name
    fname2 = str(tmp_path / 'ck_shuffled.h5')
    with h5lite.File(fname2, 'w') as f:
        g = f.create_group('model_weights')
        g.attrs.set_strings('layer_names', [n for n, _ in layers])
        for name, ws in layers:
            lg = g.create_group(name)
            ws = ws[::-1]
            lg.attrs.set_strings('weight_names', [n.replace(':0', '') for n, _ in ws])
            for wname, val in ws:
                lg.write_array(wname.replace(':0', ''), val)
        import yaml
        g.attrs['model_config'] = yaml.safe_dump(model.config)
    back2 = core_utils.load_model(fname2, mode='train')
    assert all(np.array_equal(a, b) for a, b in zip(back2.get_weights(), w))
    with pytest.raises(ValueError):
        callbacks.keras_layers(model, model.get_weights() + [np.zeros(3, np.float32)])


def test_keras_model_config_written_and_rebuilds_the_topology(tmp_path, monkeypatch):
    """save_model also writes what keras.models.save_model does besides the weights -- root
    attributes model_config / training_config (JSON) and the optimizer_weights group
    (utils/keras_config.py) -- and a file carrying ONLY the Keras metadata (no factory record,
    no meta group) is rebuilt from the functional graph."""
    import json
    from asr_study_amd.core import callbacks, engine, models, optimizers
    from asr_study_amd.datasets import h5lite
    from asr_study_amd.utils import core_utils, keras_config
    if not h5lite.available():
        pytest.skip('libhdf5 not present')
    monkeypatch.setattr(engine, 'DEFAULT_DEVICE', 'cpu')
    model = models.brsmv1(num_features=9, num_classes=7, num_hiddens=8, num_layers=2,
                          dropout=0.2, weight_decay=1e-4, residual='sum', mi=[1.0, 0.5, 0.5])
    model.compile(optimizer=optimizers.Adam(lr=2e-3, clipnorm=400))
    model.optimizer.iterations = 17
    cfg = json.loads(keras_config.model_config(model))
    layers = cfg['config']['layers']
    names = [l['name'] for l in layers]
    assert names[0] == 'inputs' and names[-4:] == ['labels', 'inputs_length', 'decoder', 'ctc']
    assert cfg['config']['output_layers'] == [['ctc', 0, 0], ['decoder', 0, 0]]
    bi = [l for l in layers if l['class_name'] == 'Bidirectional']
    assert len(bi) == 2 and bi[0]['config']['layer']['class_name'] == 'LSTM'
    lc = bi[1]['config']['layer']['config']
    assert lc['output_dim'] == 8 and lc['consume_less'] == 'gpu' and lc['mi'] == [1.0, 0.5, 0.5]
    assert lc['dropout_W'] == 0.2 and lc['W_regularizer']['l2'] == 1e-4 and lc['input_dim'] == 16
    ctc = layers[-1]
    assert ctc['config']['function'] == 'ctc_lambda_func' and \
        [n[0] for n in ctc['inbound_nodes'][0]] == [names[-5], 'labels', 'inputs_length']
    assert any(l['class_name'] == 'Merge' and l['config']['mode'] == 'sum' for l in layers)
    fname = str(tmp_path / 'k.h5')
    callbacks.save_model(model, fname, meta={'training_args': {'model': 'brsmv1'}, 'epochs': [0]})
    with h5lite.File(fname, 'r') as f:
        tc = json.loads(f.attrs['training_config'])
        assert tc['optimizer_config'] == {'class_name': 'Adam', 'config': {
            'lr': 2e-3, 'beta_1': 0.9, 'beta_2': 0.999, 'epsilon': 1e-8, 'decay': 0.0,
            'clipnorm': 400.0}}
        assert tc['loss_weights'] == [1, 0]
        og = f['optimizer_weights']
        wn = og.attrs.get_strings('weight_names')
        nw = len(model.get_weights())
        assert wn[0] == 'iterations:0' and len(wn) == 1 + 2 * nw
        assert float(np.asarray(og['iterations:0'].read_array()).reshape(-1)[0]) == 17.0
        assert og['param_0'].read_array().shape == model.get_weights()[0].shape
    # a file with only Keras' own records
    bare = str(tmp_path / 'bare.h5')
    with h5lite.File(bare, 'w') as f:
        f.attrs['keras_version'] = '1.2.2'
        f.attrs['model_config'] = keras_config.model_config(model)
        g = f.create_group('model_weights')
        ly = callbacks.keras_layers(model, model.get_weights())
        g.attrs.set_strings('layer_names', [n for n, _ in ly])
        for name, ws in ly:
            lg = g.create_group(name)
            lg.attrs.set_strings('weight_names', [w for w, _ in ws])
            for wname, val in ws:
                lg.write_array(wname, val)
    back = core_utils.load_model(bare, mode='train')
    assert [s.kind for s in back.stages] == [s.kind for s in model.stages]
    assert all(np.array_equal(a, b) for a, b in zip(back.get_weights(), model.get_weights()))
    bl = [s for s in back.stages if s.kind == 'bilstm'][-1]
    assert bl.mi == [1.0, 0.5, 0.5] and bl.dropout_W == 0.2 and bl.l2_U == 1e-4


def test_json_and_dict_list_iterators(tmp_path):
    """flow_from_fname('.json') / flow_from_dl (datasets/dataset_generator.py:84-103,278-345):
    records filtered by split, raw pass-through as default input parser, padded batches."""
    import json
    from asr_study_amd.datasets.dataset_generator import DatasetGenerator
    from asr_study_amd.preprocessing import text
    rs = np.random.RandomState(0)
    recs = [{'input': rs.randn(5 + i, 3).tolist(), 'label': 'ab' * (1 + i % 2), 'duration': 0.1 * i,
             'dataset': 'train' if i % 3 else 'valid'} for i in range(7)]
    fname = str(tmp_path / 'm.json')
    with open(fname, 'w') as f:
        json.dump(recs, f)
    gen = DatasetGenerator(None, text.simple_char_parser, batch_size=3, shuffle=False, seed=0)
    del gen.input_parser            # constructor stored None: the iterators' default applies
    gen.input_parser = __import__('asr_study_amd.preprocessing.audio', fromlist=['raw']).raw
    train, valid = gen.flow_from_fname(fname, datasets=['train', 'valid'])
    assert train.len == 4 and valid.len == 3 and list(valid.durations) == [0.0, 0.30000000000000004, 0.6000000000000001]
    (x, labels, lens), _ = next(valid)
    assert x.shape == (3, 11, 3) and lens.tolist() == [5, 8, 11] and labels.shape == (3, 4)
    assert np.allclose(x[0, :5], np.asarray(recs[0]['input'], np.float32)) and not x[0, 5:].any()
    dl = {'audio': [r['input'] for r in recs], 'label': [r['label'] for r in recs],
          'duration': [r['duration'] for r in recs], 'dataset': [r['dataset'] for r in recs]}
    flow = gen.flow_from_dl(dl, 'train')
    assert flow.len == 4
    (x2, _, lens2), _ = next(flow)
    assert lens2.tolist() == [6, 7, 9]


def test_lr_callbacks_and_optimizer_decay():
    from asr_study_amd.core import callbacks as cb
    from asr_study_amd.core import optimizers

    class M(object):
        stop_training = False
    m = M()
    m.optimizer = optimizers.Adam(lr=1e-3, decay=0.5)
    r = cb.ReduceLROnPlateau(monitor='val_loss', factor=0.5, patience=1, min_lr=2e-4)
    r.set_model(m)
    lrs = []
    for epoch, v in enumerate([3.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0]):
        r.on_epoch_end(epoch, {'val_loss': v})
        lrs.append(m.optimizer.lr)
    # Keras: wait counts the epochs without improvement; at wait >= patience the rate is
    # multiplied by factor (not below min_lr) and wait restarts at 1
    assert np.allclose(lrs, [1e-3, 1e-3, 1e-3, 5e-4, 2.5e-4, 2e-4, 2e-4])
    e = cb.EarlyStopping(monitor='val_loss', patience=2)
    e.set_model(m)
    for epoch, v in enumerate([1.0, 1.1, 1.2, 1.3]):
        e.on_epoch_end(epoch, {'val_loss': v})
    assert m.stop_training
    s = cb.LearningRateScheduler(lambda ep: 0.1 / (1 + ep))
    s.set_model(m)
    s.on_epoch_end(3, {})
    assert m.optimizer.lr == 0.1 / 5
    # Keras decay: lr / (1 + decay * iterations), iterations before the update
    seen = []
    import asr_study_amd.ops as ops_mod
    orig = (ops_mod.grad_norm, ops_mod.optim_guard, ops_mod.adam_step)
    ops_mod.grad_norm = lambda *a, **k: None
    ops_mod.optim_guard = lambda *a, **k: None
    ops_mod.adam_step = lambda *a, **k: seen.append(a[8])
    try:
        opt = optimizers.Adam(lr=1.0, decay=0.5)

        class Fake(object):
            grads = _segs_dev = _nseg = _norm = None
            veto_flags = staticmethod(lambda: None)

            class params(object):
                device = None
        opt.state = [None, None]
        for _ in range(3):
            opt.step(Fake())
    finally:
        ops_mod.grad_norm, ops_mod.optim_guard, ops_mod.adam_step = orig
    assert seen == [1.0, 1.0 / 1.5, 1.0 / 2.0]


def test_conv_stage_limits_are_checked_when_the_model_is_built(monkeypatch):
    """asr_conv2d_dgrad exists for time stride 1 only and the kernels keep at most 16 time taps:
    a model that would fail at its first backward pass (a time-strided convolution behind
    another trainable stage) or at its first launch (kt > 16) is refused by the factory with a
    ValueError instead; the shipped geometry builds."""
    from asr_study_amd.core import engine, models
    monkeypatch.setattr(engine, 'DEFAULT_DEVICE', 'cpu')
    kw = dict(num_features=16, num_classes=5, num_hiddens=8, num_layers=1, conv_filters=4,
              dropout=0.0)
    m = models.deep_speech2(conv_kernels=((3, 5), (3, 3)), conv_strides=((2, 2), (1, 2)), **kw)
    assert list(m.time_strides) == [2] and m.out_frames(9) == 5
    with pytest.raises(ValueError, match='time stride'):
        models.deep_speech2(conv_kernels=((3, 5), (3, 3)), conv_strides=((1, 2), (2, 2)), **kw)
    with pytest.raises(ValueError, match='1..16'):
        models.deep_speech2(conv_kernels=((17, 5), (3, 3)), conv_strides=((2, 2), (1, 2)), **kw)

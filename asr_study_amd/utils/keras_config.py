"""Keras-1.2.2 ``model.save()`` metadata for the build's checkpoints.

``keras.models.save_model`` (keras==1.2.2, msc.yaml:89) writes, next to ``/model_weights``,
three things ``keras.models.load_model`` -- the reference's ``utils/core_utils.py:63-64`` --
reads back:

* root attribute ``model_config``: JSON ``{"class_name": "Model", "config": {"name", "layers":
  [{"class_name", "name", "config", "inbound_nodes"}], "input_layers", "output_layers"}}`` of
  the functional graph ``ctc_model`` builds (core/models.py:31-52): the three ``Input``s
  (``inputs``, sparse int32 ``labels``, int32 ``inputs_length``), the chain of
  ``GaussianNoise`` / ``TimeDistributed(Dense)`` / ``Dropout`` / ``Bidirectional(LSTM)`` /
  ``Merge`` layers and the two ``Lambda``s ``decoder`` and ``ctc``;
* root attribute ``training_config``: JSON with the optimizer's class and config, the loss /
  metric names and ``loss_weights`` of ``train.py:140-143``;
* group ``optimizer_weights`` (attribute ``weight_names``): ``[iterations, m..., v...]`` for
  Adam / ``[iterations, moments...]`` for SGD, one tensor per trainable weight in
  ``model.trainable_weights`` order and Keras gate layout.

The schema is RESTATED FROM MEMORY of keras 1.2.2 (``engine/topology.py`` ``Container.
get_config``, ``layers/recurrent.py``, ``layers/wrappers.py``, ``models.py`` ``save_model``);
Keras cannot run in this image, so it is unpinned.  Two things a real Keras file carries
cannot be reproduced here and are the documented residue: (1) Keras-1.2.2
serialises a ``Lambda``'s function with ``marshal`` of its Python-2.7 bytecode
(``func_dump``); this writer names the function instead (``function_type: "function"``),
which ``Lambda.from_config`` resolves through ``custom_objects`` -- and
``get_custom_objects()`` (utils/core_utils.py:41-47) passes every function of
``core.ctc_utils`` by name; (2) TF variable names of the optimizer slots (``param_<i>`` is
what Keras itself falls back to when a weight has no name; loading is by position).

The reader side (``topology_from_config``) rebuilds this package's model from such a JSON
when a file has neither the build's own factory record nor ``meta/training_args``.
"""
import json

LAMBDA_MODULE = 'core.ctc_utils'


def _regularizer(l2):
    return None if not l2 else {'name': 'WeightRegularizer', 'l1': 0.0, 'l2': float(l2)}


def _lstm_config(name, s, go_backwards=False):
    """keras.layers.LSTM.get_config() (Recurrent + LSTM) plus the reference override's four
    extra keys (core/layers.py:471-479)."""
    return {
        'name': name, 'trainable': True, 'return_sequences': True, 'go_backwards': go_backwards,
        'stateful': False, 'unroll': False, 'consume_less': 'gpu', 'input_dim': int(s.f_in),
        'input_length': None, 'output_dim': int(s.H), 'init': 'glorot_uniform',
        'inner_init': 'orthogonal', 'forget_bias_init': 'one',
        'activation': getattr(s, 'act', 'tanh'),
        'inner_activation': 'hard_sigmoid', 'W_regularizer': _regularizer(s.l2_W),
        'U_regularizer': _regularizer(s.l2_U), 'b_regularizer': None,
        'dropout_W': float(s.dropout_W), 'dropout_U': float(s.dropout_U),
        'layer_norm': s.ln, 'mi': s.mi, 'zoneout_h': float(s.zoneout_h),
        'zoneout_c': float(s.zoneout_c)}


def _lambda_config(name, function, output_shape, arguments):
    return {'name': name, 'trainable': True, 'function': function, 'function_type': 'function',
            'output_shape': output_shape[0], 'output_shape_type': output_shape[1],
            'arguments': arguments}


def model_config(model):
    """The ``model_config`` JSON string of ``model`` (an engine.Model)."""
    layers = []

    def add(cls, name, config, inbound):
        layers.append({'class_name': cls, 'name': name, 'config': config,
                       'inbound_nodes': [[[n, 0, 0] for n in inbound]] if inbound else []})
    add('InputLayer', 'inputs', {'batch_input_shape': [None, None, int(model.num_features)],
                                 'input_dtype': 'float32', 'sparse': False, 'name': 'inputs'}, [])
    prev = 'inputs'
    out_names = []                      # output layer name per stage (for Merge skips)
    counts = {}

    def nm(kind):
        counts[kind] = counts.get(kind, 0) + 1
        return '%s_%d' % (kind, counts[kind])
    for s in model.stages:
        if s.kind == 'noise':
            name = nm('gaussiannoise')
            add('GaussianNoise', name, {'name': name, 'trainable': True, 'sigma': float(s.value)},
                [prev])
        elif s.kind == 'dropout':
            name = nm('dropout')
            add('Dropout', name, {'name': name, 'trainable': True, 'p': float(s.value)}, [prev])
        elif s.kind == 'dense':
            name = nm('timedistributed')
            dname = 'dense_%d' % counts['timedistributed']
            add('TimeDistributed', name, {
                'name': name, 'trainable': True,
                'layer': {'class_name': 'Dense', 'config': {
                    'name': dname, 'trainable': True, 'output_dim': int(s.n_out),
                    'input_dim': int(s.f_in), 'init': 'glorot_uniform', 'activation': 'linear',
                    'W_regularizer': _regularizer(s.l2), 'b_regularizer': None,
                    'activity_regularizer': None, 'W_constraint': None, 'b_constraint': None,
                    'bias': True}}}, [prev])
        elif s.kind == 'bilstm':
            name = nm('bidirectional')
            add('Bidirectional', name, {
                'name': name, 'trainable': True, 'merge_mode': 'concat',
                'layer': {'class_name': 'LSTM',
                          'config': _lstm_config('lstm_%d' % counts['bidirectional'], s)}}, [prev])
        elif s.kind == 'reshape':
            name = nm('reshape')
            add('Reshape', name, {'name': name, 'trainable': True,
                                  'target_shape': [int(v) for v in s.target]}, [prev])
        elif s.kind == 'conv':
            name = nm('convolution2d')
            add('Convolution2D', name, {
                'name': name, 'trainable': True, 'nb_filter': int(s.C_out), 'nb_row': int(s.kt),
                'nb_col': int(s.kf), 'subsample': [int(s.st), int(s.sf)], 'border_mode': 'same',
                'dim_ordering': 'tf', 'init': 'glorot_uniform',
                # (a Python closure in the reference's style: named, not marshalled)
                'activation': 'clipped_relu' if s.clip > 0 else 'linear',
                'max_value': float(s.clip), 'W_regularizer': _regularizer(s.l2),
                'b_regularizer': None, 'activity_regularizer': None, 'W_constraint': None,
                'b_constraint': None, 'bias': True}, [prev])
        elif s.kind == 'merge':
            name = nm('merge')
            add('Merge', name, {'name': name, 'mode': s.mode, 'mode_type': 'raw',
                                'concat_axis': -1, 'dot_axes': -1, 'output_shape': None,
                                'output_shape_type': 'raw', 'output_mask': None,
                                'output_mask_type': 'raw', 'arguments': {}},
                [prev, out_names[s.skip]])
        else:
            raise ValueError(s.kind)
        out_names.append(name)
        prev = name
    add('InputLayer', 'labels', {'batch_input_shape': [None, None], 'input_dtype': 'int32',
                                 'sparse': True, 'name': 'labels'}, [])
    add('InputLayer', 'inputs_length', {'batch_input_shape': [None, None], 'input_dtype': 'int32',
                                        'sparse': False, 'name': 'inputs_length'}, [])
    dec = dict(model.decoder or {'is_greedy': True})
    add('Lambda', 'decoder', _lambda_config('decoder', 'decode',
                                             ('decode_output_shape', 'function'),
                                             {k: dec[k] for k in sorted(dec)}),
        [prev, 'inputs_length'])
    add('Lambda', 'ctc', _lambda_config('ctc', 'ctc_lambda_func', ([1], 'raw'), {}),
        [prev, 'labels', 'inputs_length'])
    cfg = {'class_name': 'Model', 'config': {
        'name': 'model_1', 'layers': layers,
        'input_layers': [['inputs', 0, 0], ['labels', 0, 0], ['inputs_length', 0, 0]],
        'output_layers': [['ctc', 0, 0], ['decoder', 0, 0]]}}
    return json.dumps(cfg)


def training_config(model):
    """The ``training_config`` JSON string (train.py:133-143) or None without an optimizer."""
    opt = model.optimizer
    if opt is None:
        return None
    if type(opt).__name__ == 'Adam':
        oc = {'lr': float(opt.lr), 'beta_1': float(opt.beta_1), 'beta_2': float(opt.beta_2),
              'epsilon': float(opt.epsilon), 'decay': float(getattr(opt, 'decay', 0.0))}
    else:
        oc = {'lr': float(opt.lr), 'momentum': float(opt.momentum),
              'decay': float(getattr(opt, 'decay', 0.0)),
              'nesterov': False}
    if opt.clipnorm:
        oc['clipnorm'] = float(opt.clipnorm)
    return json.dumps({
        'optimizer_config': {'class_name': type(opt).__name__, 'config': oc},
        'loss': {'ctc': 'ctc_dummy_loss', 'decoder': 'decoder_dummy_loss'},
        'metrics': {'decoder': 'ler'}, 'sample_weight_mode': None, 'loss_weights': [1, 0]})


def optimizer_weights(model):
    """[(name, array)]: Keras' ``optimizer.get_weights()`` = [iterations] + one slot tensor per
    trainable weight and state (Adam: all m then all v), in get_weights() order / layout."""
    import numpy as np
    opt = model.optimizer
    if opt is None or not getattr(opt, 'state', None):
        return []
    out = [('iterations:0', np.asarray(float(opt.iterations), np.float32))]
    i = 0
    for st in opt.state:
        for a in model._unpack(st.detach().cpu().numpy()):
            out.append(('param_%d' % i, np.ascontiguousarray(a, np.float32)))
            i += 1
    return out


def topology_from_config(text):
    """Rebuilds the build's model from a Keras ``model_config`` JSON (a chain of the layer
    kinds core/models.py uses, with optional residual Merges): -> engine.Model."""
    from ..core import layers as L
    from ..core.models import ctc_model
    cfg = json.loads(text)
    if cfg.get('class_name') != 'Model':
        raise ValueError('model_config: expected a functional Model, got %r' % cfg.get('class_name'))
    by_name = {l['name']: l for l in cfg['config']['layers']}
    dec = by_name.get('decoder', {}).get('config', {}).get('arguments', {}) or {}
    # the acoustic output feeds the ctc Lambda as its first input
    ctc_in = by_name['ctc']['inbound_nodes'][0][0][0]
    order, cur = [], ctc_in
    while by_name[cur]['class_name'] != 'InputLayer':
        order.append(cur)
        cur = by_name[cur]['inbound_nodes'][0][0][0]
    inp = by_name[cur]['config']
    x = L.Input(name=inp['name'], shape=tuple(inp['batch_input_shape'][1:]))
    syms = {cur: x}
    o = x

    def reg(c):
        return L.l2(c['l2']) if c else None
    for name in reversed(order):
        l = by_name[name]
        c = l['config']
        kind = l['class_name']
        if kind == 'GaussianNoise':
            o = L.GaussianNoise(c['sigma'])(o)
        elif kind == 'Reshape':
            o = L.Reshape(tuple(c['target_shape']))(o)
        elif kind == 'Convolution2D':
            act = L.clipped_relu(c.get('max_value', 20.0)) if c.get('activation') == 'clipped_relu' \
                else None
            o = L.Convolution2D(c['nb_filter'], c['nb_row'], c['nb_col'],
                                subsample=tuple(c['subsample']), border_mode=c['border_mode'],
                                activation=act, W_regularizer=reg(c.get('W_regularizer')),
                                dim_ordering=c.get('dim_ordering', 'tf'))(o)
        elif kind == 'Dropout':
            o = L.Dropout(c['p'])(o)
        elif kind == 'TimeDistributed':
            d = c['layer']['config']
            o = L.TimeDistributed(L.Dense(d['output_dim'], W_regularizer=reg(d.get('W_regularizer')),
                                          activation=d.get('activation')))(o)
        elif kind == 'Bidirectional':
            r = c['layer']['config']
            o = L.Bidirectional(L.LSTM(
                r['output_dim'], zoneout_h=r.get('zoneout_h', 0.), zoneout_c=r.get('zoneout_c', 0.),
                layer_norm=r.get('layer_norm'), mi=r.get('mi'),
                W_regularizer=reg(r.get('W_regularizer')), U_regularizer=reg(r.get('U_regularizer')),
                dropout_W=r.get('dropout_W', 0.), dropout_U=r.get('dropout_U', 0.),
                activation=r.get('activation', 'tanh'),
                inner_activation=r.get('inner_activation', 'hard_sigmoid')),
                merge_mode=c.get('merge_mode', 'concat'))(o)
        elif kind == 'Merge':
            other = l['inbound_nodes'][0][1][0]
            o = L.merge([o, syms[other]], mode=c['mode'])
        else:
            raise NotImplementedError('model_config: layer class %r' % kind)
        syms[name] = o
    return ctc_model(x, o, **{k: v for k, v in dec.items()
                              if k in ('is_greedy', 'beam_width', 'merge_repeated', 'top_paths')})

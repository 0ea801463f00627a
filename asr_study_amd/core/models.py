"""Model factories with the reference's names and hyper-parameters
(core/models.py): ``ctc_model``, ``graves2006``, ``eyben``, ``brsmv1``.

``train.py`` resolves them by name -- ``get_from_module('core.models', 'brsmv1')
(**hparams)`` (train.py:127-129) -- and gets back an object with the Keras
``compile / fit_generator / evaluate_generator / metrics_names / optimizer.lr``
surface; here that object is core.engine.Model, which runs on the HIP kernels.
``maas`` and ``deep_speech`` reference un-imported Keras names in the reference
(NameError at call time) and are not provided.
"""
from . import ctc_utils
from .engine import Model
from .layers import (Input, GaussianNoise, TimeDistributed, Dense, LSTM, Bidirectional,
                     Dropout, Merge, merge, l2, Reshape, Convolution2D, clipped_relu)


def ctc_model(inputs, output, **kwargs):
    """Given the symbolic input and the (N, T, C) logit tensor of a user-built
    topology, returns the trainable CTC model: inputs ``[inputs, labels,
    inputs_length]``, outputs ``[ctc, decoder]`` (core/models.py:31-52).

    kwargs: ``is_greedy`` / ``beam_width`` / ``merge_repeated`` for the decoder
    (core/ctc_utils.py:8-52), ``device``, ``seed``.
    """
    root, chain = output.chain()
    if root is not inputs:
        raise ValueError('output is not connected to inputs')
    spec = []
    # outputs[i] = the symbolic tensor stage i produces (to resolve merge() skip inputs)
    outputs, cur = [], output
    while cur.producer is not None:
        outputs.append(cur)
        cur = cur.parent
    outputs = outputs[::-1]
    for layer in chain:
        if isinstance(layer, Merge):
            if layer.skip is inputs:
                raise NotImplementedError('merge with the raw model input')
            src = [i for i, t in enumerate(outputs) if t is layer.skip]
            if not src:
                raise ValueError('merge: the skip input is not on the path from inputs')
            spec.append({'type': 'merge', 'mode': layer.mode, 'skip': src[0]})
        elif isinstance(layer, Reshape):
            spec.append({'type': 'reshape', 'target': list(layer.target)})   # a view: no data moves
        elif isinstance(layer, Convolution2D):
            spec.append({'type': 'conv', 'F_in': layer.in_fc[0], 'C_in': layer.in_fc[1],
                         'C_out': layer.nb_filter, 'kt': layer.kt, 'kf': layer.kf,
                         'st': layer.st, 'sf': layer.sf, 'clip': layer.clip, 'l2': layer.l2})
        elif isinstance(layer, GaussianNoise):
            spec.append({'type': 'noise', 'value': layer.sigma})
        elif isinstance(layer, Dropout):
            spec.append({'type': 'dropout', 'value': layer.p})
        elif isinstance(layer, TimeDistributed):
            spec.append({'type': 'dense', 'n_out': layer.dense.output_dim, 'l2': layer.dense.l2})
        elif isinstance(layer, Bidirectional):
            r = layer.lstm
            spec.append({'type': 'bilstm', 'H': r.output_dim, 'dropout_W': r.dropout_W,
                         'dropout_U': r.dropout_U, 'l2_W': r.l2_W, 'l2_U': r.l2_U,
                         'mi': r.mi, 'zoneout_c': r.zoneout_c, 'zoneout_h': r.zoneout_h,
                         'layer_norm': r.layer_norm, 'activation': r.activation})
        else:
            raise NotImplementedError(type(layer).__name__)
    model = Model(spec, inputs.features, device=kwargs.get('device'),
                  seed=kwargs.get('seed', 0))
    model.decoder = ctc_utils.decoder_config(**{k: v for k, v in kwargs.items()
                                                if k in ('is_greedy', 'beam_width',
                                                         'merge_repeated', 'top_paths')})
    return model


def graves2006(num_features=26, num_hiddens=100, num_classes=28, std=.6, **kw):
    """Graves et al. 2006 (core/models.py:55-73)."""
    x = Input(name='inputs', shape=(None, num_features))
    o = x
    o = GaussianNoise(std)(o)
    o = Bidirectional(LSTM(num_hiddens, return_sequences=True, consume_less='gpu'))(o)
    o = TimeDistributed(Dense(num_classes))(o)
    model = ctc_model(x, o, **kw)
    model.config = {'name': 'graves2006', 'kwargs': dict(
        num_features=num_features, num_hiddens=num_hiddens, num_classes=num_classes, std=std)}
    return model


def eyben(num_features=39, num_hiddens=[78, 120, 27], num_classes=28, **kw):
    """Eyben et al. 2009 (core/models.py:76-103)."""
    assert len(num_hiddens) == 3
    x = Input(name='inputs', shape=(None, num_features))
    o = x
    if num_hiddens[0]:
        o = TimeDistributed(Dense(num_hiddens[0]))(o)
    if num_hiddens[1]:
        o = Bidirectional(LSTM(num_hiddens[1], return_sequences=True, consume_less='gpu'))(o)
    if num_hiddens[2]:
        o = Bidirectional(LSTM(num_hiddens[2], return_sequences=True, consume_less='gpu'))(o)
    o = TimeDistributed(Dense(num_classes))(o)
    model = ctc_model(x, o, **kw)
    model.config = {'name': 'eyben', 'kwargs': dict(
        num_features=num_features, num_hiddens=list(num_hiddens), num_classes=num_classes)}
    return model


def brsmv1(num_features=39, num_classes=28, num_hiddens=256, num_layers=5,
           dropout=0.2, zoneout=0., input_dropout=False, input_std_noise=.0,
           weight_decay=1e-4, residual=None, layer_norm=None, mi=None,
           activation='tanh', **kw):
    """BRSM v1.0 (core/models.py:217-281), same defaults."""
    x = Input(name='inputs', shape=(None, num_features))
    o = x
    if input_std_noise is not None:
        o = GaussianNoise(input_std_noise)(o)
    if residual is not None:
        o = TimeDistributed(Dense(num_hiddens * 2, W_regularizer=l2(weight_decay)))(o)
    if input_dropout:
        o = Dropout(dropout)(o)
    for i, _ in enumerate(range(num_layers)):
        new_o = Bidirectional(LSTM(num_hiddens,
                                   return_sequences=True,
                                   W_regularizer=l2(weight_decay),
                                   U_regularizer=l2(weight_decay),
                                   dropout_W=dropout,
                                   dropout_U=dropout,
                                   zoneout_c=zoneout,
                                   zoneout_h=zoneout,
                                   mi=mi,
                                   layer_norm=layer_norm,
                                   activation=activation))(o)
        if residual is not None:
            o = merge([new_o, o], mode=residual)
        else:
            o = new_o
    o = TimeDistributed(Dense(num_classes, W_regularizer=l2(weight_decay)))(o)
    model = ctc_model(x, o, **kw)
    model.config = {'name': 'brsmv1', 'kwargs': dict(
        num_features=num_features, num_classes=num_classes, num_hiddens=num_hiddens,
        num_layers=num_layers, dropout=dropout, zoneout=zoneout, input_dropout=input_dropout,
        input_std_noise=input_std_noise, weight_decay=weight_decay, residual=residual,
        layer_norm=layer_norm, mi=mi, activation=activation)}
    return model


def deep_speech2(num_features=80, num_classes=28, num_hiddens=512, num_layers=5,
                 conv_filters=32, conv_kernels=((11, 41), (11, 21)), conv_strides=((2, 2), (1, 2)),
                 max_value=20, dropout=0.2, weight_decay=1e-4, input_std_noise=.0, **kw):
    """BASELINE.json configs[2]: "DeepSpeech2-style 5xBiLSTM(512) + 2 conv front-end, 80-dim
    log-mel".  NO REFERENCE COUNTERPART: the reference lists Deep Speech 2 as TODO
    (README.md:118) and its ``deep_speech`` factory (core/models.py:148-214) is dead code
    (un-imported Keras names) without convolutions.  Built from the reference's own pieces:
    brsmv1's recurrent stack and regularisers (core/models.py:217-281), the ``clipped_relu``
    of its Deep Speech factories (:116-117), and two Keras Convolution2D layers over (time,
    frequency) with DeepSpeech2's filter shapes -- 32 x (11 x 41) stride (2, 2) and
    32 x (11 x 21) stride (1, 2), 'same' padding -- so the recurrent stack sees T/2 frames of
    F/4 * 32 features.  ``inputs_length`` is mapped to ceil(len / 2) inside the model."""
    x = Input(name='inputs', shape=(None, num_features))
    o = x
    if input_std_noise is not None:
        o = GaussianNoise(input_std_noise)(o)
    o = Reshape((-1, num_features, 1))(o)
    for (kt, kf), (st, sf) in zip(conv_kernels, conv_strides):
        o = Convolution2D(conv_filters, kt, kf, subsample=(st, sf), border_mode='same',
                          activation=clipped_relu(max_value),
                          W_regularizer=l2(weight_decay))(o)
    o = Reshape((-1, o.features))(o)
    for _ in range(num_layers):
        o = Bidirectional(LSTM(num_hiddens, return_sequences=True,
                               W_regularizer=l2(weight_decay), U_regularizer=l2(weight_decay),
                               dropout_W=dropout, dropout_U=dropout))(o)
    o = TimeDistributed(Dense(num_classes, W_regularizer=l2(weight_decay)))(o)
    model = ctc_model(x, o, **kw)
    model.config = {'name': 'deep_speech2', 'kwargs': dict(
        num_features=num_features, num_classes=num_classes, num_hiddens=num_hiddens,
        num_layers=num_layers, conv_filters=conv_filters,
        conv_kernels=[list(k) for k in conv_kernels], conv_strides=[list(k) for k in conv_strides],
        max_value=max_value, dropout=dropout, weight_decay=weight_decay,
        input_std_noise=input_std_noise)}
    return model


"""Layer descriptors with the Keras-1.2.2 names the reference's model factories use.

The reference builds a Keras functional graph (core/models.py:247-281:
Input -> GaussianNoise -> [TimeDistributed(Dense)] -> [Dropout] ->
N x Bidirectional(LSTM) -> TimeDistributed(Dense)) and hands its two ends to
``ctc_model(inputs, output)``.  Here the same calls build a tiny symbolic chain
(no tensors, no math); ``ctc_model`` turns the chain into the list of stages the
HIP engine (core/engine.py) executes.  ``LSTM`` keeps the reference override's
signature (core/layers.py:366-386: zoneout_h/zoneout_c/layer_norm/mi on top of the
Keras LSTM arguments); of the optional variants the residual ``merge``, multiplicative
integration, zoneout and layer normalisation are implemented.
"""


class l2(object):
    """keras.regularizers.l2: adds l * sum(w^2) to the loss."""

    def __init__(self, l=0.01):
        self.l2 = float(l)


class Sym(object):
    """A symbolic (N, T, features) tensor: knows the stage that produced it.  ``fc`` is set
    while the tensor is viewed as an (N, T, F, C) image (between ``Reshape`` and the flattening
    ``Reshape`` of the convolution front-end): features = F * C, channel minor."""

    def __init__(self, features, producer=None, parent=None, name=None, fc=None):
        self.features = features
        self.producer = producer
        self.parent = parent
        self.name = name
        self.fc = fc

    def chain(self):
        out, cur = [], self
        while cur.producer is not None:
            out.append(cur.producer)
            cur = cur.parent
        return cur, out[::-1]


def Input(name=None, shape=None, dtype='float32', sparse=False):
    """keras.layers.Input(name='inputs', shape=(None, num_features))."""
    return Sym(shape[-1] if shape else None, name=name)


class Layer(object):
    def __call__(self, x):
        return Sym(self.out_features(x.features), producer=self, parent=x, fc=self.out_fc(x))

    def out_fc(self, x):
        return x.fc

    def out_features(self, f):
        return f


class GaussianNoise(Layer):
    """Adds N(0, sigma) in the training phase only (identity at sigma = 0, which
    is brsmv1's default: core/models.py:250-251)."""

    def __init__(self, sigma):
        self.sigma = float(sigma or 0.0)


class Dropout(Layer):
    """Plain (per-element) inverted dropout on the layer input, training only."""

    def __init__(self, p):
        self.p = float(p)


class Dense(object):
    def __init__(self, output_dim, W_regularizer=None, activation=None, **kwargs):
        if activation not in (None, 'linear'):
            raise NotImplementedError('Dense activation %r' % (activation,))
        self.output_dim = int(output_dim)
        self.l2 = W_regularizer.l2 if W_regularizer is not None else 0.0


class TimeDistributed(Layer):
    """TimeDistributed(Dense(n)): a row-wise affine map (core/models.py:278-279)."""

    def __init__(self, layer):
        assert isinstance(layer, Dense)
        self.dense = layer

    def out_features(self, f):
        return self.dense.output_dim


class LSTM(object):
    """core/layers.py:356-479 (reference override of keras.layers.LSTM).

    Implemented: consume_less='gpu' fused layout, hard_sigmoid inner activation, the
    ``activation`` of the cell candidate and output (core/layers.py:452, :463: tanh by default;
    relu, sigmoid, hard_sigmoid, linear, softsign, softplus run on the variant kernels),
    variational dropout_W / dropout_U, W/U l2 regularisers,
    multiplicative integration (mi=[alpha, beta1, beta2] inits), zoneout_c / zoneout_h and
    layer_norm=[gain_init, bias_init] (LN of h@U, x@W and the output cell state).
    """

    def __init__(self, output_dim, zoneout_h=0., zoneout_c=0., layer_norm=None, mi=None,
                 return_sequences=True, consume_less='gpu', W_regularizer=None,
                 U_regularizer=None, dropout_W=0., dropout_U=0., activation='tanh',
                 inner_activation='hard_sigmoid', **kwargs):
        if layer_norm is not None and len(layer_norm) != 2:
            raise ValueError('layer_norm = [gain_init, bias_init]')
        self.layer_norm = None if layer_norm is None else [float(v) for v in layer_norm]
        if mi is not None and len(mi) != 3:
            raise ValueError('mi = [alpha_init, beta1_init, beta2_init]')
        self.mi = None if mi is None else [float(v) for v in mi]
        self.zoneout_h = float(zoneout_h or 0.0)
        self.zoneout_c = float(zoneout_c or 0.0)
        if inner_activation != 'hard_sigmoid':
            raise NotImplementedError('inner_activation: only hard_sigmoid is implemented')
        if activation not in ('tanh', 'relu', 'sigmoid', 'hard_sigmoid', 'linear', 'softsign',
                              'softplus'):
            raise NotImplementedError('LSTM activation %r (implemented: the Keras-1.2.2 names '
                                      'tanh, relu, sigmoid, hard_sigmoid, linear, softsign, '
                                      'softplus)' % (activation,))
        self.activation = activation
        if not return_sequences:
            raise NotImplementedError('return_sequences=False')
        self.output_dim = int(output_dim)
        self.dropout_W = float(dropout_W or 0.0)
        self.dropout_U = float(dropout_U or 0.0)
        self.l2_W = W_regularizer.l2 if W_regularizer is not None else 0.0
        self.l2_U = U_regularizer.l2 if U_regularizer is not None else 0.0


class Bidirectional(Layer):
    """keras.layers.Bidirectional(merge_mode='concat')."""

    def __init__(self, layer, merge_mode='concat'):
        assert isinstance(layer, LSTM)
        if merge_mode != 'concat':
            raise NotImplementedError('merge_mode=%r' % merge_mode)
        self.lstm = layer

    def out_features(self, f):
        return 2 * self.lstm.output_dim


class Merge(Layer):
    """keras.layers.merge([a, b], mode): element-wise 'sum' or 'ave' of two tensors of
    the same width (brsmv1's residual connection, core/models.py:273-276)."""

    def __init__(self, mode, skip):
        if mode not in ('sum', 'ave'):
            raise NotImplementedError('merge mode %r (implemented: sum, ave)' % (mode,))
        self.mode = mode
        self.skip = skip


def merge(inputs, mode=None):
    a, b = inputs
    if a.features != b.features:
        raise ValueError('merge: widths differ (%s vs %s)' % (a.features, b.features))
    return Merge(mode, b)(a)



def clipped_relu(max_value=20.0):
    """The reference's activation of its (dead) Deep Speech factories: ``relu(x,
    max_value=max_value)`` = min(max(x, 0), max_value) (core/models.py:116-117)."""
    return ('clipped_relu', float(max_value))


class Reshape(Layer):
    """keras.layers.Reshape on the feature axes only: ``Reshape((-1, F, C))`` views the
    (N, T, F*C) features as an (N, T, F, C) image for Convolution2D, ``Reshape((-1, F*C))``
    flattens it again.  Channel-minor memory order either way: no data moves."""

    def __init__(self, target_shape):
        self.target = tuple(int(v) for v in target_shape)
        if len(self.target) not in (2, 3) or self.target[0] != -1:
            raise NotImplementedError('Reshape(%r): only (-1, F, C) and (-1, F*C)' % (target_shape,))

    def out_features(self, f):
        n = int(np_prod(self.target[1:]))
        if n != f:
            raise ValueError('Reshape: %d features into %r' % (f, self.target))
        return f

    def out_fc(self, x):
        return (self.target[1], self.target[2]) if len(self.target) == 3 else None


def np_prod(t):
    out = 1
    for v in t:
        out *= int(v)
    return out


class Convolution2D(Layer):
    """keras.layers.Convolution2D(nb_filter, nb_row, nb_col, subsample=(st, sf),
    border_mode='same', dim_ordering='tf', activation=clipped_relu(...)) over (time, frequency)
    of an (N, T, F, C) tensor.  NO REFERENCE COUNTERPART (README.md:118 lists Deep Speech 2 as
    TODO): the layer of BASELINE.json configs[2]'s "2 conv front-end", defined in
    include/asr_hip.h (K13) and oracle/conv.py."""

    def __init__(self, nb_filter, nb_row, nb_col, subsample=(1, 1), border_mode='same',
                 activation=None, W_regularizer=None, dim_ordering='tf', **kwargs):
        if border_mode != 'same' or dim_ordering != 'tf':
            raise NotImplementedError("Convolution2D: border_mode='same', dim_ordering='tf' only")
        if activation is None or activation == 'linear':
            self.clip = 0.0
        elif isinstance(activation, tuple) and activation[0] == 'clipped_relu':
            self.clip = float(activation[1])
        else:
            raise NotImplementedError('Convolution2D activation %r' % (activation,))
        self.nb_filter, self.kt, self.kf = int(nb_filter), int(nb_row), int(nb_col)
        self.st, self.sf = int(subsample[0]), int(subsample[1])
        self.l2 = W_regularizer.l2 if W_regularizer is not None else 0.0

    def out_fc(self, x):
        if x.fc is None:
            raise ValueError('Convolution2D needs an (N, T, F, C) input: Reshape((-1, F, C)) first')
        return (-(-x.fc[0] // self.sf), self.nb_filter)

    def __call__(self, x):
        fc = self.out_fc(x)
        self.in_fc = x.fc
        return Sym(fc[0] * fc[1], producer=self, parent=x, fc=fc)


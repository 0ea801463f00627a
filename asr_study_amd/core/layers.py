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
    """A symbolic (N, T, features) tensor: knows the stage that produced it."""

    def __init__(self, features, producer=None, parent=None, name=None):
        self.features = features
        self.producer = producer
        self.parent = parent
        self.name = name

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
        return Sym(self.out_features(x.features), producer=self, parent=x)

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

    Implemented: consume_less='gpu' fused layout, hard_sigmoid inner activation,
    tanh activation, variational dropout_W / dropout_U, W/U l2 regularisers,
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
        if activation != 'tanh' or inner_activation != 'hard_sigmoid':
            raise NotImplementedError('only tanh / hard_sigmoid are implemented')
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

"""Checkpoint callback with the reference's surface (core/callbacks.py):
``MetaCheckpoint(filepath, training_args=args, meta=meta)`` writes, at every epoch
end, the model weights plus a ``meta`` group {``training_args`` yaml attribute,
``epochs``, one dataset per logged metric} (core/callbacks.py:36-56).  Note the
reference's constructor ignores monitor/save_best_only (:20-23), so ``best.h5`` is
simply the latest epoch; kept as is.

File layout (HDF5 through h5lite): ``/model_weights`` follows Keras 1.2.2's
``save_weights_to_hdf5_group`` (attribute ``layer_names``; one group per weight-bearing
layer with attribute ``weight_names`` and one N-D float32 dataset per weight, in the
order and gate layout of ``Model.get_weights()``: ``bidirectional_k`` = forward W, U, b
then backward W, U, b; ``timedistributed_k`` = W, b), so ``model.load_weights(file)`` of
the reference's Keras model reads it and utils/core_utils.load_model reads files the
reference wrote.  Extra attributes ``model_config`` (yaml: factory name + kwargs) let
this package rebuild the topology without Keras' JSON graph; /optimizer/...; /meta/...
as core/callbacks.py:47-56.  The root attributes ``model_config`` / ``training_config`` and
the ``optimizer_weights`` group of ``keras.models.save_model`` are emitted too
(utils/keras_config.py: schema restated from memory of keras 1.2.2, with the documented
residue that a Lambda's function is named, not marshalled)."""
import numpy as np
import yaml

from ..datasets import h5lite


class Callback(object):
    def set_model(self, model):
        self.model = model

    def on_train_begin(self):
        pass

    def on_epoch_end(self, epoch, logs):
        pass

    def on_train_end(self):
        pass


LSTM_PARTS = ('W', 'U', 'b')
MI_PARTS = ('mi_alpha', 'mi_beta1', 'mi_beta2')
# the reference iterates a dict literal {'Uh', 'Wx', 'new_c'} (core/layers.py:409): its
# Python-2 order is unspecified, so files are written in this order and READ BY NAME
LN_PARTS = ('ln_gain_Uh', 'ln_bias_Uh', 'ln_gain_Wx', 'ln_bias_Wx', 'ln_gain_new_c',
            'ln_bias_new_c')


def lstm_weight_parts(stage):
    """Per-direction weight suffixes of a BiLSTM stage in get_weights() order: Keras' W, U, b
    (base LSTM.build), then the reference's add_weight names (core/layers.py:388-422)."""
    parts = list(LSTM_PARTS)
    if getattr(stage, 'mi', None) is not None:
        parts += list(MI_PARTS)
    if getattr(stage, 'ln', None) is not None:
        parts += list(LN_PARTS)
    return parts


def keras_layers(model, weights):
    """[(layer name, [(weight name, array), ...])] in Keras-1.2.2 naming for the model's
    weight-bearing stages (get_weights() order); every array of ``weights`` is consumed."""
    it = iter(weights)
    out, nb, nd, nc = [], 0, 0, 0
    for s in model.stages:
        if s.kind == 'conv':        # keras.layers.Convolution2D: '<name>_W', '<name>_b'
            nc += 1
            out.append(('convolution2d_%d' % nc, [('convolution2d_%d_W:0' % nc, next(it)),
                                                  ('convolution2d_%d_b:0' % nc, next(it))]))
        elif s.kind == 'bilstm':
            nb += 1
            ws = []
            for d in ('forward', 'backward'):
                for part in lstm_weight_parts(s):
                    ws.append(('%s_lstm_%d_%s:0' % (d, nb, part), next(it)))
            out.append(('bidirectional_%d' % nb, ws))
        elif s.kind == 'dense':
            nd += 1
            out.append(('timedistributed_%d' % nd, [('dense_%d_W:0' % nd, next(it)),
                                                    ('dense_%d_b:0' % nd, next(it))]))
    rest = sum(1 for _ in it)
    if rest:
        raise ValueError('keras_layers: %d weight arrays left over (get_weights() and the '
                         'stage list disagree)' % rest)
    return out


def order_layer_weights(names, arrays):
    """Arrays of one Keras layer group re-ordered into this package's get_weights() order,
    by NAME: tolerant of the TF backend's ':0' suffix and of any order of the reference's
    mi / layer-norm weights inside the group.  Unknown names keep their file order."""
    def key(name):
        n = name[:-2] if name.endswith(':0') else name
        d = 0 if n.startswith('forward_') else 1 if n.startswith('backward_') else 0
        for rank, part in enumerate(LSTM_PARTS + MI_PARTS + LN_PARTS + ('W', 'b')):
            if n.endswith('_' + part):
                return (d, rank)
        return None
    keys = [key(n) for n in names]
    if any(k is None for k in keys) or len(set(keys)) != len(keys):
        return list(arrays)
    return [a for _, a in sorted(zip(keys, arrays), key=lambda ka: ka[0])]


def save_model(model, filepath, meta=None, model_config=None):
    weights = model.get_weights()
    with h5lite.File(filepath, 'w') as f:
        f.attrs['keras_version'] = '1.2.2'
        g = f.create_group('model_weights')
        layers = keras_layers(model, weights)
        g.attrs.set_strings('layer_names', [name for name, _ in layers])
        for name, ws in layers:
            lg = g.create_group(name)
            lg.attrs.set_strings('weight_names', [w for w, _ in ws])
            for wname, val in ws:
                lg.write_array(wname, val)
        g.attrs['model_config'] = yaml.safe_dump(model_config or getattr(model, 'config', {}))
        # what keras.models.save_model writes besides the weights (utils/keras_config.py):
        # the functional graph, the training configuration and the optimizer slots
        if hasattr(model, 'stages') and hasattr(model, 'num_features'):
            from ..utils import keras_config
            f.attrs['model_config'] = keras_config.model_config(model)
            tc = keras_config.training_config(model)
            if tc is not None:
                f.attrs['training_config'] = tc
                ow = keras_config.optimizer_weights(model)
                if ow:
                    og = f.create_group('optimizer_weights')
                    og.attrs.set_strings('weight_names', [n for n, _ in ow])
                    for n, a in ow:
                        og.write_array(n, a)
        if model.optimizer is not None:
            o = f.create_group('optimizer')
            state, it = model.optimizer.get_state()
            o.write_vlen_float('state', [s.reshape(-1) for s in state])
            o.attrs['iterations'] = int(it)
            o.attrs['config'] = yaml.safe_dump(
                {'class': type(model.optimizer).__name__, 'lr': model.optimizer.lr,
                 'clipnorm': model.optimizer.clipnorm,
                 'momentum': getattr(model.optimizer, 'momentum', None)})
        if meta is not None:
            m = f.create_group('meta')
            m.attrs['training_args'] = yaml.safe_dump(meta.get('training_args', {}))
            for k, v in meta.items():
                if k != 'training_args':
                    m.write_float(k, np.asarray(v, np.float32))


class MetaCheckpoint(Callback):
    def __init__(self, filepath, monitor='val_loss', save_best_only=False, mode='auto',
                 training_args=None, meta=None, **kwargs):
        self.filepath = filepath
        self.meta = meta or {'epochs': [], 'training_args': {}}
        if training_args is not None:
            args = training_args
            if hasattr(args, 'values') and callable(args.values):
                args = args.values()
            elif not isinstance(args, dict):
                args = vars(args)
            self.meta['training_args'] = {k: v for k, v in args.items()}

    def on_epoch_end(self, epoch, logs=None):
        self.meta.setdefault('epochs', []).append(epoch)
        for k, v in (logs or {}).items():
            self.meta.setdefault(k, []).append(v)
        # core/callbacks.py:45: filepath.format(epoch=epoch, **logs)
        save_model(self.model, self.filepath.format(epoch=epoch, **(logs or {})), self.meta)


class ReduceLROnPlateau(Callback):
    """keras.callbacks.ReduceLROnPlateau (Keras 1.2.2 semantics, restated): when ``monitor``
    has not improved by more than ``epsilon`` for ``patience`` epochs, multiply
    ``model.optimizer.lr`` by ``factor`` (not below ``min_lr``), then wait ``cooldown`` epochs.
    Reached through ``--lr_schedule ReduceLROnPlateau --lr_params ...`` (train.py:165-172,
    which resolves the class by name in keras.callbacks)."""

    def __init__(self, monitor='val_loss', factor=0.1, patience=10, verbose=0, mode='auto',
                 epsilon=1e-4, cooldown=0, min_lr=0, **kwargs):
        if factor >= 1.0:
            raise ValueError('ReduceLROnPlateau does not support a factor >= 1.0.')
        self.monitor, self.factor, self.patience = monitor, float(factor), int(patience)
        self.verbose, self.epsilon, self.cooldown = verbose, float(epsilon), int(cooldown)
        self.min_lr = float(min_lr)
        if mode not in ('auto', 'min', 'max'):
            mode = 'auto'
        self.maximise = mode == 'max' or (mode == 'auto' and 'acc' in monitor)
        self.on_train_begin()

    def on_train_begin(self):
        self.best = -np.inf if self.maximise else np.inf
        self.wait, self.cooldown_counter = 0, 0

    def _better(self, cur):
        return cur > self.best + self.epsilon if self.maximise else cur < self.best - self.epsilon

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs['lr'] = float(self.model.optimizer.lr)
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self._better(cur):
            self.best, self.wait = cur, 0
        elif self.cooldown_counter <= 0:
            if self.wait >= self.patience:
                old = float(self.model.optimizer.lr)
                if old > self.min_lr + 1e-4 * self.min_lr:
                    self.model.optimizer.lr = max(old * self.factor, self.min_lr)
                    if self.verbose:
                        print('Epoch %05d: reducing learning rate to %s.'
                              % (epoch, self.model.optimizer.lr))
                    self.cooldown_counter = self.cooldown
                    self.wait = 0
            self.wait += 1


class LearningRateScheduler(Callback):
    """keras.callbacks.LearningRateScheduler: ``schedule(epoch) -> lr`` at every epoch begin
    (applied at the end of the previous epoch here: the engine calls callbacks at epoch end)."""

    def __init__(self, schedule):
        self.schedule = schedule

    def on_train_begin(self):
        if getattr(self, 'model', None) is not None and self.model.optimizer is not None:
            self.model.optimizer.lr = float(self.schedule(0))

    def on_epoch_end(self, epoch, logs=None):
        self.model.optimizer.lr = float(self.schedule(epoch + 1))


class EarlyStopping(Callback):
    """keras.callbacks.EarlyStopping (monitor / min_delta / patience / mode)."""

    def __init__(self, monitor='val_loss', min_delta=0, patience=0, verbose=0, mode='auto'):
        self.monitor, self.min_delta, self.patience = monitor, float(min_delta), int(patience)
        self.maximise = mode == 'max' or (mode == 'auto' and 'acc' in monitor)
        self.on_train_begin()

    def on_train_begin(self):
        self.best = -np.inf if self.maximise else np.inf
        self.wait = 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        better = cur - self.min_delta > self.best if self.maximise else cur + self.min_delta < self.best
        if better:
            self.best, self.wait = cur, 0
        else:
            if self.wait >= self.patience:
                self.model.stop_training = True
            self.wait += 1

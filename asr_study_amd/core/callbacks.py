"""Checkpoint callback with the reference's surface (core/callbacks.py):
``MetaCheckpoint(filepath, training_args=args, meta=meta)`` writes, at every epoch
end, the model weights plus a ``meta`` group {``training_args`` yaml attribute,
``epochs``, one dataset per logged metric} (core/callbacks.py:36-56).  Note the
reference's constructor ignores monitor/save_best_only (:20-23), so ``best.h5`` is
simply the latest epoch; kept as is.

File layout (HDF5 through h5lite): /model_weights/w%03d (flattened float32, Keras
weight order, see Model.get_weights), attribute ``model_config`` (yaml: factory name
+ kwargs + shapes); /optimizer/s%d; /meta/...  Byte compatibility with Keras-1.2.2
``model.save`` files is a later-round row."""
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


def save_model(model, filepath, meta=None, model_config=None):
    weights = model.get_weights()
    with h5lite.File(filepath, 'w') as f:
        g = f.create_group('model_weights')
        g.write_vlen_float('weights', [w.reshape(-1) for w in weights])
        g.attrs['shapes'] = yaml.safe_dump([list(w.shape) for w in weights])
        g.attrs['model_config'] = yaml.safe_dump(model_config or getattr(model, 'config', {}))
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
        save_model(self.model, self.filepath, self.meta)

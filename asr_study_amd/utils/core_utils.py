"""Model (re)loading with the reference's surface (utils/core_utils.py).

``load_model(fname, return_meta, mode)``: mode 'train' rebuilds the model as
core.models defines it; mode 'eval' swaps the greedy decoder for beam search with
the reference's defaults (is_greedy=False, beam_width=400: utils/core_utils.py
:67-72) and renames the metric to beam_search_ler; ``setup_gpu`` selects the visible
device(s) (:22-35)."""
import os

import numpy as np
import yaml

from ..core.callbacks import order_layer_weights
from ..datasets import h5lite
from . import generic_utils as utils
from .hparams import HParams


def setup_gpu(gpu, allow_growth=False, log_device_placement=False):
    """--gpu '0' / '1,2' / 'all' / '-1' (utils/core_utils.py:22-35)."""
    if gpu == '-1':
        raise RuntimeError('this build has no CPU path: the hot path runs on the MI355X only')
    if gpu and gpu != 'all':
        os.environ.setdefault('HIP_VISIBLE_DEVICES', gpu)


def load_meta(model_fname):
    meta = {}
    with h5lite.File(model_fname, 'r') as f:
        g = f['meta']
        meta['training_args'] = yaml.safe_load(g.attrs['training_args'])
        for k in g.keys():
            meta[k] = list(np.asarray(g[k][:]))
    return meta


def load_model(model_fname, return_meta=False, mode='train', **kwargs):
    if mode not in ('train', 'predict', 'eval'):
        raise ValueError('mode must be one of (train, predict, eval)')
    from ..core import optimizers
    with h5lite.File(model_fname, 'r') as f:
        g = f['model_weights']
        cfg = yaml.safe_load(g.attrs['model_config']) if 'model_config' in g.attrs else None
        if 'layer_names' in g.attrs:       # Keras-1.2.2 weight layout (ours or the reference's)
            weights = []
            for lname in g.attrs.get_strings('layer_names'):
                lg = g[lname]
                names = lg.attrs.get_strings('weight_names') if 'weight_names' in lg.attrs else []
                weights += order_layer_weights(names, [lg[w].read_array() for w in names])
        else:                              # round-1 development format
            shapes = yaml.safe_load(g.attrs['shapes'])
            weights = [np.asarray(a, np.float32).reshape(s)
                       for a, s in zip(g['weights'][:], shapes)]
        opt_state = None
        if 'optimizer' in f:
            o = f['optimizer']
            opt_state = (yaml.safe_load(o.attrs['config']), o['state'][:],
                         int(o.attrs['iterations']))
        keras_json = f.attrs['model_config'] if 'model_config' in f.attrs else None
        if cfg is None and 'meta' not in f and keras_json is not None:
            # a bare keras model.save() file: rebuild the topology from its functional graph
            cfg = {'keras_json': keras_json}
        if cfg is None:
            # a file written by the reference: the topology is named in meta/training_args
            # (train.py:127-129), input / output widths are read off the weights
            targs = yaml.safe_load(f['meta'].attrs['training_args'])
            # (own name: ``kwargs`` are the caller's decoder options, read again below)
            mkw = HParams().parse(list(targs.get('model_params') or [])).values()
            mkw.setdefault('num_features', int(weights[0].shape[0]))
            mkw.setdefault('num_classes', int(weights[-1].shape[0]))
            cfg = {'name': targs['model'], 'kwargs': mkw}
    if 'keras_json' in cfg:
        from .keras_config import topology_from_config
        model = topology_from_config(cfg['keras_json'])
        model.config = {}
    else:
        factory = utils.get_from_module('core.models', cfg['name'])
        model = factory(**cfg.get('kwargs', {}))
        model.config = cfg
    model.set_weights(weights)
    if mode == 'train' and opt_state is not None:
        oc, state, it = opt_state
        if oc['class'] == 'Adam':
            opt = optimizers.Adam(lr=oc['lr'], clipnorm=oc['clipnorm'])
        else:
            opt = optimizers.SGD(lr=oc['lr'], momentum=oc['momentum'] or 0.0,
                                 clipnorm=oc['clipnorm'])
        model.compile(optimizer=opt)
        opt.set_state([np.asarray(s, np.float32) for s in state], it)
        # the noise streams are keyed by the optimisation step: a resumed run continues the
        # sequence instead of replaying it from step 0
        model._step = int(it)
    if mode in ('eval', 'predict'):
        if kwargs.get('decoder', True):
            model.decoder = dict(is_greedy=kwargs.get('is_greedy', False),
                                 beam_width=kwargs.get('beam_width', 400),
                                 merge_repeated=True)
        else:                       # predict.py --no_decoder: the network output itself
            model.decoder = None
        model.metrics_names = ['loss', 'ctc_loss', 'beam_search_loss', 'beam_search_ler']
    if return_meta:
        return model, load_meta(model_fname)
    return model

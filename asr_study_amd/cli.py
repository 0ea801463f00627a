"""Command-line front ends: the flag sets of the reference's ``train.py`` (:46-88),
``eval.py`` (:26-49) and ``extras/make_dataset.py`` (:31-48) over this package.

Flags are declared in tables so the three tools share one implementation of
"resolve a plugin by name" (model factory, feature extractor, label parser,
dataset parser) and of "merge stored / explicit arguments" on resume and eval."""
import argparse
import datetime
import logging
import os

from .utils import generic_utils as utils
from .utils.hparams import HParams

_PLUGIN_FLAGS = [
    ('--input_parser', dict(type=str, default=None)),
    ('--input_parser_params', dict(nargs='+', default=[])),
    ('--label_parser', dict(type=str, default='simple_char_parser')),
    ('--label_parser_params', dict(nargs='+', default=[])),
]
_DEVICE_FLAGS = [
    ('--gpu', dict(default='0', type=str)),
    ('--allow_growth', dict(default=False, action='store_true')),
]
TRAIN_FLAGS = [
    ('--load', dict(default=None, type=str)),
    ('--model', dict(default='brsmv1', type=str)),
    ('--model_params', dict(nargs='+', default=[])),
    ('--num_epochs', dict(default=100, type=int)),
    ('--lr', dict(default=0.001, type=float)),
    ('--momentum', dict(default=0.9, type=float)),
    ('--clipnorm', dict(default=400, type=float)),
    ('--batch_size', dict(default=32, type=int)),
    ('--opt', dict(default='adam', type=str, choices=['sgd', 'adam'])),
    ('--dataset', dict(default=None, type=str, nargs='+')),
] + _PLUGIN_FLAGS + [
    ('--lr_schedule', dict(default=None)),
    ('--lr_params', dict(nargs='+', default=[])),
    ('--save', dict(default=None, type=str)),
] + _DEVICE_FLAGS + [
    ('--verbose', dict(default=0, type=int)),
    ('--seed', dict(default=None, type=float)),
]
EVAL_FLAGS = [
    ('--model', dict(required=True, type=str)),
    ('--dataset', dict(required=True, type=str)),
    ('--subset', dict(type=str, default='test')),
    ('--batch_size', dict(default=32, type=int)),
] + _PLUGIN_FLAGS + _DEVICE_FLAGS + [
    ('--save_transcriptions', dict(default=None, type=str)),
    ('--beam_width', dict(default=400, type=int)),      # utils/core_utils.py:70-71
]
PREDICT_FLAGS = [
    ('--model', dict(required=True, type=str)),
    ('--dataset', dict(default=None, type=str)),
    ('--file', dict(default=None, type=str)),
    ('--subset', dict(type=str, default='test')),
] + _PLUGIN_FLAGS + [
    ('--no_decoder', dict(action='store_true', default=False)),
] + _DEVICE_FLAGS + [
    ('--save', dict(default=None, type=str)),
    ('--override', dict(default=False, action='store_true')),
    ('--beam_width', dict(default=400, type=int)),
]
MAKE_DATASET_FLAGS = [
    ('--parser', dict(type=str, default='dummy')),
    ('--parser_params', dict(nargs='+', default=[])),
    ('--output_file', dict(type=str, default=None)),
    ('--override', dict(action='store_true')),
] + _PLUGIN_FLAGS


def make_parser(description, table):
    ap = argparse.ArgumentParser(description=description)
    for flag, kw in table:
        ap.add_argument(flag, **kw)
    return ap


def resolve_plugins(args):
    """(feature extractor or None, label parser) named on the command line."""
    feature = utils.get_from_module('preprocessing.audio', args.input_parser,
                                    params=args.input_parser_params)
    labels = utils.get_from_module('preprocessing.text', args.label_parser,
                                   params=args.label_parser_params)
    return feature, labels


def merged_args(parsed, stored, explicit):
    """defaults < arguments stored in the checkpoint < arguments given explicitly.
    (The reference keeps only the last two, dropping un-stored defaults such as
    ``--subset``: eval.py:60.)"""
    return HParams(**vars(parsed)).update(stored).update(vars(explicit))


# ------------------------------------------------------------------ train
def _open_flows(args, data_gen):
    paths = args.dataset
    if len(paths) == 1:
        train, valid, test = data_gen.flow_from_fname(paths[0], datasets=['train', 'valid', 'test'])
        return train, valid, test
    train = data_gen.flow_from_fname(paths[0])
    valid = data_gen.flow_from_fname(paths[1])
    test = data_gen.flow_from_fname(paths[2]) if len(paths) == 3 else None
    return train, valid, test


def _new_model(args):
    from .core import optimizers
    factory = utils.get_from_module('core.models', args.model)
    model = factory(**(HParams().parse(args.model_params).values()))
    if args.opt.strip().lower() == 'sgd':
        opt = optimizers.SGD(lr=args.lr, momentum=args.momentum, clipnorm=args.clipnorm)
    else:
        opt = optimizers.Adam(lr=args.lr, clipnorm=args.clipnorm)
    # loss / metrics / loss_weights are accepted for signature parity (train.py:140-143)
    model.compile(loss={'ctc': 'ctc_dummy_loss', 'decoder': 'decoder_dummy_loss'}, optimizer=opt,
                  metrics={'decoder': 'ler'}, loss_weights=[1, 0])
    return model


def train_main(argv=None):
    parser = make_parser('Training an ASR system.', TRAIN_FLAGS)
    args = parser.parse_args(argv)
    utils.setup_logging()
    log = logging.getLogger('train')
    from . import parallel
    from .core.callbacks import MetaCheckpoint
    from .datasets.dataset_generator import DatasetGenerator
    from .utils.core_utils import setup_gpu, load_model
    rank, world = parallel.init_from_env()
    if world == 1:
        setup_gpu(args.gpu, args.allow_growth, log_device_placement=args.verbose > 1)

    first_epoch, meta = 0, None
    if args.load:
        explicit = utils.parse_nondefault_args(args, parser.parse_args([]), argv)
        model, meta = load_model(args.load, return_meta=True)
        args = merged_args(args, meta['training_args'], explicit)
        first_epoch = len(meta['epochs'])
        if explicit.lr:
            model.optimizer.lr = args.lr
    else:
        model = _new_model(args)
    if world > 1:
        parallel.broadcast_parameters(model)
    lr_callbacks = []
    if args.lr_schedule:
        # train.py:165-172: the class is looked up by name among the Keras callbacks
        from .core import callbacks as _cb
        fn = {k.lower(): getattr(_cb, k) for k in ('ReduceLROnPlateau', 'EarlyStopping')} \
            .get(str(args.lr_schedule).lower())
        if fn is None:
            raise ValueError('Learning rate schedule unrecognized')
        lr_callbacks.append(fn(**HParams().parse(args.lr_params).values()))

    out_dir = args.save or os.path.join('results', '%s_%s' % (args.model, datetime.datetime.now()))
    callbacks = []
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        callbacks = [MetaCheckpoint(os.path.join(out_dir, name), training_args=args, meta=meta)
                     for name in ('model.h5', 'best.h5')]
    callbacks = callbacks + lr_callbacks          # (every rank adjusts its own optimizer.lr)

    feature, labels = resolve_plugins(args)
    gen = DatasetGenerator(feature, labels, batch_size=args.batch_size, seed=args.seed)
    train_flow, valid_flow, test_flow = _open_flows(args, gen)
    if world > 1:       # every rank draws the same global batch and keeps its shard
        train_flow = parallel.ShardedFlow(train_flow, rank, world)
    if rank == 0:
        print(str(args.values() if isinstance(args, HParams) else vars(args)))
    model.fit_generator(train_flow, samples_per_epoch=train_flow.len, nb_epoch=args.num_epochs,
                        validation_data=valid_flow, nb_val_samples=valid_flow.len, max_q_size=10,
                        nb_worker=1, callbacks=callbacks, verbose=int(rank == 0),
                        initial_epoch=first_epoch)
    if test_flow is not None and rank == 0:
        best = load_model(os.path.join(out_dir, 'best.h5'), mode='eval')
        m = best.evaluate_generator(test_flow, test_flow.len, max_q_size=10, nb_worker=1)
        report = 'Total loss: %.4f\nCTC Loss: %.4f\nLER: %.2f%%' % (m[0], m[1], m[3] * 100)
        log.info(report)
        with open(os.path.join(out_dir, 'results.txt'), 'w') as f:
            f.write(report)
        print(report)
    parallel.finalize()


# ------------------------------------------------------------------ eval
def eval_main(argv=None):
    parser = make_parser('Evaluating an ASR system.', EVAL_FLAGS)
    args = parser.parse_args(argv)
    explicit = utils.parse_nondefault_args(
        args, parser.parse_args(['--model', args.model, '--dataset', args.dataset]), argv)
    from .datasets.dataset_generator import DatasetGenerator
    from .utils.core_utils import setup_gpu, load_model
    setup_gpu(args.gpu, args.allow_growth)
    model, meta = load_model(args.model, return_meta=True, mode='eval', beam_width=args.beam_width)
    args = merged_args(args, meta['training_args'], explicit)
    feature, labels = resolve_plugins(args)
    flow = DatasetGenerator(feature, labels, batch_size=args.batch_size, seed=0) \
        .flow_from_fname(args.dataset, datasets=args.subset)
    values = model.evaluate_generator(flow, flow.len, max_q_size=10, nb_worker=1)
    for name, v in zip(model.metrics_names, values):
        print('%s: %4f' % (name, v))
    return values


# ------------------------------------------------------------------ predict
def predict_main(argv=None):
    """predict.py: transcribe one audio file or a dataset split, one utterance per
    forward pass (batch_size=1, the reference's latency path); ``--no_decoder`` saves the
    per-frame network outputs instead (HDF5: predictions vlen-f32 + num_labels, labels)."""
    import numpy as np
    parser = make_parser('Predicting with an ASR system.', PREDICT_FLAGS)
    args = parser.parse_args(argv)
    if args.dataset is None and args.file is None:
        raise ValueError('dataset or file args must be set.')
    if args.dataset and args.file:
        print('Both dataset and file args was set. Ignoring file args.')
    explicit = utils.parse_nondefault_args(args, parser.parse_args(['--model', args.model]), argv)
    from .datasets.dataset_generator import DatasetGenerator, DatasetIterator
    from .utils.core_utils import setup_gpu, load_model
    setup_gpu(args.gpu, args.allow_growth)
    model, meta = load_model(args.model, return_meta=True, mode='predict',
                             decoder=(not args.no_decoder), beam_width=args.beam_width)
    # only the feature / label plugins are inherited from the training run (the reference
    # overlays every stored argument, predict.py:60, which would also import its --save)
    stored = {k: v for k, v in meta['training_args'].items()
              if k in ('input_parser', 'input_parser_params', 'label_parser',
                       'label_parser_params')}
    args = merged_args(args, stored, explicit)
    feature, labels = resolve_plugins(args)
    if args.dataset is not None:
        flow = DatasetGenerator(feature, labels, batch_size=1, seed=0, mode='predict',
                                shuffle=False).flow_from_fname(args.dataset, datasets=args.subset)
    else:
        flow = DatasetIterator(np.array([args.file]), None, input_parser=feature,
                               label_parser=labels, mode='predict', shuffle=False, batch_size=1)
        flow.labels = np.array([u''])
    truth = list(flow.labels) if flow.labels is not None else [u''] * flow.len
    results = []
    for index in range(flow.len):
        out = model.predict(flow.next())
        best = out[0] if args.no_decoder else labels.imap(out[0])
        results.append({'label': truth[index], 'best': best})
        print('Ground Truth: %s' % labels._sanitize(truth[index]))
        print('   Predicted: %s\n\n' % (best if not args.no_decoder else 'array%s' % (best.shape,)))
    if args.save is not None:
        if os.path.exists(args.save):
            if not args.override:
                raise IOError('Unable to create file')
            os.remove(args.save)
        if not args.no_decoder:
            raise ValueError('save param must be set if no_decoder is True')
        from .datasets import h5lite
        with h5lite.File(args.save, 'w') as f:
            f.write_vlen_float('predictions', [r['best'].reshape(-1).astype('float32')
                                               for r in results],
                               attrs={'num_labels': int(results[0]['best'].shape[-1])})
            f.write_strings('labels', [str(r['label']) for r in results])
    return results


# ------------------------------------------------------------------ make_dataset
def make_dataset_main(argv=None):
    """extras/make_dataset.py: crawl a corpus with a DatasetParser and write the HDF5
    the H5Iterator reads; features are computed on the GPU a chunk at a time."""
    parser = make_parser('Generates a preprocessed dataset (hdf5 file).', MAKE_DATASET_FLAGS)
    args = parser.parse_args(argv)
    corpus = utils.get_from_module('datasets*', args.parser, params=args.parser_params,
                                   regex=True)
    feature, labels = resolve_plugins(args)
    fmt = 'npz' if (args.output_file or '').endswith('.npz') else 'h5'
    out = corpus.to_h5(fname=args.output_file, input_parser=feature, label_parser=labels,
                       override=args.override, fmt=fmt)
    print('dataset written to', out)
    return out

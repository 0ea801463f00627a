#!/usr/bin/env python
"""Training CLI with the reference's flags (train.py:46-88) on the MI355X hot path.

    python train.py --dataset data.h5 --model brsmv1 --model_params num_hiddens 256 \
        --input_parser mfcc --num_epochs 10 --save results/run1

Launch one process per GPU for data parallelism (RCCL gradient all-reduce):
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...
"""
from __future__ import absolute_import, division, print_function

import argparse
import datetime
import logging
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from asr_study_amd.utils import generic_utils as utils          # noqa: E402
from asr_study_amd.utils.hparams import HParams                 # noqa: E402


def build_parser():
    parser = argparse.ArgumentParser(description='Training an ASR system.')
    parser.add_argument('--load', default=None, type=str)                 # resume
    parser.add_argument('--model', default='brsmv1', type=str)
    parser.add_argument('--model_params', nargs='+', default=[])
    parser.add_argument('--num_epochs', default=100, type=int)
    parser.add_argument('--lr', default=0.001, type=float)
    parser.add_argument('--momentum', default=0.9, type=float)
    parser.add_argument('--clipnorm', default=400, type=float)
    parser.add_argument('--batch_size', default=32, type=int)
    parser.add_argument('--opt', default='adam', type=str, choices=['sgd', 'adam'])
    parser.add_argument('--dataset', default=None, type=str, nargs='+')
    parser.add_argument('--input_parser', type=str, default=None)
    parser.add_argument('--input_parser_params', nargs='+', default=[])
    parser.add_argument('--label_parser', type=str, default='simple_char_parser')
    parser.add_argument('--label_parser_params', nargs='+', default=[])
    parser.add_argument('--lr_schedule', default=None)
    parser.add_argument('--lr_params', nargs='+', default=[])
    parser.add_argument('--save', default=None, type=str)
    parser.add_argument('--gpu', default='0', type=str)
    parser.add_argument('--allow_growth', default=False, action='store_true')
    parser.add_argument('--verbose', default=0, type=int)
    parser.add_argument('--seed', default=None, type=float)
    return parser


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    utils.setup_logging()
    logger = logging.getLogger(__name__)

    from asr_study_amd import parallel
    rank, world = parallel.init_from_env()
    from asr_study_amd.core import optimizers
    from asr_study_amd.core.callbacks import MetaCheckpoint
    from asr_study_amd.datasets.dataset_generator import DatasetGenerator
    from asr_study_amd.utils.core_utils import setup_gpu, load_model
    if world == 1:
        setup_gpu(args.gpu, args.allow_growth, log_device_placement=args.verbose > 1)

    epoch_offset, meta = 0, None
    if args.load:
        args_nondefault = utils.parse_nondefault_args(args, parser.parse_args([]), argv)
        model, meta = load_model(args.load, return_meta=True)
        args = HParams(**vars(args)).update(meta['training_args']).update(vars(args_nondefault))
        epoch_offset = len(meta['epochs'])
        if args_nondefault.lr:
            model.optimizer.lr = args.lr
    else:
        model_fn = utils.get_from_module('core.models', args.model)
        model = model_fn(**(HParams().parse(args.model_params).values()))
        if args.opt.strip().lower() == 'sgd':
            opt = optimizers.SGD(lr=args.lr, momentum=args.momentum, clipnorm=args.clipnorm)
        else:
            opt = optimizers.Adam(lr=args.lr, clipnorm=args.clipnorm)
        model.compile(loss={'ctc': 'ctc_dummy_loss', 'decoder': 'decoder_dummy_loss'},
                      optimizer=opt, metrics={'decoder': 'ler'}, loss_weights=[1, 0])
    if world > 1:
        parallel.broadcast_parameters(model)

    output_dir = args.save
    if output_dir is None:
        output_dir = os.path.join('results', '%s_%s' % (args.model, datetime.datetime.now()))
    callback_list = []
    if rank == 0:
        os.makedirs(output_dir, exist_ok=True)
        callback_list = [MetaCheckpoint(os.path.join(output_dir, 'model.h5'),
                                        training_args=args, meta=meta),
                         MetaCheckpoint(os.path.join(output_dir, 'best.h5'),
                                        monitor='val_decoder_ler', save_best_only=True,
                                        mode='min', training_args=args, meta=meta)]
    if args.lr_schedule:
        raise ValueError('Learning rate schedule unrecognized')

    input_parser = utils.get_from_module('preprocessing.audio', args.input_parser,
                                         params=args.input_parser_params)
    label_parser = utils.get_from_module('preprocessing.text', args.label_parser,
                                         params=args.label_parser_params)
    data_gen = DatasetGenerator(input_parser, label_parser, batch_size=args.batch_size,
                                seed=args.seed)
    train_flow, valid_flow, test_flow = None, None, None
    num_val_samples = 0
    if len(args.dataset) == 1:
        train_flow, valid_flow, test_flow = data_gen.flow_from_fname(
            args.dataset[0], datasets=['train', 'valid', 'test'])
        num_val_samples = valid_flow.len
    else:
        train_flow = data_gen.flow_from_fname(args.dataset[0])
        valid_flow = data_gen.flow_from_fname(args.dataset[1])
        num_val_samples = valid_flow.len
        if len(args.dataset) == 3:
            test_flow = data_gen.flow_from_fname(args.dataset[2])
    if world > 1:       # every rank draws the same global batch and keeps its shard
        train_flow = parallel.ShardedFlow(train_flow, rank, world)

    print(str(vars(args) if not isinstance(args, HParams) else args.values()))
    model.fit_generator(train_flow, samples_per_epoch=train_flow.len,
                        nb_epoch=args.num_epochs, validation_data=valid_flow,
                        nb_val_samples=num_val_samples, max_q_size=10, nb_worker=1,
                        callbacks=callback_list, verbose=1 if rank == 0 else 0,
                        initial_epoch=epoch_offset)

    if test_flow and rank == 0:
        del model
        model = load_model(os.path.join(output_dir, 'best.h5'), mode='eval')
        metrics = model.evaluate_generator(test_flow, test_flow.len, max_q_size=10, nb_worker=1)
        msg = 'Total loss: %.4f\nCTC Loss: %.4f\nLER: %.2f%%' % (metrics[0], metrics[1],
                                                                metrics[3] * 100)
        logger.info(msg)
        with open(os.path.join(output_dir, 'results.txt'), 'w') as f:
            f.write(msg)
        print(msg)
    parallel.finalize()


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Evaluation CLI with the reference's flags (eval.py:26-49): loads a checkpoint in
'eval' mode (CTC beam search, default width 400 as utils/core_utils.py:70-71;
``--beam_width 100`` reproduces the README figure's setting) and prints the metrics.
"""
from __future__ import absolute_import, division, print_function

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from asr_study_amd.utils import generic_utils as utils          # noqa: E402
from asr_study_amd.utils.hparams import HParams                 # noqa: E402


def main(argv=None):
    parser = argparse.ArgumentParser(description='Evaluating an ASR system.')
    parser.add_argument('--model', required=True, type=str)
    parser.add_argument('--dataset', required=True, type=str)
    parser.add_argument('--subset', type=str, default='test')
    parser.add_argument('--batch_size', default=32, type=int)
    parser.add_argument('--input_parser', type=str, default=None)
    parser.add_argument('--input_parser_params', nargs='+', default=[])
    parser.add_argument('--label_parser', type=str, default='simple_char_parser')
    parser.add_argument('--label_parser_params', nargs='+', default=[])
    parser.add_argument('--gpu', default='0', type=str)
    parser.add_argument('--allow_growth', default=False, action='store_true')
    parser.add_argument('--save_transcriptions', default=None, type=str)
    parser.add_argument('--beam_width', default=400, type=int)
    args = parser.parse_args(argv)
    args_nondefault = utils.parse_nondefault_args(
        args, parser.parse_args(['--model', args.model, '--dataset', args.dataset]), argv)

    from asr_study_amd.datasets.dataset_generator import DatasetGenerator
    from asr_study_amd.utils.core_utils import setup_gpu, load_model
    setup_gpu(args.gpu, args.allow_growth)
    model, meta = load_model(args.model, return_meta=True, mode='eval',
                             beam_width=args.beam_width)
    # defaults < arguments stored with the checkpoint < arguments given explicitly
    # (the reference drops un-stored defaults such as --subset here, eval.py:60)
    args = HParams(**vars(args)).update(meta['training_args']).update(vars(args_nondefault))
    input_parser = utils.get_from_module('preprocessing.audio', args.input_parser,
                                         params=args.input_parser_params)
    label_parser = utils.get_from_module('preprocessing.text', args.label_parser,
                                         params=args.label_parser_params)
    data_gen = DatasetGenerator(input_parser, label_parser, batch_size=args.batch_size, seed=0)
    test_flow = data_gen.flow_from_fname(args.dataset, datasets=args.subset)
    metrics = model.evaluate_generator(test_flow, test_flow.len, max_q_size=10, nb_worker=1)
    for m, v in zip(model.metrics_names, metrics):
        print('%s: %4f' % (m, v))
    return metrics


if __name__ == '__main__':
    main()

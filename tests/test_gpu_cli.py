"""-m gpu: the reference's command-line surface end to end on a dummy dataset:
make dataset (GPU MFCC -> HDF5) -> train.py (2 epochs, checkpoints) -> eval.py (beam
search) ; and BASELINE cfg5: beam-search hypotheses / LER identical to the oracle on
the held-out split."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dataset(tmp_path, fmt):
    """Built through the make_dataset command line (extras/make_dataset.py)."""
    from asr_study_amd import cli
    from asr_study_amd.datasets import h5lite
    if fmt == 'h5' and not h5lite.available():
        fmt = 'npz'
    fname = str(tmp_path / ('dummy.' + fmt))
    cli.make_dataset_main(['--parser', 'dummy', '--parser_params', 'num_speakers', '4',
                           'num_utterances_per_speaker', '6', 'max_duration', '1.2',
                           'min_duration', '0.6', 'max_label_length', '8', 'split',
                           '[0.5, 0.25]', 'seed', '3', '--input_parser', 'mfcc',
                           '--input_parser_params', 'dd', 'False', '--output_file', fname])
    return fname


def test_train_eval_cli_roundtrip(tmp_path, capsys):
    sys.path.insert(0, ROOT)
    import train
    import eval as eval_cli
    fname = _dataset(tmp_path, 'h5')
    out = str(tmp_path / 'run')
    train.main(['--dataset', fname, '--model', 'graves2006', '--model_params', 'num_hiddens', '16',
                'std', '0.0', '--num_epochs', '2', '--batch_size', '4', '--save', out,
                '--seed', '1', '--lr', '0.01'])
    assert os.path.exists(os.path.join(out, 'model.h5'))
    assert os.path.exists(os.path.join(out, 'best.h5'))
    txt = open(os.path.join(out, 'results.txt')).read()
    assert 'LER' in txt and 'CTC Loss' in txt
    m = eval_cli.main(['--model', os.path.join(out, 'best.h5'), '--dataset', fname,
                       '--beam_width', '20'])
    assert len(m) == 4 and np.isfinite(m[1]) and m[3] >= 0
    printed = capsys.readouterr().out
    assert 'beam_search_ler' in printed
    # resume: two more epochs continue from the stored epoch count
    train.main(['--load', os.path.join(out, 'model.h5'), '--dataset', fname, '--num_epochs', '3',
                '--save', out, '--batch_size', '4'])
    from asr_study_amd.utils.core_utils import load_meta
    assert len(load_meta(os.path.join(out, 'model.h5'))['epochs']) == 3
    # predict.py: one utterance per forward pass over the test split, then the raw
    # network outputs (--no_decoder --save)
    import predict as predict_cli
    from asr_study_amd.datasets import h5lite
    res = predict_cli.main(['--model', os.path.join(out, 'best.h5'), '--dataset', fname,
                            '--beam_width', '10'])
    assert len(res) > 0 and all(isinstance(r['best'], str) for r in res)
    assert 'Ground Truth:' in capsys.readouterr().out
    pred = str(tmp_path / 'pred.h5')
    res = predict_cli.main(['--model', os.path.join(out, 'best.h5'), '--dataset', fname,
                            '--no_decoder', '--save', pred])
    assert res[0]['best'].ndim == 2 and res[0]['best'].shape[1] == 28
    with h5lite.File(pred, 'r') as f:
        assert len(f['predictions'][:]) == len(res)
        assert int(f['predictions'].attrs['num_labels']) == 28


def test_deep_speech2_through_the_command_lines(tmp_path, capsys):
    """The conv front-end model through the reference's own command lines: make_dataset
    (log-mel on the GPU -> HDF5) -> train.py --model deep_speech2 (2 epochs, checkpoints with
    convolution2d_k groups) -> eval.py (beam search on ceil(len / 2) frames) -> predict.py."""
    sys.path.insert(0, ROOT)
    import train
    import eval as eval_cli
    import predict as predict_cli
    from asr_study_amd import cli
    from asr_study_amd.datasets import h5lite
    fname = str(tmp_path / ('mel.h5' if h5lite.available() else 'mel.npz'))
    cli.make_dataset_main(['--parser', 'dummy', '--parser_params', 'num_speakers', '4',
                           'num_utterances_per_speaker', '6', 'max_duration', '1.2',
                           'min_duration', '0.6', 'max_label_length', '6', 'split',
                           '[0.5, 0.25]', 'seed', '5', '--input_parser', 'logfbank',
                           '--input_parser_params', 'num_filt', '16', '--output_file', fname])
    out = str(tmp_path / 'ds2')
    train.main(['--dataset', fname, '--model', 'deep_speech2', '--model_params', 'num_features',
                '16', 'num_hiddens', '16', 'num_layers', '2', 'conv_filters', '4', '--num_epochs',
                '2', '--batch_size', '4', '--save', out, '--seed', '1', '--lr', '0.01'])
    assert os.path.exists(os.path.join(out, 'best.h5'))
    with h5lite.File(os.path.join(out, 'best.h5'), 'r') as f:
        names = f['model_weights'].attrs.get_strings('layer_names')
    assert names[:2] == ['convolution2d_1', 'convolution2d_2'] and 'bidirectional_2' in names
    m = eval_cli.main(['--model', os.path.join(out, 'best.h5'), '--dataset', fname,
                       '--beam_width', '20'])
    assert len(m) == 4 and np.isfinite(m[1]) and m[3] >= 0
    res = predict_cli.main(['--model', os.path.join(out, 'best.h5'), '--dataset', fname,
                            '--beam_width', '10'])
    assert len(res) > 0 and all(isinstance(r['best'], str) for r in res)
    capsys.readouterr()


def test_cfg5_beam_search_ler_matches_oracle(tmp_path):
    """Beam width 100 (README) and 400 (code default): identical top-1 strings and
    LER to the oracle decoder run on the same logits."""
    from asr_study_amd.core import models
    from asr_study_amd.datasets.dataset_generator import DatasetGenerator
    from asr_study_amd.preprocessing import text
    from oracle import decode as OD
    fname = _dataset(tmp_path, 'npz')
    model = models.graves2006(num_features=26, num_hiddens=12, num_classes=28, std=0.0, seed=4)
    gen = DatasetGenerator(None, text.simple_char_parser, batch_size=6, shuffle=False, seed=0)
    flow = gen.flow_from_fname(fname, datasets='test')
    (x, labels, lens), _ = next(flow)
    slab = model.to_slab(x)
    logits = model.forward(slab).cpu().numpy()
    csr = labels.tocsr()
    truth = [csr.data[csr.indptr[i]:csr.indptr[i + 1]].tolist() for i in range(len(lens))]
    for width in (100, 400):
        model.decoder = dict(is_greedy=False, beam_width=width, merge_repeated=True)
        hyp = model.predict(slab, lens)
        if width == 100:
            assert hyp == OD.beam_search_decode(logits[:, :len(lens)].astype(np.float64), lens,
                                                beam_width=width)
        else:
            # width 400 is what eval.py actually uses (utils/core_utils.py:70-71); the
            # pure-Python oracle needs ~0.35 s per frame there, so it checks the first 30
            # frames of two utterances (the decoder honours inputs_length; the beam is full
            # from the third frame on); tests/test_capi_host.py covers width 400 on the host
            short = np.minimum(np.asarray(lens)[:2], 30)
            got = model.predict(slab, list(short) + list(np.asarray(lens)[2:]))[:2]
            assert got == OD.beam_search_decode(logits[:, :2].astype(np.float64), short,
                                                beam_width=400)
        m = model.test_on_batch([('slab', slab), truth, lens])
        assert abs(m[3] - OD.ler(hyp, truth)) < 1e-6


def test_load_keras122_layout_checkpoint_and_round_trip(tmp_path):
    """A checkpoint in the reference's on-disk layout (Keras-1.2.2 weight groups + meta,
    written by h5py: tests/golden/gen_keras_h5.py) loads into the product model with the
    topology named in meta/training_args; saving it again and reloading preserves every
    weight bit for bit, and the forward pass equals the oracle's on those weights."""
    from asr_study_amd.core.callbacks import save_model
    from asr_study_amd.utils.core_utils import load_model
    from oracle import lstm as OL
    here = os.path.join(ROOT, 'tests', 'golden')
    want = np.load(os.path.join(here, 'keras122_graves_weights.npz'))
    model, meta = load_model(os.path.join(here, 'keras122_graves.h5'), return_meta=True)
    order = ['forward_lstm_1_W_0', 'forward_lstm_1_U_0', 'forward_lstm_1_b_0',
             'backward_lstm_1_W_0', 'backward_lstm_1_U_0', 'backward_lstm_1_b_0',
             'dense_1_W_0', 'dense_1_b_0']
    got = model.get_weights()
    assert len(got) == 8 and all(np.array_equal(g, want[k]) for g, k in zip(got, order))
    assert meta['training_args']['model'] == 'graves2006' and meta['epochs'] == [0, 1, 2]
    again = str(tmp_path / 'again.h5')
    save_model(model, again, meta)
    m2 = load_model(again)
    assert all(np.array_equal(a, b) for a, b in zip(m2.get_weights(), got))
    rs = np.random.RandomState(0)
    x = rs.randn(3, 20, 26).astype(np.float32)
    logits = model.forward(model.to_slab(x)).cpu().numpy()[:, :3]
    p = {'layers': [{d: {'W': want[d + '_lstm_1_W_0'].astype(np.float64),
                         'U': want[d + '_lstm_1_U_0'].astype(np.float64),
                         'b': want[d + '_lstm_1_b_0'].astype(np.float64)}
                     for d in ('forward', 'backward')}],
         'dense': {'W': want['dense_1_W_0'].astype(np.float64),
                   'b': want['dense_1_b_0'].astype(np.float64)}}
    p['layers'][0] = {'fwd': p['layers'][0]['forward'], 'bwd': p['layers'][0]['backward']}
    ref = OL.model_forward(p, np.transpose(x, (1, 0, 2)).astype(np.float64))
    ref = ref[0] if isinstance(ref, tuple) else ref
    assert np.abs(logits - ref).max() < 1e-4

"""CTC glue with the reference's names (core/ctc_utils.py).

In the reference these are Lambda bodies wrapping tf.nn.ctc_loss /
ctc_greedy_decoder / ctc_beam_search_decoder; here they call the HIP kernels (loss,
gradient, greedy, device beam search) through the C ABI, on time-major logit slabs."""
import numpy as np
import torch

from .. import ops


def decoder_config(is_greedy=True, beam_width=100, top_paths=1, merge_repeated=True):
    """Decoder kwargs of ``decode`` (core/ctc_utils.py:35-50)."""
    if top_paths != 1:
        raise NotImplementedError('top_paths != 1')
    return dict(is_greedy=is_greedy, beam_width=beam_width, merge_repeated=merge_repeated)


def decode(inputs, **kwargs):
    """(y_pred slab (T, n_pad, C) CUDA, seq_len (N,)) -> list of N label lists.
    is_greedy (default True) or beam search (beam_width 100, merge_repeated True)."""
    y_pred, seq_len = inputs
    seq = np.asarray(seq_len).reshape(-1).astype(np.int32)
    N = len(seq)
    if kwargs.get('is_greedy', True):
        dec, dlen = ops.ctc_greedy(y_pred, torch.as_tensor(seq).to(y_pred.device), N)
        dec, dlen = dec.cpu().numpy(), dlen.cpu().numpy()
        return [dec[n, :dlen[n]].tolist() for n in range(N)]
    width = int(kwargs.get('beam_width', 100))
    merge = bool(kwargs.get('merge_repeated', True))
    # host or device decoder (same strings, tests/test_gpu_beam.py): ops.beam_decoder_choice
    if ops.beam_decoder_choice(N, width, y_pred.shape[2], y_pred.is_cuda) == 'device':
        dec, dlen, _ = ops.ctc_beam_search(y_pred, torch.as_tensor(seq).to(y_pred.device), N,
                                           width, merge)
        dec, dlen = dec.cpu().numpy(), dlen.cpu().numpy()
        return [dec[n, :dlen[n]].tolist() for n in range(N)]
    hyps, _ = ops.ctc_beam_search_host(y_pred.cpu().numpy(), seq, N, width, merge)
    return hyps


def ctc_lambda_func(args):
    """(y_pred slab, labels list, inputs_length) -> per-sample CTC loss (N,) CUDA."""
    y_pred, labels, inputs_length = args
    N = len(labels)
    lmax = max([len(l) for l in labels] + [1])
    lab = np.zeros((N, lmax), np.int32)
    for n, l in enumerate(labels):
        lab[n, :len(l)] = l
    dev = y_pred.device
    return ops.ctc_loss_grad(
        y_pred, torch.from_numpy(lab).to(dev),
        torch.tensor([len(l) for l in labels], dtype=torch.int32, device=dev),
        torch.as_tensor(np.asarray(inputs_length, np.int32).reshape(-1)).to(dev), N)


def ctc_dummy_loss(y_true, y_pred):
    """Keras needed a loss callable; the model output already IS the loss."""
    return y_pred


def decoder_dummy_loss(y_true, y_pred):
    return 0.0

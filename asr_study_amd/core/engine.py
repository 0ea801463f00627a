"""Execution engine behind ``ctc_model()``: runs the stage list on the HIP kernels.

Replaces what Keras 1.2.2 + TensorFlow 1.3.0 do for the reference once
``model.compile`` / ``fit_generator`` / ``evaluate_generator`` are called
(train.py:140-143, 213-217, 223-227; eval.py:76-80): forward, CTC loss, BPTT,
global-norm clip + Adam/SGD, greedy/beam decode and LER.  Everything numeric goes
through the C ABI (asr_study_amd.ops); torch provides device buffers, streams and
the RCCL all-reduce only.

Data layout: activations are time-major slabs (T, n_pad, features) with the batch
rounded up to a multiple of 16 (padding rows stay zero / get zero gradient).  All
parameters live in ONE flat float32 buffer (so the optimiser and the all-reduce are
single launches); LSTM gate axes are stored unit-major/gate-minor (see
include/asr_hip.h) and hidden sizes are rounded up to a multiple of 4 with zero
weights (eyben's 78/27-unit layers).  get_weights()/set_weights() speak the
reference's Keras layout: per Bidirectional layer [W (in,4H), U (H,4H), b (4H)] for
the forward then the backward copy, gate blocks i,f,c,o; Dense [W, b].
"""
import math
import os
import time

import numpy as np
import torch

from .. import ops
from . import optimizers as _opt


def _pad4(n):
    return (int(n) + 3) // 4 * 4


def _gm2um(a, H, Hp):
    """(..., 4H) gate-major -> (..., 4Hp) unit-major/gate-minor, zero padded."""
    sh = a.shape[:-1]
    g = a.reshape(sh + (4, H))
    out = np.zeros(sh + (Hp, 4), a.dtype)
    out[..., :H, :] = np.swapaxes(g, -1, -2)
    return out.reshape(sh + (4 * Hp,))


def _um2gm(a, H, Hp):
    sh = a.shape[:-1]
    u = a.reshape(sh + (Hp, 4))[..., :H, :]
    return np.ascontiguousarray(np.swapaxes(u, -1, -2)).reshape(sh + (4 * H,))


# where a Model lives when the factory is given no ``device`` (host-only tests build on 'cpu':
# weights I/O works there, every compute call still needs the HIP library and a GPU)
DEFAULT_DEVICE = os.environ.get('ASR_DEVICE', 'cuda:0')


class Stage(object):
    pass


class _Feeder(object):
    """Bounded producer queue over a batch generator (Keras' GeneratorEnqueuer with one
    worker THREAD, train.py:215-216).  The generator's exception, if any, is re-raised in
    the consumer."""

    def __init__(self, generator, max_q_size=10, device=None):
        import queue
        import threading
        self._q = queue.Queue(maxsize=max(1, int(max_q_size)))
        self._stop = threading.Event()
        self._gen = generator

        def work():
            try:
                if device is not None and device.type == 'cuda':
                    torch.cuda.set_device(device)    # the current device is per thread
                while not self._stop.is_set():
                    item = next(self._gen)
                    while not self._stop.is_set():
                        try:
                            self._q.put((item, None), timeout=0.1)
                            break
                        except queue.Full:
                            continue
            except BaseException as e:           # StopIteration included: surfaces in get()
                self._q.put((None, e))
        self._t = threading.Thread(target=work, name='asr-batch-feeder', daemon=True)
        self._t.start()

    def get(self):
        item, err = self._q.get()
        if err is not None:
            raise err
        return item

    def close(self):
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()
        except Exception:
            pass
        self._t.join(timeout=5.0)


class Model(object):
    """The object ``ctc_model(inputs, output)`` returns (core/models.py:31-52)."""

    def __init__(self, spec, num_features, device=None, seed=0, lstm_mode=0):
        self.device = torch.device(device or DEFAULT_DEVICE)
        self.spec = spec
        self.num_features = int(num_features)
        self.lstm_mode = int(os.environ.get('ASR_LSTM_MODE', lstm_mode))   # 1 = stepwise kernels
        # A persistent recurrent kernel that abandoned a bounded spin demotes the model to the
        # stepwise kernels only for a while: after `_retry_gap` clean optimisation steps the
        # persistent kernels are tried again (the gap doubles with every further fallback).
        # A model that was ASKED for mode 1 stays there.
        self._mode_pinned = self.lstm_mode == 1
        self._retry_at = None
        self._retry_gap = int(os.environ.get('ASR_LSTM_RETRY_STEPS', '64'))
        self.fallbacks = 0              # timeouts survived (bench.py asserts 0)
        self.vetoed_steps = 0           # optimisation steps skipped because of them
        self._fault_gen = 0
        self.optimizer = None
        self.metrics_names = ['loss', 'ctc_loss', 'decoder_loss', 'decoder_ler']
        self.decoder = dict(is_greedy=True)
        self.stop_training = False
        self._bufs = {}
        self._step = 0
        self._enqueued = self._checked = 0   # train_on_batch steps enqueued / whose flags were read
        self._ar_ref_pad = 0            # rank-invariant reference shard (set per batch)
        self._ar_covered = []
        # weight-gradient GEMMs run on a side stream, concurrently with the next
        # layer's persistent BPTT kernel (which occupies only H/16 x chains CUs)
        import os as _os
        # ASR_OVERLAP: 1 = always, 0 = never, auto (default) = unless a layer's recurrence
        # fills every CU (cfg3: 2 directions x 4 batch tiles x 32 workgroups = 256): its waves
        # hold 480 of a SIMD's 512 registers, no GEMM wave fits beside them, and a GEMM launched
        # on the side stream only waits (measured: 51.9 vs 52.5 ms per cfg3 step, but every
        # GEMM duration in a profile doubled by the wait) -- decided per batch in forward()
        self._overlap_mode = _os.environ.get('ASR_OVERLAP', 'auto')
        self.overlap = self._overlap_mode != '0'
        self._side = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        # the GEMMs either side of a recurrence are pipelined against its last steps
        # (frames whose both directions are already final), on a third stream
        # 'auto': only when a layer's recurrence leaves at least half of the CUs to the GEMMs
        # it is pipelined against (cfg2: 64 of 256 workgroups; not cfg3's 256 of 256)
        self._pipeline_mode = _os.environ.get('ASR_PIPELINE', 'auto')
        # the recurrence is cut after split/16 of its steps (frames [T-S, S) are final then)
        self._pipe_split16 = min(15, max(9, int(_os.environ.get('ASR_PIPE_SPLIT', '13'))))
        # (ASR_PIPE_HALVES=0: the pipelined GEMMs wait for whole frames -- comparison switch)
        self._pipe_halves = _os.environ.get('ASR_PIPE_HALVES', '1') != '0'
        self.pipeline = self.overlap and self._pipeline_mode == '1'
        # ASR_BPTT_COMPACT: auto (default) = where a layer's BPTT would fill every CU (cfg3) it is
        # launched in the COMPACT geometry (asr_lstm_args.compact: H/32 workgroups per chain,
        # half the CUs, bit-identical gradients, a longer step) whenever the weight-gradient
        # GEMMs of the layer above are waiting, and those run beside it on the side stream
        # instead of behind it; 0 = never (the serial schedule of rounds 1-4), 1 = wherever the
        # compact kernel exists
        self._compact_mode = _os.environ.get('ASR_BPTT_COMPACT', 'auto')
        # ASR_BPTT_PLANES=0: BPTT always writes the fp32 dz slab and a pack pass follows (the
        # round 1-5 path; A/B switch).  Default: BPTT writes the packed planes itself where its
        # kernel can (backward()); per-stage bounds of max|dz| live in _buf('dzbound<si>')
        self._dz_planes_mode = _os.environ.get('ASR_BPTT_PLANES', '1') != '0'
        self._dz_bound_key = {}         # stage -> (fault generation, weights epoch) of its bound
        self._weights_epoch = 0         # bumped when weights are replaced from outside
        self._dz_measure_passes = 0     # measuring BPTT passes run so far (first step of a layer)
        self._pipe = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        # big GEMMs on operands packed once into split-fp16 planes (ops.pack_hl / gemm_hl);
        # ASR_GEMM_PACKED=0 keeps the convert-per-tile kernels, ASR_GEMM_PREC=0 (exact fp32) too
        # 'auto': from 512 hidden units on (measured: +4 % at 5 x BiLSTM(512) with the 256 x 256
        # tile, -6 % at 5 x BiLSTM(256), where the frame-range pipelining of the per-tile path
        # and the absent pack passes win)
        self._packed_mode = _os.environ.get('ASR_GEMM_PACKED', 'auto')
        self.packed = False                       # decided in _layout (needs the stage list)
        self._hl = {}
        # training-time noise comes from the library's counter-based streams (ops.dropout_masks
        # ...: Philox-4x32-10 keyed by this seed; stream id = 4 * stage index + kind, step =
        # the optimisation step), so a step's masks are a pure function of (seed, stage, step)
        # ... and, data parallel, of the rank: every rank draws its OWN noise for its shard
        # (the rank is mixed in when the key is used: the process group may be created later)
        self._rng_base = (int(seed) + 12345) & (2 ** 64 - 1)
        self._layout(seed)

    @property
    def rng_seed(self):
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        return (self._rng_base + rank * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)

    # ------------------------------------------------------------------ params
    def _layout(self, seed):
        rs = np.random.RandomState(seed)
        off = 0
        self.stages = []
        segs = []
        # the input features are padded to a multiple of 4 columns (zero column(s), zero
        # weight rows) so that the first layer's GEMMs qualify for the 16-byte fast path
        f_real, f_pad = self.num_features, _pad4(self.num_features)
        self.f_in_pad0 = f_pad
        init = []

        def take(n):
            nonlocal off
            o = off
            off += _pad4(n)
            return o

        for st in self.spec:
            s = Stage()
            s.kind = st['type']
            s.p_lo = off                    # this stage's slice of the flat param / grad buffers
            s.f_in, s.f_in_pad = f_real, f_pad
            if s.kind in ('noise', 'dropout'):
                s.value = st['value']
            elif s.kind == 'reshape':
                s.target = [int(v) for v in st['target']]
            elif s.kind == 'conv':
                # 2-D convolution front-end (K13): W (kt, kf, C_in, C_out) + b (C_out), Keras
                # 'tf' kernel layout; the slab keeps its channel-minor rows (F * C)
                for k in ('F_in', 'C_in', 'C_out', 'kt', 'kf', 'st', 'sf'):
                    setattr(s, k, int(st[k]))
                s.clip, s.l2 = float(st['clip']), float(st.get('l2', 0.0))
                s.F_out = -(-s.F_in // s.sf)
                if f_real != s.F_in * s.C_in or f_pad != f_real:
                    raise ValueError('conv stage: %d input features (a multiple of 4) expected, '
                                     'got %d (padded %d)' % (s.F_in * s.C_in, f_real, f_pad))
                if (s.F_out * s.C_out) % 4:
                    raise ValueError('conv stage: F_out * C_out must be a multiple of 4')
                # limits of asr_conv2d_* (csrc/conv.hip), checked when the model is BUILT rather
                # than at the first launch / the first backward pass:
                if not 1 <= s.kt <= 16:
                    raise ValueError('conv stage: kernel length over time %d not in 1..16 '
                                     '(asr_conv2d_* keeps one plane shift per time tap)' % s.kt)
                if s.st < 1 or s.sf < 1:
                    raise ValueError('conv stage: strides must be >= 1')
                if s.st > 1 and any(p.kind in ('dense', 'bilstm', 'conv') for p in self.stages):
                    raise ValueError(
                        'conv stage with time stride %d behind a trainable stage: asr_conv2d_dgrad '
                        'exists for time stride 1 only, so a time-strided convolution must be the '
                        'FIRST trainable stage (its input is data and needs no gradient)' % s.st)
                nw = s.kt * s.kf * s.C_in * s.C_out
                s.oW = take(nw)
                s.ob = take(s.C_out)
                segs += [(s.oW, _pad4(nw), s.l2), (s.ob, _pad4(s.C_out), 0.0)]
                lim = math.sqrt(6.0 / (s.kt * s.kf * (s.C_in + s.C_out)))      # glorot_uniform
                W = rs.uniform(-lim, lim, size=(s.kt, s.kf, s.C_in, s.C_out))
                init.append((s, 'conv', [W.astype(np.float32), np.zeros(s.C_out, np.float32)]))
                f_real = f_pad = s.F_out * s.C_out
            elif s.kind == 'dense':
                s.n_out = st['n_out']
                s.l2 = st.get('l2', 0.0)
                s.oW = take(f_pad * s.n_out)
                s.ob = take(s.n_out)
                segs += [(s.oW, _pad4(f_pad * s.n_out), s.l2), (s.ob, _pad4(s.n_out), 0.0)]
                lim = math.sqrt(6.0 / (f_real + s.n_out))
                W = rs.uniform(-lim, lim, size=(f_real, s.n_out))
                init.append((s, 'dense', [W.astype(np.float32), np.zeros(s.n_out, np.float32)]))
                f_real = f_pad = s.n_out
                s.in_map = None
            elif s.kind == 'bilstm':
                s.H = st['H']
                s.Hp = _pad4(s.H)
                s.dropout_W, s.dropout_U = st.get('dropout_W', 0.0), st.get('dropout_U', 0.0)
                s.l2_W, s.l2_U = st.get('l2_W', 0.0), st.get('l2_U', 0.0)
                s.mi = st.get('mi')                     # [alpha, beta1, beta2] inits or None
                s.zoneout_c = float(st.get('zoneout_c') or 0.0)
                s.zoneout_h = float(st.get('zoneout_h') or 0.0)
                s.act = st.get('activation') or 'tanh'      # core/layers.py:452, :463
                s.oW = take(f_pad * 8 * s.Hp)
                s.oU = take(2 * s.Hp * 4 * s.Hp)
                segs += [(s.oW, _pad4(f_pad * 8 * s.Hp), s.l2_W),
                         (s.oU, _pad4(2 * s.Hp * 4 * s.Hp), s.l2_U)]
                s.ln = st.get('layer_norm')             # [gain_init, bias_init] or None
                if s.ln is not None:
                    if s.Hp != s.H:
                        raise NotImplementedError('layer_norm needs num_hiddens % 4 == 0')
                    # (2, 34H): alpha, beta1, beta2, bias, LN(h@U) gain/bias, LN(x@W)
                    # gain/bias (4H each), LN(c) gain/bias (H each)  -- csrc/lstm_ln.hip
                    s.ocell = take(68 * s.Hp)
                    s.ob = None
                    segs.append((s.ocell, _pad4(68 * s.Hp), 0.0))
                elif s.mi is None:
                    s.ob = take(8 * s.Hp)
                    segs.append((s.ob, _pad4(8 * s.Hp), 0.0))
                else:       # (2, 4, 4Hp): alpha, beta1, beta2, bias per direction
                    s.omi = take(32 * s.Hp)
                    s.ob = None
                    segs.append((s.omi, _pad4(32 * s.Hp), 0.0))
                ws = []
                for _ in range(2):      # Keras-1.2.2 consume_less='gpu' init
                    lim = math.sqrt(6.0 / (f_real + 4 * s.H))
                    W = rs.uniform(-lim, lim, size=(f_real, 4 * s.H))
                    a = rs.normal(0.0, 1.0, (s.H, 4 * s.H))
                    u, _, v = np.linalg.svd(a, full_matrices=False)
                    U = 1.1 * (u if u.shape == (s.H, 4 * s.H) else v)
                    b = np.zeros(4 * s.H)
                    b[s.H:2 * s.H] = 1.0
                    ws += [W.astype(np.float32), U.astype(np.float32), b.astype(np.float32)]
                    if s.mi is not None:            # k_init: constant vectors (core/initializers.py)
                        ws += [np.full(4 * s.H, float(k), np.float32) for k in s.mi]
                    if s.ln is not None:            # gain, bias of LN(h@U), LN(x@W), LN(c)
                        for width in (4 * s.H, 4 * s.H, s.H):
                            ws += [np.full(width, float(s.ln[0]), np.float32),
                                   np.full(width, float(s.ln[1]), np.float32)]
                init.append((s, 'bilstm', ws))
                f_real, f_pad = 2 * s.H, 2 * s.Hp
            elif s.kind == 'merge':
                s.mode, s.skip = st['mode'], int(st['skip'])
                src = self.stages[s.skip]
                if (src.f_out, src.f_out_pad) != (f_real, f_pad):
                    raise ValueError('merge: widths differ (%d vs %d)' % (src.f_out, f_real))
                s.coef = 1.0 if s.mode == 'sum' else 0.5
            else:
                raise ValueError(s.kind)
            s.f_out, s.f_out_pad = f_real, f_pad
            s.p_hi = off
            self.stages.append(s)
        import os as _os
        widest = max([st.Hp for st in self.stages if st.kind == 'bilstm'] + [0])
        self.packed = (_os.environ.get('ASR_GEMM_PREC', '1') != '0' and
                       (self._packed_mode == '1' or (self._packed_mode == 'auto' and widest >= 512)))
        self.num_classes = f_real
        self.time_strides = [st.st for st in self.stages if st.kind == 'conv' and st.st > 1]
        self._convs = {}
        self.n_params = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        # gradients + 4 trailing floats: [0:2] carry this rank's recurrent-kernel timeout flags
        # through the gradient all-reduce (sum > 0 on every rank if ANY rank timed out), so a
        # veto of the update is collective and costs no collective of its own
        self._gbuf = torch.zeros(off + 4, dtype=torch.float32, device=self.device)
        self.grads = self._gbuf[:off]
        self._segments = sorted(segs)
        self._segs_dev, self._nseg = ops.make_segments(self._segments, self.device)
        self._norm = torch.zeros(2, dtype=torch.float64, device=self.device)
        weights = []
        for s, kind, ws in init:
            weights += ws
        self.set_weights(weights)

    def _real_rows(self, s):
        """Map of the padded input-feature rows of a stage to its real rows."""
        prev = None
        for st in self.stages:
            if st is s:
                break
            if st.kind in ('dense', 'bilstm', 'conv'):
                prev = st
        if prev is not None and prev.kind == 'bilstm' and prev.Hp != prev.H:
            idx = np.concatenate([np.arange(prev.H), prev.Hp + np.arange(prev.H)])
        else:
            idx = np.arange(s.f_in)
        return idx

    def set_weights(self, weights):
        """weights: flat list in the reference's Keras order (see module doc)."""
        host = self.params.detach().cpu().numpy().copy()
        it = iter(weights)
        for s in self.stages:
            if s.kind == 'conv':
                W, b = np.asarray(next(it), np.float32), np.asarray(next(it), np.float32)
                assert W.shape == (s.kt, s.kf, s.C_in, s.C_out), W.shape
                host[s.oW:s.oW + W.size] = W.ravel()
                host[s.ob:s.ob + s.C_out] = b
            elif s.kind == 'dense':
                W, b = np.asarray(next(it), np.float32), np.asarray(next(it), np.float32)
                rows = self._real_rows(s)
                Wp = np.zeros((s.f_in_pad, s.n_out), np.float32)
                Wp[rows] = W
                host[s.oW:s.oW + Wp.size] = Wp.ravel()
                host[s.ob:s.ob + s.n_out] = b
            elif s.kind == 'bilstm':
                rows = self._real_rows(s)
                Wp = np.zeros((s.f_in_pad, 2, 4 * s.Hp), np.float32)
                Up = np.zeros((2, s.Hp, 4 * s.Hp), np.float32)
                bp = np.zeros((2, 4 * s.Hp), np.float32)
                mip = np.zeros((2, 4, 4 * s.Hp), np.float32)
                for d in range(2):
                    W, U, b = [np.asarray(next(it), np.float32) for _ in range(3)]
                    Wp[rows, d] = _gm2um(W, s.H, s.Hp)
                    Up[d, :s.H] = _gm2um(U, s.H, s.Hp)
                    bp[d] = _gm2um(b, s.H, s.Hp)
                    if s.mi is not None:            # Keras order: W, U, b, alpha, beta1, beta2
                        for k in range(3):
                            mip[d, k] = _gm2um(np.asarray(next(it), np.float32), s.H, s.Hp)
                        mip[d, 3] = bp[d]
                    if s.ln is not None:            # ... then the LN pairs (Uh, Wx, new_c)
                        H = s.H
                        blk = np.zeros(34 * H, np.float32)
                        if s.mi is not None:
                            blk[:12 * H] = mip[d, :3].ravel()
                        blk[12 * H:16 * H] = bp[d]
                        for k in range(4):          # gain_u, bias_u, gain_w, bias_w
                            blk[(16 + 4 * k) * H:(20 + 4 * k) * H] = \
                                _gm2um(np.asarray(next(it), np.float32), H, H)
                        blk[32 * H:33 * H] = np.asarray(next(it), np.float32)
                        blk[33 * H:34 * H] = np.asarray(next(it), np.float32)
                        host[s.ocell + d * 34 * H:s.ocell + (d + 1) * 34 * H] = blk
                host[s.oW:s.oW + Wp.size] = Wp.ravel()
                host[s.oU:s.oU + Up.size] = Up.ravel()
                if s.ln is not None:
                    pass
                elif s.mi is None:
                    host[s.ob:s.ob + bp.size] = bp.ravel()
                else:
                    host[s.omi:s.omi + mip.size] = mip.ravel()
        self.params.copy_(torch.from_numpy(host))
        self._weights_epoch += 1        # (bounds measured under the old weights are dropped)

    def _unpack(self, flat):
        out = []
        for s in self.stages:
            if s.kind == 'conv':
                nw = s.kt * s.kf * s.C_in * s.C_out
                out += [flat[s.oW:s.oW + nw].reshape(s.kt, s.kf, s.C_in, s.C_out).copy(),
                        flat[s.ob:s.ob + s.C_out].copy()]
            elif s.kind == 'dense':
                rows = self._real_rows(s)
                W = flat[s.oW:s.oW + s.f_in_pad * s.n_out].reshape(s.f_in_pad, s.n_out)[rows]
                out += [W.copy(), flat[s.ob:s.ob + s.n_out].copy()]
            elif s.kind == 'bilstm':
                rows = self._real_rows(s)
                Wp = flat[s.oW:s.oW + s.f_in_pad * 8 * s.Hp].reshape(s.f_in_pad, 2, 4 * s.Hp)
                Up = flat[s.oU:s.oU + 2 * s.Hp * 4 * s.Hp].reshape(2, s.Hp, 4 * s.Hp)
                if s.ln is not None:
                    cp = flat[s.ocell:s.ocell + 68 * s.Hp].reshape(2, 34 * s.Hp)
                    mip = cp[:, :16 * s.Hp].reshape(2, 4, 4 * s.Hp)
                    bp = mip[:, 3]
                elif s.mi is None:
                    bp = flat[s.ob:s.ob + 8 * s.Hp].reshape(2, 4 * s.Hp)
                else:
                    mip = flat[s.omi:s.omi + 32 * s.Hp].reshape(2, 4, 4 * s.Hp)
                    bp = mip[:, 3]
                for d in range(2):
                    out += [_um2gm(Wp[rows, d], s.H, s.Hp), _um2gm(Up[d, :s.H], s.H, s.Hp),
                            _um2gm(bp[d], s.H, s.Hp)]
                    if s.mi is not None:
                        out += [_um2gm(mip[d, k], s.H, s.Hp) for k in range(3)]
                    if s.ln is not None:
                        H = s.H
                        out += [_um2gm(cp[d, (16 + 4 * k) * H:(20 + 4 * k) * H], H, H)
                                for k in range(4)]
                        out += [cp[d, 32 * H:33 * H].copy(), cp[d, 33 * H:34 * H].copy()]
        return out

    def get_weights(self):
        return self._unpack(self.params.detach().cpu().numpy())

    def get_gradients(self):
        """Last computed gradients (before clipping, without the l2 term), in the
        same order/layout as get_weights()."""
        return self._unpack(self.grads.detach().cpu().numpy())

    def count_params(self):
        return int(sum(w.size for w in self.get_weights()))

    # ------------------------------------------------------------------ compile
    def compile(self, loss=None, optimizer=None, metrics=None, loss_weights=None):
        """Keras signature (train.py:140-143).  ``loss``/``metrics`` are accepted for
        surface compatibility: the objective is always 1*mean(ctc) + 0*decoder + l2."""
        if isinstance(optimizer, str):
            optimizer = _opt.get(optimizer)
        self.optimizer = optimizer
        if optimizer is not None:
            optimizer.bind(self)

    # ------------------------------------------------------------------ buffers
    def _buf(self, name, shape, zero=False):
        """Cached float32 buffer.  zero=True fills it with zeros WHEN IT IS CREATED, on the
        current stream -- only for buffers whose first user runs on that same stream (a fill
        kernel is not ordered against the side / pipe streams, which wait on events only)."""
        key = (name, tuple(int(x) for x in shape))
        b = self._bufs.get(key)
        if b is None:
            # drop stale shapes of the same name to bound memory
            for k in [k for k in self._bufs if k[0] == name]:
                del self._bufs[k]
            b = (torch.zeros if zero else torch.empty)(key[1], dtype=torch.float32,
                                                       device=self.device)
            self._bufs[key] = b
        return b

    def _view(self, off, n):
        return self.params[off:off + n]

    def _gview(self, off, n):
        return self.grads[off:off + n]

    def _planes(self, name, rows, k):
        """Cached ops.HlPlanes buffer (rows, k) under `name` (stale shapes dropped)."""
        key = (name, int(rows), int(k))
        b = self._hl.get(key)
        if b is None:
            for kk in [kk for kk in self._hl if kk[0] == name]:
                del self._hl[kk]
            b = ops.HlPlanes(rows, k, self.device)
            self._hl[key] = b
        return b

    def _const_one(self):
        if not hasattr(self, '_one'):
            self._one = torch.ones(1, dtype=torch.float32, device=self.device)
        return self._one

    @staticmethod
    def _y_bounded(s):
        """|h| <= 1 for a BiLSTM stage whose output activation is bounded (h = o * act(c), 0 <= o
        <= 1): only then may a packed operand built from its y take 1 as the tensor bound (x the
        inverted-dropout scale, far below the 2^8 head-room of pow2_scale(1)).  relu / linear /
        softplus carry an unbounded c through: those outputs are measured."""
        return s.kind == 'bilstm' and s.act in ('tanh', 'sigmoid', 'hard_sigmoid', 'softsign')

    def _stage_packed(self, s):
        """Whether a BiLSTM stage's GEMMs run on packed operands (plain cell only)."""
        return (self.packed and s.kind == 'bilstm' and s.mi is None and s.ln is None
                and s.f_in_pad % 8 == 0 and s.Hp % 8 == 0)      # 16-byte rows of the planes

    def _pack_weights(self):
        """W of every packed stage -> planes for x@W (reduction over the input features:
        (8H, in)) and for dz@W^T (reduction over the gate columns: (in, 8H)); one pass per W,
        ~0.3 GB per step at cfg3.  Only the FIRST packed stage's planes are needed at once: its
        pack runs on the calling stream, the others' (two small kernels each, launch-bound: 0.15 ms
        of a cfg3 step when they sat in front of the first GEMM) on the side stream beside the
        first layer's work; self._w_ready[si] is the event the stage's first GEMM waits for."""
        self._w_ready = {}
        main = torch.cuda.current_stream(self.device) if self.device.type == 'cuda' else None
        packed = [(si, s) for si, s in enumerate(self.stages) if self._stage_packed(s)]

        def pack(si, s):
            n = s.f_in_pad * 8 * s.Hp
            w = self.params[s.oW:s.oW + n]
            amax = ops.absmax(w, self._buf('wamax%d' % si, (1,)))
            ops.pack_hl(self.params, s.f_in_pad, 8 * s.Hp, src_off=s.oW, absmax=amax,
                        r=self._planes('Wn%d' % si, s.f_in_pad, 8 * s.Hp),
                        c=self._planes('Wt%d' % si, 8 * s.Hp, s.f_in_pad))
        aside = (self._side is not None and main is not None and len(packed) > 1
                 and os.environ.get('ASR_PACK_W_ASIDE', '1') != '0')
        for k, (si, s) in enumerate(packed):
            if k == 0 or not aside:
                pack(si, s)
        if aside:
            start = torch.cuda.Event()
            start.record(main)              # (the weights are final on the calling stream here)
            with torch.cuda.stream(self._side):
                self._side.wait_event(start)
                for si, s in packed[1:]:
                    pack(si, s)
                    ev = torch.cuda.Event()
                    ev.record(self._side)
                    self._w_ready[si] = ev

    def _await_weights(self, si):
        """The calling stream waits for stage si's weight planes (packed on the side stream)."""
        ev = getattr(self, '_w_ready', {}).pop(si, None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def _pack_input(self, si, s, a, BW, rows, n_pad, amax):
        """The stage's input slab (rows, f_in_pad) [x B_W of each direction] -> (rows, f_in_pad)
        planes: x@W reduces over their columns, dW = x^T dz over their rows (asr_gemm_hl's
        k_major form reads them transposed out of LDS -- no second orientation is packed).
        One entry per direction when masks are on, else one shared entry."""
        r0 = self._planes('ar%d_0' % si, rows, s.f_in_pad)
        if BW is None:
            ops.pack_hl(a, rows, s.f_in_pad, absmax=amax, r=r0)
            return [r0]
        # both directions' planes in ONE pass over the slab (the source is read once)
        r1 = self._planes('ar%d_1' % si, rows, s.f_in_pad)
        ops.pack_hl(a, rows, s.f_in_pad, mask=BW[0], mask_period=n_pad, absmax=amax, r=r0,
                    mask2=BW[1], r2=r1)
        return [r0, r1]

    def _gate_gemm_hl(self, s, si, pa, zx, rows):
        """zx = (a (.) B_W) @ W + b from packed planes (both directions in one GEMM without
        masks, one GEMM per direction with them)."""
        Hp = s.Hp
        self._await_weights(si)
        Wt = self._planes('Wt%d' % si, 8 * Hp, s.f_in_pad)
        bias = self._view(s.ob, 8 * Hp)
        if len(pa) == 1:
            ops.gemm_hl(pa[0], Wt, zx, rows, 8 * Hp, s.f_in_pad, bias=bias)
            return
        for d in range(2):
            ops.gemm_hl(pa[d], Wt, zx, rows, 4 * Hp, s.f_in_pad, b_row=d * 4 * Hp,
                        c_off=d * 4 * Hp, ldc=8 * Hp, bias=bias[d * 4 * Hp:(d + 1) * 4 * Hp])

    # ------------------------------------------------------------------ forward
    def forward(self, x, training=False, masks=None, need_grad=True, n_valid=0):
        """x: (T, n_pad, F) float32 CUDA slab -> logits (T, n_pad, C).
        need_grad=False (evaluation / prediction) skips what only BPTT would read; n_valid=1
        with it (one utterance, predict.py) selects the tile-free recurrent kernel.

        masks: optional explicit variational-dropout masks (parity tests):
        {stage_index: (BW (2, n_pad, f_in_pad), BU (2, n_pad, Hp))}.
        """
        T, n_pad, F = x.shape
        assert F in (self.num_features, self.f_in_pad0) and n_pad % 16 == 0
        rows = T * n_pad
        if F != self.f_in_pad0:
            # (T, n_pad, F) -> zero-padded (T, n_pad, pad4(F)); the pad columns of the
            # buffer are written once (zeros) and never touched again
            key = ('xpad', (T, n_pad, self.f_in_pad0))
            fresh = key not in self._bufs
            xp = self._buf('xpad', (T, n_pad, self.f_in_pad0))
            if fresh:
                xp.zero_()
            xp[:, :, :F].copy_(x)
            x = xp
        a = x
        self._acts = []
        drawn = [None]
        nb = 0
        if self._overlap_mode == 'auto':
            self.overlap = not self._recurrence_fills_chip(n_pad)
        self._pipe_now = self._pipeline_on(n_pad)
        # (the packed-operand GEMMs take whole slabs: no frame-range pipelining with them)
        t_rec = self.out_frames(T)              # frames the recurrent stack sees
        pipe = (self._pipe_now and self._pipe is not None and self.lstm_mode == 0 and t_rec >= 16
                and not self.packed)
        self._pipe_now = self._pipe_now and not self.packed
        if n_valid == 1 and not need_grad:      # one utterance: the tile-free kernel, whole layers
            pipe = False
        pre = {}
        if any(self._stage_packed(st) for st in self.stages):
            self._pack_weights()

        def stage_masks(i):
            st = self.stages[i]
            if masks is not None and i in masks:
                return masks[i][:2]
            if training and (st.dropout_W > 0 or st.dropout_U > 0):
                if drawn[0] is None:
                    drawn[0] = self._draw_all_masks(n_pad)
                return drawn[0][i]
            return None, None
        for si, s in enumerate(self.stages):
            rec = {'in': a}
            T = a.shape[0]                      # (a time-strided convolution shortens the slab)
            rows = T * n_pad
            S = (self._pipe_split16 * T) // 16
            if s.kind == 'reshape':
                pass
            elif s.kind == 'conv':
                op = self._conv_op(si, s, T, n_pad)
                # clipped ReLU: applied in the GEMM epilogue, only y exists (its mask
                # 0 < z < clip reads the same from y); linear: y is z
                y = self._buf('convy%d' % si, (op.T_out, n_pad, s.F_out * s.C_out))
                nw = s.kt * s.kf * s.C_in * s.C_out
                op.fwd(a.contiguous(), self._view(s.oW, nw), self._view(s.ob, s.C_out),
                       None if s.clip > 0 else y, y, x_absmax=self._clip_bound(si))
                rec.update(op=op, z=y)
                a = y
            elif s.kind == 'noise':
                if training and s.value > 0:
                    out = self._buf('noise%d' % si, a.shape)
                    a = ops.gaussian_noise(a, out, s.value, self.rng_seed, 4 * si, self._step)
                    if s.f_out_pad != s.f_out and si == 0:
                        a[:, :, s.f_out:] = 0.0        # keep the input pad columns zero
            elif s.kind == 'dropout':
                if training and s.value > 0:
                    keep = self._buf('dropmask%d' % si, a.shape)
                    out = self._buf('dropout%d' % si, a.shape)
                    a = ops.dropout_apply(a, out, keep, s.value, 1.0 / (1.0 - s.value),
                                          self.rng_seed, 4 * si, self._step)
                    rec['mask'] = keep
            elif s.kind == 'merge':       # residual: c * (new + skip)
                out = self._buf('merge%d' % si, a.shape)
                a = ops.axpby(s.coef, a, s.coef, self._acts[s.skip]['out'], out)
            elif s.kind == 'dense':
                out = self._buf('dense%d' % si, (T, n_pad, s.n_out))
                ops.gemm(a, self.params, out, rows, s.n_out, s.f_in_pad, b_off=s.oW,
                         bias=self._view(s.ob, s.n_out))
                a = out
            elif s.kind == 'bilstm':
                Hp = s.Hp
                BW, BU = stage_masks(si)[:2]
                rec['BW'], rec['BU'] = BW, BU
                var = self._variant_args(s, si, T, n_pad, training, masks)
                rec['var'] = var
                if s.mi is None and s.ln is None:
                    zx = self._buf('zx%d_%d' % (nb % 2, Hp), (T, n_pad, 2, 4 * Hp))
                else:       # x@W is needed again by BPTT: one buffer per layer
                    zx = self._buf('zxmi%d' % si, (T, n_pad, 2, 4 * Hp))
                    rec['zx'] = zx
                nb += 1
                main = torch.cuda.current_stream(self.device)
                inner_done, halves = pre.pop(si, (None, False))
                if self._stage_packed(s):
                    # |y| < 1 behind a BiLSTM stage with a bounded activation; anything else
                    # is measured
                    prev = self.stages[si - 1] if si > 0 else None
                    bound = self._clip_bound(si)
                    amax = self._const_one() if (prev is not None and self._y_bounded(prev)) \
                        else (bound if bound is not None
                              else ops.absmax(a, self._buf('aamax%d' % si, (1,))))
                    rec['pa'] = self._pack_input(si, s, a, BW, rows, n_pad, amax)
                    self._gate_gemm_hl(s, si, rec['pa'], zx, rows)
                elif inner_done is None:
                    self._gate_gemm(a, s, zx, BW, 0, rows, n_pad)
                elif halves:
                    # frames [T-S, S) were projected while the previous layer ran, and of the
                    # others the half of the reduction that was final by then (below): what is
                    # left on the critical path is the other half of those frames
                    main.wait_event(inner_done)
                    # (r6) the two row ranges are independent and each is two launch-bound GEMMs
                    # of ~25 us: one range on the pipe stream (idle by now), the other here
                    fork = (self._pipe is not None
                            and os.environ.get('ASR_PIPE_TAIL_FORK', '1') != '0')
                    if fork:
                        ev0 = torch.cuda.Event()
                        ev0.record(main)
                        with torch.cuda.stream(self._pipe):
                            self._pipe.wait_event(ev0)
                            self._gate_gemm_half(a, s, zx, BW, S * n_pad, rows, n_pad, 0, False)
                            ev1 = torch.cuda.Event()
                            ev1.record(self._pipe)
                        self._gate_gemm_half(a, s, zx, BW, 0, (T - S) * n_pad, n_pad, 1, False)
                        main.wait_event(ev1)
                    else:
                        self._gate_gemm_half(a, s, zx, BW, 0, (T - S) * n_pad, n_pad, 1, False)
                        self._gate_gemm_half(a, s, zx, BW, S * n_pad, rows, n_pad, 0, False)
                else:       # frames [T-S, S) were projected while the previous layer ran
                    self._gate_gemm(a, s, zx, BW, 0, (T - S) * n_pad, n_pad)
                    self._gate_gemm(a, s, zx, BW, S * n_pad, rows, n_pad)
                    main.wait_event(inner_done)
                # (the single-utterance kernel writes row 0 only: no junk in the padding rows)
                one = n_valid == 1 and not need_grad
                y = self._buf('y%d' % si, (T, n_pad, 2 * Hp), zero=one)
                cell = self._buf('cell%d' % si, (T, n_pad, 2, Hp), zero=one)
                gates = self._buf('gates%d' % si, (T, n_pad, 2, 4 * Hp), zero=one)
                U = self._view(s.oU, 2 * Hp * 4 * Hp)
                nxt = self.stages[si + 1] if si + 1 < len(self.stages) else None
                if s.ln is not None:        # generic row-per-workgroup cell (csrc/lstm_ln.hip)
                    uh = self._buf('uh%d' % si, (T, n_pad, 2, 4 * Hp))
                    rec['uh'] = uh
                    ops.lstm_ln_seq_fwd(zx, U, self._view(s.ocell, 68 * Hp), uh, y, cell, gates, T,
                                        n_pad, Hp, has_mi=s.mi is not None, mask_u=BU,
                                        zone_c=var.get('zone_c'), zone_h=var.get('zone_h'),
                                        act=s.act)
                elif pipe and nxt is not None and nxt.kind == 'bilstm' and not var \
                        and nxt.mi is None and nxt.ln is None:
                    # after S = 13T/16 steps the frames [T-S, S) of y are final in BOTH
                    # directions: the next layer's input projection of those frames runs
                    # on the pipe stream while this recurrence finishes its last steps
                    # (nothing runs beside the first S steps: the library's choice of geometry;
                    # the GEMMs of the next layer run beside the last T - S: sixteen units per
                    # workgroup there, i.e. the fewest CUs)
                    ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, Hp, mask_u=BU,
                                     mode=self.lstm_mode, steps=(0, S))
                    ev = torch.cuda.Event()
                    ev.record(main)
                    zx_n = self._buf('zx%d_%d' % (nb % 2, nxt.Hp), (T, n_pad, 2, 4 * nxt.Hp))
                    with torch.cuda.stream(self._pipe):
                        self._pipe.wait_event(ev)
                        BWn = stage_masks(si + 1)[0]
                        self._gate_gemm(y, nxt, zx_n, BWn, (T - S) * n_pad, S * n_pad, n_pad)
                        # (the next stage's input must be exactly [y_f, y_b] of this one)
                        halves = self._pipe_halves and nxt.f_in_pad == 2 * Hp
                        if halves:
                            # x = [y_f, y_b]: after S steps y_f is final on the frames [0, T-S)
                            # too and y_b on [S, T) -- their halves of the reduction
                            # (+ bias) go ahead as well
                            self._gate_gemm_half(y, nxt, zx_n, BWn, 0, (T - S) * n_pad, n_pad,
                                                 0, True)
                            self._gate_gemm_half(y, nxt, zx_n, BWn, S * n_pad, rows, n_pad,
                                                 1, True)
                        done = torch.cuda.Event()
                        done.record(self._pipe)
                    pre[si + 1] = (done, halves)
                    rec['ws'] = ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, Hp, mask_u=BU,
                                                 mode=self.lstm_mode, steps=(S, T - S), units=16)
                else:
                    rec['ws'] = ops.lstm_seq_fwd(
                        zx, U, y, cell, gates, T, n_pad, Hp, mask_u=BU, mode=self.lstm_mode,
                        n_valid=n_valid if not need_grad else 0, **var)
                rec.update(y=y, cell=cell, gates=gates)
                a = y
            rec['out'] = a
            self._acts.append(rec)
        return a

    def _clip_bound(self, si):
        """1-element device tensor >= max|input of stage si| when that input is a clipped-ReLU
        convolution's output (0 <= y <= clip), seen through reshapes; else None (measure)."""
        j = si - 1
        while j >= 0 and self.stages[j].kind == 'reshape':
            j -= 1
        if j < 0 or self.stages[j].kind != 'conv' or not self.stages[j].clip > 0:
            return None
        key = ('clipbound', float(self.stages[j].clip))
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.full((1,), float(self.stages[j].clip),
                                              dtype=torch.float32, device=self.device)
        return t

    def _conv_op(self, si, s, T, n_pad):
        """The ops.Conv2d of stage si for this slab shape (owns the layer's workspace: the
        packed planes of x and dz live there between the forward and backward calls)."""
        key = (si, int(T), int(n_pad))
        op = self._convs.get(key)
        if op is None:
            for k in [k for k in self._convs if k[0] == si]:
                del self._convs[k]
            op = self._convs[key] = ops.Conv2d(T, n_pad, s.F_in, s.C_in, s.C_out, s.kt, s.kf,
                                               s.st, s.sf, s.clip, self.device)
        return op

    def out_frames(self, T):
        """Frames of the logits for T input frames (ceil(T / st) per time-strided layer)."""
        for st in self.time_strides:
            T = -(-int(T) // st)
        return int(T)

    def out_lengths(self, lens):
        """inputs_length -> lengths on the logits' time axis (host array or device tensor)."""
        for st in self.time_strides:
            if torch.is_tensor(lens):
                lens = torch.div(lens + (st - 1), st, rounding_mode='floor').to(lens.dtype)
            else:
                lens = -(-np.asarray(lens) // st)
        return lens

    def _recurrence_fills_chip(self, n_pad):
        if self.device.type != 'cuda':
            return False
        if not hasattr(self, '_num_cu'):
            self._num_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        widest = max([(st.Hp + 15) // 16 for st in self.stages if st.kind == 'bilstm'] + [0])
        return 2 * (n_pad // 16) * widest >= self._num_cu

    def _bptt_compact(self, s, n_pad):
        """True if stage s's BPTT may run in the compact geometry with GEMMs on the CUs it
        leaves free (plain cell, persistent kernels, H = 256 / 512, at most half of the CUs)."""
        if self._compact_mode == '0' or self.device.type != 'cuda' or self.lstm_mode != 0:
            return False
        if s.kind != 'bilstm' or s.mi is not None or s.ln is not None or s.Hp not in (256, 512):
            return False
        if s.zoneout_c > 0 or s.zoneout_h > 0 or s.act != 'tanh':
            return False
        if os.environ.get('ASR_LSTM_PREC', '1') == '0':
            return False
        if not hasattr(self, '_num_cu'):
            self._num_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        if self._compact_mode == '1':
            return True
        return (self._recurrence_fills_chip(n_pad)
                and 2 * (n_pad // 16) * (s.Hp // 32) <= self._num_cu // 2)

    def _pipeline_on(self, n_pad):
        if not self.overlap or self._pipeline_mode == '0':
            return False
        if self._pipeline_mode == '1':
            return True
        if not hasattr(self, '_num_cu'):
            self._num_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        widest = max([(st.Hp + 15) // 16 for st in self.stages if st.kind == 'bilstm'] + [0])
        return 0 < 2 * (n_pad // 16) * widest <= self._num_cu // 2

    def _variant_args(self, s, si, T, n_pad, training, masks):
        """Keyword arguments of the optional cell variants for ops.lstm_seq_fwd (and, with
        the backward extras added later, lstm_seq_bwd): {} for the plain cell."""
        var = {}
        if s.act != 'tanh' and s.ln is None:        # any other activation: the variant kernels
            var['act'] = s.act
        if s.mi is not None and s.ln is None:
            var['mi'] = self._view(s.omi, 32 * s.Hp)
            var['uh'] = self._buf('uh%d' % si, (T, n_pad, 2, 4 * s.Hp))
        if s.zoneout_c > 0 or s.zoneout_h > 0:
            if masks is not None and si in masks and len(masks[si]) == 4:
                kc, kh = masks[si][2], masks[si][3]          # explicit (T, 2, Hp) coefficients
            else:
                def coef(level, name):
                    if not (0 < level < 1):
                        return None
                    buf = self._buf('%s%d' % (name, si), (T, 2, s.Hp))
                    if training:    # keep mask, shared over the batch, fresh every frame
                        ops.dropout_masks(buf, level, 1.0, self.rng_seed,
                                          4 * si + (2 if name == 'zonec' else 3), self._step)
                    else:           # test phase: (h - h_prev) * (1 - level) + h_prev
                        buf.fill_(1.0 - level)
                    return buf
                kc, kh = coef(s.zoneout_c, 'zonec'), coef(s.zoneout_h, 'zoneh')
            if kc is not None:
                var['zone_c'] = kc
            if kh is not None:
                var['zone_h'] = kh
        return var

    def _gate_gemm(self, a, s, zx, BW, r0, r1, n_pad):
        """zx[r0:r1] = (a[r0:r1] (.) B_W) @ W + b over slab rows [r0, r1) (whole frames)."""
        m, Hp = r1 - r0, s.Hp
        if m <= 0:
            return
        # with multiplicative integration the bias enters inside the cell (z = alpha Wx Uh
        # + beta1 Uh + beta2 Wx + b), so the projection is the bare product
        bias = self._view(s.ob, 8 * Hp) if (s.mi is None and s.ln is None) else None
        if BW is None:
            ops.gate_gemm('fwd', m, n_pad, s.f_in_pad, 8 * Hp, self.params, s.oW, 8 * Hp, 8 * Hp,
                          x=a, x_off=r0 * s.f_in_pad, bias=bias, zx=zx, z_off=r0 * 8 * Hp)
            return
        for d in range(2):      # each direction has its own input mask (two Keras layers)
            ops.gate_gemm('fwd', m, n_pad, s.f_in_pad, 4 * Hp, self.params, s.oW + d * 4 * Hp,
                          8 * Hp, 8 * Hp, x=a, x_off=r0 * s.f_in_pad,
                          bias=None if bias is None else bias[d * 4 * Hp:(d + 1) * 4 * Hp],
                          mask_w=BW[d], zx=zx, z_off=r0 * 8 * Hp + d * 4 * Hp)

    def _gate_gemm_half(self, a, s, zx, BW, r0, r1, n_pad, half, first):
        """One half of the reduction of the projection over slab rows [r0, r1): the input is
        [y_f, y_b] of the BiLSTM stage below, half 0 = its forward direction's columns, 1 = the
        backward direction's.  first: zx = partial + b, else zx += partial."""
        m, Hp, kh = r1 - r0, s.Hp, s.f_in_pad // 2
        if m <= 0:
            return
        bias = self._view(s.ob, 8 * Hp) if first else None
        for d in (range(2) if BW is not None else (None,)):
            n, c0 = (8 * Hp, 0) if d is None else (4 * Hp, d * 4 * Hp)
            ops.gemm(a, self.params, zx, m, n, kh, lda=s.f_in_pad, ldb=8 * Hp, ldc=8 * Hp,
                     a_off=r0 * s.f_in_pad + half * kh, b_off=s.oW + half * kh * 8 * Hp + c0,
                     c_off=r0 * 8 * Hp + c0, beta=0.0 if first else 1.0,
                     bias=None if bias is None else bias[c0:c0 + n],
                     a_scale=None if d is None else BW[d], a_scale_period=n_pad,
                     a_scale_off=half * kh, a_scale_ld=s.f_in_pad)

    def _dx_gemm_dir(self, dz, s, dx, BW, r0, r1, n_pad, zmx, d, beta):
        """Direction d's term of _dx_gemm over slab rows [r0, r1): dx = beta dx + B_W[d] (.)
        (dz_d @ W_d^T)."""
        m, Hp = r1 - r0, s.Hp
        if m <= 0:
            return
        ops.gate_gemm('dgrad', m, n_pad, s.f_in_pad, 4 * Hp, self.params, s.oW + d * 4 * Hp,
                      8 * Hp, 8 * Hp, mask_w=None if BW is None else BW[d], dz=dz,
                      z_off=r0 * 8 * Hp + d * 4 * Hp, dz_absmax=zmx, dx=dx,
                      dx_off=r0 * s.f_in_pad, dx_beta=beta)

    def _dx_gemm(self, dz, s, dx, BW, r0, r1, n_pad, zmx):
        """dx[r0:r1] = sum_d B_W[d] (.) (dz_d[r0:r1] @ W_d^T) over slab rows [r0, r1)."""
        m, Hp = r1 - r0, s.Hp
        if m <= 0:
            return
        if BW is None:
            ops.gate_gemm('dgrad', m, n_pad, s.f_in_pad, 8 * Hp, self.params, s.oW, 8 * Hp,
                          8 * Hp, dz=dz, z_off=r0 * 8 * Hp, dz_absmax=zmx, dx=dx,
                          dx_off=r0 * s.f_in_pad)
            return
        for d in range(2):
            ops.gate_gemm('dgrad', m, n_pad, s.f_in_pad, 4 * Hp, self.params, s.oW + d * 4 * Hp,
                          8 * Hp, 8 * Hp, mask_w=BW[d], dz=dz, z_off=r0 * 8 * Hp + d * 4 * Hp,
                          dz_absmax=zmx, dx=dx, dx_off=r0 * s.f_in_pad,
                          dx_beta=0.0 if d == 0 else 1.0)

    def _draw_all_masks(self, n_pad):
        """Variational-dropout masks of every BiLSTM stage for one batch (core/models.py:265-266:
        one mask per batch and direction, constant over time, inverted scaling), from the
        library's counter-based streams: {stage: (B_W (2, n_pad, in), B_U (2, n_pad, H))} with
        stream id 4 * stage (B_W) / 4 * stage + 1 (B_U) at step self._step."""
        out = {}
        for si, s in enumerate(self.stages):
            if s.kind != 'bilstm' or not (s.dropout_W > 0 or s.dropout_U > 0):
                continue
            BW = self._buf('BW%d' % si, (2, n_pad, s.f_in_pad))
            BU = self._buf('BU%d' % si, (2, n_pad, s.Hp))
            for k, (t, p) in enumerate(((BW, s.dropout_W), (BU, s.dropout_U))):
                if p > 0:
                    ops.dropout_masks(t, p, 1.0 / (1.0 - p), self.rng_seed, 4 * si + k, self._step)
                else:
                    t.fill_(1.0)
            out[si] = (BW, BU)
        return out

    # ------------------------------------------------------------------ backward
    def backward(self, dlogits):
        """dlogits (T, n_pad, C) -> fills self.grads (raw, no l2, no clip)."""
        T, n_pad, _ = dlogits.shape
        rows = T * n_pad
        da = dlogits
        split = 'auto'
        pending = []
        if not hasattr(self, '_dz_free'):
            self._dz_free = [None, None]

        # ASR_AR_OVERLAP: auto (default) = per-layer asynchronous all-reduces beside the BPTT of
        # the layers below ONLY while a recurrence leaves CUs free (cfg2: 64 of 256).  Where a
        # recurrence fills the chip with no compact schedule (ASR_BPTT_COMPACT=0) its
        # spin-waiting workgroups must all be resident, and an RCCL kernel that takes CUs first
        # stalls the whole chain: there the gradients are reduced by ONE collective over the flat
        # buffer after BPTT (110.6 MB: 0.6-2.5 ms of a ~40 ms step).  1 = always overlap, 0 = never.  The decision must be the SAME on every
        # rank (mismatched collective sequences hang RCCL): it is taken on the rank-invariant
        # reference shard pad16(ceil(n_global / world)), never on this rank's own n_pad -- the
        # shards of a ragged last batch differ by one utterance and can straddle a multiple of 16.
        # Round 6: the COMPACT backward schedule leaves half of the CUs to the side stream during
        # every BPTT below the top layer, so the same per-layer collectives run there too: layer
        # l + 1's bucket goes out on the communicator's stream once its weight-gradient GEMMs have
        # finished on the side stream, i.e. beside the compact BPTT of layer l - 1 (the top
        # layer's chip-filling BPTT comes first and has no collective beside it); what is left
        # for the end of the step is the bottom layer + Dense + the flag slots.  compact_any is
        # a function of the reference shard as well.
        ar_mode = os.environ.get('ASR_AR_OVERLAP', 'auto')
        n_ref = self._ar_ref_pad if self._ar_ref_pad else n_pad
        compact_any = any(self._bptt_compact(st, n_ref) for st in self.stages)
        reduce_now = (self._dist_active() and ar_mode != '0'
                      and (ar_mode == '1' or not self._recurrence_fills_chip(n_ref)
                           or compact_any))
        self._ar_covered = []
        self._ar_decision = reduce_now

        def reduce_async(lo, hi, after):
            # this slice of the gradients is final once the work on stream `after` is done: its
            # all-reduce (on the communicator's own stream) runs beside the BPTT of the layers below
            from ..parallel import grad_comm
            grad_comm(self.device).allreduce_after(self.grads[lo:hi], after)
            self._ar_covered.append((lo, hi))

        def flush_side():
            # enqueue deferred weight-gradient work on the side stream (called right
            # AFTER the next layer's BPTT kernel has been launched on the main stream)
            while pending:
                fn, ready, par, rng = pending.pop(0)
                with torch.cuda.stream(self._side):
                    self._side.wait_event(ready)
                    fn('gemm_side')
                    ev = torch.cuda.Event()
                    ev.record(self._side)
                    self._dz_free[par] = ev
                if reduce_now and rng is not None:
                    reduce_async(rng[0], rng[1], self._side)

        # the side stream is in use if the recurrences leave CUs free by themselves (cfg2) or
        # are made to (compact BPTT geometry, cfg3)
        # (decided, like the collective schedule, on the RANK-INVARIANT reference shard: every
        # rank then walks the same branches below, whatever its own shard of a ragged batch)
        overlap = self.overlap or compact_any
        self._compact_launches = 0
        # split-K of the weight-gradient GEMMs that run beside a compact BPTT: sized to the CUs
        # that are free there (ASR_SIDE_SLOTS, measurement switch; 0 = as on the whole chip)
        side_slots = int(os.environ.get('ASR_SIDE_SLOTS', '0'))
        split_side = ('auto:%d' % side_slots) if (compact_any and not self.overlap and side_slots) \
            else split
        skip_grads = {}        # stage index -> gradient to add to that stage's OUTPUT
        for si in range(len(self.stages) - 1, -1, -1):
            s = self.stages[si]
            rec = self._acts[si]
            a_in = rec['in']
            if si in skip_grads:            # the residual branch rejoins here
                g = skip_grads.pop(si)
                da = ops.axpby(1.0, da, 1.0, g, self._buf('dskip%d' % si, da.shape))
            if s.kind == 'merge':
                if s.coef != 1.0:
                    da = ops.axpby(s.coef, da, 0.0, da, self._buf('dmerge%d' % si, da.shape))
                skip_grads[s.skip] = da
                continue
            first = not any(st.kind in ('dense', 'bilstm', 'conv') for st in self.stages[:si])
            if s.kind in ('noise', 'reshape'):
                continue
            if s.kind == 'conv':
                flush_side()
                op, z = rec['op'], rec['z']
                nw = s.kt * s.kf * s.C_in * s.C_out
                da = da.contiguous()
                dz_ready = False
                if not first:
                    dx = self._buf('da_s%d' % si, a_in.shape)
                    op.dgrad(da, z, self._view(s.oW, nw), dx)
                    dz_ready = True
                # (x planes: still in the layer's workspace from the forward call)
                op.wgrad(a_in, da, z, self._gview(s.oW, nw), self._gview(s.ob, s.C_out),
                         reuse_x=True, reuse_dz=dz_ready)
                if not first:
                    da = dx
                continue
            if s.kind == 'dropout':
                if first:               # nothing trainable upstream: no gradient needed there
                    continue
                if 'mask' in rec:
                    da = ops.mul(da, rec['mask'], self._buf('ddrop%d' % si, da.shape))
                continue
            if s.kind == 'dense':
                gmx = ops.absmax(da, self._buf('damax%d' % si, (1,)))
                ops.gemm(a_in, da, self.grads, s.f_in_pad, s.n_out, rows, trans_a=True,
                         c_off=s.oW, split_k=split, b_absmax=gmx)
                ops.colsum(da, rows, s.n_out, s.n_out, self._gview(s.ob, s.n_out))
                if not first:
                    dx = self._buf('da_s%d' % si, (T, n_pad, s.f_in_pad))
                    ops.gemm(da, self.params, dx, rows, s.f_in_pad, s.n_out, trans_b=True,
                             b_off=s.oW, a_absmax=gmx)
                    da = dx
            elif s.kind == 'bilstm':
                Hp = s.Hp
                BW, BU = rec['BW'], rec['BU']
                # two dz buffers alternate so that the side stream may still read layer
                # l's dz while layer l-1's BPTT writes the other one
                par = self._dz_parity = 1 - getattr(self, '_dz_parity', 0)
                dz = self._buf('dz%d' % par, (T, n_pad, 2, 4 * Hp))
                main = torch.cuda.current_stream(self.device)
                if overlap and self._dz_free[par] is not None:
                    main.wait_event(self._dz_free[par])
                U = self._view(s.oU, 2 * Hp * 4 * Hp)
                zmx = self._buf('dzmax%d' % par, (1,))
                var = dict(rec.get('var') or {})
                gsrc = dz                  # slab the dW / dX GEMMs read
                pgrad = None               # (per-row parameter-gradient sums, rows, cols, offset)
                if s.ln is not None:
                    gsrc = self._buf('dwx%d' % par, (T, n_pad, 2, 4 * Hp))
                    dpar = self._buf('dcellp%d' % si, (n_pad, 2, 34 * Hp))
                    pgrad = (dpar, n_pad, 68 * Hp, s.ocell)
                elif s.mi is not None:
                    gsrc = self._buf('dwx%d' % par, (T, n_pad, 2, 4 * Hp))
                    dmi = self._buf('dmi%d' % si, (n_pad // 16, 2, 4, 4 * Hp))
                    var.update(wx=rec['zx'], dwx=gsrc, dmi=dmi)
                    pgrad = (dmi, n_pad // 16, 32 * Hp, s.omi)
                else:
                    # plain cell: BPTT leaves the bias gradient as per-batch-tile sums of dz
                    # (accumulated in registers over the steps), so nothing re-reads the dz
                    # slab for it; the tiles are added up on the side stream
                    dbp = self._buf('dbpart%d' % si, (n_pad // 16, 2, 4 * Hp))
                    pgrad = (dbp, n_pad // 16, 8 * Hp, s.ob)
                planes_now, pdz_r = False, None
                pipe_b = (getattr(self, '_pipe_now', False) and self._pipe is not None
                          and not first and self.lstm_mode == 0 and T >= 16 and not var
                          and s.ln is None)
                S = (self._pipe_split16 * T) // 16
                dx = None
                if pipe_b:
                    # after S BPTT steps the gate gradients of frames [T-S, S) are final
                    # in both directions: their dX GEMMs overlap the last steps
                    dx = self._buf('da_s%d' % si, (T, n_pad, s.f_in_pad))
                    ops.lstm_seq_bwd(da, U, rec['cell'], rec['gates'], dz, T, n_pad, Hp,
                                     mask_u=BU, mode=self.lstm_mode, dz_absmax=zmx, steps=(0, S),
                                     db_part=pgrad[0])
                    ev = torch.cuda.Event()
                    ev.record(main)
                    flush_side()    # previous layer's dW/dU/db now overlap this BPTT
                    with torch.cuda.stream(self._pipe):
                        self._pipe.wait_event(ev)
                        self._dx_gemm(dz, s, dx, BW, (T - S) * n_pad, S * n_pad, n_pad, zmx)
                        if self._pipe_halves:
                            # the forward direction's BPTT (frames T-1 .. 0) has also finished
                            # [S, T), the backward direction's [0, T-S): their terms of dx
                            self._dx_gemm_dir(dz, s, dx, BW, S * n_pad, rows, n_pad, zmx, 0, 0.0)
                            self._dx_gemm_dir(dz, s, dx, BW, 0, (T - S) * n_pad, n_pad, zmx, 1,
                                              0.0)
                        dx_inner = torch.cuda.Event()
                        dx_inner.record(self._pipe)
                    rec['ws_b'] = ops.lstm_seq_bwd(da, U, rec['cell'], rec['gates'], dz, T, n_pad,
                                                   Hp, mask_u=BU, mode=self.lstm_mode,
                                                   dz_absmax=zmx, steps=(S, T - S),
                                                   db_part=pgrad[0])
                elif s.ln is not None:
                    ops.lstm_ln_seq_bwd(da, rec['zx'], U, self._view(s.ocell, 68 * Hp), rec['uh'],
                                        rec['y'], rec['cell'], rec['gates'], dz, gsrc, pgrad[0], T,
                                        n_pad, Hp, has_mi=s.mi is not None, mask_u=BU,
                                        zone_c=var.get('zone_c'), zone_h=var.get('zone_h'),
                                        act=s.act)
                    # one pre-scale for the gradient GEMMs: max over both gradient slabs
                    ops.absmax(dz, zmx)
                    tmp = ops.absmax(gsrc, self._buf('dzmax_t', (1,)))
                    torch.maximum(zmx, tmp, out=zmx)
                    flush_side()
                else:
                    if s.mi is None:
                        var['db_part'] = pgrad[0]
                    # compact geometry only while there is work to run beside it (the top
                    # layer's BPTT has none: it keeps the whole chip and the shorter step)
                    cmp_now = bool(pending) and not self.overlap and self._bptt_compact(s, n_ref)
                    self._compact_launches += int(cmp_now)
                    # BPTT writes dz as packed planes itself where its kernel can (the plain cell
                    # at H = 256 / 512 on the two-dimensional-split kernels): the pack pass over
                    # the fp32 slab (2.1 GB per layer at cfg3) disappears.  The planes' scale has
                    # to be known before the pass: a per-layer bound kept at ~64 x the measured
                    # max|dz| by asr_lstm_dz_guard (hysteresis: identical inputs see identical
                    # scales); a layer without a bound yet -- first step, after a fault or new
                    # weights -- runs one MEASURING pass first (fp32 slab, discarded).
                    plain = not any(var.get(k) is not None
                                    for k in ('act', 'zone_c', 'zone_h', 'mi', 'uh'))
                    planes_now = (self._dz_planes_mode and self._stage_packed(s) and plain
                                  and self.lstm_mode == 0
                                  and ops.lstm_dz_hl_supported(T, n_pad, Hp, compact=cmp_now))
                    if planes_now:
                        pdz_r = self._planes('dzr%d' % par, rows, 8 * Hp)
                        bound = self._buf('dzbound%d' % si, (1,))
                        key = (self._fault_gen, self._weights_epoch)
                        if self._dz_bound_key.get(si) != key:
                            bound.zero_()
                            ops.lstm_seq_bwd(da, U, rec['cell'], rec['gates'], dz, T, n_pad, Hp,
                                             mask_u=BU, mode=self.lstm_mode, dz_absmax=zmx,
                                             compact=cmp_now, **var)
                            ops.lstm_dz_guard(zmx, bound, False)
                            self._dz_bound_key[si] = key
                            self._dz_measure_passes += 1
                        rec['ws_b'] = ops.lstm_seq_bwd(
                            da, U, rec['cell'], rec['gates'], None, T, n_pad, Hp, mask_u=BU,
                            mode=self.lstm_mode, dz_absmax=zmx, compact=cmp_now,
                            dz_planes=pdz_r, dz_bound=bound, **var)
                        ops.lstm_dz_guard(zmx, bound, True)
                    else:
                        rec['ws_b'] = ops.lstm_seq_bwd(
                            da, U, rec['cell'], rec['gates'], dz, T, n_pad, Hp, mask_u=BU,
                            mode=self.lstm_mode, dz_absmax=zmx, compact=cmp_now, **var)
                    flush_side()    # previous layer's dW/dU/db now overlap this BPTT
                y = rec['y']
                hl = self._stage_packed(s)
                if hl and not planes_now:
                    # dz -> (rows, 8H) planes, once: dX reduces over their gate columns, the
                    # weight gradients over their rows (k_major); the scale is the BPTT kernel's
                    # own max|dz|
                    pdz_r = self._planes('dzr%d' % par, rows, 8 * Hp)
                    ops.pack_hl(dz, rows, 8 * Hp, absmax=zmx, r=pdz_r)
                # bound of |y| for its planes (dU): 1 behind a bounded activation, else measured
                # (here, on the main stream, before the side stream's packs read it)
                ymx = self._const_one() if (not hl or self._y_bounded(s)) \
                    else ops.absmax(y, self._buf('yamax%d' % si, (1,)))

                def grads_U_hl(wsn, s=s, y=y, BU=BU, Hp=Hp, pdz_r=pdz_r, ymx=ymx):
                    # dU[d] = (h_prev (.) B_U)^T dz[d], reduced over the plane rows; h_prev = y
                    # one frame earlier in the direction's processing order = a row offset
                    kk = (T - 1) * n_pad
                    for d in range(2):
                        if kk <= 0:
                            self._gview(s.oU + d * Hp * 4 * Hp, Hp * 4 * Hp).zero_()
                            continue
                        # (one pair of buffers per stream: the bottom layer's gradients run on
                        # the main stream while the side stream may still read its own pair)
                        yu = self._planes('yu%d%s' % (d, '' if wsn == 'gemm_side' else '_m'),
                                          rows, Hp)
                        ops.pack_hl(y, rows, Hp, ld=2 * Hp, src_off=d * Hp,
                                    mask=None if BU is None else BU[d], mask_period=n_pad,
                                    absmax=ymx, r=yu)
                        ops.gemm_hl(yu, pdz_r, self.grads, Hp, 4 * Hp, kk,
                                    a_row=0 if d == 0 else n_pad, b_k=d * 4 * Hp,
                                    b_row=n_pad if d == 0 else 0, c_off=s.oU + d * Hp * 4 * Hp,
                                    split_k=split_side if wsn == 'gemm_side' else split,
                                    ws_name=wsn, k_major=True)

                def grads_W_hl(wsn, s=s, pa=rec.get('pa'), Hp=Hp, pdz_r=pdz_r, pgrad=pgrad):
                    sp = split_side if wsn == 'gemm_side' else split
                    if len(pa) == 1:
                        ops.gemm_hl(pa[0], pdz_r, self.grads, s.f_in_pad, 8 * Hp, rows,
                                    c_off=s.oW, split_k=sp, ws_name=wsn, k_major=True)
                    else:
                        for d in range(2):
                            ops.gemm_hl(pa[d], pdz_r, self.grads, s.f_in_pad, 4 * Hp, rows,
                                        b_k=d * 4 * Hp, c_off=s.oW + d * 4 * Hp, ldc=8 * Hp,
                                        split_k=sp, ws_name=wsn, k_major=True)
                    buf, nrow, ncol, goff = pgrad
                    ops.colsum(buf, nrow, ncol, ncol, self._gview(goff, ncol), ws_name=wsn + '_cs')

                def grads_U(wsn, s=s, dz=dz, y=y, BU=BU, Hp=Hp, zmx=zmx):
                    # dU[d] = (h_prev (.) B_U)^T dz[d]: h_prev is y shifted by one step in
                    # the direction's processing order (zero at its first step)
                    kk = (T - 1) * n_pad
                    for d in range(2):
                        a_off = d * Hp + (0 if d == 0 else n_pad * 2 * Hp)
                        b_off = d * 4 * Hp + (n_pad * 8 * Hp if d == 0 else 0)
                        if kk > 0:
                            ops.gemm(y, dz, self.grads, Hp, 4 * Hp, kk, trans_a=True, lda=2 * Hp,
                                     ldb=8 * Hp, ldc=4 * Hp, a_off=a_off, b_off=b_off,
                                     c_off=s.oU + d * Hp * 4 * Hp, split_k=split,
                                     a_scale=None if BU is None else BU[d],
                                     a_scale_period=n_pad, ws_name=wsn, b_absmax=zmx)
                        else:
                            self._gview(s.oU + d * Hp * 4 * Hp, Hp * 4 * Hp).zero_()

                def grads_W(wsn, s=s, dz=gsrc, a_in=a_in, BW=BW, Hp=Hp, zmx=zmx, pgrad=pgrad):
                    # dW = (x (.) B_W)^T d(x@W); the bias (or MI / LN parameter) gradients come
                    # from BPTT's per-tile partial sums, never from a pass over the dz slab
                    if BW is None:
                        ops.gate_gemm('wgrad', rows, n_pad, s.f_in_pad, 8 * Hp, self.params, s.oW,
                                      8 * Hp, 8 * Hp, x=a_in, dz=dz, dz_absmax=zmx,
                                      dW=self.grads, dw_off=s.oW, split_k=split, ws_name=wsn)
                    else:
                        for d in range(2):
                            ops.gate_gemm('wgrad', rows, n_pad, s.f_in_pad, 4 * Hp, self.params,
                                          s.oW + d * 4 * Hp, 8 * Hp, 8 * Hp, x=a_in,
                                          mask_w=BW[d], dz=dz, z_off=d * 4 * Hp, dz_absmax=zmx,
                                          dW=self.grads, dw_off=s.oW + d * 4 * Hp, split_k=split,
                                          ws_name=wsn)
                    buf, nrow, ncol, goff = pgrad
                    ops.colsum(buf, nrow, ncol, ncol, self._gview(goff, ncol), ws_name=wsn + '_cs')

                if hl:
                    grads_U, grads_W = grads_U_hl, grads_W_hl

                def weight_grads(wsn, gu=grads_U, gw=grads_W):
                    gu(wsn)
                    gw(wsn)

                if pipe_b and self._pipe_halves:
                    main.wait_event(dx_inner)
                    # (r6: two independent row ranges, one launch-bound GEMM each: two streams)
                    if os.environ.get('ASR_PIPE_TAIL_FORK', '1') != '0':
                        ev0 = torch.cuda.Event()
                        ev0.record(main)
                        with torch.cuda.stream(self._pipe):
                            self._pipe.wait_event(ev0)
                            self._dx_gemm_dir(dz, s, dx, BW, 0, (T - S) * n_pad, n_pad, zmx, 0, 1.0)
                            ev1 = torch.cuda.Event()
                            ev1.record(self._pipe)
                        self._dx_gemm_dir(dz, s, dx, BW, S * n_pad, rows, n_pad, zmx, 1, 1.0)
                        main.wait_event(ev1)
                    else:
                        self._dx_gemm_dir(dz, s, dx, BW, S * n_pad, rows, n_pad, zmx, 1, 1.0)
                        self._dx_gemm_dir(dz, s, dx, BW, 0, (T - S) * n_pad, n_pad, zmx, 0, 1.0)
                    da = dx
                elif pipe_b:
                    self._dx_gemm(dz, s, dx, BW, 0, (T - S) * n_pad, n_pad, zmx)
                    self._dx_gemm(dz, s, dx, BW, S * n_pad, rows, n_pad, zmx)
                    main.wait_event(dx_inner)
                    da = dx
                elif not first and hl:
                    dx = self._buf('da_s%d' % si, (T, n_pad, s.f_in_pad))
                    Wn = self._planes('Wn%d' % si, s.f_in_pad, 8 * Hp)
                    if BW is None:
                        ops.gemm_hl(pdz_r, Wn, dx, rows, s.f_in_pad, 8 * Hp)
                    else:       # dx = sum_d B_W[d] (.) (dz_d @ W_d^T)
                        for d in range(2):
                            ops.gemm_hl(pdz_r, Wn, dx, rows, s.f_in_pad, 4 * Hp, a_k=d * 4 * Hp,
                                        b_k=d * 4 * Hp, c_scale=BW[d], c_scale_period=n_pad,
                                        beta=0.0 if d == 0 else 1.0)
                    da = dx
                elif not first:
                    dx = self._buf('da_s%d' % si, (T, n_pad, s.f_in_pad))
                    self._dx_gemm(gsrc, s, dx, BW, 0, rows, n_pad, zmx)
                    da = dx
                # (compact schedule: is there a BiLSTM stage below whose BPTT this layer's
                # weight gradients could run beside?  A convolution / Dense / nothing below:
                # chip-filling GEMMs either way -- one stream, as in the serial schedule)
                below = any(st.kind == 'bilstm' for st in self.stages[:si])
                if not overlap:
                    weight_grads('gemm')
                    if reduce_now and not first:
                        # no side stream (a recurrence fills the chip): this layer's gradients
                        # are final on the main stream; their all-reduce runs beside the layers
                        # below instead of after them all
                        reduce_async(s.p_lo, s.p_hi, main)
                elif (first or not below) and not self.overlap:
                    # (compact schedule: the tail's GEMMs each fill the chip -- one stream)
                    weight_grads('gemm')
                    if reduce_now and not first:
                        reduce_async(s.p_lo, s.p_hi, main)
                    if not first:
                        da = dx
                elif first:
                    # nothing left to hide behind: share the tail between both streams
                    ready = torch.cuda.Event()
                    ready.record(main)
                    pending.append((grads_U, ready, par, None))
                    flush_side()
                    grads_W('gemm')
                else:
                    # the side stream starts once the dX GEMMs (critical path) are done,
                    # i.e. together with the next layer's BPTT kernel
                    ready = torch.cuda.Event()
                    ready.record(main)
                    pending.append((weight_grads, ready, par, (s.p_lo, s.p_hi)))
                    da = dx
        flush_side()
        if overlap and self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)
        return da

    # ------------------------------------------------------------------ batches
    def _prep_labels(self, labels, seq_len, T):
        N = len(labels)
        lmax = max([len(l) for l in labels] + [1])
        lab = np.zeros((N, lmax), np.int32)
        seq_len = self.out_lengths(np.asarray(seq_len).reshape(-1))   # on the logits' time axis
        for n, l in enumerate(labels):
            l = np.asarray(l, np.int64).reshape(-1)
            need = len(l) + int(np.sum(l[1:] == l[:-1])) if len(l) else 0
            if need > int(seq_len[n]):
                raise ValueError('Not enough time for target transition sequence '
                                 '(required: %d, available: %d) in sample %d'
                                 % (need, int(seq_len[n]), n))
            lab[n, :len(l)] = l
        dev = self.device
        return (torch.from_numpy(lab).to(dev),
                torch.tensor([len(l) for l in labels], dtype=torch.int32, device=dev),
                torch.as_tensor(np.asarray(seq_len, np.int32)).to(dev))

    def to_slab(self, x_ntf):
        """(N, T, F) batch-major host/GPU array (the reference's layout) -> (T, n_pad, F)."""
        t = torch.as_tensor(np.asarray(x_ntf, np.float32)) if not torch.is_tensor(x_ntf) else x_ntf
        t = t.to(self.device, dtype=torch.float32)
        N, T, F = t.shape
        n_pad = ops.pad16(N)
        slab = torch.zeros((T, n_pad, F), dtype=torch.float32, device=self.device)
        slab[:, :N] = t.permute(1, 0, 2)
        return slab

    def _unpack_inputs(self, inputs):
        x, labels, lens = inputs[0], inputs[1], inputs[2]
        if hasattr(labels, 'tocsr'):            # scipy.sparse (the reference's batches)
            csr = labels.tocsr()
            N = x.shape[0] if not torch.is_tensor(x) or x.dim() == 3 else len(lens)
            labels = [csr.data[csr.indptr[i]:csr.indptr[i + 1]] if i < csr.shape[0] else []
                      for i in range(len(np.asarray(lens).reshape(-1)))]
        lens = np.asarray(lens).reshape(-1)
        if torch.is_tensor(x) and x.dim() == 3 and x.shape[1] % 16 == 0 and \
                getattr(x, '_asr_time_major', False):
            slab = x
        elif isinstance(x, tuple) and x[0] == 'slab':
            slab = x[1]
        else:
            slab = self.to_slab(x)
        return slab, [np.asarray(l).reshape(-1) for l in labels], lens

    def loss_and_grads(self, slab, labels, seq_len, training=True, masks=None, n_global=None,
                       n_ref=None):
        """One forward + CTC + backward.  Returns per-sample CTC loss (device) and
        logits; self.grads holds d(mean ctc)/d params."""
        lab, lab_len, sl = self._prep_labels(labels, seq_len, slab.shape[0])
        return self.loss_and_grads_device(slab, lab, lab_len, sl, len(labels), training, masks,
                                          n_global, n_ref)

    def loss_and_grads_device(self, slab, lab, lab_len, sl, N, training=True, masks=None,
                              n_global=None, n_ref=None):
        """Same with labels (N, l_max) / label_len / seq_len already on the device.
        n_global: samples of the GLOBAL batch (gradient scale 1/n_global; 0 = this rank holds a
        zero-weight dummy); n_ref: the global batch size again, for decisions every rank must
        take alike (it survives n_global = 0)."""
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        ng = int(n_ref or n_global or N * world)
        self._ar_ref_pad = ops.pad16((ng + world - 1) // world)
        logits = self.forward(slab, training=training, masks=masks)
        dlog = self._buf('dlogits', logits.shape)
        # n_global = 0: a zero-weight dummy shard (parallel.ShardedBatch.n_local == 0)
        ctc = ops.ctc_loss_grad(logits, lab, lab_len, sl, N, grad=dlog,
                                grad_scale=0.0 if n_global == 0 else 1.0 / float(n_global or N))
        self.backward(dlog)
        return ctc, logits, sl

    def train_step_device(self, slab, lab, lab_len, sl, N, world=1):
        """A full optimisation step with every input resident in HBM and no host
        synchronisation: forward, CTC, BPTT, (RCCL all-reduce), clip + update, greedy
        decode for the LER metric.  Returns device tensors (ctc, decoded, lengths)."""
        self._maybe_retry_persistent()
        sl = self.out_lengths(sl)
        ctc, logits, sl = self.loss_and_grads_device(slab, lab, lab_len, sl, N, training=True,
                                                     n_global=N * world)
        self._allreduce()
        self._step += 1
        self.optimizer.step(self)
        dec, dlen = ops.ctc_greedy(logits, sl, N)
        return ctc, dec, dlen

    def _dist_active(self):
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and (
            dist.get_world_size() > 1 or os.environ.get('ASR_FORCE_ALLREDUCE') == '1')

    # ------------------------------------------------------------------ timeouts
    def _collect_flags(self):
        """This rank's recurrent-kernel timeout flags -> the two flag slots behind the
        gradients (as 0.0 / 1.0), enqueued on the current stream."""
        ops.collect_timeout_flags(self._gbuf[self.n_params:], self.device)

    def veto_flags(self):
        """What the optimiser's guard looks at: data parallel, the all-reduced flag slots
        (non-zero on EVERY rank when any rank timed out, so all ranks skip the same update);
        single process, None = the workspaces' own sticky words."""
        return self._gbuf[self.n_params:self.n_params + 2] if self._dist_active() else None

    def _flag_snapshot(self):
        v = self.veto_flags()
        return ops.lstm_timeout_flags(self.device) if v is None else v.clone()

    def _maybe_retry_persistent(self):
        """After a fallback the stepwise kernels run for `_retry_gap` clean steps, then the
        persistent ones get another chance (every rank takes the same decision: the detection
        is collective)."""
        if (self.lstm_mode == 1 and not self._mode_pinned and self._retry_at is not None
                and self._step >= self._retry_at):
            self.lstm_mode = 0
            self._retry_at = None

    def _handle_timeout(self, flags, gen, train=True):
        """flags: a step's snapshot of the (collective) timeout flags, gen: self._fault_gen
        when that step was enqueued.  Returns True when the step was vetoed (its update did not
        happen and its activations are invalid): the caller drops its metrics.  Bookkeeping:
        the sticky flags stay set until they are cleared HERE, so every step enqueued between
        the fault and its detection -- in the lagged (sync=False) loop: this one and the one
        already in flight behind it -- was vetoed on the device; all of them are taken back
        from the optimiser's iteration count (Adam bias correction, decay) and from the
        noise-stream step AT ONCE, so no later step re-uses an iteration number.
        Evaluation (train=False) decides on this rank's own flags: no collective may be issued
        from test_on_batch (rank 0 alone runs the final test), and a rank that evaluates on the
        stepwise kernels for a while computes the same arithmetic as the others."""
        inflight = self._enqueued - self._checked       # train steps not yet checked (>= 1)
        if train:
            self._checked += 1
        if not bool(flags.any().item()):
            return False
        from .._lib import AsrHipError
        if gen < self._fault_gen:
            return True                 # enqueued before the fallback took effect: known
        if self.lstm_mode == 1:
            raise AsrHipError('a recurrent LSTM kernel reported a timeout in stepwise mode: '
                              'device fault')
        if train and self.optimizer is not None:
            n = max(1, inflight)
            self.optimizer.iterations = max(0, self.optimizer.iterations - n)
            self._step = max(0, self._step - n)
            self.vetoed_steps += n
        # A persistent kernel abandoned a bounded spin (a peer workgroup was not co-resident).
        # The update of every step enqueued since was vetoed on the device (ops.optim_guard),
        # so the weights are intact: clear the flags and go on with the stepwise kernels (one
        # launch per step, identical arithmetic, no co-residency) for a while.
        import logging
        logging.getLogger(__name__).warning(
            'persistent LSTM kernel timed out waiting for a peer workgroup; the affected '
            'step(s) were skipped; stepwise kernels (mode 1) for the next %d steps',
            self._retry_gap)
        torch.cuda.synchronize(self.device)
        ops.clear_timeout_flags(self.device)
        self._gbuf[self.n_params:].zero_()
        self.lstm_mode = 1
        self.fallbacks += 1
        self._fault_gen += 1
        self._retry_at = self._step + self._retry_gap
        self._retry_gap = min(2 * self._retry_gap, 1 << 14)
        return True

    def _allreduce(self):
        """Sums the gradients over the ranks: RCCL through the library's C ABI (asr_comm_*,
        parallel.CapiComm).  Layers whose weight gradients were finished during BPTT were already
        reduced there, asynchronously on the communicator's stream (backward(), only where a
        recurrence leaves CUs free); the rest of the flat buffer -- everything, where BPTT fills
        the chip -- is reduced here in place, TOGETHER with the two timeout-flag slots behind
        it, and the current stream then waits for all of them."""
        if not self._dist_active():
            return 1
        from ..parallel import grad_comm, world_size
        comm = grad_comm(self._gbuf.device)
        self._collect_flags()
        total = self.n_params + 4
        covered = sorted(getattr(self, '_ar_covered', []))
        pos = 0
        for lo, hi in covered + [(total, total)]:
            if lo > pos:
                comm.allreduce_after(self._gbuf[pos:lo], None)   # behind the current stream's work
            pos = max(pos, hi)
        comm.join(None)
        self._ar_covered = []
        return world_size()

    def train_on_batch(self, inputs, outputs=None, masks=None, sync=True):
        """One optimisation step on a batch ``[x, labels, inputs_length]``.

        Returns [loss, ctc_loss, decoder_loss, decoder_ler] when ``sync`` (like
        Keras), else the device tensors needed to compute them later."""
        assert self.optimizer is not None, 'compile() first'
        self._maybe_retry_persistent()
        slab, labels, lens = self._unpack_inputs(inputs)
        N = len(labels)
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # a rank's shard of a global batch says how many samples the GLOBAL batch had (shards
        # are unequal when the batch is not divisible by the world; parallel.ShardedBatch)
        n_global = n_ref = getattr(inputs, 'n_global', N * world)
        if getattr(inputs, 'n_local', N) == 0:
            n_global = 0
        gen = self._fault_gen
        ctc, logits, sl = self.loss_and_grads(slab, labels, lens, training=True, masks=masks,
                                              n_global=n_global, n_ref=n_ref)
        self._allreduce()
        self._step += 1
        self._enqueued += 1
        self.optimizer.step(self)
        dec, dlen = ops.ctc_greedy(logits, sl, N)
        if not sync:
            # snapshots of this step's small result tensors (the buffers behind them are
            # reused by the next step): metrics can then be fetched one step later
            return (ctc.clone(), dec.clone(), dlen.clone(), self._norm[1:2].clone(),
                    self._flag_snapshot(), gen)
        m = self._metrics(ctc, dec, dlen, labels, gen=gen)
        if m is None:
            # the step was vetoed (timeout): run it again, on the stepwise kernels
            return self.train_on_batch(inputs, outputs, masks=masks, sync=True)
        return m

    def _lagged(self, pending):
        """Metrics of a step enqueued earlier, or None if its update was vetoed."""
        (ctc, dec, dlen, pen, flags, gen), labels, _ = pending
        return self._metrics(ctc, dec, dlen, labels, pen=pen, flags=flags, gen=gen)

    def _metrics(self, ctc, dec, dlen, labels, hyps=None, pen=None, flags=None, gen=None,
                 train=True):
        ctc_h = ctc.cpu().numpy().astype(np.float64)
        # the persistent recurrent kernels bound every spin; one that gave up has left
        # invalid activations behind: handled here, at the step's host synchronisation point
        flags = self._flag_snapshot() if flags is None else flags
        if self._handle_timeout(flags, self._fault_gen if gen is None else gen, train):
            return None
        if pen is not None:
            pen = float(pen.item())
        else:
            pen = float(self._norm[1].item()) if self.optimizer is not None else 0.0
        if hyps is None:
            dec_h, dlen_h = dec.cpu().numpy(), dlen.cpu().numpy()
            hyps = [dec_h[n, :dlen_h[n]].tolist() for n in range(len(labels))]
        ler = float(np.mean(ops.edit_distance_host(hyps, [list(l) for l in labels])))
        ctc_mean = float(np.mean(ctc_h))
        return [ctc_mean + pen, ctc_mean, 0.0, ler]

    def test_on_batch(self, inputs, outputs=None):
        slab, labels, lens = self._unpack_inputs(inputs)
        N = len(labels)
        lab, lab_len, sl = self._prep_labels(labels, lens, slab.shape[0])
        logits = self.forward(slab, training=False, need_grad=False, n_valid=N)
        ctc = ops.ctc_loss_grad(logits, lab, lab_len, sl, N, grad=None)
        hyps = None
        dec = dlen = None
        if self.decoder.get('is_greedy', True):
            dec, dlen = ops.ctc_greedy(logits, sl, N)
        else:
            dec, dlen = self._beam(logits, sl, N)
        # l2 penalty of the current weights (reported in 'loss', as Keras does)
        if self.optimizer is None or self._step == 0:
            w2 = 0.0
            flat = self.params
            for off, n, l2 in self._segments:
                if l2:
                    w2 += l2 * float((flat[off:off + n].double() ** 2).sum().item())
            self._norm[1] = w2
        m = self._metrics(ctc, dec, dlen, labels, hyps, train=False,
                          flags=ops.lstm_timeout_flags(self.device))
        if m is None:       # a forward kernel timed out: evaluate the batch again, stepwise
            return self.test_on_batch(inputs, outputs)
        return m

    def predict(self, x, inputs_length=None):
        """Decoded label sequences for a batch (greedy or beam, per self.decoder), or the
        (N, T, C) logits when the model was loaded without a decoder (predict.py's
        --no_decoder).  ``x`` may also be the ``[inputs, inputs_length]`` list a
        predict-mode DatasetIterator yields (predict.py:88)."""
        if isinstance(x, list) and len(x) == 2 and inputs_length is None:
            x, inputs_length = x
        if isinstance(x, tuple) and x[0] == 'slab':
            x = x[1]
        slab = x if (torch.is_tensor(x) and x.dim() == 3 and x.shape[1] % 16 == 0) else self.to_slab(x)
        N = len(inputs_length) if inputs_length is not None else slab.shape[1]
        lens = np.asarray(inputs_length if inputs_length is not None else [slab.shape[0]] * N).reshape(-1)
        logits = self.forward(slab, training=False, need_grad=False, n_valid=N)
        if self.decoder is None:
            return logits[:, :N].permute(1, 0, 2).contiguous().cpu().numpy()
        sl = torch.as_tensor(np.asarray(self.out_lengths(lens)).astype(np.int32)).to(self.device)
        if self.decoder.get('is_greedy', True):
            dec, dlen = ops.ctc_greedy(logits, sl, N)
            dec, dlen = dec.cpu().numpy(), dlen.cpu().numpy()
            return [dec[n, :dlen[n]].tolist() for n in range(N)]
        dec, dlen = self._beam(logits, sl, N)
        dec, dlen = dec.cpu().numpy(), dlen.cpu().numpy()
        return [dec[n, :dlen[n]].tolist() for n in range(N)]

    def _beam(self, logits, seq_len_dev, N):
        """core/ctc_utils.py:48-50 (K9): the library's host decoder (decode_host.cpp: one
        utterance per host thread, on a copy of the logits) or the device decoder (beam.hip: the
        logits stay in HBM) -- same strings either way; ops.beam_decoder_choice picks (host
        while every utterance gets its own host thread, ASR_BEAM=device / host force one)."""
        width = int(self.decoder.get('beam_width', 100))
        merge = self.decoder.get('merge_repeated', True)
        if ops.beam_decoder_choice(N, width, logits.shape[2]) == 'device':
            dec, dlen, _ = ops.ctc_beam_search(logits, seq_len_dev, N, width, merge)
            return dec, dlen
        lens = seq_len_dev.cpu().numpy()
        hyps, _ = ops.ctc_beam_search_host(logits.cpu().numpy(), lens, N, width, merge)
        T = logits.shape[0]
        dec = np.full((N, T), -1, np.int32)
        for n, h in enumerate(hyps):
            dec[n, :len(h)] = h
        return torch.from_numpy(dec), torch.tensor([len(h) for h in hyps], dtype=torch.int32)

    # ------------------------------------------------------------------ loops
    def fit_generator(self, generator, samples_per_epoch, nb_epoch, verbose=1, callbacks=None,
                      validation_data=None, nb_val_samples=None, max_q_size=10, nb_worker=1,
                      initial_epoch=0, **kwargs):
        """Keras-1.2.2 signature (train.py:213-217).  Like Keras' generator queue
        (``nb_worker=1``: ONE producer thread, ``max_q_size`` batches deep) the batches
        are drawn on a background thread, so HDF5 reads / padding / label parsing overlap
        the GPU step; the per-batch metrics of step i are fetched while step i+1 runs
        (the step itself never synchronises with the host)."""
        callbacks = callbacks or []
        history = []
        for cb in callbacks:
            cb.set_model(self)
            cb.on_train_begin()
        feeder = _Feeder(generator, max_q_size, self.device)
        try:
            for epoch in range(initial_epoch, nb_epoch):
                t0 = time.time()
                seen, seen_local, sums = 0, 0, np.zeros(4)
                pending = None              # (device results, labels, n) of the previous step

                def account(pend):
                    # a step whose update was vetoed (recurrent-kernel timeout) has no valid
                    # activations: it counts for nothing
                    m = self._lagged(pend)
                    if m is None:
                        return 0
                    sums[:] += np.array(m) * pend[2]
                    return pend[2]
                while seen < samples_per_epoch:
                    inputs, outputs = feeder.get()
                    slab, labels, lens = self._unpack_inputs(inputs)
                    # an epoch counts GLOBAL samples; this rank's metrics weigh its own
                    n = getattr(inputs, 'n_local', len(labels))
                    batch = [('slab', slab), labels, lens]
                    if hasattr(inputs, 'n_global'):
                        from ..parallel import ShardedBatch
                        batch = ShardedBatch(batch, inputs.n_global, inputs.n_local)
                    res = self.train_on_batch(batch, outputs, sync=False)
                    if pending is not None:
                        seen_local += account(pending)
                    pending = (res, labels, n)
                    seen += getattr(inputs, 'n_global', len(labels))
                if pending is not None:
                    seen_local += account(pending)
                # the training-side logs are GLOBAL means: callbacks that monitor them (LR
                # schedules, early stopping) must take the same decision on every rank
                from ..parallel import reduce_metrics
                logs = dict(zip(self.metrics_names, reduce_metrics(sums, seen_local)))
                if validation_data is not None:
                    val = self.evaluate_generator(validation_data, nb_val_samples)
                    for k, v in zip(self.metrics_names, val):
                        logs['val_' + k] = v
                if verbose:
                    shown = ' - '.join('%s: %.4f' % (k, logs[k]) for k in
                                       ('loss', 'decoder_ler', 'val_loss', 'val_decoder_ler')
                                       if k in logs)
                    print('Epoch %d/%d - %.0fs - %s' % (epoch + 1, nb_epoch, time.time() - t0,
                                                        shown))
                history.append(logs)
                for cb in callbacks:
                    cb.on_epoch_end(epoch, logs)
                if self.stop_training:
                    break
        finally:
            feeder.close()
        for cb in callbacks:
            cb.on_train_end()
        return history

    def evaluate_generator(self, generator, val_samples, max_q_size=10, nb_worker=1, **kwargs):
        """Returns [loss, ctc_loss, decoder_loss, <decoder>_ler] averaged with batch-size
        weights (train.py:223-227, eval.py:76-80).  Batches are drawn on a producer thread
        (Keras' generator queue), so host reads overlap the forward pass and the decoder."""
        seen, sums = 0, np.zeros(4)
        feeder = _Feeder(generator, max_q_size, self.device)
        try:
            while seen < val_samples:
                inputs, outputs = feeder.get()
                n = len(np.asarray(inputs[2]).reshape(-1))
                sums += np.array(self.test_on_batch(inputs, outputs)) * n
                seen += n
        finally:
            feeder.close()
        return (sums / max(seen, 1)).tolist()

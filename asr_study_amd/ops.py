"""Thin torch-tensor level wrappers over the C ABI (libasr_hip.so).

torch is used ONLY as a device-memory / stream provider: every function here takes
``tensor.data_ptr()`` and the current HIP stream handle and calls the C ABI through
ctypes.  No torch.nn, no torch math on the hot path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_f32(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), \
                'expected contiguous float32 CUDA tensor'


class _Workspaces(object):
    """Caller-owned scratch buffers, grown on demand, one per op family."""

    def __init__(self):
        self.bufs = {}

    def get(self, name, nbytes, device):
        key = (name, str(device))
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            # zero-filled: the recurrent kernels keep a sticky timeout flag in the first
            # bytes of their workspace (asr_lstm_status); a workspace that grows in the
            # middle of a step (a wider layer, a larger batch) inherits the flag
            new = torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            if buf is not None:
                new[:256].copy_(buf[:256])
            buf = self.bufs[key] = new
        return buf


WS = _Workspaces()


def pad16(n):
    return (int(n) + 15) // 16 * 16


# --------------------------------------------------------------------------- GEMM
def _resolve_split(split_k, M, N, K, tile=128):
    if split_k == 'auto':
        # long-K / small-output GEMMs (dW, dU): split K until ~1024 workgroups' worth of
        # `tile` x `tile` tiles exist (256 workgroups of the 256 x 256 kernel)
        tiles = ((int(M) + tile - 1) // tile) * ((int(N) + tile - 1) // tile)
        split_k = max(1, min(64, (1024 + tiles - 1) // tiles, int(K) // 256))
    return int(split_k)


def _resolve_split_hl(split_k, M, N, K, tile=0, batch=1):
    """'auto' for asr_gemm_hl: split K until the launch fills the chip ONCE -- a workgroup of the
    256 x 256 kernel has a CU to itself (256 slots), the 128 x 128 kernel runs two per CU.  (A
    launch a little over one round costs two: 312 workgroups of a 640 x 2048 weight gradient.)"""
    slots_cap = 0
    if isinstance(split_k, str) and split_k.startswith('auto:'):
        # 'auto:<CUs>': a launch that shares the chip (the side stream beside a compact BPTT
        # has half of the CUs): one round of THAT many workgroups
        slots_cap = int(split_k[5:])
        split_k = 'auto'
    if split_k != 'auto':
        return int(split_k)
    big = int(tile) != 128 and int(M) >= 256 and int(N) >= 256
    tl, slots = (256, 256) if big else (128, 512)
    if slots_cap:
        slots = slots_cap if big else 2 * slots_cap
    tiles = ((int(M) + tl - 1) // tl) * ((int(N) + tl - 1) // tl) * max(1, int(batch))
    return max(1, min(64, slots // tiles, int(K) // 256))


def gemm(A, B, Cm, M, N, K, trans_a=False, trans_b=False, lda=None, ldb=None, ldc=None,
         alpha=1.0, beta=0.0, bias=None, a_scale=None, a_scale_period=0, c_scale=None,
         c_scale_period=0, split_k=0, a_off=0, b_off=0, c_off=0, ws_name='gemm',
         precision=-1, a_absmax=None, b_absmax=None, a_scale_off=0, a_scale_ld=None):
    """C[M,N] = alpha * opA(A) @ opB(B) + beta*C (+bias), see include/asr_hip.h.

    A/B/Cm are float32 CUDA tensors used as raw storage; *_off are element
    offsets into them (sub-matrix views without torch slicing semantics)."""
    lib = L.load()
    a = L.GemmArgs()
    a.M, a.N, a.K = int(M), int(N), int(K)
    a.trans_a, a.trans_b = int(bool(trans_a)), int(bool(trans_b))
    a.A = A.data_ptr() + 4 * int(a_off)
    a.B = B.data_ptr() + 4 * int(b_off)
    a.C = Cm.data_ptr() + 4 * int(c_off)
    a.lda = int(lda if lda is not None else (M if trans_a else K))
    a.ldb = int(ldb if ldb is not None else (K if trans_b else N))
    a.ldc = int(ldc if ldc is not None else N)
    a.alpha, a.beta = float(alpha), float(beta)
    a.bias = bias.data_ptr() if bias is not None else None
    a.a_scale = a_scale.data_ptr() + 4 * int(a_scale_off) if a_scale is not None else None
    a.a_scale_period = int(a_scale_period)
    a.a_scale_ld = int(a_scale_ld if a_scale_ld is not None else a_scale.shape[-1]) \
        if a_scale is not None else 0
    a.c_scale = c_scale.data_ptr() if c_scale is not None else None
    a.c_scale_period = int(c_scale_period)
    a.c_scale_ld = int(c_scale.shape[-1]) if c_scale is not None else 0
    a.split_k = _resolve_split(split_k, M, N, K)
    a.precision = int(precision)
    a.a_absmax = a_absmax.data_ptr() if a_absmax is not None else None
    a.b_absmax = b_absmax.data_ptr() if b_absmax is not None else None
    nbytes = lib.asr_gemm_workspace_bytes(C.byref(a))
    ws = WS.get(ws_name, nbytes, Cm.device) if nbytes else None
    L.check(lib.asr_gemm(C.byref(a), _ptr(ws), nbytes, _stream()), 'asr_gemm')


_GATE_ROLES = {'fwd': 0, 'dgrad': 1, 'wgrad': 2}


def gate_gemm(role, rows, n_pad, in_dim, gate_dim, W, w_off, ldw, ldz, x=None, x_off=0,
              ldx=None, bias=None, mask_w=None, zx=None, dz=None, z_off=0, dz_absmax=None,
              dx=None, dx_off=0, dx_beta=0.0, dW=None, dw_off=0, db=None, split_k=0,
              precision=-1, ws_name='gemm'):
    """One of the three GEMMs of an LSTM layer's input projection (asr_gemm_gate_fwd /
    _dgrad / _wgrad, include/asr_hip.h): tensors are raw float32 storage, *_off element
    offsets into them (x_off into x, z_off into zx or dz, w_off into W, dw_off into dW)."""
    lib = L.load()
    g = L.GateGemmArgs()
    g.rows, g.n_pad, g.in_dim, g.gate_dim = int(rows), int(n_pad), int(in_dim), int(gate_dim)
    g.x = None if x is None else x.data_ptr() + 4 * int(x_off)
    g.ldx = int(ldx if ldx is not None else in_dim)
    g.W = W.data_ptr() + 4 * int(w_off)
    g.ldw, g.ldz = int(ldw), int(ldz)
    g.bias = None if bias is None else bias.data_ptr()
    g.mask_w = None if mask_w is None else mask_w.data_ptr()
    g.zx = None if zx is None else zx.data_ptr() + 4 * int(z_off)
    g.dz = None if dz is None else dz.data_ptr() + 4 * int(z_off)
    g.dz_absmax = None if dz_absmax is None else dz_absmax.data_ptr()
    g.dx = None if dx is None else dx.data_ptr() + 4 * int(dx_off)
    g.dx_beta = float(dx_beta)
    g.dW = None if dW is None else dW.data_ptr() + 4 * int(dw_off)
    g.db = None if db is None else db.data_ptr()
    g.split_k = _resolve_split(split_k, in_dim, gate_dim, rows)
    g.precision = int(precision)
    kind = _GATE_ROLES[role]
    nbytes = lib.asr_gemm_gate_workspace_bytes(C.byref(g), kind)
    ws = WS.get(ws_name, nbytes, W.device) if nbytes else None
    fn = (lib.asr_gemm_gate_fwd, lib.asr_gemm_gate_dgrad, lib.asr_gemm_gate_wgrad)[kind]
    L.check(fn(C.byref(g), _ptr(ws), nbytes, _stream()), 'asr_gemm_gate_' + role)


def absmax(x, out=None):
    """max |x| of a contiguous float32 CUDA tensor -> 1-element float32 tensor."""
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    L.check(L.load().asr_absmax(_ptr(x), x.numel(), _ptr(out), _stream()), 'asr_absmax')
    return out


def colsum(X, M, N, ldx, out, beta=0.0, x_off=0, ws_name='colsum'):
    lib = L.load()
    nbytes = lib.asr_colsum_workspace_bytes(int(M), int(N))
    ws = WS.get(ws_name, nbytes, out.device)
    L.check(lib.asr_colsum(C.c_void_p(X.data_ptr() + 4 * int(x_off)), int(M), int(N),
                           int(ldx), _ptr(out), float(beta), _ptr(ws), nbytes, _stream()),
            'asr_colsum')


# --------------------------------------------------------------------------- LSTM
# LDS a recurrent workgroup reserves (asr_lstm_args.lds_reserve_kb; 0 = the library's 96 KB:
# no GEMM workgroup fits beside it).
LSTM_LDS_KB = 0


def cu_masked_stream(device, n_cus, total_cus):
    """A torch stream confined to n_cus of the device's total_cus compute units
    (asr_stream_create_cu_mask).  The mask alternates groups of 8 CUs, which selects the same
    number of CUs on every XCD whether the runtime numbers them XCD-interleaved or XCD-major."""
    words = (int(total_cus) + 31) // 32
    bits = [0] * words
    chosen = 0
    for pass_ in range(2):                      # even groups of 8 first, then odd ones
        for i in range(int(total_cus)):
            if chosen >= n_cus:
                break
            if ((i // 8) % 2 == pass_) and not (bits[i // 32] >> (i % 32)) & 1:
                bits[i // 32] |= 1 << (i % 32)
                chosen += 1
    arr = (C.c_uint32 * words)(*bits)
    out = C.c_void_p()
    L.check(L.load().asr_stream_create_cu_mask(arr, words, C.byref(out)), 'asr_stream_create_cu_mask')
    return torch.cuda.ExternalStream(out.value, device=device)


# asr_lstm_args.activation: the reference LSTM's `activation` hyper-parameter (Keras-1.2.2 names)
ACTIVATION_IDS = {'tanh': 0, 'relu': 1, 'sigmoid': 2, 'hard_sigmoid': 3, 'linear': 4,
                  'softsign': 5, 'softplus': 6}


def activation_id(act):
    if isinstance(act, int):
        return act
    try:
        return ACTIVATION_IDS[act or 'tanh']
    except KeyError:
        raise ValueError('LSTM activation %r: one of %s' % (act, sorted(ACTIVATION_IDS)))


def _lstm_args(T, n_pad, H, U, mask_u=None, zx=None, y=None, cell=None, gates=None,
               dy=None, dz=None, mode=0, dz_absmax=None, steps=None, mi=None, uh=None,
               zone_c=None, zone_h=None, wx=None, dwx=None, dmi=None, db_part=None, act=0):
    a = L.LstmArgs()
    a.activation = activation_id(act)
    a.T, a.n_pad, a.H, a.mode = int(T), int(n_pad), int(H), int(mode)
    a.step_begin, a.step_count = (0, 0) if steps is None else (int(steps[0]), int(steps[1]))
    a.U = U.data_ptr()
    for name, t in (('mask_u', mask_u), ('zx', zx), ('y', y), ('cell', cell),
                    ('gates', gates), ('dy', dy), ('dz', dz), ('dz_absmax', dz_absmax),
                    ('mi', mi), ('uh', uh), ('zone_c', zone_c), ('zone_h', zone_h),
                    ('wx', wx), ('dwx', dwx), ('dmi', dmi), ('db_part', db_part)):
        setattr(a, name, t.data_ptr() if t is not None else None)
    return a


def lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H, mask_u=None, mode=0, check=False,
                 steps=None, mi=None, uh=None, zone_c=None, zone_h=None, n_valid=0, act=0,
                 units=0):
    """steps=(begin, count): only that slice of the recurrence (consecutive slices from 0
    on the same stream continue one sequence); None = all T steps."""
    lib = L.load()
    _check_f32(zx, U, y, cell, gates, mask_u)
    a = _lstm_args(T, n_pad, H, U, mask_u, zx=zx, y=y, cell=cell, gates=gates, mode=mode,
                   steps=steps, mi=mi, uh=uh, zone_c=zone_c, zone_h=zone_h, act=act)
    a.n_valid = int(n_valid)            # 1: single-utterance kernel (row 0 only)
    a.fwd_units = int(units)            # 0 = the library decides, 8 / 16 = units per workgroup
    a.lds_reserve_kb = LSTM_LDS_KB
    nbytes = lib.asr_lstm_workspace_bytes(C.byref(a), 0)
    ws = WS.get('lstm_fwd', nbytes, zx.device)
    L.check(lib.asr_lstm_seq_fwd(C.byref(a), _ptr(ws), nbytes, _stream()), 'asr_lstm_seq_fwd')
    if check:
        L.check(lib.asr_lstm_status(_ptr(ws), _stream()), 'asr_lstm_status(fwd)')
    return ws


def lstm_seq_bwd(dy, U, cell, gates, dz, T, n_pad, H, mask_u=None, mode=0, check=False,
                 dz_absmax=None, steps=None, mi=None, uh=None, zone_c=None, zone_h=None,
                 wx=None, dwx=None, dmi=None, db_part=None, compact=False, act=0,
                 dz_planes=None, dz_bound=None):
    """db_part: optional (n_pad/16, 2, 4H) buffer receiving the per-batch-tile sums of dz over
    samples and steps (bias-gradient partials; accumulated across the slices of a sequence).
    compact: H/32 workgroups per chain (half the CUs per layer, asr_lstm_args.compact).
    dz_planes (HlPlanes of (T n_pad, 8H)) + dz_bound (device float): the gate gradients leave
    as packed planes pre-scaled for *dz_bound INSTEAD of the fp32 slab `dz` (may be None then;
    asr_lstm_args.dz_hl, see lstm_dz_hl_supported / lstm_dz_guard)."""
    lib = L.load()
    _check_f32(dy, U, cell, gates, dz, mask_u)
    a = _lstm_args(T, n_pad, H, U, mask_u, cell=cell, gates=gates, dy=dy, dz=dz, mode=mode,
                   dz_absmax=dz_absmax, steps=steps, mi=mi, uh=uh, zone_c=zone_c,
                   zone_h=zone_h, wx=wx, dwx=dwx, dmi=dmi, db_part=db_part, act=act)
    a.lds_reserve_kb = LSTM_LDS_KB
    a.compact = 1 if compact else 0
    if dz_planes is not None:
        assert dz_bound is not None and dz_planes.rows == T * n_pad and dz_planes.ld == 8 * H
        a.dz_hl = dz_planes.hl.data_ptr()
        a.dz_bound = dz_bound.data_ptr()
        a.dz_scale_out = dz_planes.scale.data_ptr()
    nbytes = lib.asr_lstm_workspace_bytes(C.byref(a), 1)
    ws = WS.get('lstm_bwd', nbytes, dy.device)
    L.check(lib.asr_lstm_seq_bwd(C.byref(a), _ptr(ws), nbytes, _stream()), 'asr_lstm_seq_bwd')
    if check:
        L.check(lib.asr_lstm_status(_ptr(ws), _stream()), 'asr_lstm_status(bwd)')
    return ws


def lstm_dz_hl_supported(T, n_pad, H, mode=0, compact=False, act=0):
    """Whether lstm_seq_bwd would write packed planes for this geometry (asr_lstm_dz_hl_supported:
    the plain cell on the two-dimensional-split BPTT kernels, persistent mode, split-fp16)."""
    a = L.LstmArgs()
    a.T, a.n_pad, a.H, a.mode = int(T), int(n_pad), int(H), int(mode)
    a.compact = 1 if compact else 0
    a.activation = activation_id(act)
    a.lds_reserve_kb = LSTM_LDS_KB
    return bool(L.load().asr_lstm_dz_hl_supported(C.byref(a)))


def lstm_dz_guard(dz_absmax, dz_bound, planes_used, device=None):
    """Device-side upkeep of a layer's dz bound behind a BPTT pass (asr_lstm_dz_guard): keeps or
    renews *dz_bound from the measured *dz_absmax; with planes_used, a maximum outside the
    planes' safe range raises the BPTT workspace's sticky flag (the step is vetoed and re-run)."""
    ws = WS.get('lstm_bwd', 0, dz_absmax.device) if planes_used else None
    L.check(L.load().asr_lstm_dz_guard(_ptr(dz_absmax), _ptr(dz_bound), 1 if planes_used else 0,
                                       _ptr(ws) if ws is not None else None, _stream()),
            'asr_lstm_dz_guard')


def lstm_status(ws):
    L.check(L.load().asr_lstm_status(_ptr(ws), _stream()), 'asr_lstm_status')


def lstm_timeout_flags(device):
    """The sticky timeout flags of the forward / BPTT workspaces as one int32 tensor of two
    elements on the device (a snapshot enqueued on the current stream, no synchronisation);
    non-zero = a persistent kernel abandoned a bounded spin since the flag was cleared."""
    return torch.stack([WS.get(name, 0, device)[:4].view(torch.int32)[0]
                        for name in ('lstm_fwd', 'lstm_bwd')])


def lstm_fast_chains(ws):
    return L.load().asr_lstm_fast_chains(_ptr(ws), _stream())


def lstm_profile(ws):
    out = (C.c_longlong * 24)()
    L.check(L.load().asr_lstm_profile(_ptr(ws), _stream(), out), 'asr_lstm_profile')
    return [[out[6 * w + i] for i in range(6)] for w in range(4)]


def lstm_plan(T, n_pad, H, backward):
    lib = L.load()
    a = L.LstmArgs()
    a.T, a.n_pad, a.H = int(T), int(n_pad), int(H)
    ks, r, blocks, cpl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    L.check(lib.asr_lstm_plan(C.byref(a), int(backward), C.byref(ks), C.byref(r),
                              C.byref(blocks), C.byref(cpl)), 'asr_lstm_plan')
    return dict(k_split=ks.value, k_per_lane=r.value, blocks=blocks.value,
                chains_per_launch=cpl.value)


# --------------------------------------------------------------------------- CTC
def ctc_loss_grad(logits, labels, label_len, seq_len, N, grad=None, grad_scale=1.0,
                  loss=None):
    """logits (T, n_pad, C) float32; labels (N, l_max) int32; returns loss (N,)."""
    lib = L.load()
    T, n_pad, Cc = logits.shape
    l_max = int(labels.shape[1])
    _check_f32(logits, grad)
    if loss is None:
        loss = torch.empty(N, dtype=torch.float32, device=logits.device)
    nbytes = lib.asr_ctc_workspace_bytes(T, int(N), n_pad, Cc, l_max)
    ws = WS.get('ctc', nbytes, logits.device)
    L.check(lib.asr_ctc_loss_grad(_ptr(logits), _ptr(labels), _ptr(label_len), _ptr(seq_len),
                                  T, int(N), n_pad, Cc, l_max, float(grad_scale), _ptr(loss),
                                  _ptr(grad), _ptr(ws), nbytes, _stream()),
            'asr_ctc_loss_grad')
    return loss


def ctc_greedy(logits, seq_len, N):
    lib = L.load()
    T, n_pad, Cc = logits.shape
    dec = torch.empty((int(N), T), dtype=torch.int32, device=logits.device)
    dlen = torch.empty(int(N), dtype=torch.int32, device=logits.device)
    L.check(lib.asr_ctc_greedy(_ptr(logits), _ptr(seq_len), T, int(N), n_pad, Cc, _ptr(dec),
                               _ptr(dlen), _stream()), 'asr_ctc_greedy')
    return dec, dlen


def ctc_beam_search(logits, seq_len, N, beam_width=100, merge_repeated=True):
    """K9 on the device: logits (T, n_pad, C) float32 and seq_len (N,) int32 DEVICE tensors ->
    (decoded (N, T) int32 padded with -1, decoded_len (N,), log_score (N,)) device tensors; same
    results as ctc_beam_search_host (asr_ctc_beam_device).  The prefix-tree workspace is sized
    for the hard bound (T * width child blocks per utterance) and is not initialised."""
    lib = L.load()
    T, n_pad, Cc = logits.shape
    _check_f32(logits)
    dev = logits.device
    dec = torch.empty((int(N), T), dtype=torch.int32, device=dev)
    dlen = torch.empty(int(N), dtype=torch.int32, device=dev)
    score = torch.empty(int(N), dtype=torch.float32, device=dev)
    nbytes = lib.asr_ctc_beam_device_workspace_bytes(T, int(N), Cc, int(beam_width))
    key = ('beam', str(dev))
    ws = WS.bufs.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = WS.bufs[key] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
    L.check(lib.asr_ctc_beam_device(_ptr(logits), _ptr(seq_len), T, int(N), n_pad, Cc,
                                    int(beam_width), int(bool(merge_repeated)), _ptr(dec),
                                    _ptr(dlen), _ptr(score), _ptr(ws), nbytes, _stream()),
            'asr_ctc_beam_device')
    return dec, dlen, score


HOST_BEAM_MAX_UTTERANCES = 64


def beam_decoder_choice(N, width, classes, on_device=True):
    """Which K9 decoder serves a batch of N utterances: 'host' (decode_host.cpp: one utterance
    per host thread, after a D2H copy of the logits) or 'device' (beam.hip: one wave per
    utterance, logits stay in HBM).  Both give the same strings (tests/test_gpu_beam.py); the
    default is decided on measurements (profiles/r6_beam_crossover.md, tools/beam_crossover.py:
    N x 999 frames x 28 classes on a 256-thread host).  The device decoder's time is FLAT in N up
    to one wave per CU (0.118 s at 64, 256 and 0.126 s at 512 utterances, width 100) while the
    host decoder's grows faster than N once more than ~64 threads decode at once (0.08 s at 64,
    0.82 s at 256, 2.1 s at 512; on 8 host threads 0.35 s at 64): ASR_BEAM=auto (default) takes
    the host decoder for N <= min(USABLE host threads, 64) and the device decoder beyond -- the
    BASELINE eval batch (64 utterances) decodes on the host of a many-core box, larger batches and
    small hosts on the device; ASR_BEAM=device / host force one.  The device kernel handles
    widths <= 1024 and <= 64 classes."""
    import os
    mode = os.environ.get('ASR_BEAM', 'auto')
    device_ok = on_device and int(width) <= 1024 and int(classes) <= 64
    if mode == 'host' or not device_ok:
        return 'host'
    if mode == 'device':
        return 'device'
    return 'host' if int(N) <= min(usable_host_threads(), HOST_BEAM_MAX_UTTERANCES) else 'device'


def usable_host_threads():
    """Host threads this PROCESS may run on: the affinity mask where the platform has one (a
    container or a taskset limits it below os.cpu_count()), else the CPU count."""
    import os
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def ctc_beam_counters(logits_shape, N, beam_width, utterance=0, device='cuda:0'):
    """Work counters of the last ctc_beam_search call of that shape (see
    asr_ctc_beam_device_counters): dict of seconds per phase and event counts."""
    lib = L.load()
    T, n_pad, Cc = logits_shape
    ws = WS.bufs[('beam', str(torch.device(device)))]
    out = (C.c_longlong * 7)()
    L.check(lib.asr_ctc_beam_device_counters(_ptr(ws), int(T), int(N), int(Cc), int(beam_width),
                                             int(utterance), out, _stream()),
            'asr_ctc_beam_device_counters')
    return dict(update_s=out[0] * 1e-8, rank_s=out[1] * 1e-8, turns_s=out[2] * 1e-8,
                handover_s=out[3] * 1e-8, turns=out[4], insertions=out[5], blocks=out[6])


def ctc_beam_search_host(logits_host, seq_len_host, N, beam_width=100, merge_repeated=True):
    """logits_host: (T, n_pad, C) float32 numpy array (already on the host)."""
    lib = L.load()
    logits_host = np.ascontiguousarray(logits_host, dtype=np.float32)
    seq = np.ascontiguousarray(seq_len_host, dtype=np.int32)
    T, n_pad, Cc = logits_host.shape
    dec = np.empty((int(N), T), dtype=np.int32)
    dlen = np.empty(int(N), dtype=np.int32)
    score = np.empty(int(N), dtype=np.float32)
    L.check(lib.asr_ctc_beam(
        logits_host.ctypes.data_as(C.c_void_p), seq.ctypes.data_as(C.c_void_p), T, int(N),
        n_pad, Cc, int(beam_width), int(bool(merge_repeated)),
        dec.ctypes.data_as(C.c_void_p), dlen.ctypes.data_as(C.c_void_p),
        score.ctypes.data_as(C.c_void_p)), 'asr_ctc_beam')
    return [dec[n, :dlen[n]].tolist() for n in range(int(N))], score


def edit_distance_host(hyps, truths):
    """Lists of int lists -> per-sample normalised Levenshtein (numpy float32)."""
    lib = L.load()
    N = len(hyps)

    def pack(seqs):
        ld = max([len(s) for s in seqs] + [1])
        a = np.zeros((N, ld), np.int32)
        ln = np.zeros(N, np.int32)
        for i, s in enumerate(seqs):
            a[i, :len(s)] = s
            ln[i] = len(s)
        return a, ln, ld

    h, hl, hld = pack(hyps)
    t, tl, tld = pack(truths)
    out = np.empty(N, np.float32)
    L.check(lib.asr_edit_distance(
        h.ctypes.data_as(C.c_void_p), hl.ctypes.data_as(C.c_void_p), hld,
        t.ctypes.data_as(C.c_void_p), tl.ctypes.data_as(C.c_void_p), tld, N,
        out.ctypes.data_as(C.c_void_p)), 'asr_edit_distance')
    return out


# --------------------------------------------------------------------------- optimiser
def make_segments(entries, device):
    """entries: list of (offset, length, l2) -> device buffer of asr_segment."""
    arr = (L.Segment * len(entries))()
    for i, (off, ln, l2) in enumerate(entries):
        arr[i].offset, arr[i].len, arr[i].l2, arr[i].reserved = int(off), int(ln), float(l2), 0.0
    raw = np.frombuffer(memoryview(arr), dtype=np.uint8).copy()
    return torch.from_numpy(raw).to(device), len(entries)


def grad_norm(params, grads, segs, n_seg, norm_out):
    lib = L.load()
    n = params.numel()
    nbytes = lib.asr_optim_workspace_bytes(n)
    ws = WS.get('optim', nbytes, params.device)
    L.check(lib.asr_grad_norm(_ptr(params), _ptr(grads), n, _ptr(segs), n_seg, _ptr(norm_out),
                              _ptr(ws), nbytes, _stream()), 'asr_grad_norm')


def adam_step(params, grads, m, v, segs, n_seg, norm, clipnorm, lr, step, beta1=0.9,
              beta2=0.999, eps=1e-8):
    L.check(L.load().asr_adam_step(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(),
                                   _ptr(segs), n_seg, _ptr(norm), float(clipnorm), float(lr),
                                   float(beta1), float(beta2), float(eps), int(step),
                                   _stream()), 'asr_adam_step')


def clip_adam_step(params, grads, m, v, segs, n_seg, norm_out, clipnorm, lr, step, beta1=0.9,
                   beta2=0.999, eps=1e-8):
    """Global norm (-> norm_out, two float64) + clipped Adam update in one library call."""
    lib = L.load()
    n = params.numel()
    nbytes = lib.asr_optim_workspace_bytes(n)
    ws = WS.get('optim', nbytes, params.device)
    L.check(lib.asr_clip_adam_step(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), n, _ptr(segs),
                                   n_seg, _ptr(norm_out), float(clipnorm), float(lr),
                                   float(beta1), float(beta2), float(eps), int(step), _ptr(ws),
                                   nbytes, _stream()), 'asr_clip_adam_step')


def optim_guard(norm, device, flags=None):
    """Vetoes the update that follows if a persistent recurrent kernel has flagged a timeout;
    device-side only.  flags=None: the sticky words at the head of this process' lstm_fwd /
    lstm_bwd workspaces; data parallel: a 2-element device tensor holding the all-reduced
    flags (any non-zero bit pattern vetoes), so every rank skips the same update."""
    if flags is None:
        a, b = WS.get('lstm_fwd', 0, device), WS.get('lstm_bwd', 0, device)
        pa, pb = _ptr(a), _ptr(b)
    else:
        assert flags.numel() >= 2 and flags.element_size() == 4 and flags.is_contiguous()
        pa, pb = C.c_void_p(flags.data_ptr()), C.c_void_p(flags.data_ptr() + 4)
    L.check(L.load().asr_optim_guard(_ptr(norm), pa, pb, _stream()), 'asr_optim_guard')


def collect_timeout_flags(out, device):
    """out[0:2] (float32, device) <- 1.0 where the forward / BPTT workspace's sticky timeout
    word is set, else 0.0 (asr_timeout_flags): the form that travels through the gradient
    all-reduce."""
    L.check(L.load().asr_timeout_flags(_ptr(WS.get('lstm_fwd', 0, device)),
                                       _ptr(WS.get('lstm_bwd', 0, device)), _ptr(out), _stream()),
            'asr_timeout_flags')


def debug_occupy(blocks, lds_bytes, seconds):
    """Tests only: holds `blocks` 256-thread workgroups (lds_bytes of LDS each) on the device
    for `seconds` on the current stream (asr_debug_occupy)."""
    L.check(L.load().asr_debug_occupy(int(blocks), int(lds_bytes), float(seconds), _stream()),
            'asr_debug_occupy')


def clear_timeout_flags(device):
    for name in ('lstm_fwd', 'lstm_bwd'):
        WS.get(name, 0, device)[:4].zero_()


def clip_sgd_step(params, grads, vel, segs, n_seg, norm_out, clipnorm, lr, momentum=0.9):
    lib = L.load()
    n = params.numel()
    nbytes = lib.asr_optim_workspace_bytes(n)
    ws = WS.get('optim', nbytes, params.device)
    L.check(lib.asr_clip_sgd_step(_ptr(params), _ptr(grads), _ptr(vel), n, _ptr(segs), n_seg,
                                  _ptr(norm_out), float(clipnorm), float(lr), float(momentum),
                                  _ptr(ws), nbytes, _stream()), 'asr_clip_sgd_step')


def sgd_step(params, grads, vel, segs, n_seg, norm, clipnorm, lr, momentum=0.9):
    L.check(L.load().asr_sgd_step(_ptr(params), _ptr(grads), _ptr(vel), params.numel(),
                                  _ptr(segs), n_seg, _ptr(norm), float(clipnorm), float(lr),
                                  float(momentum), _stream()), 'asr_sgd_step')


# --------------------------------------------------------------------------- front-end
def frontend_features(cfg, audio, offsets, lengths, host_lengths, n_pad, tables, t_out):
    """cfg: _lib.FrontendCfg; audio float32 CUDA (concatenated); offsets/lengths
    int32 CUDA; tables: dict(window, mel, mel_range, dct) CUDA tensors.
    Returns (out (t_out, n_pad, F) float32, out_frames int32 (n_utt,))."""
    lib = L.load()
    n_utt = len(host_lengths)
    hl = (C.c_int * n_utt)(*[int(x) for x in host_lengths])
    ffull = lib.asr_frontend_num_feats(C.byref(cfg))
    f_out = ffull * (2 * cfg.num_context + 1)
    max_frames = max(lib.asr_frontend_num_frames(int(x), cfg.frame_len, cfg.frame_step)
                     for x in host_lengths)
    nbytes = lib.asr_frontend_workspace_bytes(C.byref(cfg), n_utt, max_frames)
    ws = WS.get('frontend', nbytes, audio.device)
    out = torch.empty((int(t_out), int(n_pad), f_out), dtype=torch.float32, device=audio.device)
    frames = torch.empty(n_utt, dtype=torch.int32, device=audio.device)
    head = (C.byref(cfg), _ptr(audio), _ptr(offsets), _ptr(lengths), hl, n_utt, int(n_pad),
            _ptr(tables['window']), _ptr(tables['mel']), _ptr(tables['mel_range']))
    tail = (_ptr(out), int(t_out), _ptr(frames), _ptr(ws), nbytes, _stream())
    if cfg.kind == 0:
        L.check(lib.asr_frontend_mfcc_batch(*(head + (_ptr(tables['dct']),) + tail)),
                'asr_frontend_mfcc_batch')
    else:
        L.check(lib.asr_frontend_logfbank_batch(*(head + tail)), 'asr_frontend_logfbank_batch')
    return out, frames


def axpby(a, x, b, y, out=None):
    """out = a * x + b * y over contiguous float32 CUDA tensors of equal size (the residual
    merge of brsmv1 and its gradient); out defaults to a new tensor."""
    _check_f32(x, y, out)
    assert x.numel() == y.numel()
    if out is None:
        out = torch.empty_like(x)
    L.check(L.load().asr_axpby(x.numel(), float(a), _ptr(x), float(b), _ptr(y), _ptr(out),
                               _stream()), 'asr_axpby')
    return out


# --------------------------------------------------------------------------- LN cell
def _lstm_ln_args(T, n_pad, H, U, cellp, wx, uh, y, cell, gates, has_mi, mask_u=None,
                  zone_c=None, zone_h=None, dy=None, duh=None, dwx=None, dparams=None):
    a = L.LstmLnArgs()
    a.T, a.n_pad, a.H, a.has_mi = int(T), int(n_pad), int(H), int(bool(has_mi))
    for name, t in (('U', U), ('mask_u', mask_u), ('cellp', cellp), ('zone_c', zone_c),
                    ('zone_h', zone_h), ('wx', wx), ('uh', uh), ('y', y), ('cell', cell),
                    ('gates', gates), ('dy', dy), ('duh', duh), ('dwx', dwx),
                    ('dparams', dparams)):
        setattr(a, name, t.data_ptr() if t is not None else None)
    return a


def lstm_ln_seq_fwd(wx, U, cellp, uh, y, cell, gates, T, n_pad, H, has_mi=False, mask_u=None,
                    zone_c=None, zone_h=None, act=0):
    """Layer-normalised cell, forward over the whole sequence (include/asr_hip.h K5-LN)."""
    _check_f32(wx, U, cellp, uh, y, cell, gates, mask_u, zone_c, zone_h)
    a = _lstm_ln_args(T, n_pad, H, U, cellp, wx, uh, y, cell, gates, has_mi, mask_u, zone_c,
                      zone_h)
    a.activation = activation_id(act)
    L.check(L.load().asr_lstm_ln_seq_fwd(C.byref(a), _stream()), 'asr_lstm_ln_seq_fwd')


def lstm_ln_seq_bwd(dy, wx, U, cellp, uh, y, cell, gates, duh, dwx, dparams, T, n_pad, H,
                    has_mi=False, mask_u=None, zone_c=None, zone_h=None, act=0):
    _check_f32(dy, wx, U, cellp, uh, y, cell, gates, duh, dwx, dparams, mask_u, zone_c, zone_h)
    a = _lstm_ln_args(T, n_pad, H, U, cellp, wx, uh, y, cell, gates, has_mi, mask_u, zone_c,
                      zone_h, dy=dy, duh=duh, dwx=dwx, dparams=dparams)
    a.activation = activation_id(act)
    lib = L.load()
    nbytes = lib.asr_lstm_ln_workspace_bytes(C.byref(a))
    ws = WS.get('lstm_ln', nbytes, dy.device)
    L.check(lib.asr_lstm_ln_seq_bwd(C.byref(a), _ptr(ws), nbytes, _stream()),
            'asr_lstm_ln_seq_bwd')


# --------------------------------------------------------------------------- packed operands
class HlPlanes(object):
    """The two fp16 planes (hi, lo) of a matrix with ``ld`` reduction indices per row,
    interleaved in groups of 16 -- ``hl[row, g, 0]`` = hi[16g:16g+16], ``hl[row, g, 1]`` = lo --
    so that a 32-deep slab of a row is one 128-byte line; plus the device float holding the
    power-of-two scale they were packed with (include/asr_hip.h, asr_pack_hl)."""

    def __init__(self, rows, k, device):
        self.rows, self.k = int(rows), int(k)
        self.ld = (self.k + 31) // 32 * 32
        self.hl = torch.empty((self.rows, self.ld // 16, 2, 16), dtype=torch.float16,
                              device=device)
        self.scale = torch.ones(1, dtype=torch.float32, device=device)

    @property
    def hi(self):
        """(rows, ld) view-copy of the hi plane (tests / debugging)."""
        return self.hl[:, :, 0, :].reshape(self.rows, self.ld)

    @property
    def lo(self):
        return self.hl[:, :, 1, :].reshape(self.rows, self.ld)

    def ptr(self, row=0, k=0):
        """Address of (row, reduction index k); k a multiple of 16."""
        assert k % 16 == 0
        return self.hl.data_ptr() + 2 * (int(row) * 2 * self.ld + (int(k) // 16) * 32)


def pack_hl(src, rows, cols, ld=None, src_off=0, mask=None, mask_period=0, absmax=None,
            r=None, c=None, mask2=None, r2=None):
    """src (rows, cols) float32 (row stride ld, element offset src_off) [* mask[(row % period)]]
    -> r planes (rows, cols as K) and / or c planes (cols, rows as K); both get the same scale.
    mask2 / r2: a second set of row planes of the same source under another mask, written in the
    same pass (the source is read once)."""
    lib = L.load()
    a = L.PackArgs()
    a.src = src.data_ptr() + 4 * int(src_off)
    a.rows, a.cols, a.ld = int(rows), int(cols), int(ld if ld is not None else cols)
    a.mask = mask.data_ptr() if mask is not None else None
    a.mask_period = int(mask_period)
    a.mask_ld = int(mask.shape[-1]) if mask is not None else 0
    a.absmax = absmax.data_ptr() if absmax is not None else None
    first = r if r is not None else c
    a.scale_out = first.scale.data_ptr()
    if r is not None:
        assert r.rows >= rows and r.k == cols
        a.r_hl, a.ldk_r = r.hl.data_ptr(), r.ld
    if c is not None:
        assert c.rows >= cols and c.k == rows
        a.c_hl, a.ldk_c = c.hl.data_ptr(), c.ld
    if r2 is not None:
        assert r is not None and mask is not None and mask2 is not None
        assert r2.rows >= rows and r2.k == cols and mask2.shape[-1] == mask.shape[-1]
        a.mask2, a.r2_hl = mask2.data_ptr(), r2.hl.data_ptr()
    L.check(lib.asr_pack_hl(C.byref(a), _stream()), 'asr_pack_hl')
    if r2 is not None:
        r2.scale = r.scale                # one scale for both sets
    if r is not None and c is not None:
        c.scale = r.scale               # one scale for both orientations
    return r, c


def gemm_hl(A, B, Cm, M, N, K, a_row=0, a_k=0, b_row=0, b_k=0, c_off=0, ldc=None, alpha=1.0,
            beta=0.0, bias=None, c_scale=None, c_scale_period=0, split_k=0, ws_name='gemm',
            tile=0, k_major=False, a_seg_k=0, a_seg_rows=None, batch_rows=None):
    """C[M,N] (float32 storage Cm, element offset c_off) = alpha * A @ B^T (+bias)(*c_scale)
    + beta*C from packed planes: A rows [a_row, a_row+M), reduction range [a_k, a_k+K) of its
    planes; B rows [b_row, b_row+N), range [b_k, b_k+K).
    k_major: C = alpha * A^T @ B from planes whose ROWS are the reduction index: A plane rows
    [a_row, a_row+K), columns [a_k, a_k+M); B rows [b_row, b_row+K), columns [b_k, b_k+N)."""
    lib = L.load()
    g = L.GemmHlArgs()
    g.M, g.N, g.K = int(M), int(N), int(K)
    g.a_hl, g.lda = A.ptr(a_row, a_k), A.ld
    g.b_hl, g.ldb = B.ptr(b_row, b_k), B.ld
    g.a_scale, g.b_scale = A.scale.data_ptr(), B.scale.data_ptr()
    g.C = Cm.data_ptr() + 4 * int(c_off)
    g.ldc = int(ldc if ldc is not None else N)
    g.alpha, g.beta = float(alpha), float(beta)
    g.bias = bias.data_ptr() if bias is not None else None
    g.c_scale = c_scale.data_ptr() if c_scale is not None else None
    g.c_scale_period = int(c_scale_period)
    g.c_scale_ld = int(c_scale.shape[-1]) if c_scale is not None else 0
    g.split_k = _resolve_split_hl(split_k, M, N, K, tile,
                                  len(batch_rows) if batch_rows is not None else 1)
    g.tile = int(tile)
    g.k_major = 1 if k_major else 0
    if batch_rows is not None:      # k_major batch sharing B: C_b = A_b^T B, rows b*M.. of Cm
        g.batch = len(batch_rows)
        for i, r in enumerate(batch_rows):
            g.a_batch_row[i] = int(r)
    if a_seg_k:         # segmented reduction range of A (include/asr_hip.h): K = len(rows) * a_seg_k
        g.a_seg_k = int(a_seg_k)
        for i, r in enumerate(a_seg_rows):
            g.a_seg_row[i] = int(r)
    nbytes = lib.asr_gemm_hl_workspace_bytes(C.byref(g))
    ws = WS.get(ws_name, nbytes, Cm.device) if nbytes else None
    L.check(lib.asr_gemm_hl(C.byref(g), _ptr(ws), nbytes, _stream()), 'asr_gemm_hl')


# --------------------------------------------------------------------------- conv front-end
class Conv2d(object):
    """One layer of the 2-D convolution front-end (K13, include/asr_hip.h asr_conv2d_*) on
    time-major slabs: x (T_in, n_pad, F_in * C_in) -> y (T_out, n_pad, F_out * C_out),
    'same' padding, strides (st, sf), clipped ReLU.  Owns its workspace (the packed planes of x
    and of dz live there between the forward and the backward calls of a step)."""

    def __init__(self, T_in, n_pad, F_in, C_in, C_out, kt, kf, st, sf, clip, device):
        lib = L.load()
        a = self.args = L.Conv2dArgs()
        a.T_in, a.n_pad, a.F_in, a.C_in = int(T_in), int(n_pad), int(F_in), int(C_in)
        a.C_out, a.kt, a.kf, a.st, a.sf = int(C_out), int(kt), int(kf), int(st), int(sf)
        a.clip = float(clip)
        t, f = C.c_int(), C.c_int()
        L.check(lib.asr_conv2d_out_shape(C.byref(a), C.byref(t), C.byref(f)), 'asr_conv2d_out_shape')
        self.T_out, self.F_out = t.value, f.value
        self.ws_bytes = lib.asr_conv2d_workspace_bytes(C.byref(a))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.device = device

    def _call(self, fn, name, **ptrs):
        a = self.args
        for k in ('x', 'W', 'bias', 'z', 'y', 'dy', 'dx', 'dW', 'db', 'x_absmax'):
            t = ptrs.get(k)
            setattr(a, k, t.data_ptr() if t is not None else None)
        a.reuse_x, a.reuse_dz = int(ptrs.get('reuse_x', 0)), int(ptrs.get('reuse_dz', 0))
        L.check(fn(C.byref(a), _ptr(self.ws), self.ws_bytes, _stream()), name)

    def fwd(self, x, W, bias, z, y, x_absmax=None):
        """z=None (clip > 0): the clipped ReLU is applied in the GEMM epilogue and only y is
        written -- pass y as `z` to dgrad / wgrad then (the mask 0 < z < clip reads the same
        from y).  x_absmax: 1-element device tensor >= max|x| (skips the measuring pass)."""
        _check_f32(x, W, bias, z, y, x_absmax)
        self._call(L.load().asr_conv2d_fwd, 'asr_conv2d_fwd', x=x, W=W, bias=bias, z=z, y=y,
                   x_absmax=x_absmax)
        return y

    def dgrad(self, dy, z, W, dx, reuse_dz=False):
        _check_f32(dy, z, W, dx)
        self._call(L.load().asr_conv2d_dgrad, 'asr_conv2d_dgrad', dy=dy, z=z, W=W, dx=dx,
                   reuse_dz=reuse_dz)
        return dx

    def wgrad(self, x, dy, z, dW, db, reuse_x=False, reuse_dz=False, x_absmax=None):
        _check_f32(x, dy, z, dW, db, x_absmax)
        self._call(L.load().asr_conv2d_wgrad, 'asr_conv2d_wgrad', x=x, dy=dy, z=z, dW=dW, db=db,
                   reuse_x=reuse_x, reuse_dz=reuse_dz, x_absmax=x_absmax)


# --------------------------------------------------------------------------- random streams
def dropout_masks(out, p, scale, seed, stream_id, step):
    """out (any shape, contiguous float32) <- keep mask of the stream (seed, stream_id, step):
    scale where u >= p, else 0 (include/asr_hip.h K12)."""
    _check_f32(out)
    L.check(L.load().asr_dropout_masks(_ptr(out), out.numel(), float(p), float(scale),
                                       int(seed) & (2 ** 64 - 1), int(stream_id), int(step),
                                       _stream()), 'asr_dropout_masks')
    return out


def dropout_apply(x, out, mask_out, p, scale, seed, stream_id, step):
    _check_f32(x, out, mask_out)
    L.check(L.load().asr_dropout_apply(_ptr(x), _ptr(out), _ptr(mask_out), x.numel(), float(p),
                                       float(scale), int(seed) & (2 ** 64 - 1), int(stream_id),
                                       int(step), _stream()), 'asr_dropout_apply')
    return out


def gaussian_noise(x, out, sigma, seed, stream_id, step):
    """out = x + sigma * N(0, 1) (x may be None: pure noise)."""
    _check_f32(x, out)
    L.check(L.load().asr_gaussian_noise(_ptr(x), _ptr(out), out.numel(), float(sigma),
                                        int(seed) & (2 ** 64 - 1), int(stream_id), int(step),
                                        _stream()), 'asr_gaussian_noise')
    return out


def random_words(n, seed, stream_id, step, device):
    out = torch.empty(int(n), dtype=torch.int32, device=device)
    L.check(L.load().asr_random_words(_ptr(out), int(n), int(seed) & (2 ** 64 - 1),
                                      int(stream_id), int(step), _stream()), 'asr_random_words')
    return out


def mul(x, y, out=None):
    _check_f32(x, y, out)
    assert x.numel() == y.numel()
    if out is None:
        out = torch.empty_like(x)
    L.check(L.load().asr_mul(x.numel(), _ptr(x), _ptr(y), _ptr(out), _stream()), 'asr_mul')
    return out

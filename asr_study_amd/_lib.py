"""ctypes binding of libasr_hip.so (the C ABI declared in include/asr_hip.h).

This is the only place Python touches the native library.  There is NO fallback:
if the shared object is missing or fails to load, importing the product raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libasr_hip.so')

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)
void_p = C.c_void_p


class FrontendCfg(C.Structure):
    _fields_ = [('kind', C.c_int), ('frame_len', C.c_int), ('frame_step', C.c_int),
                ('nfft', C.c_int), ('num_filt', C.c_int), ('num_cep', C.c_int),
                ('append_energy', C.c_int), ('d', C.c_int), ('dd', C.c_int),
                ('stride', C.c_int), ('num_context', C.c_int),
                ('mean_norm', C.c_int), ('var_norm', C.c_int), ('reserved', C.c_int),
                ('pre_emph', C.c_double), ('eps', C.c_double)]


class GemmArgs(C.Structure):
    _fields_ = [('M', C.c_int), ('N', C.c_int), ('K', C.c_int),
                ('trans_a', C.c_int), ('trans_b', C.c_int),
                ('A', void_p), ('lda', C.c_int),
                ('B', void_p), ('ldb', C.c_int),
                ('C', void_p), ('ldc', C.c_int),
                ('alpha', C.c_float), ('beta', C.c_float),
                ('bias', void_p),
                ('a_scale', void_p), ('a_scale_period', C.c_int), ('a_scale_ld', C.c_int),
                ('c_scale', void_p), ('c_scale_period', C.c_int), ('c_scale_ld', C.c_int),
                ('split_k', C.c_int), ('precision', C.c_int),
                ('a_absmax', void_p), ('b_absmax', void_p)]


class PackArgs(C.Structure):
    _fields_ = [('src', void_p), ('rows', C.c_int), ('cols', C.c_int), ('ld', C.c_int),
                ('mask', void_p), ('mask_period', C.c_int), ('mask_ld', C.c_int),
                ('absmax', void_p), ('scale_out', void_p),
                ('r_hl', void_p), ('ldk_r', C.c_int),
                ('c_hl', void_p), ('ldk_c', C.c_int),
                ('mask2', void_p), ('r2_hl', void_p)]


class GemmHlArgs(C.Structure):
    _fields_ = [('M', C.c_int), ('N', C.c_int), ('K', C.c_int),
                ('a_hl', void_p), ('lda', C.c_int),
                ('b_hl', void_p), ('ldb', C.c_int),
                ('a_scale', void_p), ('b_scale', void_p),
                ('C', void_p), ('ldc', C.c_int),
                ('alpha', C.c_float), ('beta', C.c_float),
                ('bias', void_p),
                ('c_scale', void_p), ('c_scale_period', C.c_int), ('c_scale_ld', C.c_int),
                ('split_k', C.c_int), ('tile', C.c_int), ('k_major', C.c_int),
                ('a_seg_k', C.c_int), ('a_seg_row', C.c_longlong * 16),
                ('batch', C.c_int), ('a_batch_row', C.c_longlong * 16),
                ('clamp_hi', C.c_float)]


class Conv2dArgs(C.Structure):
    _fields_ = [('T_in', C.c_int), ('n_pad', C.c_int), ('F_in', C.c_int), ('C_in', C.c_int),
                ('C_out', C.c_int), ('kt', C.c_int), ('kf', C.c_int), ('st', C.c_int),
                ('sf', C.c_int), ('clip', C.c_float),
                ('x', void_p), ('W', void_p), ('bias', void_p), ('z', void_p), ('y', void_p),
                ('dy', void_p), ('dx', void_p), ('dW', void_p), ('db', void_p),
                ('reuse_x', C.c_int), ('reuse_dz', C.c_int), ('x_absmax', void_p)]


class LstmArgs(C.Structure):
    _fields_ = [('T', C.c_int), ('n_pad', C.c_int), ('H', C.c_int), ('mode', C.c_int),
                ('U', void_p), ('mask_u', void_p),
                ('zx', void_p), ('y', void_p), ('cell', void_p), ('gates', void_p),
                ('dy', void_p), ('dz', void_p), ('dz_absmax', void_p),
                ('step_begin', C.c_int), ('step_count', C.c_int),
                ('mi', void_p), ('uh', void_p), ('zone_c', void_p), ('zone_h', void_p),
                ('wx', void_p), ('dwx', void_p), ('dmi', void_p), ('db_part', void_p),
                ('n_valid', C.c_int), ('lds_reserve_kb', C.c_int), ('compact', C.c_int),
                ('activation', C.c_int), ('fwd_units', C.c_int),
                ('dz_hl', void_p), ('dz_bound', void_p), ('dz_scale_out', void_p)]


class LstmLnArgs(C.Structure):
    _fields_ = [('T', C.c_int), ('n_pad', C.c_int), ('H', C.c_int), ('has_mi', C.c_int),
                ('U', void_p), ('mask_u', void_p), ('cellp', void_p),
                ('zone_c', void_p), ('zone_h', void_p),
                ('wx', void_p), ('uh', void_p), ('y', void_p), ('cell', void_p),
                ('gates', void_p), ('dy', void_p), ('duh', void_p), ('dwx', void_p),
                ('dparams', void_p), ('activation', C.c_int)]


class GateGemmArgs(C.Structure):
    _fields_ = [('rows', C.c_int), ('n_pad', C.c_int), ('in_dim', C.c_int),
                ('gate_dim', C.c_int),
                ('x', void_p), ('ldx', C.c_int),
                ('W', void_p), ('ldw', C.c_int),
                ('bias', void_p), ('mask_w', void_p),
                ('zx', void_p), ('ldz', C.c_int),
                ('dz', void_p), ('dz_absmax', void_p),
                ('dx', void_p), ('dx_beta', C.c_float),
                ('dW', void_p), ('db', void_p),
                ('split_k', C.c_int), ('precision', C.c_int)]


class Segment(C.Structure):
    _fields_ = [('offset', C.c_int64), ('len', C.c_int64), ('l2', C.c_float),
                ('reserved', C.c_float)]


ABI_VERSION = 106      # include/asr_hip.h ASR_HIP_ABI_VERSION: the struct layouts bound above

# name -> (restype, argtypes); also the list the "exports every symbol" test walks
SIGNATURES = {
    'asr_last_error': (C.c_char_p, []),
    'asr_version': (C.c_int, []),
    'asr_device_info': (C.c_int, [c_int_p, c_int_p, C.c_char_p, C.c_int]),
    'asr_frontend_num_frames': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'asr_frontend_num_feats': (C.c_int, [C.POINTER(FrontendCfg)]),
    'asr_frontend_workspace_bytes': (C.c_size_t, [C.POINTER(FrontendCfg), C.c_int, C.c_int]),
    'asr_frontend_features': (C.c_int, [C.POINTER(FrontendCfg), void_p, void_p, void_p,
                                        c_int_p, C.c_int, C.c_int, void_p, void_p, void_p,
                                        void_p, void_p, C.c_int, void_p, void_p, C.c_size_t,
                                        void_p]),
    'asr_gemm_workspace_bytes': (C.c_size_t, [C.POINTER(GemmArgs)]),
    'asr_gemm': (C.c_int, [C.POINTER(GemmArgs), void_p, C.c_size_t, void_p]),
    'asr_pack_hl': (C.c_int, [C.POINTER(PackArgs), void_p]),
    'asr_gemm_hl_workspace_bytes': (C.c_size_t, [C.POINTER(GemmHlArgs)]),
    'asr_gemm_hl': (C.c_int, [C.POINTER(GemmHlArgs), void_p, C.c_size_t, void_p]),
    'asr_gemm_hl_profile': (C.c_int, [C.c_int, C.POINTER(C.c_longlong), void_p]),
    'asr_conv2d_out_shape': (C.c_int, [C.POINTER(Conv2dArgs), c_int_p, c_int_p]),
    'asr_conv2d_workspace_bytes': (C.c_size_t, [C.POINTER(Conv2dArgs)]),
    'asr_conv2d_fwd': (C.c_int, [C.POINTER(Conv2dArgs), void_p, C.c_size_t, void_p]),
    'asr_conv2d_dgrad': (C.c_int, [C.POINTER(Conv2dArgs), void_p, C.c_size_t, void_p]),
    'asr_conv2d_wgrad': (C.c_int, [C.POINTER(Conv2dArgs), void_p, C.c_size_t, void_p]),
    'asr_absmax': (C.c_int, [void_p, C.c_int64, void_p, void_p]),
    'asr_dropout_masks': (C.c_int, [void_p, C.c_int64, C.c_float, C.c_float, C.c_uint64,
                                    C.c_uint32, C.c_uint32, void_p]),
    'asr_dropout_apply': (C.c_int, [void_p, void_p, void_p, C.c_int64, C.c_float, C.c_float,
                                    C.c_uint64, C.c_uint32, C.c_uint32, void_p]),
    'asr_gaussian_noise': (C.c_int, [void_p, void_p, C.c_int64, C.c_float, C.c_uint64,
                                     C.c_uint32, C.c_uint32, void_p]),
    'asr_random_words': (C.c_int, [void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_uint32, void_p]),
    'asr_mul': (C.c_int, [C.c_int64, void_p, void_p, void_p, void_p]),
    'asr_stream_create_cu_mask': (C.c_int, [void_p, C.c_int, C.POINTER(void_p)]),
    'asr_stream_destroy': (C.c_int, [void_p]),
    'asr_optim_guard': (C.c_int, [void_p, void_p, void_p, void_p]),
    'asr_timeout_flags': (C.c_int, [void_p, void_p, void_p, void_p]),
    'asr_debug_occupy': (C.c_int, [C.c_int, C.c_int, C.c_double, void_p]),
    'asr_colsum_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'asr_colsum': (C.c_int, [void_p, C.c_int, C.c_int, C.c_int, void_p, C.c_float, void_p,
                             C.c_size_t, void_p]),
    'asr_lstm_workspace_bytes': (C.c_size_t, [C.POINTER(LstmArgs), C.c_int]),
    'asr_lstm_seq_fwd': (C.c_int, [C.POINTER(LstmArgs), void_p, C.c_size_t, void_p]),
    'asr_lstm_seq_bwd': (C.c_int, [C.POINTER(LstmArgs), void_p, C.c_size_t, void_p]),
    'asr_lstm_status': (C.c_int, [void_p, void_p]),
    'asr_lstm_fast_chains': (C.c_int, [void_p, void_p]),
    'asr_lstm_trace': (C.c_int, [C.POINTER(C.c_longlong), C.c_size_t, void_p]),
    'asr_lstm_profile': (C.c_int, [void_p, void_p, C.POINTER(C.c_longlong)]),
    'asr_lstm_plan': (C.c_int, [C.POINTER(LstmArgs), C.c_int, c_int_p, c_int_p, c_int_p,
                                c_int_p]),
    'asr_lstm_dz_hl_supported': (C.c_int, [C.POINTER(LstmArgs)]),
    'asr_lstm_dz_guard': (C.c_int, [void_p, void_p, C.c_int, void_p, void_p]),
    'asr_ctc_workspace_bytes': (C.c_size_t, [C.c_int] * 5),
    'asr_ctc_loss_grad': (C.c_int, [void_p, void_p, void_p, void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_float, void_p, void_p,
                                    void_p, C.c_size_t, void_p]),
    'asr_ctc_greedy': (C.c_int, [void_p, void_p, C.c_int, C.c_int, C.c_int, C.c_int, void_p,
                                 void_p, void_p]),
    'asr_ctc_beam_search_host': (C.c_int, [void_p, void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, void_p, void_p,
                                           void_p]),
    'asr_ctc_beam_device_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'asr_ctc_beam_device': (C.c_int, [void_p, void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, void_p, void_p, void_p, void_p,
                                      C.c_size_t, void_p]),
    'asr_ctc_beam_device_counters': (C.c_int, [void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, void_p, void_p]),
    'asr_edit_distance_host': (C.c_int, [void_p, void_p, C.c_int, void_p, void_p, C.c_int,
                                         C.c_int, void_p]),
    'asr_optim_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'asr_grad_norm': (C.c_int, [void_p, void_p, C.c_int64, void_p, C.c_int, void_p, void_p,
                                C.c_size_t, void_p]),
    'asr_adam_step': (C.c_int, [void_p, void_p, void_p, void_p, C.c_int64, void_p, C.c_int,
                                void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_int, void_p]),
    'asr_sgd_step': (C.c_int, [void_p, void_p, void_p, C.c_int64, void_p, C.c_int, void_p,
                               C.c_float, C.c_float, C.c_float, void_p]),
    'asr_axpby': (C.c_int, [C.c_int64, C.c_float, void_p, C.c_float, void_p, void_p, void_p]),
    'asr_comm_unique_id': (C.c_int, [void_p]),
    'asr_comm_init': (C.c_int, [void_p, C.c_int, C.c_int, C.POINTER(void_p)]),
    'asr_comm_allreduce_sum': (C.c_int, [void_p, void_p, C.c_int64, void_p]),
    'asr_comm_destroy': (C.c_int, [void_p]),
    'asr_lstm_ln_workspace_bytes': (C.c_size_t, [C.POINTER(LstmLnArgs)]),
    'asr_lstm_ln_seq_fwd': (C.c_int, [C.POINTER(LstmLnArgs), void_p]),
    'asr_lstm_ln_seq_bwd': (C.c_int, [C.POINTER(LstmLnArgs), void_p, C.c_size_t, void_p]),
    # operation-level entry points (csrc/roles.cpp)
    'asr_frontend_mfcc_batch': (C.c_int, [C.POINTER(FrontendCfg), void_p, void_p, void_p,
                                          c_int_p, C.c_int, C.c_int, void_p, void_p, void_p,
                                          void_p, void_p, C.c_int, void_p, void_p, C.c_size_t,
                                          void_p]),
    'asr_frontend_logfbank_batch': (C.c_int, [C.POINTER(FrontendCfg), void_p, void_p, void_p,
                                              c_int_p, C.c_int, C.c_int, void_p, void_p,
                                              void_p, void_p, C.c_int, void_p, void_p,
                                              C.c_size_t, void_p]),
    'asr_gemm_gate_workspace_bytes': (C.c_size_t, [C.POINTER(GateGemmArgs), C.c_int]),
    'asr_gemm_gate_fwd': (C.c_int, [C.POINTER(GateGemmArgs), void_p, C.c_size_t, void_p]),
    'asr_gemm_gate_dgrad': (C.c_int, [C.POINTER(GateGemmArgs), void_p, C.c_size_t, void_p]),
    'asr_gemm_gate_wgrad': (C.c_int, [C.POINTER(GateGemmArgs), void_p, C.c_size_t, void_p]),
    'asr_ctc_beam': (C.c_int, [void_p, void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, void_p, void_p, void_p]),
    'asr_edit_distance': (C.c_int, [void_p, void_p, C.c_int, void_p, void_p, C.c_int,
                                    C.c_int, void_p]),
    'asr_clip_adam_step': (C.c_int, [void_p, void_p, void_p, void_p, C.c_int64, void_p,
                                     C.c_int, void_p, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_int, void_p, C.c_size_t,
                                     void_p]),
    'asr_clip_sgd_step': (C.c_int, [void_p, void_p, void_p, C.c_int64, void_p, C.c_int,
                                    void_p, C.c_float, C.c_float, C.c_float, void_p,
                                    C.c_size_t, void_p]),
    'asr_comm_allreduce': (C.c_int, [void_p, void_p, C.c_int64, void_p]),
}

_lib = None


class AsrHipError(RuntimeError):
    pass


def load():
    """Load libasr_hip.so (once).  Raises if it is not built -- there is no
    Python/CPU fallback for the hot path."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    LIB_PATH = os.environ.get('ASR_LIB_PATH', LIB_PATH)      # (A/B runs of two builds)
    if not os.path.exists(LIB_PATH):
        raise AsrHipError(
            'libasr_hip.so not found at %s -- build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or '
            '`python asr_study_amd/build.py`; the hot path has no CPU fallback.' % LIB_PATH)
    # torch bundles its own HIP runtime; it must be in the process before this
    # library so that both resolve to the SAME libamdhip64 (one device context).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.asr_version() != ABI_VERSION:
        raise AsrHipError('%s reports ABI version %d, this package binds version %d '
                          '(include/asr_hip.h ASR_HIP_ABI_VERSION): rebuild the library'
                          % (LIB_PATH, lib.asr_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().asr_last_error()
        raise AsrHipError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))

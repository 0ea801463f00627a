"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every
symbol include/asr_hip.h declares, and its host-side entry points (beam search,
edit distance) agree with the oracle.  No device compute here."""
import os
import re

import numpy as np
import pytest

from asr_study_amd import _lib as L
from oracle import decode as OD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.load()


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, 'include', 'asr_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(asr_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), 'libasr_hip.so does not export %s' % name
        assert name in L.SIGNATURES, 'no ctypes signature for %s' % name
    assert lib.asr_version() >= 100


def test_frontend_frame_count(lib):
    for n, want in ((160000, 999), (16000, 99), (400, 1), (401, 2), (2, 1)):
        assert lib.asr_frontend_num_frames(n, 400, 160) == want


def test_edit_distance_host(lib):
    from asr_study_amd import ops
    hyps = [[1, 2, 3], [], [5], [1, 1, 1, 1], []]
    truths = [[1, 3], [1, 2], [], [1], []]
    got = ops.edit_distance_host(hyps, truths)
    want = [OD.normalized_edit_distance(h, t) for h, t in zip(hyps, truths)]
    assert np.allclose(got, want) or (np.isinf(got[2]) and np.isinf(want[2]))
    assert got[0] == 0.5 and got[1] == 1.0 and np.isinf(got[2]) and got[3] == 3.0 and got[4] == 0


@pytest.mark.parametrize('beam_width', [1, 4, 25, 100])
def test_beam_search_host_matches_oracle(lib, beam_width):
    from asr_study_amd import ops
    rs = np.random.RandomState(beam_width)
    T, N, n_pad, C = 40, 5, 16, 8
    logits = (rs.randn(T, n_pad, C) * 3).astype(np.float32)
    seq = np.array([40, 31, 1, 17, 40], np.int32)
    paths, score = ops.ctc_beam_search_host(logits, seq, N, beam_width, True)
    for n in range(N):
        want, sc = OD.beam_search_decode_one(logits[:seq[n], n], beam_width,
                                             merge_repeated=True, dtype=np.float64)
        assert paths[n] == want[0], (n, paths[n], want[0])
        assert abs(score[n] - sc[0]) < 1e-3 * max(1.0, abs(sc[0]))
    plain, _ = ops.ctc_beam_search_host(logits, seq, N, beam_width, False)
    for n in range(N):
        want, _ = OD.beam_search_decode_one(logits[:seq[n], n], beam_width,
                                            merge_repeated=False, dtype=np.float64)
        assert plain[n] == want[0]


@pytest.mark.parametrize('width,T', [(100, 60), (400, 36)])
def test_beam_search_host_28_classes_width_100_and_400(lib, width, T):
    """28 classes at the README's width (100) and at the width eval.py really uses (400,
    utils/core_utils.py:70-71): the beam is full from the third frame on."""
    from asr_study_amd import ops
    rs = np.random.RandomState(7 + width)
    N, n_pad, C = 2, 16, 28
    logits = (rs.randn(T, n_pad, C) * 2).astype(np.float32)
    seq = np.array([T, 3 * T // 4], np.int32)
    paths, _ = ops.ctc_beam_search_host(logits, seq, N, width, True)
    for n in range(N):
        want, _ = OD.beam_search_decode_one(logits[:seq[n], n], width, merge_repeated=True,
                                            dtype=np.float64)
        assert paths[n] == want[0]


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(L.AsrHipError):
        L.load()


def test_beam_search_host_tie_breaking_follows_creation_order():
    """Quantised logits force exact score ties: eviction order and final ranking then depend
    on the entries' creation rank (children are ranked as if all C-1 were created when their
    parent first expanded, as TensorFlow's decoder allocates them).  Compared with the
    oracle run in float64 (the library computes in double)."""
    from asr_study_amd import ops
    from oracle import decode as OD
    for seed in range(0, 400, 2):
        rs = np.random.RandomState(seed)
        C, T, W = rs.randint(3, 8), rs.randint(3, 16), rs.randint(1, 9)
        q = 1 + seed % 3
        lg = np.zeros((T, 16, C), np.float32)
        lg[:, 0] = np.round(rs.randn(T, C) * 1.5 * q) / q
        got, _ = ops.ctc_beam_search_host(lg, np.array([T]), 1, W, True)
        want, _ = OD.beam_search_decode_one(lg[:T, 0].astype(np.float64), W, dtype=np.float64)
        assert got[0] == want[0], seed


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The argument structs of include/asr_hip.h are mirrored by hand in _lib.py: compile a
    C program that prints sizeof / a few offsetof of every struct with gcc and compare with
    ctypes (catches ABI drift when a field is added on one side only)."""
    import ctypes as C
    import shutil
    import subprocess
    from asr_study_amd import _lib
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / 'layout.c'
    src.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "asr_hip.h"
int main(void) {
  printf("frontend %zu %zu\\n", sizeof(asr_frontend_cfg), offsetof(asr_frontend_cfg, eps));
  printf("gemm %zu %zu %zu\\n", sizeof(asr_gemm_args), offsetof(asr_gemm_args, bias),
         offsetof(asr_gemm_args, b_absmax));
  printf("lstm %zu %zu %zu %zu %zu\\n", sizeof(asr_lstm_args), offsetof(asr_lstm_args, dz_absmax),
         offsetof(asr_lstm_args, step_begin), offsetof(asr_lstm_args, dmi),
         offsetof(asr_lstm_args, db_part) + 1000 * offsetof(asr_lstm_args, n_valid));
  printf("pack %zu %zu %zu\\n", sizeof(asr_pack_args), offsetof(asr_pack_args, scale_out),
         offsetof(asr_pack_args, ldk_c));
  printf("gemmhl %zu %zu %zu %zu\\n", sizeof(asr_gemm_hl_args), offsetof(asr_gemm_hl_args, b_scale),
         offsetof(asr_gemm_hl_args, bias), offsetof(asr_gemm_hl_args, split_k) + 1000 * offsetof(asr_gemm_hl_args, tile));
  printf("gemmhl2 %zu %zu %zu\\n", offsetof(asr_gemm_hl_args, a_seg_row), offsetof(asr_gemm_hl_args, a_batch_row),
         offsetof(asr_gemm_hl_args, clamp_hi));
  printf("conv %zu %zu %zu %zu %zu\\n", sizeof(asr_conv2d_args), offsetof(asr_conv2d_args, clip),
         offsetof(asr_conv2d_args, y), offsetof(asr_conv2d_args, reuse_dz), offsetof(asr_conv2d_args, x_absmax));
  printf("segment %zu %zu\\n", sizeof(asr_segment), offsetof(asr_segment, l2));
  printf("lstmln %zu %zu %zu\\n", sizeof(asr_lstm_ln_args), offsetof(asr_lstm_ln_args, cellp),
         offsetof(asr_lstm_ln_args, dparams));
  printf("gate %zu %zu %zu %zu\\n", sizeof(asr_gate_gemm_args), offsetof(asr_gate_gemm_args, zx),
         offsetof(asr_gate_gemm_args, dx_beta), offsetof(asr_gate_gemm_args, precision));
  return 0;
}
''')
    exe = tmp_path / 'layout'
    subprocess.check_call([gcc, '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)])
    out = dict((ln.split()[0], [int(v) for v in ln.split()[1:]])
               for ln in subprocess.check_output([str(exe)]).decode().splitlines())
    F, G, Ls, S = _lib.FrontendCfg, _lib.GemmArgs, _lib.LstmArgs, _lib.Segment
    assert out['frontend'] == [C.sizeof(F), F.eps.offset]
    assert out['gemm'] == [C.sizeof(G), G.bias.offset, G.b_absmax.offset]
    assert out['lstm'] == [C.sizeof(Ls), Ls.dz_absmax.offset, Ls.step_begin.offset, Ls.dmi.offset,
                           Ls.db_part.offset + 1000 * Ls.n_valid.offset]
    P, GH = _lib.PackArgs, _lib.GemmHlArgs
    assert out['pack'] == [C.sizeof(P), P.scale_out.offset, P.ldk_c.offset]
    assert out['gemmhl'] == [C.sizeof(GH), GH.b_scale.offset, GH.bias.offset, GH.split_k.offset + 1000 * GH.tile.offset]
    assert out['gemmhl2'] == [GH.a_seg_row.offset, GH.a_batch_row.offset, GH.clamp_hi.offset]
    CV = _lib.Conv2dArgs
    assert out['conv'] == [C.sizeof(CV), CV.clip.offset, CV.y.offset, CV.reuse_dz.offset,
                           CV.x_absmax.offset]
    assert out['segment'] == [C.sizeof(S), S.l2.offset]
    LN = _lib.LstmLnArgs
    assert out['lstmln'] == [C.sizeof(LN), LN.cellp.offset, LN.dparams.offset]
    GG = _lib.GateGemmArgs
    assert out['gate'] == [C.sizeof(GG), GG.zx.offset, GG.dx_beta.offset, GG.precision.offset]


def test_gemm_hl_argument_checks_need_no_gpu(lib):
    """asr_gemm_hl rejects bad arguments before it touches the device: the two operand forms
    have their own leading-dimension rules (row-major planes: ld >= K; k_major planes, whose
    rows are the reduction index: ld >= M / N columns), planes start at a 64-byte group."""
    import ctypes as C
    g = L.GemmHlArgs()
    buf = (C.c_char * 4096)()
    base = (C.addressof(buf) + 63) & ~63
    g.M, g.N, g.K = 64, 64, 128
    g.a_hl, g.b_hl, g.C = base, base, base
    g.lda, g.ldb, g.ldc = 64, 64, 64            # fine for k_major (>= M, N), too short for row-major
    g.alpha = 1.0
    g.k_major = 0
    assert lib.asr_gemm_hl(C.byref(g), None, 0, None) != 0
    assert b'leading' in lib.asr_last_error()
    g.k_major = 1
    g.lda = 48                                  # < M
    assert lib.asr_gemm_hl(C.byref(g), None, 0, None) != 0
    assert b'k_major' in lib.asr_last_error()
    g.lda = 64
    g.a_hl = base + 16                          # not at a reduction group
    assert lib.asr_gemm_hl(C.byref(g), None, 0, None) != 0
    assert b'64-byte' in lib.asr_last_error()


def test_gemm_hl_clamp_needs_the_segmented_form(lib):
    """clamp_hi (the conv front-end's clipped ReLU in the GEMM epilogue) exists in the segmented
    instantiation only: anything else is refused before the device is touched."""
    import ctypes as C
    g = L.GemmHlArgs()
    buf = (C.c_char * 4096)()
    base = (C.addressof(buf) + 63) & ~63
    g.M, g.N, g.K = 64, 64, 128
    g.a_hl, g.b_hl, g.C = base, base, base
    g.lda, g.ldb, g.ldc = 128, 128, 64
    g.alpha = 1.0
    g.clamp_hi = 20.0
    assert lib.asr_gemm_hl(C.byref(g), None, 0, None) != 0
    assert b'clamp_hi' in lib.asr_last_error()


def test_auto_split_fills_the_chip_once():
    """ops._resolve_split_hl: a workgroup of the 256 x 256 kernel has a CU to itself (256 slots),
    the 128 x 128 kernel runs two per CU (512)."""
    from asr_study_amd.ops import _resolve_split_hl as rs
    assert rs('auto', 1024, 4096, 63936) == 4            # dW of a cfg3 layer: 64 tiles
    assert rs('auto', 512, 2048, 63936) == 16            # dU of one direction: 16 tiles
    assert rs('auto', 640, 2048, 32000) == 10            # 24 tiles: 240 workgroups, not 312
    assert rs('auto', 80, 4096, 63936) == 16             # M < 256: the 128 x 128 kernel, 512 slots
    assert rs('auto', 1280, 640, 32000, batch=11) == 1   # more tiles than slots
    assert rs('auto', 256, 256, 512) == 2                # never shorter than 256 reduction indices
    assert rs(7, 1024, 4096, 63936) == 7

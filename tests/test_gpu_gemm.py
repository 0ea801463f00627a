"""-m gpu: fp32 MFMA GEMM through the C ABI vs float64 NumPy (exact-fp32 MFMA:
tolerance = fp32 accumulation round-off, 2e-6 * K * max|a||b|)."""
import numpy as np
import pytest
import torch

from tests.gpu_util import to_dev, report

pytestmark = pytest.mark.gpu

CASES = [
    # M, N, K, trans_a, trans_b, split_k
    (300, 200, 64, False, False, 0),
    (257, 130, 39, False, False, 0),      # first-layer K=39, unaligned lda
    (128, 128, 16, True, False, 0),
    (77, 28, 100, False, True, 0),        # dense C=28 style, trans_b
    (39, 1024, 999, True, False, 4),      # dW style: K long, split-K
    (512, 96, 700, True, True, 3),
    (1000, 260, 28, False, True, 0),      # K=28 (dlogits @ Wd^T)
]


@pytest.mark.parametrize('M,N,K,ta,tb,sk', CASES)
def test_gemm_variants(M, N, K, ta, tb, sk):
    from asr_study_amd import ops
    rs = np.random.RandomState(M + N + K)
    A = rs.randn(K, M) if ta else rs.randn(M, K)
    B = rs.randn(N, K) if tb else rs.randn(K, N)
    C0 = rs.randn(M, N)
    bias = rs.randn(N)
    want = 0.75 * ((A.T if ta else A) @ (B.T if tb else B)) + 0.5 * C0 + bias
    Cd = to_dev(C0.astype(np.float32))
    ops.gemm(to_dev(A.astype(np.float32)), to_dev(B.astype(np.float32)), Cd, M, N, K,
             trans_a=ta, trans_b=tb, alpha=0.75, beta=0.5, bias=to_dev(bias.astype(np.float32)),
             split_k=sk)
    torch.cuda.synchronize()
    err = report('gemm %dx%dx%d ta=%d tb=%d sk=%d' % (M, N, K, ta, tb, sk), Cd.cpu().numpy(), want)
    assert err < 3e-6 * K * 12 + 1e-5


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric B catches swapped row/col in the C write."""
    from asr_study_amd import ops
    M = N = K = 96
    B = np.arange(K * N, dtype=np.float32).reshape(K, N) / 100.0
    Cd = torch.zeros((M, N), dtype=torch.float32, device='cuda:0')
    ops.gemm(to_dev(np.eye(M, dtype=np.float32)), to_dev(B), Cd, M, N, K, precision=0)
    torch.cuda.synchronize()
    assert np.array_equal(Cd.cpu().numpy(), B)          # exact-fp32 MFMA path: bit exact
    ops.gemm(to_dev(np.eye(M, dtype=np.float32)), to_dev(B), Cd, M, N, K, precision=1)
    torch.cuda.synchronize()
    assert np.abs(Cd.cpu().numpy() - B).max() < 2e-6 * B.max()   # split-fp16: 22-bit operands


def test_gemm_masks_and_strided_views():
    """Variational-dropout masks on A rows / C rows and sub-matrix offsets (the
    dU = H_prev^T dZ pattern: A is a column slice of y shifted by one time step)."""
    from asr_study_amd import ops
    rs = np.random.RandomState(0)
    T, NB, H = 9, 16, 20
    y = rs.randn(T * NB, 2 * H).astype(np.float32)            # (T*N, 2H)
    dz = rs.randn(T * NB, 2 * 4 * H).astype(np.float32)       # (T*N, 2, 4H)
    mask = ((rs.rand(NB, H) > 0.3) / 0.7).astype(np.float32)
    # dU_fwd = sum_t (y[t-1,:, :H] * mask)^T dz[t, :, 0:4H]
    want = np.zeros((H, 4 * H))
    for t in range(1, T):
        hm = y[(t - 1) * NB:t * NB, :H].astype(np.float64) * mask
        want += hm.T @ dz[t * NB:(t + 1) * NB, :4 * H]
    out = torch.zeros((H, 4 * H), dtype=torch.float32, device='cuda:0')
    ops.gemm(to_dev(y), to_dev(dz), out, H, 4 * H, (T - 1) * NB, trans_a=True, trans_b=False,
             lda=2 * H, ldb=8 * H, ldc=4 * H, a_scale=to_dev(mask), a_scale_period=NB,
             b_off=NB * 8 * H, split_k=2)
    torch.cuda.synchronize()
    assert report('gemm dU-style', out.cpu().numpy(), want) < 1e-3
    # dX = (dz @ W^T) * maskW   (c_scale, trans_b)
    W = rs.randn(24, 8 * H).astype(np.float32)
    mw = ((rs.rand(NB, 24) > 0.2) / 0.8).astype(np.float32)
    want = (dz.astype(np.float64) @ W.T.astype(np.float64)) * np.tile(mw, (T, 1))
    out = torch.zeros((T * NB, 24), dtype=torch.float32, device='cuda:0')
    ops.gemm(to_dev(dz), to_dev(W), out, T * NB, 24, 8 * H, trans_b=True, c_scale=to_dev(mw),
             c_scale_period=NB)
    torch.cuda.synchronize()
    assert report('gemm dX-style', out.cpu().numpy(), want) < 1e-3
    # colsum
    cs = torch.zeros(8 * H, dtype=torch.float32, device='cuda:0')
    ops.colsum(to_dev(dz), T * NB, 8 * H, 8 * H, cs)
    torch.cuda.synchronize()
    assert report('colsum', cs.cpu().numpy(), dz.astype(np.float64).sum(0)) < 1e-4


@pytest.mark.parametrize('M,N,K,ta,tb,sk', CASES)
def test_gemm_split_fp16_variants(M, N, K, ta, tb, sk):
    """precision=1: split-fp16 MFMA path; tolerance 2e-6 relative to sum|a||b| terms
    (22-bit operands, fp32 accumulation)."""
    from asr_study_amd import ops
    rs = np.random.RandomState(M * 3 + N + K)
    A = rs.randn(K, M) if ta else rs.randn(M, K)
    B = rs.randn(N, K) if tb else rs.randn(K, N)
    A *= 1e-5                                   # gradient-like magnitudes need the pre-scale
    A[0, 0] = 3e-3                              # one outlier sets the scale
    C0 = rs.randn(M, N) * 1e-5
    bias = rs.randn(N) * 1e-5
    want = 0.75 * ((A.T if ta else A) @ (B.T if tb else B)) + 0.5 * C0 + bias
    Ad = to_dev(A.astype(np.float32))
    Cd = to_dev(C0.astype(np.float32))
    ops.gemm(Ad, to_dev(B.astype(np.float32)), Cd, M, N, K, trans_a=ta, trans_b=tb, alpha=0.75,
             beta=0.5, bias=to_dev(bias.astype(np.float32)), split_k=sk, precision=1,
             a_absmax=ops.absmax(Ad))
    torch.cuda.synchronize()
    err = report('gemm16 %dx%dx%d ta=%d tb=%d sk=%d' % (M, N, K, ta, tb, sk), Cd.cpu().numpy(), want)
    assert err < 2e-6 * np.abs(want).max()


def test_absmax():
    from asr_study_amd import ops
    x = torch.randn(1000003, device='cuda:0') * 1e-4
    x[77777] = -0.5
    assert abs(float(ops.absmax(x[:1000000].contiguous())) - 0.5) < 1e-7


@pytest.mark.parametrize('masked', [False, True])
def test_gate_projection_roles(masked):
    """asr_gemm_gate_fwd / _dgrad / _wgrad (one direction's block of a fused two-direction
    layout: ldw = ldz = 2 * gate_dim) against float64 NumPy."""
    from asr_study_amd import ops
    rs = np.random.RandomState(7 + masked)
    T, n_pad, F, G = 11, 16, 40, 48          # gate_dim G = 4H of one direction
    rows = T * n_pad
    x = rs.randn(rows, F)
    W = rs.randn(F, 2 * G) * 0.3
    b = rs.randn(2 * G)
    dz = rs.randn(rows, 2 * G)
    mask = (rs.rand(n_pad, F) > 0.3) / 0.7 if masked else None
    mrows = np.tile(mask, (T, 1)) if masked else 1.0
    xd, Wd, bd, dzd = [to_dev(a.astype(np.float32)) for a in (x, W, b, dz)]
    md = to_dev(mask.astype(np.float32)) if masked else None
    for d in range(2):
        cols = slice(d * G, (d + 1) * G)
        zx = torch.zeros((rows, 2 * G), dtype=torch.float32, device='cuda:0')
        ops.gate_gemm('fwd', rows, n_pad, F, G, Wd, d * G, 2 * G, 2 * G, x=xd, bias=bd[cols],
                      mask_w=md, zx=zx, z_off=d * G)
        torch.cuda.synchronize()
        got = zx.cpu().numpy()
        assert np.abs(got[:, cols] - ((x * mrows) @ W[:, cols] + b[cols])).max() < 2e-4
        assert not got[:, (1 - d) * G:(2 - d) * G].any()      # the other direction untouched
        dx = to_dev(np.ones((rows, F), np.float32))
        ops.gate_gemm('dgrad', rows, n_pad, F, G, Wd, d * G, 2 * G, 2 * G, mask_w=md, dz=dzd,
                      z_off=d * G, dx=dx, dx_beta=1.0)
        torch.cuda.synchronize()
        want = 1.0 + (dz[:, cols] @ W[:, cols].T) * mrows
        assert np.abs(dx.cpu().numpy() - want).max() < 2e-4
        dW = torch.zeros((F, 2 * G), dtype=torch.float32, device='cuda:0')
        db = torch.zeros(G, dtype=torch.float32, device='cuda:0')
        ops.gate_gemm('wgrad', rows, n_pad, F, G, Wd, d * G, 2 * G, 2 * G, x=xd, mask_w=md,
                      dz=dzd, z_off=d * G, dW=dW, dw_off=d * G, db=db, split_k=3)
        torch.cuda.synchronize()
        assert np.abs(dW.cpu().numpy()[:, cols] - (x * mrows).T @ dz[:, cols]).max() < 1e-3
        assert np.abs(db.cpu().numpy() - dz[:, cols].sum(0)).max() < 1e-4


def _hl_ref(x, s):
    """float64 value represented by the (hi, lo) planes of x * s."""
    xs = (x * s).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


@pytest.mark.parametrize('rows,cols,ld,period', [(300, 200, 200, 0), (128, 64, 64, 0),
                                                 (999 * 16, 80, 80, 16), (70, 1024, 1040, 32),
                                                 (257, 40, 40, 0), (480, 72, 72, 48),
                                                 (330, 64, 64, 80)])
def test_pack_hl_planes_both_orientations(rows, cols, ld, period):
    """asr_pack_hl: hi/lo planes equal the reference split of (src * mask) * scale bit for bit,
    in the row orientation (K = columns) and the transposed one (K = rows); padding is zero;
    the scale is the power of two that maps max|src| into [2^8, 2^9)."""
    from asr_study_amd import ops
    rs = np.random.RandomState(rows + cols)
    src = (rs.randn(rows, ld) * 3e-3).astype(np.float32)
    mask = ((rs.rand(period, cols) > 0.2) / 0.8).astype(np.float32) if period else None
    sd = to_dev(src)
    amax = ops.absmax(sd)
    r = ops.HlPlanes(rows, cols, 'cuda:0')
    c = ops.HlPlanes(cols, rows, 'cuda:0')
    for t in (r.hl, c.hl):
        t.fill_(7.0)
    ops.pack_hl(sd, rows, cols, ld=ld, mask=to_dev(mask) if period else None, mask_period=period,
                absmax=amax, r=r, c=c)
    torch.cuda.synchronize()
    s = float(r.scale.cpu().numpy()[0])
    m = np.abs(src).max()
    assert 2.0 ** 8 <= m * s < 2.0 ** 9 and np.log2(s) == np.round(np.log2(s))
    x = src[:, :cols].copy()
    if period:
        x = x * mask[np.arange(rows) % period]
    hi, lo = _hl_ref(x, np.float32(s))
    got_hi, got_lo = r.hi.cpu().numpy(), r.lo.cpu().numpy()
    assert np.array_equal(got_hi[:, :cols], hi) and np.array_equal(got_lo[:, :cols], lo)
    assert not got_hi[:, cols:].any() and not got_lo[:, cols:].any()
    ch, cl = c.hi.cpu().numpy(), c.lo.cpu().numpy()
    assert np.array_equal(ch[:, :rows], hi.T) and np.array_equal(cl[:, :rows], lo.T)
    assert not ch[:, rows:].any() and not cl[:, rows:].any()
    # 22-bit operands: hi + lo reproduces x * s to 2^-22 of the maximum
    back = hi.astype(np.float64) + lo.astype(np.float64)
    assert np.abs(back - x.astype(np.float64) * s).max() < 2.0 ** -21 * m * s


@pytest.mark.parametrize('rows,cols,ld,period', [(999 * 16, 80, 80, 16), (333, 100, 104, 48),
                                                 (64 * 50, 1024, 1024, 64)])
def test_pack_hl_two_masks_in_one_pass(rows, cols, ld, period):
    """asr_pack_args.mask2 / r2_hl (the two directions' dropout masks of a BiLSTM input, the
    slab read once): both plane sets equal, bit for bit, the planes of two separate packs, and
    share one scale; padding columns are zero in both."""
    from asr_study_amd import ops
    rs = np.random.RandomState(rows + cols + 1)
    src = (rs.randn(rows, ld) * 0.3).astype(np.float32)
    m1 = ((rs.rand(period, cols) > 0.2) / 0.8).astype(np.float32)
    m2 = ((rs.rand(period, cols) > 0.2) / 0.8).astype(np.float32)
    sd, d1, d2 = to_dev(src), to_dev(m1), to_dev(m2)
    amax = ops.absmax(sd)
    want = []
    for m in (d1, d2):
        r = ops.HlPlanes(rows, cols, 'cuda:0')
        r.hl.fill_(5.0)
        ops.pack_hl(sd, rows, cols, ld=ld, mask=m, mask_period=period, absmax=amax, r=r)
        want.append((r.hl.clone(), float(r.scale.item())))
    ra, rb = ops.HlPlanes(rows, cols, 'cuda:0'), ops.HlPlanes(rows, cols, 'cuda:0')
    for t in (ra.hl, rb.hl):
        t.fill_(7.0)
    ops.pack_hl(sd, rows, cols, ld=ld, mask=d1, mask_period=period, absmax=amax, r=ra,
                mask2=d2, r2=rb)
    torch.cuda.synchronize()
    assert torch.equal(ra.hl, want[0][0]) and torch.equal(rb.hl, want[1][0])
    assert float(ra.scale.item()) == want[0][1] == float(rb.scale.item())
    x2 = src[:, :cols] * m2[np.arange(rows) % period]
    hi, lo = _hl_ref(x2, np.float32(want[1][1]))
    assert np.array_equal(rb.hi.cpu().numpy()[:, :cols], hi)
    assert np.array_equal(rb.lo.cpu().numpy()[:, :cols], lo)


@pytest.mark.parametrize('M,N,K,sk', [(300, 200, 64, 0), (128, 128, 32, 0), (257, 132, 40, 0),
                                      (700, 512, 160, 0), (1024, 768, 2048, 4),
                                      (80, 1024, 999 * 16, 0), (1000, 260, 96, 0),
                                      (512, 96, 704, 3), (63, 2048, 1024, 0)])
def test_gemm_hl_matches_float64(M, N, K, sk):
    """asr_gemm_hl on packed planes vs float64: the error of a split-fp16 product is 2^-22
    relative per term (tolerance as for asr_gemm precision 1); with bias, alpha, beta and a
    C-row mask, interior and edge tiles, K not a multiple of the 32-wide slab, split-K."""
    from asr_study_amd import ops
    rs = np.random.RandomState(M + N + K)
    A = rs.randn(M, K).astype(np.float32)
    B = (rs.randn(K, N) * 0.05).astype(np.float32)
    C0 = rs.randn(M, N).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    cm = ((rs.rand(16, N) > 0.3) / 0.7).astype(np.float32)
    pa = ops.HlPlanes(M, K, 'cuda:0')
    pb = ops.HlPlanes(N, K, 'cuda:0')
    Ad, Bd = to_dev(A), to_dev(B)
    ops.pack_hl(Ad, M, K, absmax=ops.absmax(Ad), r=pa)
    ops.pack_hl(Bd, K, N, absmax=ops.absmax(Bd), c=pb)         # B^T planes: (N, K)
    Cd = to_dev(C0)
    split = 0 if sk == 0 and K < 4096 else (sk or 'auto')
    ops.gemm_hl(pa, pb, Cd, M, N, K, alpha=0.75, beta=0.5, bias=to_dev(bias),
                c_scale=to_dev(cm), c_scale_period=16, split_k=split)
    torch.cuda.synchronize()
    want = (0.75 * (A.astype(np.float64) @ B.astype(np.float64)) + bias) * cm[np.arange(M) % 16] \
        + 0.5 * C0
    err = report('gemm_hl %dx%dx%d sk=%s' % (M, N, K, split), Cd.cpu().numpy(), want)
    assert err < 2e-6 * np.abs(A).max() * np.abs(B).max() * K + 2e-5


@pytest.mark.parametrize('M,N,K,sk', [(700, 512, 160, 0), (1024, 768, 2048, 4)])
def test_gemm_hl_small_tile_kernel_on_large_outputs(M, N, K, sk):
    """asr_gemm_hl_args.tile = 128 (the 256-thread / 64 KB instance of the kernel) against the
    default 256 x 256 tile: same planes, same products, fp32 sums in the same slab order."""
    from asr_study_amd import ops
    rs = np.random.RandomState(M + K)
    A = rs.randn(M, K).astype(np.float32)
    B = (rs.randn(K, N) * 0.05).astype(np.float32)
    pa = ops.HlPlanes(M, K, 'cuda:0')
    pb = ops.HlPlanes(N, K, 'cuda:0')
    Ad, Bd = to_dev(A), to_dev(B)
    ops.pack_hl(Ad, M, K, absmax=ops.absmax(Ad), r=pa)
    ops.pack_hl(Bd, K, N, absmax=ops.absmax(Bd), c=pb)
    outs = []
    for tile in (0, 128):
        Cd = torch.empty((M, N), dtype=torch.float32, device='cuda:0')
        ops.gemm_hl(pa, pb, Cd, M, N, K, split_k=sk, tile=tile)
        torch.cuda.synchronize()
        outs.append(Cd.cpu().numpy())
    want = A.astype(np.float64) @ B.astype(np.float64)
    for o in outs:
        assert np.abs(o - want).max() < 2e-6 * np.abs(A).max() * np.abs(B).max() * K + 2e-5
    assert np.array_equal(outs[0], outs[1])


def test_gemm_hl_sub_views_and_k_offsets():
    """Row / reduction-range offsets into packed planes: the dU = h_prev^T dz pattern (A's K
    range shifted by one frame against B's) and a column slice of a wider operand."""
    from asr_study_amd import ops
    rs = np.random.RandomState(5)
    T, NB, H = 12, 16, 24
    rows = T * NB
    y = rs.randn(rows, 2 * H).astype(np.float32)
    dz = (rs.randn(rows, 8 * H) * 1e-4).astype(np.float32)
    yd, dzd = to_dev(y), to_dev(dz)
    py = ops.HlPlanes(2 * H, rows, 'cuda:0')
    pdz = ops.HlPlanes(8 * H, rows, 'cuda:0')
    ops.pack_hl(yd, rows, 2 * H, absmax=ops.absmax(yd), c=py)
    ops.pack_hl(dzd, rows, 8 * H, absmax=ops.absmax(dzd), c=pdz)
    K = (T - 1) * NB
    for d in range(2):
        out = torch.zeros((H, 4 * H), dtype=torch.float32, device='cuda:0')
        a_k, b_k = (0, NB) if d == 0 else (NB, 0)
        ops.gemm_hl(py, pdz, out, H, 4 * H, K, a_row=d * H, a_k=a_k, b_row=d * 4 * H, b_k=b_k)
        torch.cuda.synchronize()
        ys = y[a_k:a_k + K, d * H:(d + 1) * H].astype(np.float64)
        zs = dz[b_k:b_k + K, d * 4 * H:(d + 1) * 4 * H].astype(np.float64)
        want = ys.T @ zs
        assert report('gemm_hl dU d=%d' % d, out.cpu().numpy(), want) < \
            2e-6 * np.abs(y).max() * np.abs(dz).max() * K


@pytest.mark.parametrize('M,N,K,sk,tile', [(300, 200, 64, 0, 0), (128, 128, 32, 0, 0), (260, 132, 41, 0, 0),
                                           (700, 512, 160, 0, 0), (1024, 768, 2048, 4, 0),
                                           (80, 1024, 999 * 16, 0, 0), (512, 96, 703, 3, 0),
                                           (1024, 2048, 4000, 'auto', 0), (700, 512, 160, 0, 128)])
def test_gemm_hl_k_major_matches_float64(M, N, K, sk, tile):
    """asr_gemm_hl with k_major: C = A^T B from the (rows = reduction index) planes that x@W and
    dz@W^T use, fragments transposed out of LDS by ds_read_b64_tr_b16 -- what the weight gradients
    x^T dz / h^T dz run on.  Sub-matrices by plane-row and column-group offsets (the one-frame
    shift of dU, one direction's gate columns of dz), any K (row tail masked), split-K, both
    tiles; bias / alpha / beta as in the row-major form."""
    from asr_study_amd import ops
    rs = np.random.RandomState(M + N + K)
    r0, ca, cb = 5, 32, 48                      # plane-row offset, column offsets (groups of 16)
    A = rs.randn(K + r0 + 3, M + ca + 16).astype(np.float32)
    B = (rs.randn(K + r0 + 3, N + cb + 32) * 0.05).astype(np.float32)
    C0 = rs.randn(M, N).astype(np.float32)
    pa = ops.HlPlanes(A.shape[0], A.shape[1], 'cuda:0')
    pb = ops.HlPlanes(B.shape[0], B.shape[1], 'cuda:0')
    Ad, Bd = to_dev(A), to_dev(B)
    ops.pack_hl(Ad, A.shape[0], A.shape[1], absmax=ops.absmax(Ad), r=pa)
    ops.pack_hl(Bd, B.shape[0], B.shape[1], absmax=ops.absmax(Bd), r=pb)
    Cd = to_dev(C0)
    ops.gemm_hl(pa, pb, Cd, M, N, K, a_row=r0, a_k=ca, b_row=r0 + 2, b_k=cb, alpha=0.75, beta=0.5,
                split_k=sk, tile=tile, k_major=True)
    torch.cuda.synchronize()
    As = A[r0:r0 + K, ca:ca + M].astype(np.float64)
    Bs = B[r0 + 2:r0 + 2 + K, cb:cb + N].astype(np.float64)
    want = 0.75 * (As.T @ Bs) + 0.5 * C0
    err = report('gemm_hl k_major %dx%dx%d sk=%s tile=%d' % (M, N, K, sk, tile), Cd.cpu().numpy(), want)
    assert err < 2e-6 * np.abs(A).max() * np.abs(B).max() * K + 2e-5


@pytest.mark.parametrize('M,N,K,opts', [
    (8192, 4352, 160, 'plain'),           # 32 x 17 interior tiles, the shortest K loop (5 slabs)
    (8292, 4200, 328, 'plain'),           # edge tiles in both directions, K tail of 8
    (16384, 2304, 1024, 'bias'),          # x@W form
    (8292, 4200, 256, 'mask'),            # dX form: old C, row mask, alpha / beta
    (8192, 4352, 192, 'view'),            # output = a column range of a wider matrix (ldc > N)
])
def test_gemm_hl_persistent_form_is_bit_identical(M, N, K, opts, monkeypatch):
    """gemm_hlp_kernel (round 6: one workgroup per CU walks the output tiles, the K loop runs on
    into the next tile, dynamic per-XCD tile hand-out) against gemm_hlx_kernel
    (ASR_GEMM_PERSIST=0): same planes, same products in the same order -> every output bit equal;
    launched four times in a row (the control block re-arms itself) and against float64."""
    from asr_study_amd import ops
    rs = np.random.RandomState(M + N + K)
    A = rs.randn(M, K).astype(np.float32)
    B = (rs.randn(K, N) * 0.05).astype(np.float32)
    pa = ops.HlPlanes(M, K, 'cuda:0')
    pb = ops.HlPlanes(N, K, 'cuda:0')
    Ad, Bd = to_dev(A), to_dev(B)
    ops.pack_hl(Ad, M, K, absmax=ops.absmax(Ad), r=pa)
    ops.pack_hl(Bd, K, N, absmax=ops.absmax(Bd), c=pb)
    ldc = N + 64 if opts == 'view' else N
    C0 = rs.randn(M, ldc).astype(np.float32)
    bias = to_dev(rs.randn(N).astype(np.float32))
    cm_h = ((rs.rand(16, N) > 0.3) / 0.7).astype(np.float32)
    cm = to_dev(cm_h)
    kw = {}
    if opts == 'bias':
        kw = dict(bias=bias)
    elif opts == 'mask':
        kw = dict(alpha=0.75, beta=0.5, bias=bias, c_scale=cm, c_scale_period=16)
    elif opts == 'view':
        kw = dict(c_off=32, ldc=ldc)

    def run(persist, reps=1):
        monkeypatch.setenv('ASR_GEMM_PERSIST', persist)
        outs = []
        for _ in range(reps):
            Cd = to_dev(C0)
            ops.gemm_hl(pa, pb, Cd, M, N, K, **kw)
            torch.cuda.synchronize()
            outs.append(Cd)
        return outs
    plain = run('0')[0]
    pers = run('1', reps=4)
    for o in pers:
        assert torch.equal(o, plain)
    got = plain.cpu().numpy()
    Aw, Bw = A.astype(np.float64), B.astype(np.float64)
    if opts == 'mask':
        want = (0.75 * (Aw @ Bw) + bias.cpu().numpy()) * cm_h[np.arange(M) % 16] + 0.5 * C0
    elif opts == 'view':
        prod = Aw @ Bw
        for r in range(0, M, 997):                                    # sampled rows
            assert np.abs(got.reshape(-1)[32 + r * ldc:32 + r * ldc + N] - prod[r]).max() < \
                2e-6 * np.abs(A).max() * np.abs(B).max() * K + 2e-5
        # nothing outside the view was touched
        keep = np.ones(M * ldc, bool)
        for r in range(M):
            keep[32 + r * ldc:32 + r * ldc + N] = False
        assert np.array_equal(got.reshape(-1)[keep], C0.reshape(-1)[keep])
        return
    else:
        want = Aw @ Bw + (bias.cpu().numpy() if opts == 'bias' else 0.0)
    assert np.abs(got - want).max() < 2e-6 * np.abs(A).max() * np.abs(B).max() * K + 2e-5


def test_gemm_hl_persistent_launches_on_two_streams_do_not_share_a_control_block(monkeypatch):
    """Two persistent launches in flight at once (two streams) take different control blocks of
    the ring: both results equal the plain kernel's."""
    from asr_study_amd import ops
    M, N, K = 8192, 4352, 512
    rs = np.random.RandomState(5)
    Ad = to_dev(rs.randn(M, K).astype(np.float32))
    Bd = to_dev((rs.randn(K, N) * 0.05).astype(np.float32))
    pa = ops.HlPlanes(M, K, 'cuda:0')
    pb = ops.HlPlanes(N, K, 'cuda:0')
    ops.pack_hl(Ad, M, K, absmax=ops.absmax(Ad), r=pa)
    ops.pack_hl(Bd, K, N, absmax=ops.absmax(Bd), c=pb)
    monkeypatch.setenv('ASR_GEMM_PERSIST', '0')
    want = torch.empty((M, N), dtype=torch.float32, device='cuda:0')
    ops.gemm_hl(pa, pb, want, M, N, K)
    torch.cuda.synchronize()
    monkeypatch.setenv('ASR_GEMM_PERSIST', '1')
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [torch.zeros((M, N), dtype=torch.float32, device='cuda:0') for _ in range(6)]
    for i, o in enumerate(outs):
        with torch.cuda.stream(s1 if i % 2 == 0 else s2):
            ops.gemm_hl(pa, pb, o, M, N, K)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)


@pytest.mark.parametrize('rows,cols,ld,off,period,two', [
    (999 * 16, 80, 80, 0, 16, True),        # first-layer input, both directions' masks
    (64 * 37, 1024, 1024, 0, 64, True),     # a BiLSTM(512) layer input
    (64 * 37, 512, 1024, 512, 64, False),   # y (.) B_U: the second half of strided rows
    (257, 16, 16, 0, 0, False),             # one group per row, no mask
    (1000, 4096, 4096, 0, 0, False),        # dz-like, no mask
    (333, 48, 64, 8, 48, True),             # period not a power of two, offset rows
])
def test_pack_hl_streaming_row_kernel_is_bit_identical(rows, cols, ld, off, period, two,
                                                       monkeypatch):
    """pack_rows_kernel (r6: row planes only -- four columns per thread, DPP quad exchange, 16-byte
    stores, no LDS) against the tiled pack_hl_kernel (ASR_PACK_ROWS=0) and the NumPy split: same
    planes, same scale, bit for bit; one and two masks, strided / offset sources."""
    from asr_study_amd import ops
    rs = np.random.RandomState(rows + cols + off)
    src = (rs.randn(rows, ld) * 0.3).astype(np.float32)
    m1 = ((rs.rand(max(period, 1), cols) > 0.2) / 0.8).astype(np.float32) if period else None
    m2 = ((rs.rand(max(period, 1), cols) > 0.2) / 0.8).astype(np.float32) if two else None
    sd = to_dev(src)
    d1 = to_dev(m1) if period else None
    d2 = to_dev(m2) if two else None
    amax = ops.absmax(sd)

    def run(mode):
        monkeypatch.setenv('ASR_PACK_ROWS', mode)
        ra, rb = ops.HlPlanes(rows, cols, 'cuda:0'), ops.HlPlanes(rows, cols, 'cuda:0')
        ra.hl.fill_(7.0)
        rb.hl.fill_(7.0)
        ops.pack_hl(sd, rows, cols, ld=ld, src_off=off, mask=d1, mask_period=period, absmax=amax,
                    r=ra, mask2=d2, r2=rb if two else None)
        torch.cuda.synchronize()
        return ra, rb
    a0, b0 = run('0')
    a1, b1 = run('1')
    assert torch.equal(a0.hl, a1.hl) and float(a0.scale.item()) == float(a1.scale.item())
    if two:
        assert torch.equal(b0.hl, b1.hl)
    x = src[:, off:off + cols].copy()
    if period:
        x = x * m1[np.arange(rows) % period]
    hi, lo = _hl_ref(x, np.float32(a1.scale.item()))
    assert np.array_equal(a1.hi.cpu().numpy()[:, :cols], hi)
    assert np.array_equal(a1.lo.cpu().numpy()[:, :cols], lo)

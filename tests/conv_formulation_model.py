"""CPU model (float64 NumPy) of HOW csrc/conv.hip computes the convolution -- the banded matrix
per time tap, the phase layout of the input planes, the per-tap row shifts of the segmented
GEMM, the padded dz planes of the dgrad, the per-tap K-major weight-gradient GEMMs and the fold
of the band gradient -- index for index as the kernels do it (make_geo, tap_row, conv_band_kernel,
conv_pack_kernel, conv_band_reduce_kernel).  tests/test_conv_formulation_model.py checks it
against oracle/conv.py, so an indexing mistake shows up without a GPU."""
import numpy as np


def floordiv(a, b):
    return a // b          # Python's // already floors


class Geo(object):
    def __init__(self, T_in, n_pad, F_in, C_in, C_out, kt, kf, st, sf):
        self.__dict__.update(T_in=T_in, n_pad=n_pad, F_in=F_in, C_in=C_in, C_out=C_out, kt=kt,
                             kf=kf, st=st, sf=sf)
        self.T_out = -(-T_in // st)
        self.F_out = -(-F_in // sf)
        self.pt = max((self.T_out - 1) * st + kt - T_in, 0) // 2
        self.pf = max((self.F_out - 1) * sf + kf - F_in, 0) // 2
        self.Ki, self.Ko = F_in * C_in, self.F_out * C_out
        self.Ki_p, self.Ko_p = -(-self.Ki // 32) * 32, -(-self.Ko // 32) * 32
        self.qmin = floordiv(0 - self.pt, st)
        qmax = floordiv(kt - 1 - self.pt, st)
        self.S = self.T_out + qmax - self.qmin
        self.padb, self.pada = max(kt - 1 - self.pt, 0), self.pt
        self.M = self.T_out * n_pad

    def tap_row(self, dt):
        q = floordiv(dt - self.pt, self.st)
        r = (dt - self.pt) - q * self.st
        return (r * self.S + (q - self.qmin)) * self.n_pad


def band(g, W):
    """band_f (kt*Ki_p, Ko) and band_dg (Ki, kt*Ko_p) as conv_band_kernel writes them."""
    bf = np.zeros((g.kt * g.Ki_p, g.Ko))
    bd = np.zeros((g.Ki, g.kt * g.Ko_p))
    for dt in range(g.kt):
        for k in range(g.Ki):
            fi, ci = divmod(k, g.C_in)
            for fo in range(g.F_out):
                df = fi - g.sf * fo + g.pf
                if 0 <= df < g.kf:
                    bf[dt * g.Ki_p + k, fo * g.C_out:(fo + 1) * g.C_out] = W[dt, df, ci]
                    bd[k, dt * g.Ko_p + fo * g.C_out:dt * g.Ko_p + (fo + 1) * g.C_out] = W[dt, df, ci]
    return bf, bd


def pack_x(g, x):
    """Phase layout: row (r, s, n) <- frame st (s + qmin) + r, zeros outside; width Ki_p."""
    out = np.zeros((g.st * g.S * g.n_pad, g.Ki_p))
    for r in range(g.st):
        for s in range(g.S):
            t = g.st * (s + g.qmin) + r
            if 0 <= t < g.T_in:
                base = (r * g.S + s) * g.n_pad
                out[base:base + g.n_pad, :g.Ki] = x[t]
    return out


def pack_dz(g, dy, z, clip):
    dz = dy * ((z > 0) & (z < clip)) if clip > 0 else dy.copy()
    out = np.zeros(((g.padb + g.T_out + g.pada) * g.n_pad, g.Ko_p))
    out[g.padb * g.n_pad:(g.padb + g.T_out) * g.n_pad, :g.Ko] = dz.reshape(g.M, g.Ko)
    return out, dz


def seg_gemm(A, lda_k, seg_rows, B, M):
    """C = sum_i A[seg_rows[i] : seg_rows[i] + M, :lda_k] @ B[:, i*lda_k:(i+1)*lda_k]^T."""
    C = np.zeros((M, B.shape[0]))
    for i, r in enumerate(seg_rows):
        C += A[r:r + M, :lda_k] @ B[:, i * lda_k:(i + 1) * lda_k].T
    return C


def forward(g, x, W, b, clip):
    bf, _ = band(g, W)
    xp = pack_x(g, x)
    Bpl = bf.T                                    # c planes: (Ko rows, K = kt*Ki_p)
    z = seg_gemm(xp, g.Ki_p, [g.tap_row(dt) for dt in range(g.kt)], Bpl, g.M)
    z = z + np.tile(b, g.F_out)[None]
    z = z.reshape(g.T_out, g.n_pad, g.Ko)
    return (np.clip(z, 0, clip) if clip > 0 else z), z, xp


def dgrad(g, dy, z, W, clip):
    assert g.st == 1
    _, bd = band(g, W)
    dzp, _ = pack_dz(g, dy, z, clip)
    rows = [(g.padb - (g.kt - 1 - g.pt) + (g.kt - 1 - dt)) * g.n_pad for dt in range(g.kt)]
    return seg_gemm(dzp, g.Ko_p, rows, bd, g.M).reshape(g.T_in, g.n_pad, g.Ki)


def wgrad(g, xp, dy, z, clip):
    dzp, dz = pack_dz(g, dy, z, clip)
    dzr = dzp[g.padb * g.n_pad:g.padb * g.n_pad + g.M, :g.Ko]
    dband = np.zeros((g.kt, g.Ki, g.Ko))
    for dt in range(g.kt):
        r = g.tap_row(dt)
        dband[dt] = xp[r:r + g.M, :g.Ki].T @ dzr
    dW = np.zeros((g.kt, g.kf, g.C_in, g.C_out))
    for dt in range(g.kt):
        for df in range(g.kf):
            for fo in range(g.F_out):
                fi = g.sf * fo + df - g.pf
                if 0 <= fi < g.F_in:
                    dW[dt, df] += dband[dt, fi * g.C_in:(fi + 1) * g.C_in,
                                        fo * g.C_out:(fo + 1) * g.C_out]
    db = dz.reshape(g.M, g.F_out, g.C_out).sum(axis=(0, 1))
    return dW, db


# ---- frequency blocks (make_geo's nblk / nwblk rules, rect_geo) -------------------------------

def col_blocks(g):
    """Column blocks of the forward GEMM (256-column tiles of output frequencies):
    [(fi0, nfi, fo0, nfo)]."""
    whole = [(0, g.F_in, 0, g.F_out)]
    if g.C_in % 32 or g.C_out > 256 or 256 % g.C_out:
        return whole
    cb = 256 // g.C_out
    while -(-g.F_out // cb) > 4:
        cb *= 2
    nb = -(-g.F_out // cb)
    out, cost = [], 0
    for b in range(nb):
        fo0 = b * cb
        nfo = min(cb, g.F_out - fo0)
        lo = max(g.sf * fo0 - g.pf, 0)
        hi = min(g.sf * (fo0 + nfo - 1) - g.pf + g.kf, g.F_in)
        hi = max(hi, lo + 1)
        out.append((lo, hi - lo, fo0, nfo))
        cost += -(-(nfo * g.C_out) // 256) * (hi - lo)
    if nb > 1 and cost < -(-g.Ko // 256) * g.F_in:
        return out
    return whole


def row_blocks(g):
    """Row blocks of the dgrad and weight-gradient GEMMs (tile-aligned input-frequency
    ranges)."""
    whole = [(0, g.F_in, 0, g.F_out)]
    if g.C_in % 32 or g.C_out % 32 or 256 % g.C_in:
        return whole
    fpb = 256 // g.C_in
    while -(-g.F_in // fpb) > 8:
        fpb *= 2
    nb = -(-g.F_in // fpb)
    out, tiles = [], 0
    for b in range(nb):
        fi0 = b * fpb
        fi1 = min(fi0 + fpb, g.F_in) - 1
        lo = max(-((-(fi0 + g.pf - (g.kf - 1))) // g.sf), 0)
        hi = min((fi1 + g.pf) // g.sf, g.F_out - 1)
        hi = max(hi, lo)
        out.append((fi0, fi1 - fi0 + 1, lo, hi - lo + 1))
        tiles += -(-((fi1 - fi0 + 1) * g.C_in) // 256) * -(-((hi - lo + 1) * g.C_out) // 256)
    if nb > 1 and tiles < -(-g.Ki // 256) * -(-g.Ko // 256):
        return out
    return whole


def rect_geo(g, fi0, nfi, fo0, nfo):
    import copy
    q = copy.copy(g)
    q.F_in, q.F_out = nfi, nfo
    q.pf = g.pf + fi0 - g.sf * fo0
    q.Ki, q.Ko = nfi * g.C_in, nfo * g.C_out
    q.Ki_p, q.Ko_p = -(-q.Ki // 32) * 32, -(-q.Ko // 32) * 32
    return q


def forward_blocks(g, x, W, b, clip):
    """asr_conv2d_fwd with column blocks: block GEMMs on column ranges of the same x planes."""
    xp = pack_x(g, x)
    z = np.full((g.M, g.Ko), np.nan)
    blocks = col_blocks(g)
    for fi0, nfi, fo0, nfo in blocks:
        gb = rect_geo(g, fi0, nfi, fo0, nfo)
        bf, _ = band(gb, W)
        kseg = g.Ki_p if len(blocks) == 1 else gb.Ki
        A = xp[:, fi0 * g.C_in:]
        zb = seg_gemm(A, kseg, [g.tap_row(dt) for dt in range(g.kt)],
                      np.concatenate([bf[dt * gb.Ki_p:dt * gb.Ki_p + kseg] for dt in range(g.kt)]).T
                      if kseg != gb.Ki_p else bf.T, g.M)
        z[:, fo0 * g.C_out:fo0 * g.C_out + gb.Ko] = zb + np.tile(b, nfo)[None]
    z = z.reshape(g.T_out, g.n_pad, g.Ko)
    return (np.clip(z, 0, clip) if clip > 0 else z), z, xp


def dgrad_blocks(g, dy, z, W, clip):
    """asr_conv2d_dgrad with row blocks: block GEMMs on column ranges of the dz planes, each
    writing its own column range of dx."""
    assert g.st == 1
    dzp, _ = pack_dz(g, dy, z, clip)
    rows = [(g.padb - (g.kt - 1 - g.pt) + (g.kt - 1 - dt)) * g.n_pad for dt in range(g.kt)]
    dx = np.full((g.M, g.Ki), np.nan)
    blocks = row_blocks(g)
    for fi0, nfi, fo0, nfo in blocks:
        gb = rect_geo(g, fi0, nfi, fo0, nfo)
        _, bd = band(gb, W)
        kseg = g.Ko_p if len(blocks) == 1 else gb.Ko
        B = np.concatenate([bd[:, dt * gb.Ko_p:dt * gb.Ko_p + kseg] for dt in range(g.kt)], axis=1)
        dx[:, fi0 * g.C_in:fi0 * g.C_in + gb.Ki] = seg_gemm(dzp[:, fo0 * g.C_out:], kseg, rows, B, g.M)
    return dx.reshape(g.T_in, g.n_pad, g.Ki)


def wgrad_blocks(g, xp, dy, z, clip):
    dzp, dz = pack_dz(g, dy, z, clip)
    dzr = dzp[g.padb * g.n_pad:g.padb * g.n_pad + g.M]
    dW = np.zeros((g.kt, g.kf, g.C_in, g.C_out))
    for fi0, nfi, fo0, nfo in row_blocks(g):
        gb = rect_geo(g, fi0, nfi, fo0, nfo)
        for dt in range(g.kt):
            r = g.tap_row(dt)
            dband = xp[r:r + g.M, fi0 * g.C_in:fi0 * g.C_in + gb.Ki].T @ \
                dzr[:, fo0 * g.C_out:fo0 * g.C_out + gb.Ko]
            for df in range(g.kf):                  # conv_band_reduce_kernel on the block geometry
                for fo in range(gb.F_out):
                    fi = gb.sf * fo + df - gb.pf
                    if 0 <= fi < gb.F_in:
                        dW[dt, df] += dband[fi * g.C_in:(fi + 1) * g.C_in,
                                            fo * g.C_out:(fo + 1) * g.C_out]
    db = dz.reshape(g.M, g.F_out, g.C_out).sum(axis=(0, 1))
    return dW, db

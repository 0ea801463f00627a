"""Oracle: the library's counter-based random streams (csrc/random.hip) restated in NumPy.
TEST INFRASTRUCTURE ONLY.

The reference draws its training noise with TF-1.3 random ops -- GaussianNoise
(core/models.py:250-251), input Dropout (:257-258), the variational dropout masks of every
LSTM (:265-266: ``dropout_W`` / ``dropout_U``, one mask per batch, inverted scaling) and the
zoneout keep masks (core/layers_utils.py:34-42) -- whose stream cannot be replayed outside
TF: PARITY UNPINNED by the reference; parity is defined on the build's own stream instead,
and this module is its independent restatement.

Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
SC'11), pinned on the Random123 known-answer vectors in tests/test_oracle_rng.py.  Element
4 b + j of a tensor is word j of the block with counter (b & 0xffffffff, stream id, step,
b >> 32) under the key (seed & 0xffffffff, seed >> 32)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counters (uint64 arrays holding 32-bit values) -> 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, np.uint64) & MASK for c in (c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def words(n, seed, stream_id, step):
    """The first n 32-bit words of the stream (seed, stream_id, step)."""
    nb = (int(n) + 3) // 4
    b = np.arange(nb, dtype=np.uint64)
    w = philox4x32_10(b & MASK, np.full(nb, stream_id, np.uint64), np.full(nb, step, np.uint64),
                      b >> np.uint64(32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return np.stack(w, axis=1).reshape(-1)[:n]


def uniform(n, seed, stream_id, step):
    """u = (word >> 8) * 2^-24 in [0, 1), float32 (exact)."""
    return ((words(n, seed, stream_id, step) >> np.uint32(8)).astype(np.float32)
            * np.float32(2.0 ** -24))


def keep_mask(n, p, scale, seed, stream_id, step):
    """Inverted-dropout keep mask: scale where u >= p, else 0 (float32)."""
    u = uniform(n, seed, stream_id, step)
    return np.where(u >= np.float32(p), np.float32(scale), np.float32(0)).astype(np.float32)


def normal(n, seed, stream_id, step):
    """Standard normals by Box-Muller on word pairs (float64 arithmetic of the float32
    uniforms the kernel forms): words (w0, w1) -> sqrt(-2 ln u1) (cos, sin)(2 pi u2),
    u1 = (w0 + 1) 2^-32 in (0, 1], u2 = w1 2^-32."""
    w = words(((int(n) + 3) // 4) * 4, seed, stream_id, step).reshape(-1, 2)
    u1 = ((w[:, 0].astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -32)).astype(np.float64)
    u2 = (w[:, 1].astype(np.float32) * np.float32(2.0 ** -32)).astype(np.float64)
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = np.float64(np.float32(6.283185307179586)) * u2
    z = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1).reshape(-1)
    return z[:n]

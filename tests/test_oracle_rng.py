"""oracle/rng.py against the Random123 known-answer vectors of Philox-4x32-10 (the generator
csrc/random.hip implements; kat_vectors of the Random123 distribution), plus the stream
layout and the statistics of the derived masks / normals."""
import numpy as np

from oracle import rng


def test_philox4x32_10_known_answers():
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff),
         (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = rng.philox4x32_10(*[np.array([c], np.uint64) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want


def test_stream_layout_and_statistics():
    w = rng.words(10, seed=(7 << 32) | 5, stream_id=3, step=9)
    b1 = rng.philox4x32_10(np.array([1], np.uint64), np.array([3], np.uint64),
                           np.array([9], np.uint64), np.array([0], np.uint64), 5, 7)
    assert [int(x) for x in w[4:8]] == [int(g[0]) for g in b1]
    assert not np.array_equal(w, rng.words(10, (7 << 32) | 5, 3, 10))       # step matters
    assert not np.array_equal(w, rng.words(10, (7 << 32) | 5, 4, 9))        # stream matters
    m = rng.keep_mask(200000, 0.2, 1.25, seed=1, stream_id=0, step=0)
    assert set(np.unique(m)) == {np.float32(0), np.float32(1.25)}
    assert abs((m > 0).mean() - 0.8) < 5e-3
    z = rng.normal(200000, seed=2, stream_id=1, step=0)
    assert abs(z.mean()) < 1e-2 and abs(z.std() - 1.0) < 1e-2
    u = rng.uniform(1000, 3, 0, 0)
    assert u.dtype == np.float32 and 0.0 <= u.min() and u.max() < 1.0

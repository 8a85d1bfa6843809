"""Known-answer test of the numpy Philox4x32-10 restatement (oracle/philox.py) against the published
Random123 vectors (kat_vectors of the Random123 distribution, philox4x32 10 rounds)."""
import numpy as np

from oracle import philox


def _run(counter, key):
    # the restatement takes (low 64 bits, offset = high 64 bits, seed): feed the full 128-bit counter
    g = np.array([counter[0] | (counter[1] << 32)], dtype=np.uint64)
    off = counter[2] | (counter[3] << 32)
    seed = key[0] | (key[1] << 32)
    return [int(x) for x in philox.philox4x32_10(g, off, seed)[0]]


def test_random123_known_answers():
    assert _run((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _run((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _run((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_range_and_layout():
    u = philox.fill_uniform(1001, 7, 42)
    assert u.shape == (1001,) and u.dtype == np.float32
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(float(u.mean()) - 0.5) < 0.05
    s = philox.sampler_uniforms(3, 175, 7, 42)
    assert s.shape == (3, 175)

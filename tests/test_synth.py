"""Host-side checks of the synthetic-input protocol (legosnark_amd/synth.py)."""
import numpy as np

from legosnark_amd import synth
from legosnark_amd.curve import MONT, R

M64 = (1 << 64) - 1


def ref_xoshiro(seed, count):
    """Scalar xoshiro256** seeded by splitmix64 (the public reference algorithms)."""
    def sm():
        nonlocal seed
        seed = (seed + 0x9E3779B97F4A7C15) & M64
        z = seed
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)
    s = [sm() for _ in range(4)]
    rotl = lambda x, k: ((x << k) | (x >> (64 - k))) & M64
    out = []
    for _ in range(count):
        out.append((rotl((s[1] * 5) & M64, 7) * 9) & M64)
        t = (s[1] << 17) & M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45)
    return out


def test_xoshiro_single_lane_matches_reference_algorithm():
    g = synth.Xoshiro256ss(seed=synth.SEED, lanes=1)
    assert [int(x) for x in g.u64(50)] == ref_xoshiro(synth.SEED, 50)


def test_xoshiro_lanes_are_interleaved_single_streams():
    g = synth.Xoshiro256ss(seed=7, lanes=8)
    got = g.u64(64).reshape(8, 8)
    sm = synth._splitmix64(7, 32)
    # lane 3 = a single-stream generator whose state is splitmix outputs 12..15
    h = synth.Xoshiro256ss(seed=0, lanes=1)
    h.s = [np.array([sm[12 + k]], dtype=np.uint64) for k in range(4)]
    assert [int(x) for x in got[:, 3]] == [int(x) for x in h.u64(8)]


def test_uniform_fr_is_reduced_and_deterministic():
    a = synth.Xoshiro256ss().uniform_fr(5000)
    b = synth.Xoshiro256ss().uniform_fr(5000)
    assert np.array_equal(a, b)
    vals = [synth.limbs_to_int(x) for x in a]
    assert max(vals) < R and len(set(vals)) == 5000
    assert max(vals) > R * 0.99 and min(vals) < R * 0.01          # spread over the whole range


def test_arith_fr_mont_and_dot_and_sum():
    a, b, n = 0x1234567890ABCDEF << 150 | 77, R - 12345, 5000
    x = synth.arith_fr_mont(a, b, n)
    for i in (0, 1, 1023, 1024, 1025, 2047, 2048, 4999):
        assert synth.limbs_to_int(x[i]) == (a + i * b) % R * MONT % R, i
    s = synth.Xoshiro256ss(seed=3).uniform_fr(n)
    rinv = pow(MONT, -1, R)
    want = sum(synth.limbs_to_int(s[i]) * rinv % R * ((a + i * b) % R) for i in range(n)) % R
    assert synth.fr_dot_mont(s, x) == want
    assert synth.fr_sum_mont(s) == sum(synth.limbs_to_int(v) for v in s) * rinv % R
    sm = synth.small_fr_mont(np.arange(300))
    assert all(synth.limbs_to_int(sm[i]) == i * MONT % R for i in range(300))

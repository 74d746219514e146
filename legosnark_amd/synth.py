"""Synthetic inputs of the measurement protocol -- host plumbing.

* PRNG: splitmix64-seeded xoshiro256** with seed 0x4C45474F534E4152 ("LEGOSNAR").  One
  splitmix64 sequence seeds LANES independent xoshiro256** states (lane j takes outputs
  4j..4j+3); output t*LANES + j is the t-th output of lane j.  LANES = 1 is the classic single
  stream; the default (4096) only exists so that numpy can step all lanes at once.
* Scalars: 4 limbs, top limb masked to 62 bits, rejected when >= r: uniform in [0, r), the
  distribution of LFr::random_element (src/examples/cplink.cc:44-49).  The limbs are handed to
  the library as libff's in-memory (Montgomery) representation; a bijection of [0, r), so the
  field elements are uniform as well, and value(limbs) = limbs * 2^-256 mod r.
* Bases: P_i = (a + i*b) * G with a, b from the PRNG, so that an MSM result can be checked by
  an O(n) computation in Fr that touches no elliptic-curve code:
      sum_i s_i * P_i == ((sum_i s_i * (a + i*b)) mod r) * G.
All arrays are numpy uint64 (n, 4), little-endian limbs.
"""
import numpy as np

from .curve import MONT, R

SEED = 0x4C45474F534E4152
LANES = 4096
_M64 = (1 << 64) - 1
_R_LIMBS = np.array([(R >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def _splitmix64(state, count):
    out = np.empty(count, dtype=np.uint64)
    for i in range(count):
        state = (state + 0x9E3779B97F4A7C15) & _M64
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        out[i] = z ^ (z >> 31)
    return out


def _rotl(x, k):
    return (x << np.uint64(k)) | (x >> np.uint64(64 - k))


class Xoshiro256ss:
    """LANES interleaved xoshiro256** generators seeded from one splitmix64 sequence."""

    def __init__(self, seed=SEED, lanes=LANES):
        s = _splitmix64(seed & _M64, 4 * lanes).reshape(lanes, 4)
        self.s = [s[:, k].copy() for k in range(4)]
        self.lanes = lanes

    def _step(self):
        s0, s1, s2, s3 = self.s
        with np.errstate(over="ignore"):
            res = _rotl(s1 * np.uint64(5), 7) * np.uint64(9)
        t = s1 << np.uint64(17)
        s2 = s2 ^ s0
        s3 = s3 ^ s1
        s1 = s1 ^ s2
        s0 = s0 ^ s3
        s2 = s2 ^ t
        s3 = _rotl(s3, 45)
        self.s = [s0, s1, s2, s3]
        return res

    def u64(self, count):
        steps = (count + self.lanes - 1) // self.lanes
        out = np.empty((steps, self.lanes), dtype=np.uint64)
        for t in range(steps):
            out[t] = self._step()
        return out.reshape(-1)[:count]

    def uniform_fr(self, n):
        """(n, 4) limbs of integers uniform in [0, r): mask to 254 bits, reject >= r."""
        out = np.empty((n, 4), dtype=np.uint64)
        have = 0
        while have < n:
            want = int((n - have) * 1.4) + 64
            c = self.u64(4 * want).reshape(want, 4).copy()
            c[:, 3] &= np.uint64((1 << 62) - 1)
            keep = c[_less_than_r(c)]
            take = min(len(keep), n - have)
            out[have:have + take] = keep[:take]
            have += take
        return out

    def fr_int(self):
        """One integer uniform in [0, r)."""
        return limbs_to_int(self.uniform_fr(1)[0])


def _less_than_r(c):
    lt = np.zeros(len(c), dtype=bool)
    eq = np.ones(len(c), dtype=bool)
    for k in (3, 2, 1, 0):
        lt |= eq & (c[:, k] < _R_LIMBS[k])
        eq &= c[:, k] == _R_LIMBS[k]
    return lt


def limbs_to_int(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    return sum(int(a[i]) << (64 * i) for i in range(a.size))


def int_to_limbs(x):
    return np.array([(x >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)


def _add256(a, b):
    """(n,4) + (n,4) or (4,) -> (n,4) sum mod 2^256 and the carry out (bool array)."""
    out = np.empty_like(a)
    carry = np.zeros(len(a), dtype=np.uint64)
    b = np.broadcast_to(b, a.shape)
    for k in range(4):
        s = a[:, k] + b[:, k]
        c1 = s < a[:, k]
        s2 = s + carry
        c2 = s2 < s
        out[:, k] = s2
        carry = (c1 | c2).astype(np.uint64)
    return out, carry.astype(bool)


def _addmod_r(a, b):
    """a + b mod r for reduced limb arrays (values < r < 2^254: no carry out of 256 bits)."""
    s, _ = _add256(a, b)
    ge = ~_less_than_r(s)
    neg_r = int_to_limbs((1 << 256) - R)
    d, _ = _add256(s, neg_r)              # s - r mod 2^256
    s[ge] = d[ge]
    return s


def arith_fr_mont(a, b, n):
    """Montgomery limbs of the field elements a + i*b, i < n (a, b python ints): the scalars
    x_i that turn the generator into the bases P_i = x_i * G."""
    A, B = a % R * MONT % R, b % R * MONT % R
    out = np.empty((n, 4), dtype=np.uint64)
    blk = min(n, 1024)
    cur = A
    for i in range(blk):                  # first block by plain integers
        out[i] = int_to_limbs(cur)
        cur = (cur + B) % R
    done = blk
    while done < n:                       # then double: x_{i+done} = x_i + done*b
        take = min(done, n - done)
        step = int_to_limbs(done * B % R)
        out[done:done + take] = _addmod_r(out[:take], step)
        done += take
    return out


def _ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return (a[:, 0].astype(object) | (a[:, 1].astype(object) << 64) | (a[:, 2].astype(object) << 128)
            | (a[:, 3].astype(object) << 192))


def fr_dot_mont(a_limbs, b_limbs):
    """sum_i value(a_i) * value(b_i) mod r as a python int, where value(x) = x * 2^-256 mod r
    (the field element a libff-layout Montgomery limb array stands for)."""
    acc = int(np.dot(_ints(a_limbs), _ints(b_limbs)))
    rinv = pow(MONT, -1, R)
    return acc * rinv * rinv % R


def fr_sum_mont(a_limbs):
    """sum_i value(a_i) mod r."""
    a = np.asarray(a_limbs, dtype=np.uint64).reshape(-1, 4)
    tot = 0
    for k in range(4):
        lo = int(np.sum(a[:, k] & np.uint64(0xFFFFFFFF), dtype=np.uint64))
        hi = int(np.sum(a[:, k] >> np.uint64(32), dtype=np.uint64))
        tot += (lo + (hi << 32)) << (64 * k)
    return tot * pow(MONT, -1, R) % R


def small_fr_mont(values):
    """Montgomery limbs of small non-negative integers (u[i] = i style inputs,
    src/examples/hadamard.cc:130-135): values is a 1-D integer array < 2^62."""
    v = np.asarray(values, dtype=np.uint64)
    n = len(v)
    # x * 2^256 mod r = x * M mod r with M = 2^256 mod r; do it with python ints in blocks
    M = MONT % R
    out = np.empty((n, 4), dtype=np.uint64)
    red = (v.astype(object) * M) % R
    for k in range(4):
        out[:, k] = ((red >> (64 * k)) & _M64).astype(np.uint64)
    return out

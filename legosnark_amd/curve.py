"""alt_bn128 constants and byte-layout helpers (host plumbing, pure Python ints).

Layout is libff's: 4 x u64 little-endian limbs, Montgomery form, R = 2^256
(SURVEY.md section 8 header)."""
import numpy as np

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
MONT = 1 << 256

G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


def limbs(x):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def fq_mont(x):
    return limbs(x % P * MONT % P)


def fr_mont(x):
    return limbs(x % R * MONT % R)


def generator(group):
    """libff G::one() as Jacobian Montgomery limbs: (12,) for g1, (24,) for g2."""
    if group == "g1":
        return np.concatenate([fq_mont(1), fq_mont(2), fq_mont(1)])
    if group == "g2":
        (x0, x1), (y0, y1) = G2_GEN
        return np.concatenate([fq_mont(x0), fq_mont(x1), fq_mont(y0), fq_mont(y1), fq_mont(1), fq_mont(0)])
    raise ValueError(group)


def infinity(group):
    """libff G::zero() = (0, 1, 0)."""
    w = 12 if group == "g1" else 24
    out = np.zeros(w, dtype=np.uint64)
    out[w // 3: w // 3 + 4] = fq_mont(1)
    return out

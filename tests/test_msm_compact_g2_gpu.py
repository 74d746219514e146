"""GPU parity tests for the four-launch MSM pipeline on G2 (legosnark_amd/csrc/msm_compact.hip instantiated for CurveG2)
and the one-kernel G2 table builder it runs over, through the C-ABI, against the oracle (libff-algorithm restatement,
oracle/bn254.c) on the same inputs, bit-exact after affine normalisation.  The callers it is for: the G2 half of
CommScheme::commit (/root/reference/src/prototools/commit.h:155) and InterpCommScheme::commit
(src/gadgets/lipmaa.cc:27) at the sizes of the small provers (hadamard 12 / 16)."""
import os
import random

import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu
R = o.R
TABLES_ENABLED = os.environ.get("LSA_PRECOMPUTE", "1")[:1] != "0"


def canon(pt):
    return o.g2_canonical_affine(pt)


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")


def k_g2(k):
    return canon(o.g2_mul(o.generator("g2"), o.fr_mont(k % R)))


@pytest.fixture(scope="module")
def handle(lsa):
    """2^14 + 37 un-normalised G2 bases (a + i*b)*G2 with an infinity and a repeated point among them: the handle gets
    its 26 copies from the one-kernel G2 builder (n <= 2^16) and every MSM of up to 2^14 pairs takes the compact pipeline."""
    n = (1 << 14) + 37
    a, b = 0x2F1E3D4C5B6A << 61 | 5, 0x79 << 101 | 17
    bases = o.arith_bases("g2", a, b, n)
    bases[9] = 0                                # infinity
    bases[11] = bases[10]                       # P + P inside a bucket when the digits agree
    ks = [(a + i * b) % R for i in range(n)]
    ks[9] = 0
    ks[11] = ks[10]
    lsa.set_table_threshold(0)
    B = lsa.Bases("g2", bases)
    if not B.has_table():
        pytest.skip("tables disabled (LSA_PRECOMPUTE=0)")
    yield lsa, B, bases, ks
    B.close()


def test_g2_size_sweep_1_to_2_14_vs_oracle(handle):
    """Every power of two and its neighbours from 1 to 2^14, prefixes and offset ranges of the same handle: the
    known-discrete-log identity for every size, the oracle's multi_exp_with_mixed_addition for a handful."""
    lsa, B, bases, ks = handle
    sc, ints = o.random_scalars(len(ks), seed=4343)
    d_s = dev(sc)
    sizes = sorted({m for k in range(15) for m in ((1 << k) - 1, 1 << k, (1 << k) + 1)} | {3, 255, 257, 1000, 5000, 12345} - {0})
    sizes = [m for m in sizes if m <= 1 << 14]
    for m in sizes:
        got = canon(B.msm(d_s, n=m))
        assert got == k_g2(sum(s * x for s, x in zip(ints[:m], ks[:m]))), m
        if m in (1, 2, 3, 64, 257, 1000):
            assert got == canon(o.multi_exp("g2", bases[:m], sc[:m], mode="mixed")), m
    for first, m in ((7, 100), (300, 4096), (16000, 400), (1, 1 << 14)):
        got = canon(B.msm(d_s[first:first + m], n=m, first=first))
        assert got == k_g2(sum(s * x for s, x in zip(ints[first:first + m], ks[first:first + m]))), (first, m)


def test_g2_degenerate_scalars(handle):
    """0, 1, r - 1, the same scalar everywhere, every digit equal, u[i] = i and i^2 (src/examples/hadamard.cc:130-135),
    31-bit values (src/examples/matrixsc.cc:50-53), mostly zeros."""
    lsa, B, bases, ks = handle
    rng = np.random.default_rng(6)
    for m in (1, 2, 700, 4096, 1 << 14):
        shapes = {
            "zeros": [0] * m,
            "ones": [1] * m,
            "minus one": [R - 1] * m,
            "repeated": [0x1234567890ABCDEF1234567890ABCDEF % R] * m,
            "equal digits": [sum(5 << p for p in range(0, 250, 10)) % R] * m,
            "i": list(range(m)),
            "i^2": [i * i for i in range(m)],
            "31-bit": [int(x) for x in rng.integers(0, 1 << 31, size=m)],
            "sparse": [(int(x) if i % 97 == 0 else 0) for i, x in enumerate(rng.integers(0, 1 << 62, size=m))],
            "mixed": [0, 1, R - 1, 2, R - 2][:m] + [int(x) for x in rng.integers(0, 1 << 62, size=max(0, m - 5))],
        }
        for name, vals in shapes.items():
            got = canon(B.msm(dev(o.fr_mont_array(vals)), n=m))
            assert got == k_g2(sum(s * x for s, x in zip(vals, ks[:m]))), (m, name)


def test_g2_generator_copies_and_cancelling_points(lsa):
    """CommScheme's G2 bases are n copies of the generator (src/prototools/commit.h:134-138): every addition inside a
    bucket is P + P or 2P + P; pairs (P, -P) with equal scalars cancel to the point at infinity."""
    lsa.set_table_threshold(0)
    g = o.generator("g2")
    for n in (1, 5, 300, 4096):
        bases = np.tile(g, (n, 1))
        B = lsa.Bases("g2", bases)
        if not B.has_table():
            pytest.skip("tables disabled")
        sc, ints = o.random_scalars(n, seed=78 + n)
        assert canon(B.msm(dev(sc))) == k_g2(sum(ints)), n
        assert canon(B.msm(dev(o.fr_mont_array([1] * n)))) == k_g2(n), n
        B.close()
    n = 256
    pts = o.arith_bases("g2", 99, 5, n // 2)
    neg = pts.copy()
    for i in range(n // 2):                     # -P: (X, -Y, Z) in Fq2 Montgomery limbs
        for c in (8, 12):
            y = o.limbs_to_int(pts[i, c:c + 4])
            neg[i, c:c + 4] = o.int_to_limbs((o.P - y) % o.P)
    B = lsa.Bases("g2", np.concatenate([pts, neg]))
    sc, _ = o.random_scalars(n // 2, seed=3)
    assert canon(B.msm(dev(np.concatenate([sc, sc])))) is None
    B.close()


def test_g2_one_kernel_table_entries(lsa):
    """The copies 2^(pos_j) * P_i from the one-kernel G2 builder, read back through MSMs whose scalars are a single
    digit at one window position -- the result is exactly one table entry (an infinity among the points keeps
    infinite copies) -- and through a random MSM against the oracle."""
    lsa.set_table_threshold(0)
    n = 300
    bases = o.arith_bases("g2", 12345, 678, n)
    bases[5] = 0
    B = lsa.Bases("g2", bases)
    if not B.has_table():
        pytest.skip("tables disabled")
    rng = random.Random(2)
    for pos in (0, 10, 20, 49, 100, 137, 196, 245, 253):
        i = rng.randrange(6, n)
        vals = [0] * n
        vals[i] = (1 << pos) % R
        vals[5] = 12345                          # the infinity's copies stay infinity
        assert canon(B.msm(dev(o.fr_mont_array(vals)))) == k_g2((12345 + 678 * i) * (1 << pos)), (pos, i)
    sc, _ = o.random_scalars(n, seed=21)
    assert canon(B.msm(dev(sc))) == canon(o.multi_exp("g2", bases, sc, mode="mixed"))
    B.close()


def test_g2_compact_agrees_with_the_general_pipeline(handle):
    """The same handle, the same scalars through a segmented call of one segment (never compact)."""
    lsa, B, bases, ks = handle
    import torch
    sc, _ = o.random_scalars(len(ks), seed=124)
    d_s = dev(sc)
    for m in (1, 77, 1024, 1 << 14):
        a = canon(B.msm(d_s, n=m))
        seg = torch.zeros((1, 24), dtype=torch.int64, device="cuda:0")
        B.msm_segments_async(d_s, np.array([0, m], dtype=np.uint64), seg)
        lsa.synchronize()
        assert a == canon(seg.cpu().numpy().view(np.uint64)[0]), m


def test_g2_pipelined_compact_calls_keep_their_results_apart(handle):
    """Back-to-back asynchronous G2 calls of changing sizes share the tail slots with each other (and would with G1
    calls); two calls that write the same output buffer land in call order."""
    import torch
    lsa, B, bases, ks = handle
    sc, ints = o.random_scalars(len(ks), seed=98)
    d_s = dev(sc)
    sizes = [4096, 1, 300, 1 << 14, 17, 2048, 5, 9000, 64, 1000]
    outs = torch.zeros((len(sizes), 24), dtype=torch.int64, device="cuda:0")
    same = torch.zeros(24, dtype=torch.int64, device="cuda:0")
    for rep in range(3):
        for i, m in enumerate(sizes):
            B.msm_async(d_s, outs[i], n=m)
            B.msm_async(d_s, same, n=m)
    lsa.synchronize()
    got = outs.cpu().numpy().view(np.uint64)
    for i, m in enumerate(sizes):
        assert canon(got[i]) == k_g2(sum(s * x for s, x in zip(ints[:m], ks[:m]))), (i, m)
    m = sizes[-1]
    assert canon(same.cpu().numpy().view(np.uint64)) == k_g2(sum(s * x for s, x in zip(ints[:m], ks[:m])))


def test_g2_prefix_table_behind_the_host_entry_point(lsa):
    """multiExpMA<G2> on host vectors (lsa_g2_msm): the first request is served from the plain layout, the second one
    that fits builds the copies of the entry's points in one kernel and runs over them (lsa_msm_host_stats: table == 2);
    every result equals the oracle's; a vector modified in place is a miss."""
    lsa.set_table_threshold(0)
    lsa.crs_cache_clear()
    n = 3000
    bases = o.arith_bases("g2", 434343, 37, n)
    sc, _ = o.random_scalars(n, seed=9)
    want = canon(o.multi_exp("g2", bases, sc, mode="mixed"))
    assert canon(lsa.msm("g2", bases, sc)) == want
    first = lsa.msm_host_stats()
    assert first["cache_hit"] == 0
    seen_table = False
    for m in (2048, 1024, 512, 64, 7, 1, 2048):
        got = canon(lsa.msm("g2", bases[:m], sc[:m]))
        st = lsa.msm_host_stats()
        assert got == canon(o.multi_exp("g2", bases[:m], sc[:m], mode="mixed")), m
        seen_table = seen_table or st["table"] == 2
    if TABLES_ENABLED and os.environ.get("LSA_CRS_PREFIX_TABLE", "1") != "0" and os.environ.get("LSA_NO_COMPACT_G2") is None:
        assert seen_table
        assert canon(lsa.msm("g2", bases, sc)) == want and lsa.msm_host_stats()["table"] == 2
    bases2 = bases.copy()
    bases2[3] = bases2[4]
    assert canon(lsa.msm("g2", bases2[:512], sc[:512])) == canon(o.multi_exp("g2", bases2[:512], sc[:512], mode="mixed"))
    lsa.crs_cache_clear()

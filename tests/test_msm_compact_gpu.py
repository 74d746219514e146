"""GPU parity tests for the four-launch MSM pipeline (legosnark_amd/csrc/msm_compact.hip) and the one-kernel table
builder it runs over, through the C-ABI, against the oracle (libff-algorithm restatement, oracle/bn254.c) on the
same inputs, bit-exact after affine normalisation.  The callers it is for: CPPoly::prove's ladder of MSMs over
prefixes 2^(d-1) .. 1 of its key vector (/root/reference/src/gadgets/poly.h:76-88) and multiExpMA on short vectors
(src/utils/globl.h:63-78)."""
import os
import random

import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu
R = o.R
TABLES_ENABLED = os.environ.get("LSA_PRECOMPUTE", "1")[:1] != "0"


def canon(pt):
    return o.g1_canonical_affine(pt)


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")


@pytest.fixture(scope="module")
def handle(lsa):
    """2^14 + 37 un-normalised bases (a + i*b)*G with an infinity and a repeated point among them: the handle gets its
    copies from the one-kernel builder (n <= 2^16) and every MSM of up to 2^14 pairs takes the compact pipeline."""
    n = (1 << 14) + 37
    a, b = 0x1F2E3D4C5B6A << 60 | 11, 0x77 << 100 | 13
    bases = o.arith_bases("g1", a, b, n)
    bases[9] = 0                                # infinity
    bases[11] = bases[10]                       # P + P inside a bucket when the digits agree
    ks = [(a + i * b) % R for i in range(n)]
    ks[9] = 0
    ks[11] = ks[10]
    lsa.set_table_threshold(0)
    B = lsa.Bases("g1", bases)
    if not B.has_table():
        pytest.skip("tables disabled (LSA_PRECOMPUTE=0)")
    yield lsa, B, bases, ks
    B.close()


def test_size_sweep_1_to_2_14_vs_oracle(handle):
    """Every power of two and its neighbours from 1 to 2^14 (tile boundaries at multiples of 256, one item per bucket up to
    several chunks per bucket), prefixes and offset ranges of the same handle."""
    lsa, B, bases, ks = handle
    sc, ints = o.random_scalars(len(ks), seed=4242)
    d_s = dev(sc)
    sizes = sorted({m for k in range(15) for m in ((1 << k) - 1, 1 << k, (1 << k) + 1)} | {3, 255, 257, 1000, 5000, 12345} - {0})
    sizes = [m for m in sizes if m <= 1 << 14]
    g = o.generator("g1")
    for m in sizes:
        got = canon(B.msm(d_s, n=m))
        k = sum(s * x for s, x in zip(ints[:m], ks[:m])) % R
        assert got == canon(o.g1_mul(g, o.fr_mont(k))), m
        if m in (1, 2, 3, 64, 257, 1000, 4096):
            assert got == canon(o.multi_exp("g1", bases[:m], sc[:m], mode="mixed")), m
    for first, m in ((7, 100), (300, 4096), (16000, 400), (1, 1 << 14)):
        got = canon(B.msm(d_s[first:first + m], n=m, first=first))
        k = sum(s * x for s, x in zip(ints[first:first + m], ks[first:first + m])) % R
        assert got == canon(o.g1_mul(g, o.fr_mont(k))), (first, m)


def test_degenerate_scalars(handle):
    """0, 1, r - 1, the same scalar everywhere, every digit equal, u[i] = i and i^2 (src/examples/hadamard.cc:130-135),
    31-bit values (src/examples/matrixsc.cc:50-53), mostly zeros: buckets with one chunk, with hundreds, with none."""
    lsa, B, bases, ks = handle
    rng = np.random.default_rng(5)
    g = o.generator("g1")
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
            k = sum(s * x for s, x in zip(vals, ks[:m])) % R
            assert got == canon(o.g1_mul(g, o.fr_mont(k))), (m, name)


def test_generator_copies_and_cancelling_points(lsa):
    """CommScheme's bases are n copies of the generator (src/prototools/commit.h:134-138): every addition inside a bucket
    is P + P or 2P + P; and pairs (P, -P) with equal scalars cancel to the point at infinity."""
    lsa.set_table_threshold(0)
    g = o.generator("g1")
    for n in (1, 5, 300, 4096):
        bases = np.tile(g, (n, 1))
        B = lsa.Bases("g1", bases)
        if not B.has_table():
            pytest.skip("tables disabled")
        sc, ints = o.random_scalars(n, seed=77 + n)
        assert canon(B.msm(dev(sc))) == canon(o.g1_mul(g, o.fr_mont(sum(ints) % R))), n
        ones = o.fr_mont_array([1] * n)
        assert canon(B.msm(dev(ones))) == canon(o.g1_mul(g, o.fr_mont(n % R))), n
        B.close()
    n = 512
    pts = o.arith_bases("g1", 99, 5, n // 2)
    aff = [o.g1_canonical_affine(p) for p in pts]
    neg = o.g1_array([(x, o.P - y) for x, y in aff])
    B = lsa.Bases("g1", np.concatenate([pts, neg]))
    sc, _ = o.random_scalars(n // 2, seed=3)
    assert canon(B.msm(dev(np.concatenate([sc, sc])))) is None
    B.close()


def test_pipelined_compact_calls_keep_their_results_apart(handle):
    """Back-to-back asynchronous calls of changing sizes share four tail slots; two calls that write the same output
    buffer must land in call order."""
    import torch
    lsa, B, bases, ks = handle
    sc, ints = o.random_scalars(len(ks), seed=99)
    d_s = dev(sc)
    sizes = [4096, 1, 300, 1 << 14, 17, 2048, 5, 9000, 64, 1000]
    outs = torch.zeros((len(sizes), 12), dtype=torch.int64, device="cuda:0")
    same = torch.zeros(12, dtype=torch.int64, device="cuda:0")
    for rep in range(3):
        for i, m in enumerate(sizes):
            B.msm_async(d_s, outs[i], n=m)
            B.msm_async(d_s, same, n=m)
    lsa.synchronize()
    g = o.generator("g1")
    got = outs.cpu().numpy().view(np.uint64)
    for i, m in enumerate(sizes):
        k = sum(s * x for s, x in zip(ints[:m], ks[:m])) % R
        assert canon(got[i]) == canon(o.g1_mul(g, o.fr_mont(k))), (i, m)
    k = sum(s * x for s, x in zip(ints[:sizes[-1]], ks[:sizes[-1]])) % R
    assert canon(same.cpu().numpy().view(np.uint64)) == canon(o.g1_mul(g, o.fr_mont(k)))


def test_compact_agrees_with_the_general_pipeline(handle, monkeypatch):
    """The same handle, the same scalars: LSA_COMPACT_MAX is read once per process, so the general pipeline is reached
    through a segmented call of one segment (never compact) -- both must give the same point."""
    lsa, B, bases, ks = handle
    import torch
    sc, _ = o.random_scalars(len(ks), seed=123)
    d_s = dev(sc)
    for m in (1, 77, 1024, 1 << 14):
        a = canon(B.msm(d_s, n=m))
        seg = torch.zeros((1, 12), dtype=torch.int64, device="cuda:0")
        B.msm_segments_async(d_s, np.array([0, m], dtype=np.uint64), seg)
        lsa.synchronize()
        assert a == canon(seg.cpu().numpy().view(np.uint64)[0]), m


def test_one_kernel_table_equals_the_stepwise_table(lsa):
    """The copies 2^(pos_j) * P_i from the one-kernel builder (handles of up to 2^16 points) against the 25-step builder
    (explicit threshold above 2^16 points is not needed: G2 and large handles still use it): compared through MSMs whose
    scalars are single digits at every window position -- the result is exactly one table entry."""
    lsa.set_table_threshold(0)
    n = 300
    bases = o.arith_bases("g1", 12345, 678, n)
    B = lsa.Bases("g1", bases)
    if not B.has_table():
        pytest.skip("tables disabled")
    g = o.generator("g1")
    rng = random.Random(1)
    for pos in (0, 10, 20, 49, 100, 137, 196, 245, 253):
        i = rng.randrange(n)
        vals = [0] * n
        vals[i] = (1 << pos) % R
        got = canon(B.msm(dev(o.fr_mont_array(vals))))
        assert got == canon(o.g1_mul(g, o.fr_mont((12345 + 678 * i) * (1 << pos) % R))), (pos, i)
    B.close()


def test_prefix_table_behind_the_host_entry_point(lsa):
    """multiExpMA on host vectors (lsa_g1_msm): the second request over a prefix of a cached vector builds the copies of
    the entry's first points in one kernel and runs over them (lsa_msm_host_stats: table == 2); results equal the oracle's."""
    lsa.set_table_threshold(0)
    lsa.crs_cache_clear()
    n = 5000
    bases = o.arith_bases("g1", 424242, 31, n)
    sc, _ = o.random_scalars(n, seed=8)
    assert canon(lsa.msm("g1", bases, sc)) == canon(o.multi_exp("g1", bases, sc, mode="mixed"))
    for m in (4096, 1024, 512, 64, 7, 1, 4096):
        got = canon(lsa.msm("g1", bases[:m], sc[:m]))
        st = lsa.msm_host_stats()
        assert got == canon(o.multi_exp("g1", bases[:m], sc[:m], mode="mixed")), m
        if TABLES_ENABLED and os.environ.get("LSA_CRS_PREFIX_TABLE", "1") != "0":
            assert st["cache_hit"] == 1 and st["table"] == 2, (m, st)
    # the whole vector again: served by the same prefix table (5000 <= 2^14)
    got = canon(lsa.msm("g1", bases, sc))
    assert got == canon(o.multi_exp("g1", bases, sc, mode="mixed"))
    # a vector modified in place is a miss, never a stale table
    bases2 = bases.copy()
    bases2[3] = bases2[4]
    got = canon(lsa.msm("g1", bases2[:512], sc[:512]))
    assert got == canon(o.multi_exp("g1", bases2[:512], sc[:512], mode="mixed"))
    lsa.crs_cache_clear()


def test_compact_at_its_upper_end_over_a_stepwise_table(lsa):
    """n = 2^17 (512 tiles, 52 mixed additions per lane) and sizes around it on a handle of 2^17 + 300 points whose copies
    come from the 25-step builder (more than 2^16 points): uniform scalars, u[i] = i, one repeated scalar; checked by
    the known-discrete-log identity."""
    n = (1 << 17) + 300
    a, b = 0xABCDEF << 70 | 3, 0x13579B << 33 | 1
    bases = o.arith_bases("g1", a, b, n)
    lsa.set_table_threshold(1)
    try:
        B = lsa.Bases("g1", bases)
        if not B.has_table():
            pytest.skip("tables disabled")
        g = o.generator("g1")
        sc, ints = o.random_scalars(n, seed=17)
        d_s = dev(sc)
        for m in ((1 << 17), (1 << 17) - 1, 70000, (1 << 16) + 1):
            k = sum(s * (a + i * b) for i, s in enumerate(ints[:m])) % R
            assert canon(B.msm(d_s, n=m)) == canon(o.g1_mul(g, o.fr_mont(k))), m
        m = 1 << 17
        for name, vals in (("i", list(range(m))), ("repeated", [0x1234567890ABCDEF % R] * m)):
            k = sum(s * (a + i * b) for i, s in enumerate(vals)) % R
            assert canon(B.msm(dev(o.fr_mont_array(vals)), n=m)) == canon(o.g1_mul(g, o.fr_mont(k))), name
        B.close()
    finally:
        lsa.set_table_threshold(0)


def test_compact_and_general_pipelines_interleaved(lsa):
    """Asynchronous calls that alternate between the two pipelines on one handle (2^18 + 7 points: calls above 2^17 pairs take
    the general pipeline, the others the compact one) plus a segmented call, sharing the eight tail slots and writing some
    results to the same buffer: every result by the discrete-log identity, the shared buffer holds the LAST call's point."""
    import torch
    n = (1 << 18) + 7
    a, b = 0x5A5A5A << 90 | 7, 0x3C3C << 50 | 9
    bases = o.arith_bases("g1", a, b, n)
    lsa.set_table_threshold(1)
    try:
        B = lsa.Bases("g1", bases)
        if not B.has_table():
            pytest.skip("tables disabled")
        sc, ints = o.random_scalars(n, seed=606)
        d_s = dev(sc)
        sizes = [n, 5000, 1 << 18, 1, 100000, (1 << 17) + 1, 64, 1 << 17, 200000, 777]
        outs = torch.zeros((len(sizes), 12), dtype=torch.int64, device="cuda:0")
        same = torch.zeros(12, dtype=torch.int64, device="cuda:0")
        seg = torch.zeros((3, 12), dtype=torch.int64, device="cuda:0")
        for rep in range(2):
            for i, m in enumerate(sizes):
                B.msm_async(d_s, outs[i], n=m)
                B.msm_async(d_s, same, n=m)
                if i == 4:
                    B.msm_segments_async(d_s, np.array([0, 300, 300, 5000], dtype=np.uint64), seg)
        lsa.synchronize()
        g = o.generator("g1")
        pre = [0]
        for i in range(n):
            pre.append((pre[-1] + ints[i] * (a + i * b)) % R)
        got = outs.cpu().numpy().view(np.uint64)
        for i, m in enumerate(sizes):
            assert canon(got[i]) == canon(o.g1_mul(g, o.fr_mont(pre[m]))), (i, m)
        assert canon(same.cpu().numpy().view(np.uint64)) == canon(o.g1_mul(g, o.fr_mont(pre[sizes[-1]])))
        sg = seg.cpu().numpy().view(np.uint64)
        assert canon(sg[0]) == canon(o.g1_mul(g, o.fr_mont(pre[300])))
        assert canon(sg[1]) is None
        k2 = sum(ints[300 + i] * (a + i * b) for i in range(4700)) % R            # a segment multiplies a PREFIX of the bases
        assert canon(sg[2]) == canon(o.g1_mul(g, o.fr_mont(k2)))
        B.close()
    finally:
        lsa.set_table_threshold(0)

"""GPU parity tests for the MSM path, through the C-ABI (ctypes), bit-exact after
affine normalisation against (i) the committed golden vectors, (ii) the oracle on
seeded inputs at sizes it finishes in seconds, (iii) the known-discrete-log identity at
BASELINE.json's full size n = 2^20."""
import random

import os

import numpy as np
import pytest

import oracle_lib as o
from conftest import g1_dec, g2_dec

pytestmark = pytest.mark.gpu
TABLES_ENABLED = os.environ.get("LSA_PRECOMPUTE", "1")[:1] != "0"     # the library's opt-out
P, R = o.P, o.R


def canon(group, pt):
    return o.g1_canonical_affine(pt) if group == "g1" else o.g2_canonical_affine(pt)


def rand_z(group, rng):
    return rng.randrange(2, P) if group == "g1" else (rng.randrange(2, P), rng.randrange(P))


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_golden_vectors(lsa, golden, group):
    dec = g1_dec if group == "g1" else g2_dec
    arr = o.g1_array if group == "g1" else o.g2_array
    rng = random.Random(21)
    for e in golden[group + "_msm"]:
        pts = [dec(p) for p in e["bases"]]
        sc = o.fr_mont_array([int(s, 16) for s in e["scalars"]])
        want = dec(e["result"])
        # normalised (Z=1) and un-normalised (random Z) libff inputs
        for zs in (None, [rand_z(group, rng) for _ in pts]):
            got = lsa.msm(group, arr(pts, zs), sc)
            assert canon(group, got) == want, e["name"]


@pytest.mark.parametrize("group", ["g1", "g2"])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 65, 1000, 1026, 4097])
def test_msm_vs_oracle_seeded(lsa, group, n):
    a, b = 0xA5A5A5A5A5A5 << 40 | 3, 0x5A5A5A5A << 90 | 7
    bases = o.arith_bases(group, a, b, n)        # un-normalised Jacobian
    sc, ints = o.random_scalars(n, seed=1000 + n)
    want = canon(group, o.multi_exp(group, bases, sc, chunks=1, mode="mixed"))
    got = canon(group, lsa.msm(group, bases, sc))
    assert got == want
    k = sum(s * (a + i * b) for i, s in enumerate(ints)) % R
    mul = o.g1_mul if group == "g1" else o.g2_mul
    assert got == canon(group, mul(o.generator(group), o.fr_mont(k)))


def test_msm_empty_and_truncation(lsa):
    z = np.zeros((0, 12), dtype=np.uint64)
    assert canon("g1", lsa.msm("g1", z, np.zeros((0, 4), dtype=np.uint64))) is None
    # multiExpMA truncates to min(|gs|,|xs|)  (src/utils/globl.h:66)
    bases = o.arith_bases("g1", 5, 9, 10)
    sc, _ = o.random_scalars(6, seed=2)
    want = canon("g1", o.multi_exp("g1", bases[:6], sc, mode="mixed"))
    assert canon("g1", lsa.msm("g1", bases, sc)) == want


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_degenerate_inputs(lsa, group):
    """All-equal bases (CommScheme: n copies of the generator, commit.h:134-138),
    all-equal scalars, zeros/ones, infinity bases, small scalars (hadamard.cc:130-135)."""
    n = 3000
    g = o.generator(group)
    w = 12 if group == "g1" else 24
    same = np.tile(g, (n, 1))
    sc, ints = o.random_scalars(n, seed=77)
    mul = o.g1_mul if group == "g1" else o.g2_mul
    # n copies of the generator
    got = canon(group, lsa.msm(group, same, sc))
    assert got == canon(group, mul(g, o.fr_mont(sum(ints) % R)))
    # u[i] = i and i^2
    small = o.fr_mont_array([i for i in range(n)])
    assert canon(group, lsa.msm(group, same, small)) == canon(group, mul(g, o.fr_mont(n * (n - 1) // 2)))
    sq = o.fr_mont_array([i * i for i in range(n)])
    assert canon(group, lsa.msm(group, same, sq)) == canon(group, mul(g, o.fr_mont(sum(i * i for i in range(n)) % R)))
    # all scalars equal (one heavy bucket per window)
    bases = o.arith_bases(group, 11, 13, n)
    eq = np.tile(o.fr_mont(0xABCDEF0123456789ABCDEF), (n, 1))
    assert canon(group, lsa.msm(group, bases, eq)) == canon(group, o.multi_exp(group, bases, eq, mode="mixed"))
    # zeros, ones, r-1 and infinity bases mixed in
    sc2 = sc.copy()
    sc2[::7] = o.fr_mont(0)
    sc2[1::7] = o.fr_mont(1)
    sc2[2::7] = o.fr_mont(R - 1)
    b2 = bases.copy()
    inf = np.zeros(w, dtype=np.uint64)
    inf[w // 3: w // 3 + 4] = o.fq_mont(1)
    b2[3::11] = inf
    assert canon(group, lsa.msm(group, b2, sc2)) == canon(group, o.multi_exp(group, b2, sc2, mode="mixed"))
    # all-zero scalars -> O
    assert canon(group, lsa.msm(group, bases, np.zeros((n, 4), dtype=np.uint64))) is None


def test_msm_cplink_prover_shape(lsa):
    """SubspaceSnark::prove (subspace.cc:78-85): P has 2N+2 entries (trailing N are O),
    w = (0, rF, u) has N+2 entries -> MSM over min() = N+2 pairs, w[0] = 0."""
    N = 1 << 10
    P_ = np.zeros((2 * N + 2, 12), dtype=np.uint64)
    P_[:N + 2] = o.arith_bases("g1", 99, 1234567, N + 2)
    P_[N + 2:, 4:8] = o.fq_mont(1)   # libff zero() = (0,1,0)
    w, _ = o.random_scalars(N + 2, seed=9)
    w[0] = o.fr_mont(0)
    want = canon("g1", o.multi_exp("g1", P_, w, mode="mixed"))
    assert canon("g1", lsa.msm("g1", P_, w)) == want


def test_msm_resident_api_and_linearity_full_size(lsa):
    """n = 2^20 (BASELINE.json config 2): device-resident bases + scalars, checked by the
    known-discrete-log identity and by linearity MSM(s) + MSM(t) == MSM(s + t)."""
    import torch
    n = 1 << 20
    a, b = 0x1F2E3D4C5B6A7988 << 64 | 0x123, 0x0FEDCBA987654321 << 32 | 0x77
    bases = o.arith_bases("g1", a, b, n)
    B = lsa.Bases("g1", bases)
    rng = np.random.default_rng(123)
    raw = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)   # canonical < 2^254 .. reduce below
    raw[:, 3] &= np.uint64((1 << 60) - 1)                           # < 2^252 < r: canonical as is
    raw_t = np.roll(raw, 1, axis=0).copy()

    def to_ints(arr):
        return [int(arr[i, 0]) | int(arr[i, 1]) << 64 | int(arr[i, 2]) << 128 | int(arr[i, 3]) << 192 for i in range(n)]
    si, ti = to_ints(raw), to_ints(raw_t)
    mont_s = o.fr_mont_array(si)
    mont_t = o.fr_mont_array(ti)
    mont_st = o.fr_mont_array([(x + y) % R for x, y in zip(si, ti)])
    dev = torch.device("cuda:0")
    d_s = torch.from_numpy(mont_s.view(np.int64)).to(dev)
    d_t = torch.from_numpy(mont_t.view(np.int64)).to(dev)
    d_st = torch.from_numpy(mont_st.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    r_s, r_t, r_st = B.msm(d_s), B.msm(d_t), B.msm(d_st)
    k = sum(s * (a + i * b) for i, s in enumerate(si)) % R
    assert canon("g1", r_s) == canon("g1", o.g1_mul(o.generator("g1"), o.fr_mont(k)))
    assert canon("g1", o.g1_add(r_s, r_t)) == canon("g1", r_st)
    # sub-range run
    r_half = B.msm(d_s[: n // 2], n=n // 2)
    k2 = sum(s * (a + i * b) for i, s in enumerate(si[: n // 2])) % R
    assert canon("g1", r_half) == canon("g1", o.g1_mul(o.generator("g1"), o.fr_mont(k2)))
    B.close()


def test_normalize(lsa):
    pts = o.arith_bases("g1", 3, 5, 100)
    pts[7] = 0
    pts[7, 4:8] = o.fq_mont(1)
    out = lsa.normalize("g1", pts)
    for i in range(100):
        assert canon("g1", out[i]) == canon("g1", pts[i])
        if i != 7:
            assert np.array_equal(out[i, 8:12], o.fq_mont(1))


def test_sharded_path_on_one_gpu_with_virtual_peer(lsa):
    """Exercises the multi-GPU step (library stream <-> torch stream hand-offs, gather buffer,
    lsa_g1_sum fold) on one GPU: the collective is replaced by a stand-in that supplies the
    peer's partial, computed beforehand by the same library on the other index range."""
    import torch
    from legosnark_amd import sharded
    n = 5000
    bases = o.arith_bases("g1", 424242, 31, n)
    sc, _ = o.random_scalars(n, seed=8)
    lo0, hi0 = sharded.shard_range(n, 2, 0)
    lo1, hi1 = sharded.shard_range(n, 2, 1)
    assert (lo0, hi0, hi1) == (0, n // 2, n) and lo1 == hi0
    peer = lsa.msm("g1", bases[lo1:hi1], sc[lo1:hi1])
    d_peer = torch.from_numpy(peer.view(np.int64)).to("cuda:0")

    class FakeDist:
        def all_gather_into_tensor(self, out, inp):
            out.view(2, 12)[0].copy_(inp)
            out.view(2, 12)[1].copy_(d_peer)

    B = lsa.Bases("g1", bases[lo0:hi0])
    job = sharded.make_gpu_sharded(lsa, "g1", B, 2, 0, dist=FakeDist())
    d_s = torch.from_numpy(sc[lo0:hi0].view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    for _ in range(7):                                                  # more calls than rotating buffer sets
        res = job.run(d_s)
    got = job.result_host(res)
    want = o.multi_exp("g1", bases, sc, chunks=2, mode="multi_exp")     # libff chunked sum
    assert canon("g1", got) == canon("g1", want)
    B.close()


@pytest.mark.parametrize("table_threshold", [1 << 30, 0, 1, 4000])
def test_pipelined_async_calls_with_changing_sizes(lsa, table_threshold):
    """Back-to-back lsa_msm_run_async calls overlap one call's tail (reduce + fold, internal
    stream) with the next call's front.  Results must not depend on that: different sizes
    (different workspace layouts) and output slots, checked against synchronous runs.
    With a table threshold the handle also carries the pre-shifted window copies, and calls
    switch between the merged path (idle device) and the plain one (queued calls)."""
    import torch
    n = 20000
    bases = o.arith_bases("g1", 777, 12345, n)
    lsa.set_table_threshold(table_threshold)
    B = lsa.Bases("g1", bases)
    # (G1 handles of up to 2^16 points carry the copies by default -- built in one kernel -- unless a threshold says otherwise)
    assert B.has_table() == (table_threshold != 1 << 30 and TABLES_ENABLED)
    sizes = [n, 37, 4096, 1, 9000, 12, 20000, 300, 1025, 5, 16384, 2]
    scs = []
    for i, m in enumerate(sizes):
        sc, _ = o.random_scalars(m, seed=500 + i)
        scs.append(torch.from_numpy(sc.view(np.int64)).to("cuda:0"))
    outs = torch.zeros((len(sizes), 12), dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    for rep in range(3):
        for i, m in enumerate(sizes):
            B.msm_async(scs[i], outs[i], n=m)
    lsa.synchronize()
    got = outs.cpu().numpy().view(np.uint64)
    for i, m in enumerate(sizes):
        want = B.msm(scs[i], n=m)
        assert canon("g1", got[i]) == canon("g1", want), (i, m)
        ref = o.multi_exp("g1", bases[:m], scs[i].cpu().numpy().view(np.uint64), mode="mixed")
        assert canon("g1", got[i]) == canon("g1", ref), (i, m)
    B.close()
    lsa.set_table_threshold(0)


@pytest.mark.parametrize("group,n", [("g1", 1), ("g1", 3000), ("g2", 700)])
def test_preshifted_window_tables_vs_oracle(lsa, group, n):
    """Resident bases with the pre-shifted copies 2^(16k)*P (merged windows, msm.hip): forced on
    at small n, full range and sub-ranges, against the oracle; then the same handle below the
    threshold takes the plain path and must agree."""
    import torch
    lsa.set_table_threshold(1)
    try:
        bases = o.arith_bases(group, 4242 + n, 17, n)
        if n > 10:
            bases[5] = 0                                     # infinity among the bases
        B = lsa.Bases(group, bases)
        assert B.has_table() == TABLES_ENABLED
        sc, _ = o.random_scalars(n, seed=900 + n)
        if n > 10:
            sc[0] = o.fr_mont(0); sc[1] = o.fr_mont(1); sc[2] = o.fr_mont(o.R - 1)
        d_s = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        torch.cuda.synchronize()
        got = B.msm(d_s)
        want = o.multi_exp(group, bases, sc, mode="mixed")
        assert canon(group, got) == canon(group, want)
        if n > 10:
            first, m = 7, n - 100
            got = B.msm(d_s[first:first + m], n=m, first=first)
            want = o.multi_exp(group, bases[first:first + m], sc[first:first + m], mode="mixed")
            assert canon(group, got) == canon(group, want)
        lsa.set_table_threshold(1 << 30)                     # same handle, plain path
        got = B.msm(d_s)
        assert canon(group, got) == canon(group, o.multi_exp(group, bases, sc, mode="mixed"))
        B.close()
    finally:
        lsa.set_table_threshold(0)


@pytest.mark.parametrize("use_table", [True, False])
def test_skewed_scalars_at_the_table_threshold(lsa, use_table):
    """n = 2^19 + 3 (just above the default table threshold) with the scalar shapes of the
    reference's examples: u[i] = i and i^2 (src/examples/hadamard.cc:130-135), 31-bit values
    (src/examples/matrixsc.cc:50-53), one repeated scalar, mostly zeros.  Heavy buckets and
    partly filled windows, on the pre-shifted-table path and on the plain one; checked by the
    known-discrete-log identity."""
    import torch
    n = (1 << 19) + 3
    a, b = 0xA5A5A5A5A5A5A5A5A5A5 << 40 | 0x31, 0x1234567 << 20 | 0x5
    bases = o.arith_bases("g1", a, b, n)
    lsa.set_table_threshold(0 if use_table else 1 << 30)
    try:
        B = lsa.Bases("g1", bases)
        assert B.has_table() == (use_table and TABLES_ENABLED)
        rng = np.random.default_rng(9)
        shapes = {
            "i": [i for i in range(n)],
            "i^2": [i * i for i in range(n)],
            "31-bit": [int(x) for x in rng.integers(0, 1 << 31, size=n)],
            "repeated": [0x1234567890ABCDEF1234567890ABCDEF % R] * n,
            "sparse": [(int(x) if i % 1000 == 0 else 0) for i, x in enumerate(rng.integers(0, 1 << 62, size=n))],
        }
        g = o.generator("g1")
        for name, sc in shapes.items():
            d_s = torch.from_numpy(o.fr_mont_array(sc).view(np.int64)).to("cuda:0")
            torch.cuda.synchronize()
            got = B.msm(d_s)
            k = sum(s * (a + i * b) for i, s in enumerate(sc)) % R
            assert canon("g1", got) == canon("g1", o.g1_mul(g, o.fr_mont(k))), name
        B.close()
    finally:
        lsa.set_table_threshold(0)


def test_msm_random_sweep_vs_oracle(lsa):
    """Many small random instances (sizes, both groups, un-normalised bases, scalars of random bit
    lengths, sprinkled zeros / ones / infinities) against the oracle: a net for rare paths of the
    lazy 29-bit arithmetic (zero tests, P + P, P + (-P)) that fixed-size tests may never hit."""
    rng = random.Random(20261001)
    for case in range(36):
        group = "g1" if case % 3 else "g2"
        n = rng.randrange(1, 1200 if group == "g1" else 300)
        bases = o.arith_bases(group, rng.randrange(1, R), rng.randrange(1, R), n)
        bits = rng.choice([1, 8, 31, 64, 127, 128, 200, 254])
        sc, _ = o.random_scalars(n, seed=rng.randrange(1 << 30), bits=bits)
        for _ in range(rng.randrange(0, 4)):
            i = rng.randrange(n)
            sc[i] = o.fr_mont(rng.choice([0, 1, R - 1, 2]))
        if n > 3 and rng.random() < 0.5:
            bases[rng.randrange(n)] = 0                    # infinity
        if n > 3 and rng.random() < 0.5:
            i, j = rng.randrange(n), rng.randrange(n)
            bases[i] = bases[j]                            # repeated base (P + P in a bucket when the digits agree)
            sc[i] = sc[j]
        got = lsa.msm(group, bases, sc)
        want = o.multi_exp(group, bases, sc, mode="mixed")
        assert canon(group, got) == canon(group, want), (case, group, n, bits)


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_segmented_msm_over_prefixes_vs_oracle(lsa, group):
    """lsa_msm_run_segments_async: several MSMs over prefixes of one table-carrying handle in one
    pass (CPPoly::prove's ladder shape, src/gadgets/poly.h:77-86), every segment against the
    oracle; empty segments, a one-element segment, zeros / ones / r-1 among the scalars, an
    infinity among the bases, a non-zero `first`."""
    import torch
    n = 3000 if group == "g1" else 600
    lsa.set_table_threshold(1)
    try:
        bases = o.arith_bases(group, 9001, 77, n)
        bases[11] = 0                                            # infinity
        B = lsa.Bases(group, bases)
        assert B.has_table() == TABLES_ENABLED
        if not TABLES_ENABLED:
            pytest.skip("tables disabled")
        lens = [n, n // 2, 0, 1, 37, 0, n // 4, 5, 64, n - 1]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        sc, _ = o.random_scalars(int(offs[-1]), seed=4242)
        sc[0] = o.fr_mont(0); sc[1] = o.fr_mont(1); sc[2] = o.fr_mont(o.R - 1); sc[n] = o.fr_mont(0)
        d_s = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        w = 12 if group == "g1" else 24
        outs = torch.zeros((len(lens), w), dtype=torch.int64, device="cuda:0")
        for first in (0, 3):
            ok_lens = [min(m, n - first) for m in lens]
            offs2 = np.concatenate([[0], np.cumsum(ok_lens)]).astype(np.uint64)
            B.msm_segments_async(d_s, offs2, outs, first=first)
            lsa.synchronize()
            got = outs.cpu().numpy().view(np.uint64)
            for j, m in enumerate(ok_lens):
                lo = int(offs2[j])
                want = o.multi_exp(group, bases[first:first + m], sc[lo:lo + m], mode="mixed")
                assert canon(group, got[j]) == canon(group, want), (first, j, m)
        # the same slices one call at a time (narrow digits, one segment) must agree as well
        for j, m in enumerate(lens):
            if m:
                lo = int(offs[j])
                assert canon(group, B.msm(d_s[lo:lo + m], n=m)) == canon(group, o.multi_exp(group, bases[:m], sc[lo:lo + m], mode="mixed"))
        B.close()
    finally:
        lsa.set_table_threshold(0)


def _table_positions(n_table):
    """Bit positions of the pre-shifted copies (mirror of table_grid() in csrc/msm.hip)."""
    nbig = 12 if n_table >= 6 << 20 else 13
    base, rem = divmod(255, nbig)
    pos, bit = [], 0
    for k in range(nbig):
        w = base + (1 if k < rem else 0)
        pos += [bit, bit + (w + 1) // 2]
        bit += w
    return pos


@pytest.mark.parametrize("n,step", [(6000, 1), (70000, 2)])
def test_every_scalar_and_every_digit_equal(lsa, n, step):
    """The worst case for the sort's u16 counters: all n scalars are the same value AND all of its
    digits (26 narrow ones at n < 2^16, 13 wide ones above) are equal, so every entry of a tile lands
    in ONE bin -- 26 x 2048 or 13 x 4096 = 53248 per tile.  Plus the heavy-bucket path end to end."""
    import torch
    lsa.set_table_threshold(1)
    try:
        bases = o.arith_bases("g1", 31, 7, n)
        B = lsa.Bases("g1", bases)
        if not B.has_table():
            pytest.skip("tables disabled")
        pos = _table_positions(n)[::step]
        for d in (5, 1):
            s = sum(d << p for p in pos) % R
            sc = np.tile(o.fr_mont(s), (n, 1))
            got = B.msm(torch.from_numpy(sc.view(np.int64)).to("cuda:0"))
            k = s * sum(31 + 7 * i for i in range(n)) % R
            assert canon("g1", got) == canon("g1", o.g1_mul(o.generator("g1"), o.fr_mont(k))), (n, d)
        B.close()
    finally:
        lsa.set_table_threshold(0)


@pytest.mark.parametrize("n", [2500, 70000])
def test_commitment_pair_shares_the_sort(lsa, n):
    """lsa_commit_run_async = CommScheme::commit's G1 + G2 MSM over one scalar vector
    (src/prototools/commit.h:154-155) with one shared scalar sort: both results against the oracle
    (narrow digits at n = 2500, wide digits at n = 70000), then again through handles without
    copies (two independent calls inside)."""
    import torch
    b1 = o.arith_bases("g1", 5, 3, n)
    b2 = o.arith_bases("g2", 7, 11, n)
    sc, ints = o.random_scalars(n, seed=n)
    d_s = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    k1 = sum(s * (5 + 3 * i) for i, s in enumerate(ints)) % R
    k2 = sum(s * (7 + 11 * i) for i, s in enumerate(ints)) % R
    want1 = canon("g1", o.g1_mul(o.generator("g1"), o.fr_mont(k1)))
    want2 = canon("g2", o.g2_mul(o.generator("g2"), o.fr_mont(k2)))
    for thr in (1, 1 << 30):
        lsa.set_table_threshold(thr)
        try:
            B1, B2 = lsa.Bases("g1", b1), lsa.Bases("g2", b2)
            o1 = torch.zeros(12, dtype=torch.int64, device="cuda:0")
            o2 = torch.zeros(24, dtype=torch.int64, device="cuda:0")
            for _ in range(3):                                   # back to back: tails overlap the next pair
                lsa.commit_async(B1, B2, d_s, o1, o2)
            lsa.synchronize()
            assert canon("g1", o1.cpu().numpy().view(np.uint64)) == want1, thr
            assert canon("g2", o2.cpu().numpy().view(np.uint64)) == want2, thr
            m = n - 37                                           # a prefix
            lsa.commit_async(B1, B2, d_s, o1, o2, n=m)
            lsa.synchronize()
            assert canon("g1", o1.cpu().numpy().view(np.uint64)) == canon("g1", o.multi_exp("g1", b1[:m], sc[:m], mode="mixed"))
            assert canon("g2", o2.cpu().numpy().view(np.uint64)) == canon("g2", o.multi_exp("g2", b2[:m], sc[:m], mode="mixed"))
            B1.close(); B2.close()
        finally:
            lsa.set_table_threshold(0)


def test_table_handle_size_boundaries_and_random_segments(lsa):
    """One table-carrying handle of 70000 points: prefixes at the tile (2048 / 4096 scalars) and
    narrow/wide (2^16) boundaries with uniform, 31-bit and all-ones scalars, then randomly cut
    segment lists -- all by the discrete-log identity (bases (a + i*b)*G)."""
    import torch
    rng = random.Random(77)
    N = 70000
    a, b = 0x1234567 << 100 | 5, 0x7654321 << 64 | 9
    bases = o.arith_bases("g1", a, b, N)
    lsa.set_table_threshold(1)
    try:
        B = lsa.Bases("g1", bases)
        if not B.has_table():
            pytest.skip("tables disabled")
        sc, ints = o.random_scalars(N, seed=5)
        small = [rng.randrange(1 << 31) for _ in range(N)]
        ones = [1] * N
        g = o.generator("g1")

        def want(vals, lo, m):
            return canon("g1", o.g1_mul(g, o.fr_mont(sum(v * (a + i * b) for i, v in enumerate(vals[lo:lo + m])) % R)))

        for vals, arr in ((ints, sc), (small, o.fr_mont_array(small)), (ones, o.fr_mont_array(ones))):
            d = torch.from_numpy(arr.view(np.int64)).to("cuda:0")
            for m in (1023, 1024, 2047, 2048, 2049, 4095, 4096, 4097, 65535, 65536, 65537, N):
                assert canon("g1", B.msm(d[:m], n=m)) == want(vals, 0, m), m
        d = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        for _ in range(4):
            nseg = rng.randrange(1, 40)
            lens = [rng.choice([0, 1, 2, 63, 64, 65, rng.randrange(1, 3000)]) for _ in range(nseg)]
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            outs = torch.zeros((nseg, 12), dtype=torch.int64, device="cuda:0")
            B.msm_segments_async(d, offs, outs)
            lsa.synchronize()
            got = outs.cpu().numpy().view(np.uint64)
            for j, m in enumerate(lens):
                lo = int(offs[j])
                k = sum(ints[lo + i] * (a + i * b) for i in range(m)) % R
                assert canon("g1", got[j]) == canon("g1", o.g1_mul(g, o.fr_mont(k))), (nseg, j, m)
        # a handle without copies refuses segmented calls loudly
        lsa.set_table_threshold(1 << 30)
        B2 = lsa.Bases("g1", bases[:100])
        with pytest.raises(lsa.LsaError):
            B2.msm_segments_async(d, np.array([0, 10, 20], dtype=np.uint64), torch.zeros((2, 12), dtype=torch.int64, device="cuda:0"))
        B2.close()
        B.close()
    finally:
        lsa.set_table_threshold(0)

"""GPU parity tests for libff's G2_precomp path (C-ABI): the line-coefficient tables of
precompute_G2, Miller loops over them, whole verifier checks as one product with conjugated terms,
the device's table cache and shared accumulators.  Everything is a canonical field element, so it
is compared byte-for-byte with the oracle (oracle/bn254.c: oracle_precompute_g2 /
oracle_miller_loop restate libff's alt_bn128_ate_precompute_G2 / alt_bn128_ate_miller_loop).
Reference call sites: /root/reference/src/gadgets/subspace.cc:48,66-70,152-166,
src/gadgets/lipmaa.cc:187-207, src/gadgets/poly.h:97-121, src/utils/globl.h:94-105."""
import os
import random

import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu
P = o.P


def _g2_points(n, seed):
    """n G2 points: normalised, un-normalised (random Z) and one at infinity."""
    rng = random.Random(seed)
    qs = o.arith_bases("g2", 4242 + seed, 17, n)            # un-normalised Jacobian
    if n >= 2:
        qs[1] = o.generator("g2")                            # Z = 1
    if n >= 4:
        qs[3] = 0                                            # infinity: libff precomputes from (0, 1, 0)
    return qs


def test_g2_precompute_matches_the_oracle(lsa):
    assert lsa.lib().lsa_g2_precomp_bytes() == 8 * o.G2_PRECOMP_WORDS == (2 + 3 * 102) * 64
    for n in (1, 5, 6, 13):
        qs = _g2_points(n, n)
        got = lsa.g2_precompute(qs)
        for i in range(n):
            assert np.array_equal(got[i], o.precompute_g2(qs[i])), (n, i)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 17, 33, 64, 70, 128, 129])
def test_miller_loop_over_tables(lsa, n):
    ps = o.arith_bases("g1", 900 + n, 13, n)
    if n >= 3:
        ps[2] = 0                                            # P at infinity: precompute_G1 gives (0, 1)
    qs = _g2_points(min(n, 6), 100 + n)
    tabs = lsa.g2_precompute(qs)
    idx = [i % len(qs) for i in range(n)]
    got = lsa.miller_loop_precomp(ps, tabs, idx)
    for i in range(n):
        assert np.array_equal(got[i], o.miller_loop_precomp(ps[i], o.precompute_g2(qs[idx[i]]))), i
    # the same Miller values from the points themselves (precompute_G2 done on the device)
    assert np.array_equal(got, lsa.miller_loop(ps, qs[idx]))
    assert np.array_equal(got, o.miller_loop_batch(ps, qs[idx]))


def _oracle_terms(ps, qs, flags, off, final_exp):
    f = o.miller_loop_batch(ps, qs)
    out = []
    for j in range(len(off) - 1):
        acc = o.fq12_one()
        for i in range(int(off[j]), int(off[j + 1])):
            acc = o.fq12_mul(acc, o.fq12_unitary_inverse(f[i]) if flags[i] & 1 else f[i])
        out.append(o.final_exponentiation(acc) if final_exp else acc)
    return np.array(out)


@pytest.mark.parametrize("chunk", [0, 1, 2, 3, 4])
def test_products_with_conjugated_terms_and_shared_accumulators(lsa, chunk):
    """lsa_pairing_terms: segments of 0..9 terms, tables and points mixed, conjugated terms; the pairs of a
    product share accumulators of `chunk` pairs (0: automatic) -- the value must not depend on it."""
    rng = random.Random(31 + chunk)
    n = 41
    ps = o.arith_bases("g1", 77, 5, n)
    qs_distinct = _g2_points(7, 9)
    idx = [rng.randrange(7) for _ in range(n)]
    qs = qs_distinct[idx]
    tabs = lsa.g2_precompute(qs_distinct)
    flags = np.array([rng.randrange(2) for _ in range(n)], dtype=np.uint8)
    cuts = sorted(rng.sample(range(1, n), 8))
    off = np.array([0, 0] + cuts + [cuts[-1], n], dtype=np.uint64)     # an empty segment first and in the middle
    lsa.pairing_set_chunk(chunk)
    try:
        for final_exp in (False, True):
            want = _oracle_terms(ps, qs, flags, off, final_exp)
            got_pts = lsa.pairing_terms(ps, off, g2=qs, flags=flags, final_exp=final_exp)
            assert np.array_equal(got_pts, want)
            got_tab = lsa.pairing_terms(ps, off, tables=tabs, index=idx, flags=flags, final_exp=final_exp)
            assert np.array_equal(got_tab, want)
            mixed = [i if k % 2 else -1 for k, i in enumerate(idx)]     # every other term by point
            got_mix = lsa.pairing_terms(ps, off, g2=qs, tables=tabs, index=mixed, flags=flags, final_exp=final_exp)
            assert np.array_equal(got_mix, want)
        # the existing shapes run through the same path
        assert np.array_equal(lsa.pairing_product(ps, qs), o.pairing_product(ps, qs))
        assert np.array_equal(lsa.pairing_product_segments(ps, qs, off), _oracle_terms(ps, qs, np.zeros(n, np.uint8), off, True))
    finally:
        lsa.pairing_set_chunk(0)


@pytest.mark.parametrize("nseg,per", [(20, 2), (33, 2), (45, 3), (64, 2), (100, 2), (128, 1), (140, 2)])
def test_split_miller_loops_every_group_size(lsa, nseg, per):
    """Miller loops over resident tables on the row engine: up to 32 / 64 / 128 accumulators are shared out over 8 / 4 / 2
    workgroups each -- contiguous ranges of the loop's steps, cut even by rt_split (csrc/tmiller.h) -- and beyond that one
    workgroup runs the whole loop.  Products of `per` terms (accumulators of two pairs and of one), conjugated terms, raw
    Miller products and GT values against the oracle."""
    rng = random.Random(1000 + nseg)
    n = nseg * per
    ps = o.arith_bases("g1", 4242 + nseg, 17, n)
    qs_distinct = _g2_points(5, 21)
    idx = [rng.randrange(5) for _ in range(n)]
    qs = qs_distinct[idx]
    tabs = lsa.g2_precompute(qs_distinct)
    flags = np.array([rng.randrange(2) for _ in range(n)], dtype=np.uint8)
    off = np.arange(0, n + 1, per, dtype=np.uint64)
    want = _oracle_terms(ps, qs, flags, off, False)
    got = lsa.pairing_terms(ps, off, tables=tabs, index=idx, flags=flags, final_exp=False)
    assert np.array_equal(got, want)
    gt = lsa.pairing_terms(ps, off, tables=tabs, index=idx, flags=flags, final_exp=True)
    for j in range(0, nseg, max(1, nseg // 6)):
        assert np.array_equal(gt[j], o.final_exponentiation(want[j])), j


def test_a_verifier_check_is_one_product(lsa):
    """simple_pairing_check (src/utils/globl.h:94-105): final_exponentiation(e(a1,a2) * e(b1,b2).unitary_inverse())
    == 1 iff the pairings agree; here as ONE product with a conjugated term."""
    g1, g2 = o.generator("g1"), o.generator("g2")
    a, b = 123456789, 987654321
    a1 = o.g1_mul(g1, o.fr_mont(a * b % o.R))
    b1 = o.g1_mul(g1, o.fr_mont(a))
    b2 = o.g2_mul(g2, o.fr_mont(b))
    ps = np.array([a1, b1])
    qs = np.array([g2, b2])
    off = np.array([0, 2], dtype=np.uint64)
    out = lsa.pairing_terms(ps, off, g2=qs, flags=[0, 1])
    assert np.array_equal(out[0], o.fq12_one())
    bad = lsa.pairing_terms(np.array([a1, g1]), off, g2=qs, flags=[0, 1])
    assert not np.array_equal(bad[0], o.fq12_one())
    assert np.array_equal(bad[0], _oracle_terms(np.array([a1, g1]), qs, [0, 1], off, True)[0])


def test_table_cache_hits_evictions_and_off(lsa):
    ps = o.arith_bases("g1", 5, 3, 6)
    qs = _g2_points(6, 77)
    want = o.miller_loop_batch(ps, qs)
    try:
        lsa.g2_table_cache(4096)
        s0 = lsa.g2_table_cache_stats()
        assert s0["resident"] == 0
        assert np.array_equal(lsa.miller_loop(ps, qs), want)
        s1 = lsa.g2_table_cache_stats()
        assert s1["misses"] - s0["misses"] == 6 and s1["resident"] == 6
        assert np.array_equal(lsa.miller_loop(ps, qs), want)          # second sight: every table is resident
        s2 = lsa.g2_table_cache_stats()
        assert s2["hits"] - s1["hits"] == 6 and s2["misses"] == s1["misses"]
        # a changed point is a different key, never a stale table
        qs2 = qs.copy()
        qs2[0] = o.g2_mul(o.generator("g2"), o.fr_mont(99))
        assert np.array_equal(lsa.miller_loop(ps, qs2), o.miller_loop_batch(ps, qs2))
        # capacity 2: six tables in one call -- four of them live in scratch for that call; LRU eviction across calls
        lsa.g2_table_cache(2)
        assert np.array_equal(lsa.miller_loop(ps, qs), want)
        assert lsa.g2_table_cache_stats()["resident"] == 2
        assert np.array_equal(lsa.miller_loop(ps[3:], qs[3:]), want[3:])
        assert np.array_equal(lsa.miller_loop(ps, qs), want)
        assert lsa.g2_table_cache_stats()["evictions"] > 0
        # off
        lsa.g2_table_cache(0)
        assert np.array_equal(lsa.miller_loop(ps, qs), want)
        assert lsa.g2_table_cache_stats()["resident"] == 0
        # tables passed as blobs are cached by content too
        lsa.g2_table_cache(16)
        tabs = lsa.g2_precompute(qs)
        a = lsa.miller_loop_precomp(ps, tabs)
        b = lsa.miller_loop_precomp(ps, tabs.copy())                   # other addresses, same bytes
        assert np.array_equal(a, want) and np.array_equal(b, want)
        assert lsa.g2_table_cache_stats()["resident"] == 6
    finally:
        lsa.g2_table_cache(4096)


def test_large_batch_bypasses_the_cache_and_shares_accumulators(lsa):
    """2000 terms (> 1024: tables in scratch, no fingerprints) in 300 products, four pairs per accumulator."""
    rng = random.Random(5)
    n = 2000
    ps = o.arith_bases("g1", 1, 1, n)
    qd = _g2_points(5, 3)
    idx = np.array([rng.randrange(5) for _ in range(n)])
    qs = qd[idx]
    cuts = sorted(rng.sample(range(1, n), 299))
    off = np.array([0] + cuts + [n], dtype=np.uint64)
    fd = {}
    f = np.zeros((n, 48), dtype=np.uint64)
    for i in range(n):                                                  # the oracle needs ~1 ms per loop: 5 tables
        f[i] = o.miller_loop_precomp(ps[i], fd.setdefault(int(idx[i]), o.precompute_g2(qd[idx[i]])))
    want = np.array([o.fq12_product(f[int(off[j]):int(off[j + 1])]) for j in range(300)])
    lsa.pairing_set_chunk(4)
    try:
        got = lsa.pairing_product_segments(ps, qs, off, final_exp=False)
    finally:
        lsa.pairing_set_chunk(0)
    assert np.array_equal(got, want)
    assert np.array_equal(lsa.pairing_product_segments(ps, qs, off, final_exp=False), want)


def test_a_failed_call_leaves_no_half_built_table_behind(lsa):
    """A call that names blobs 0 .. k-1 and then a term with neither a table nor a point fails with LSA_ERR_INVALID --
    and must not leave the keys of the blobs it had already looked at pointing at unwritten tables: the next valid call
    with the same blobs gives the oracle's Miller values, and the cache holds no more tables than that call built.
    Then: a prefetch that promises more tables than the cache can pin is still served correctly."""
    import legosnark_amd
    ps = o.arith_bases("g1", 31, 5, 5)
    qs = _g2_points(5, 404)
    want = o.miller_loop_batch(ps, qs)
    try:
        lsa.g2_table_cache(64)
        tabs = lsa.g2_precompute(qs)
        off = np.arange(6, dtype=np.uint64)
        with pytest.raises(legosnark_amd.LsaError):
            lsa.pairing_terms(ps, off, g2=None, tables=tabs, index=[0, 1, 2, -1, 4], final_exp=False)      # term 3: nothing
        assert lsa.g2_table_cache_stats()["resident"] == 0                    # validated before the cache was touched
        got = lsa.pairing_terms(ps, off, tables=tabs, index=[0, 1, 2, 3, 4], final_exp=False)
        assert np.array_equal(got, want)
        assert lsa.g2_table_cache_stats()["resident"] == 5
        # points: a second, valid call over the same Q after an invalid one
        lsa.g2_table_cache(64)
        with pytest.raises(legosnark_amd.LsaError):
            lsa.pairing_terms(ps, np.array([0, 3, 2, 5], dtype=np.uint64), g2=qs, final_exp=False)       # offsets decrease
        assert np.array_equal(lsa.miller_loop(ps, qs), want)
        # a tiny cache and prefetches one by one: promised slots are pinned until built, never handed to another point
        lsa.g2_table_cache(8)
        more = o.arith_bases("g2", 777, 3, 12)
        pm = o.arith_bases("g1", 8, 9, 12)
        for q in more:
            lsa.g2_tables_prefetch(q.reshape(1, 24))
        assert np.array_equal(lsa.miller_loop(pm, more), o.miller_loop_batch(pm, more))
    finally:
        lsa.g2_table_cache(4096)


@pytest.mark.parametrize("switch", ["LSA_MILLER_ROWS=0", "LSA_MILLER_SPLIT=1", "LSA_MILLER_TAB=4"])
def test_the_older_kernels_behind_their_switches_give_the_same_bytes(lsa, switch):
    """k_miller_wtab (two-phase rounds), the row-engine Miller loop on ONE workgroup instead of eight, k_miller_tab (four
    accumulators per wavefront: what table-resident batches of more than 2048 accumulators take) stay in the library behind A/B
    switches that are read once per process: a child process under the switch must return the bytes this process does
    (a two-term check with a conjugated term, and the Miller values alone)."""
    import subprocess
    import sys
    ps = o.arith_bases("g1", 321, 7, 2)
    qs = _g2_points(2, 55)
    tabs = lsa.g2_precompute(qs)
    off = np.array([0, 2], dtype=np.uint64)
    flags = np.array([0, 1], dtype=np.uint8)
    here = lsa.pairing_terms(ps, off, tables=tabs, index=[0, 1], flags=flags, final_exp=True)
    mill = lsa.miller_loop_precomp(ps, tabs, [0, 1])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the child rebuilds the same inputs through this module's helpers
    code = (
        "import sys, os, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import legosnark_amd as lsa, oracle_lib as o\n"
        "import test_pairing_precomp_gpu as t\n"
        "lsa.init(0)\n"
        "ps = o.arith_bases('g1', 321, 7, 2); qs = t._g2_points(2, 55); tabs = lsa.g2_precompute(qs)\n"
        "a = lsa.pairing_terms(ps, np.array([0, 2], dtype=np.uint64), tables=tabs, index=[0, 1], flags=np.array([0, 1], dtype=np.uint8), final_exp=True)\n"
        "b = lsa.miller_loop_precomp(ps, tabs, [0, 1])\n"
        "sys.stdout.write(a.tobytes().hex() + ' ' + b.tobytes().hex())\n"
    ) % (root, os.path.join(root, "tests"))
    env = dict(os.environ)
    k, v = switch.split("=")
    env[k] = v
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    a_hex, b_hex = out.stdout.strip().split()[-2:]
    assert a_hex == here.tobytes().hex()
    assert b_hex == mill.tobytes().hex()

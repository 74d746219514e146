"""The CRS cache behind lsa_g1_msm / lsa_g2_msm (the host-vector path multiExpMA takes,
/root/reference/src/utils/globl.h:74-77): hits on an unchanged vector, never a stale result."""
import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu


def canon(group, pt):
    return o.g1_canonical_affine(pt) if group == "g1" else o.g2_canonical_affine(pt)


@pytest.fixture()
def fresh_cache(lsa):
    lsa.crs_cache_configure(lsa.CRS_CACHE_FULL, 8 << 30)
    lsa.crs_cache_clear()
    lsa.crs_cache_table_after(1)          # these tests want the copies at the first re-use (the default waits for the 23rd hit)
    yield lsa
    lsa.crs_cache_configure(lsa.CRS_CACHE_FULL, 8 << 30)
    lsa.crs_cache_clear()
    lsa.crs_cache_table_after(23)
    lsa.set_table_threshold(0)


def delta(lsa, before):
    now = lsa.crs_cache_stats()
    return now["hits"] - before["hits"], now["misses"] - before["misses"]


@pytest.mark.parametrize("group,n", [("g1", 5000), ("g2", 1500)])
def test_hit_on_same_vector_miss_on_any_mutation(fresh_cache, group, n):
    lsa = fresh_cache
    bases = np.ascontiguousarray(o.arith_bases(group, 31337, 17, n))
    sc, _ = o.random_scalars(n, seed=5)
    want = canon(group, o.multi_exp(group, bases, sc, mode="mixed"))
    s0 = lsa.crs_cache_stats()
    assert canon(group, lsa.msm(group, bases, sc)) == want
    assert delta(lsa, s0) == (0, 1) and lsa.msm_host_stats()["cache_hit"] == 0
    assert canon(group, lsa.msm(group, bases, sc)) == want
    assert delta(lsa, s0) == (1, 1) and lsa.msm_host_stats()["cache_hit"] == 1
    # other scalars, same bases: hit
    sc2, _ = o.random_scalars(n, seed=6)
    assert canon(group, lsa.msm(group, bases, sc2)) == canon(group, o.multi_exp(group, bases, sc2, mode="mixed"))
    assert delta(lsa, s0) == (2, 1)
    # mutate ONE point in place, at a position no sampling scheme would look at: miss + right answer
    i = 2917 % n
    bases[i] = o.arith_bases(group, 999, 1, 1)[0]
    want2 = canon(group, o.multi_exp(group, bases, sc, mode="mixed"))
    assert want2 != want
    assert canon(group, lsa.msm(group, bases, sc)) == want2
    assert delta(lsa, s0) == (2, 2)
    # flip a single bit of a single limb: still a miss (the point is off the curve now, so only
    # the cache decision is checked; the bit is restored afterwards)
    bases[i + 1, 0] ^= np.uint64(1)
    s1 = lsa.crs_cache_stats()
    lsa.msm(group, bases, sc)
    assert delta(lsa, s1) == (0, 1)
    bases[i + 1, 0] ^= np.uint64(1)
    assert canon(group, lsa.msm(group, bases, sc)) == want2
    assert delta(lsa, s1) == (0, 2)
    # equal content at another address: the cache is content-addressed (full fingerprints), the copy hits the entry
    other = bases.copy()
    s1 = lsa.crs_cache_stats()
    assert canon(group, lsa.msm(group, other, sc)) == canon(group, o.multi_exp(group, bases, sc, mode="mixed"))
    assert delta(lsa, s1) == (1, 0)


def test_prefix_requests_cppoly_ladder_shape(fresh_cache):
    """CPPoly::prove passes g1s with shorter and shorter scalar vectors (poly.h:77-86):
    multiExpMA truncates to the prefix (globl.h:66).  Prefixes that are multiples of the
    fingerprint unit hit the long entry; others are verified as their own entry."""
    lsa = fresh_cache
    n = 8192
    bases = np.ascontiguousarray(o.arith_bases("g1", 4242, 7, n))
    sc, _ = o.random_scalars(n, seed=11)
    lsa.msm("g1", bases, sc)
    s0 = lsa.crs_cache_stats()
    for m in (4096, 2048, 1024):
        got = lsa.msm("g1", bases, sc[:m])
        assert canon("g1", got) == canon("g1", o.multi_exp("g1", bases[:m], sc[:m], mode="mixed")), m
    assert delta(lsa, s0) == (3, 0)
    m = 3001                                              # not a multiple of 64: separate entry, still right
    got = lsa.msm("g1", bases, sc[:m])
    assert canon("g1", got) == canon("g1", o.multi_exp("g1", bases[:m], sc[:m], mode="mixed"))
    assert delta(lsa, s0) == (3, 1)
    # a mutation beyond the prefix does not disturb the prefix; inside it does
    bases[5000] = bases[1]
    s1 = lsa.crs_cache_stats()
    lsa.msm("g1", bases, sc[:4096])
    assert delta(lsa, s1) == (1, 0)
    bases[100] = bases[1]
    got = lsa.msm("g1", bases, sc[:4096])
    assert canon("g1", got) == canon("g1", o.multi_exp("g1", bases[:4096], sc[:4096], mode="mixed"))
    assert delta(lsa, s1) == (1, 1)
    # small requests: never inserted, but looked up as prefixes of an entry -- whole 64-point units, or fewer than 64
    # points against the per-point fingerprints of the entry's first unit (the tail of CPPoly::prove's ladder);
    # 100 points = one unit and a partial one: not comparable, uploaded as before
    s1 = lsa.crs_cache_stats()
    got = lsa.msm("g1", bases, sc[:100])
    assert canon("g1", got) == canon("g1", o.multi_exp("g1", bases[:100], sc[:100], mode="mixed"))
    assert delta(lsa, s1) == (0, 0) and lsa.msm_host_stats()["cache_hit"] == 0
    for m in (512, 64, 37, 2, 1):
        s1 = lsa.crs_cache_stats()
        got = lsa.msm("g1", bases, sc[:m])
        assert canon("g1", got) == canon("g1", o.multi_exp("g1", bases[:m], sc[:m], mode="mixed")), m
        assert delta(lsa, s1) == (1, 0) and lsa.msm_host_stats()["cache_hit"] == 1, m
    # the same prefixes at another address (a copy): content-addressed, still hits
    head = bases[:512].copy()
    s1 = lsa.crs_cache_stats()
    assert canon("g1", lsa.msm("g1", head, sc[:37])) == canon("g1", o.multi_exp("g1", head[:37], sc[:37], mode="mixed"))
    assert delta(lsa, s1) == (1, 0)
    # one changed point inside the requested prefix: no hit, right answer; beyond it: hit
    head[20] = head[3]
    s1 = lsa.crs_cache_stats()
    assert canon("g1", lsa.msm("g1", head, sc[:37])) == canon("g1", o.multi_exp("g1", head[:37], sc[:37], mode="mixed"))
    assert lsa.msm_host_stats()["cache_hit"] == 0
    assert canon("g1", lsa.msm("g1", head, sc[:512])) == canon("g1", o.multi_exp("g1", head[:512], sc[:512], mode="mixed"))
    assert lsa.msm_host_stats()["cache_hit"] == 0
    assert canon("g1", lsa.msm("g1", head, sc[:16])) == canon("g1", o.multi_exp("g1", head[:16], sc[:16], mode="mixed"))
    assert lsa.msm_host_stats()["cache_hit"] == 1
    assert delta(lsa, s1) == (1, 0)
    # ... and on the entry's pre-shifted copies once it has them
    lsa.crs_cache_wait_tables()
    lsa.msm("g1", bases, sc)                      # (switches the entry if the build has just finished)
    if lsa.msm_host_stats()["table"] == 1:
        for m in (256, 5):
            got = lsa.msm("g1", bases, sc[:m])
            assert canon("g1", got) == canon("g1", o.multi_exp("g1", bases[:m], sc[:m], mode="mixed")), m
            st = lsa.msm_host_stats()
            assert st["cache_hit"] == 1 and st["table"] == 1, m


def test_table_built_on_first_reuse_and_lru_eviction(fresh_cache):
    lsa = fresh_cache
    lsa.set_table_threshold(2048)
    n = 4096
    sc, _ = o.random_scalars(n, seed=3)
    vecs = [np.ascontiguousarray(o.arith_bases("g1", 100 + k, 3, n)) for k in range(3)]
    wants = [canon("g1", o.multi_exp("g1", v, sc, mode="mixed")) for v in vecs]
    assert canon("g1", lsa.msm("g1", vecs[0], sc)) == wants[0]
    assert lsa.msm_host_stats()["table"] == 0             # cold: plain layout
    assert canon("g1", lsa.msm("g1", vecs[0], sc)) == wants[0]
    st = lsa.msm_host_stats()
    # first re-use: the pre-shifted copies are STARTED in the background; this call ran on the plain layout
    assert st["cache_hit"] == 1 and st["table"] == 0 and st["table_building"] == 1
    lsa.crs_cache_wait_tables()
    assert canon("g1", lsa.msm("g1", vecs[0], sc)) == wants[0]          # the same point over the copies
    st = lsa.msm_host_stats()
    assert st["cache_hit"] == 1 and st["table"] == 1 and st["table_building"] == 0
    one = lsa.crs_cache_stats()["resident_bytes"]
    assert one % (n * 64) == 0 and one // (n * 64) >= 12      # one 64-byte copy of every point per window (+ the plain one)
    # short prefixes of the vector (below the cache's own minimum) run over the copies too
    for m in (512, 64, 5, 1):
        assert canon("g1", lsa.msm("g1", vecs[0], sc[:m])) == canon("g1", o.multi_exp("g1", vecs[0][:m], sc[:m], mode="mixed")), m
        st = lsa.msm_host_stats()
        assert st["cache_hit"] == 1 and st["table"] == 1, m
    # without waiting: calls issued while the build runs switch over on their own, every result the same point
    lsa.crs_cache_clear()
    seen_table = False
    for i in range(600):          # (the builder yields to lsa_stream(): how many calls it takes depends on how fast they are)
        assert canon("g1", lsa.msm("g1", vecs[0], sc)) == wants[0], i
        seen_table = seen_table or lsa.msm_host_stats()["table"] == 1
        if seen_table and i >= 40:
            break
    assert seen_table
    # a policy of three hits: the build starts at the third re-use
    lsa.crs_cache_clear()
    lsa.crs_cache_table_after(3)
    try:
        for i in range(4):
            assert canon("g1", lsa.msm("g1", vecs[0], sc)) == wants[0]
            assert lsa.msm_host_stats()["table_building"] == (1 if i == 3 else 0), i
    finally:
        lsa.crs_cache_table_after(1)
    lsa.crs_cache_wait_tables()
    # budget for two such entries: the least recently used one goes
    lsa.crs_cache_configure(lsa.CRS_CACHE_FULL, 2 * one + 1024)
    for k in (1, 2):
        for _ in range(2):
            assert canon("g1", lsa.msm("g1", vecs[k], sc)) == wants[k]
        lsa.crs_cache_wait_tables()
        assert canon("g1", lsa.msm("g1", vecs[k], sc)) == wants[k]
    st = lsa.crs_cache_stats()
    assert st["entries"] == 2 and st["resident_bytes"] <= 2 * one + 1024
    s0 = lsa.crs_cache_stats()
    assert canon("g1", lsa.msm("g1", vecs[0], sc)) == wants[0]          # evicted: miss, right answer
    assert delta(lsa, s0) == (0, 1)


def test_a_copy_of_a_cached_vector_at_another_address_is_a_hit(fresh_cache):
    """The reference passes multiExpMA copies of its key vectors (CommScheme::getBases1() returns g1s by value,
    src/prototools/commit.h:145-147): with full fingerprints the cache is content-addressed, so the same bytes at a
    new address hit (whole vector and 64-aligned prefix); one changed point is a miss."""
    lsa = fresh_cache
    n = 4096
    bases = np.ascontiguousarray(o.arith_bases("g1", 5, 9, n))
    sc, _ = o.random_scalars(n, seed=4)
    want = canon("g1", o.multi_exp("g1", bases, sc, mode="mixed"))
    s0 = lsa.crs_cache_stats()
    assert canon("g1", lsa.msm("g1", bases, sc)) == want
    copy1 = bases.copy()
    assert copy1.ctypes.data != bases.ctypes.data
    assert canon("g1", lsa.msm("g1", copy1, sc)) == want
    assert delta(lsa, s0) == (1, 1) and lsa.crs_cache_stats()["entries"] == 1
    copy2 = bases[:1024].copy()                                    # a prefix of 16 units at yet another address
    assert canon("g1", lsa.msm("g1", copy2, sc[:1024])) == canon("g1", o.multi_exp("g1", copy2, sc[:1024], mode="mixed"))
    assert delta(lsa, s0) == (2, 1)
    copy3 = bases.copy()
    copy3[7] = copy3[8]
    assert canon("g1", lsa.msm("g1", copy3, sc)) == canon("g1", o.multi_exp("g1", copy3, sc, mode="mixed"))
    assert delta(lsa, s0) == (2, 2)


def test_sampled_mode_and_off(fresh_cache):
    lsa = fresh_cache
    n = 3000
    bases = np.ascontiguousarray(o.arith_bases("g1", 77, 5, n))
    sc, _ = o.random_scalars(n, seed=9)
    want = canon("g1", o.multi_exp("g1", bases, sc, mode="mixed"))
    lsa.crs_cache_configure(lsa.CRS_CACHE_SAMPLED)
    s0 = lsa.crs_cache_stats()
    assert canon("g1", lsa.msm("g1", bases, sc)) == want
    assert canon("g1", lsa.msm("g1", bases, sc)) == want
    assert delta(lsa, s0) == (1, 1)
    bases[n - 1] = bases[0]                               # the last point is always sampled
    assert canon("g1", lsa.msm("g1", bases, sc)) == canon("g1", o.multi_exp("g1", bases, sc, mode="mixed"))
    assert delta(lsa, s0) == (1, 2)
    lsa.crs_cache_configure(lsa.CRS_CACHE_OFF)
    s0 = lsa.crs_cache_stats()
    for _ in range(2):
        assert canon("g1", lsa.msm("g1", bases, sc)) == canon("g1", o.multi_exp("g1", bases, sc, mode="mixed"))
    assert delta(lsa, s0) == (0, 0) and lsa.crs_cache_stats()["entries"] == 0

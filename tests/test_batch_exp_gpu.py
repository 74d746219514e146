"""GPU parity tests for fixed-base batch_exp and the point-sum kernel (C-ABI)."""
import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu
R = o.R


@pytest.mark.parametrize("group", ["g1", "g2"])
@pytest.mark.parametrize("n", [1, 9, 300, 70000])
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

"""GPU parity tests for the CPlink key-generation operators:
variable-base batch scalar multiplication and mtxmultiexp on a column-sparse G1 matrix,
through the C-ABI, against the oracle's restatement of sparsemexpG / libff scalar * point.
Group elements are compared after affine normalisation."""
import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu
R = o.R


def canon_all(pts):
    return [o.g1_canonical_affine(p) for p in np.asarray(pts, dtype=np.uint64).reshape(-1, 12)]


def test_scalar_mul_batch_vs_oracle(lsa):
    n = 300
    pts = o.arith_bases("g1", 424242, 1717, n)                # un-normalised Jacobian
    sc, vals = o.random_scalars(n, seed=4242)
    # edge scalars: 0, 1, 2, r-1, r-2, 2^127 boundaries, small ones (one GLV half zero)
    special = [0, 1, 2, R - 1, R - 2, (1 << 127) - 1, 1 << 127, (1 << 128) + 5, 15, 16, 0x88888888, 8, 9]
    for i, v in enumerate(special):
        sc[i] = o.fr_mont(v)
    pts[40] = 0                                                # infinity (Z = 0)
    pts[41] = o.generator("g1")
    got = lsa.scalar_mul_batch(pts, sc)
    want = o.g1_mul_batch(pts, sc)
    assert canon_all(got) == canon_all(want)


def test_scalar_mul_batch_device_buffers_and_grid_stride(lsa):
    """More items than resident lanes (512 blocks x 256) so that the persistent loop wraps;
    checked by the known-discrete-log identity: pts[i] = (a + i b) G, so
    scalars[i] * pts[i] = (s_i (a + i b) mod r) * G = batch_exp(G, s_i (a + i b))."""
    import torch
    n = 512 * 256 + 1000
    a, b = 987654321, 1234577
    g = o.generator("g1")
    coef = [(a + i * b) % R for i in range(n)]
    d_coef = torch.from_numpy(o.fr_mont_array(coef).view(np.int64)).to("cuda:0")
    d_pts = lsa.batch_exp("g1", g, d_coef)
    sc, vals = o.random_scalars(n, seed=7)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    d_out = lsa.scalar_mul_batch(d_pts, d_sc)
    prod = [(int(vals[i]) * coef[i]) % R for i in range(n)]
    d_prod = torch.from_numpy(o.fr_mont_array(prod).view(np.int64)).to("cuda:0")
    d_want = lsa.batch_exp("g1", g, d_prod)
    lsa.synchronize()
    got = lsa.normalize("g1", d_out.cpu().numpy().view(np.uint64))
    want = lsa.normalize("g1", d_want.cpu().numpy().view(np.uint64))
    assert np.array_equal(got, want)


def _cplink_matrix(N, seed):
    """The CPlink relation matrix of /root/reference/src/examples/cplink.cc:21-41 in CSC form:
    2 rows, column 0 = (h), column 1 = (F[0]) in row 1, columns 2..N+1 = (bases1[i]; F[i+1]),
    the remaining N columns empty."""
    h = o.arith_bases("g1", seed, 3, 1)[0]
    bases1 = o.arith_bases("g1", seed + 1, 5, N)
    F = o.arith_bases("g1", seed + 2, 7, N + 1)
    vals, rows, col_ptr = [h, F[0]], [0, 1], [0, 1, 2]
    for i in range(N):
        vals += [bases1[i], F[i + 1]]
        rows += [0, 1]
        col_ptr.append(len(vals))
    col_ptr += [len(vals)] * N
    return np.array(vals, dtype=np.uint64), np.array(rows, dtype=np.uint32), np.array(col_ptr, dtype=np.uint64)


def test_sparse_matrix_msm_cplink_shape(lsa):
    N = 64
    vals, rows, col_ptr = _cplink_matrix(N, 99)
    k, _ = o.random_scalars(2, seed=11)
    got = lsa.sparse_matrix_msm(vals, rows, col_ptr, k)
    want = o.mtxmultiexp(vals, rows, col_ptr, k)
    assert len(got) == 2 * N + 2
    assert canon_all(got) == canon_all(want)
    assert all(c is None for c in canon_all(got)[N + 2:])      # empty columns -> infinity


def test_sparse_matrix_msm_general_columns(lsa):
    """Ragged columns (0..9 non-zeros), generator-valued and infinity-valued entries (the
    zero / one branches of sparsemexpG), repeated rows, cancelling entries."""
    rng = np.random.default_rng(5)
    nrows, ncols = 7, 40
    pool = o.arith_bases("g1", 31337, 11, 64)
    gen = o.generator("g1")
    vals, rows, col_ptr = [], [], [0]
    for j in range(ncols):
        for _ in range(j % 10):
            t = rng.integers(0, 10)
            if t == 0:
                vals.append(np.zeros(12, dtype=np.uint64))       # zero entry
            elif t == 1:
                vals.append(gen)
            else:
                vals.append(pool[rng.integers(0, 64)])
            rows.append(int(rng.integers(0, nrows)))
        col_ptr.append(len(vals))
    # column with P and -P under the same exponent -> infinity
    neg = pool[3].copy()
    y = o.limbs_to_int(neg[4:8])
    neg[4:8] = o.int_to_limbs((o.P - y) % o.P)
    vals += [pool[3], neg]
    rows += [2, 2]
    col_ptr.append(len(vals))
    k, _ = o.random_scalars(nrows, seed=12)
    k[5] = o.fr_mont(0)
    k[6] = o.fr_mont(1)
    vals = np.array(vals, dtype=np.uint64)
    got = lsa.sparse_matrix_msm(vals, rows, col_ptr, k)
    want = o.mtxmultiexp(vals, rows, col_ptr, k)
    assert canon_all(got) == canon_all(want)
    assert o.g1_canonical_affine(got[-1]) == o.g1_canonical_affine(np.zeros(12, dtype=np.uint64))


def test_sparse_matrix_msm_rejects_bad_input(lsa):
    import legosnark_amd
    vals = o.arith_bases("g1", 1, 1, 2)
    k, _ = o.random_scalars(2, seed=1)
    with pytest.raises(legosnark_amd.LsaError):
        lsa.sparse_matrix_msm(vals, [0, 2], [0, 2], k)           # row index out of range
    with pytest.raises(legosnark_amd.LsaError):
        lsa.sparse_matrix_msm(vals, [0, 1], [1, 2], k)           # col_ptr[0] != 0
    assert len(lsa.sparse_matrix_msm(vals[:0], [], [0], k)) == 0

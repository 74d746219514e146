#!/usr/bin/env python3
"""Times the CPlink key-generation operators on one MI355X: variable-base batch scalar
multiplication (device buffers) and mtxmultiexp on the CPlink relation matrix (host CSC
arrays), next to the oracle's sparsemexpG restatement on a bounded sample of columns."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import legosnark_amd as lsa  # noqa: E402
import oracle_lib as o  # noqa: E402


def main():
    lsa.init(0)
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << logn
    g = o.generator("g1")
    sc, _ = o.random_scalars(n, seed=1)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    d_pts = lsa.batch_exp("g1", g, d_sc)
    sc2, _ = o.random_scalars(n, seed=2)
    d_sc2 = torch.from_numpy(sc2.view(np.int64)).to("cuda:0")
    d_out = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    lsa.scalar_mul_batch(d_pts, d_sc2, out=d_out)
    lsa.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        lsa.scalar_mul_batch(d_pts, d_sc2, out=d_out)
    lsa.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"op": "lsa_g1_scalar_mul_batch (device buffers)", "n": n, "ms": dt * 1e3, "scalar_muls_per_s": n / dt}))

    # CPlink matrix, N = n/2 so that nnz ~ n
    N = n // 2
    pts = d_pts.cpu().numpy().view(np.uint64).reshape(-1, 12)
    vals = np.empty((2 * N + 2, 12), dtype=np.uint64)
    vals[:] = pts[: 2 * N + 2] if len(pts) >= 2 * N + 2 else np.resize(pts, (2 * N + 2, 12))
    rows = np.tile(np.array([0, 1], dtype=np.uint32), N + 1)
    col_ptr = np.concatenate([np.array([0, 1, 2], dtype=np.uint64), 2 + 2 * np.arange(1, N + 1, dtype=np.uint64),
                              np.full(N, 2 * N + 2, dtype=np.uint64)])
    k, _ = o.random_scalars(2, seed=3)
    t0 = time.perf_counter()
    out = lsa.sparse_matrix_msm(vals, rows, col_ptr, k)
    dt = time.perf_counter() - t0
    print(json.dumps({"op": "lsa_g1_sparse_matrix_msm (CPlink matrix, host buffers)", "N": N, "nnz": int(col_ptr[-1]), "ms": dt * 1e3}))
    # oracle on a bounded sample of columns
    m = 256
    t0 = time.perf_counter()
    want = o.mtxmultiexp(vals[: 2 * m + 2], rows[: 2 * m + 2], col_ptr[: m + 3], k)
    dtc = time.perf_counter() - t0
    ok = all(o.g1_canonical_affine(out[j]) == o.g1_canonical_affine(want[j]) for j in range(m + 2))
    print(json.dumps({"op": "oracle sparsemexpG restatement, 1 core", "columns": m + 2, "ms_per_column": dtc * 1e3 / (m + 2),
                      "extrapolated_s_for_N": dtc / (m + 2) * (N + 2), "matches_gpu": ok}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Times the Fr-vector kernels on device-resident vectors: the CPpoly witness recursion and
evalMLE at d = 20 and d = 24, with the algorithmic HBM bytes of the first (largest) round, and
the oracle's loops on one host core beside them."""
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
    for d in (20, 24):
        N = 1 << d
        g = torch.Generator(device="cuda:0").manual_seed(d)
        d_v = torch.randint(0, 1 << 62, (N, 4), dtype=torch.int64, device="cuda:0", generator=g)
        d_v[:, 3] &= (1 << 60) - 1            # any value < r is a valid Montgomery residue
        r, _ = o.random_scalars(d, seed=9)
        d_r = torch.from_numpy(r.view(np.int64)).to("cuda:0")
        d_w = torch.empty_like(d_v)
        d_out = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        for name, fn in (("cppoly_witness", lambda: lsa.cppoly_witness(d_v, d_r, out=d_w)),
                         ("eval_mle", lambda: lsa.eval_mle_device(d_v, d_r, d_out))):
            fn()
            lsa.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                fn()
            lsa.synchronize()
            dt = (time.perf_counter() - t0) / reps
            # all rounds: pairs read 32*N*(1 + 1/2 + ...) = 64 N, write 64 N; halves read 64 N write 32 N
            bytes_total = (128 if name == "cppoly_witness" else 96) * N
            print(json.dumps({"op": name, "d": d, "ms": dt * 1e3, "algorithmic_GBps": bytes_total / dt / 1e9,
                              "frac_of_8TBps": bytes_total / dt / 8e12}))
        if d == 20:
            v = d_v.cpu().numpy().view(np.uint64)
            t0 = time.perf_counter()
            w = o.fr_cppoly_witness(v, r)
            t1 = time.perf_counter()
            e = o.fr_eval_mle(v, r)
            t2 = time.perf_counter()
            ok = np.array_equal(w, d_w.cpu().numpy().view(np.uint64)) and np.array_equal(e, d_out.cpu().numpy().view(np.uint64))
            print(json.dumps({"op": "oracle loops, 1 core", "d": d, "cppoly_witness_ms": (t1 - t0) * 1e3,
                              "eval_mle_ms": (t2 - t1) * 1e3, "matches_gpu": bool(ok)}))


if __name__ == "__main__":
    main()

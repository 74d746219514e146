#!/usr/bin/env python3
"""tools/h2d_probe.py -- on the GPU box: what the host-vector MSM entry point pays to bring a witness to the device,
for a vector the caller has just filled (a fresh allocation every call, like the std::vector<Fr> temporaries of
/root/reference/src/gadgets/poly.h:77-86) and for one buffer passed again and again; run once per LSA_H2D mode."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import legosnark_amd as lsa  # noqa: E402
import oracle_lib as o  # noqa: E402


def main():
    lsa.init(0)
    out = {"mode": os.environ.get("LSA_H2D", "auto"), "threads": os.environ.get("LSA_H2D_THREADS"), "chunk_kb": os.environ.get("LSA_H2D_CHUNK_KB")}
    for log2n in [int(x) for x in os.environ.get("PROBE_LOG2N", "16,18,20").split(",")]:
        n = 1 << log2n
        bases = np.ascontiguousarray(o.arith_bases("g1", 7, 3, n))
        sc, _ = o.random_scalars(n, seed=1)
        lsa.crs_cache_table_after(1)
        lsa.msm("g1", bases, sc)
        lsa.msm("g1", bases, sc)
        lsa.crs_cache_wait_tables()
        lsa.msm("g1", bases, sc)
        fresh, again, total_fresh, total_again = [], [], [], []
        bufs = []
        for i in range(12):
            buf = np.empty_like(sc)                  # new allocations at distinct addresses, written by the CPU just now
            buf[...] = sc
            bufs.append(buf)
        for buf in bufs:
            t0 = time.perf_counter()
            lsa.msm("g1", bases, buf)
            total_fresh.append((time.perf_counter() - t0) * 1e3)
            fresh.append(lsa.msm_host_stats()["h2d_scalars_ms"])
        del bufs
        for i in range(12):
            t0 = time.perf_counter()
            lsa.msm("g1", bases, sc)
            total_again.append((time.perf_counter() - t0) * 1e3)
            again.append(lsa.msm_host_stats()["h2d_scalars_ms"])
        mib = n * 32 / 2**20
        out["2^%d" % log2n] = {"MiB": mib, "fresh_h2d_ms": round(float(np.median(fresh)), 3), "fresh_GBps": round(mib / 1024 / (np.median(fresh) * 1e-3), 1),
                               "fresh_call_ms": round(float(np.median(total_fresh)), 3),
                               "same_buffer_h2d_ms": round(float(np.median(again)), 3), "same_buffer_call_ms": round(float(np.median(total_again)), 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

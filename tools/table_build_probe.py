"""tools/table_build_probe.py -- GPU box: what the pre-shifted copies of a SMALL vector cost to build (synchronously, in
bases_create) and what an MSM over them gains, per vector size."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import legosnark_amd as lsa
import oracle_lib as o
lsa.init(0)
dev = torch.device("cuda:0")
out = {}
for log2n in (12, 14, 16, 18):
    n = 1 << log2n
    bases = np.ascontiguousarray(o.arith_bases("g1", 11, 3, n))
    sc, _ = o.random_scalars(n, seed=2)
    d_s = torch.from_numpy(sc.view(np.int64)).to(dev)
    res = {}
    for tag, thr in (("plain", 1 << 30), ("copies", 1024)):
        lsa.set_table_threshold(thr)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            B = lsa.Bases("g1", bases)
            lsa.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        res[tag + "_create_ms"] = round(sorted(ts)[1], 3)
        B.msm(d_s)
        tm = []
        for _ in range(7):
            t0 = time.perf_counter(); B.msm(d_s); tm.append((time.perf_counter() - t0) * 1e3)
        res[tag + "_msm_ms"] = round(sorted(tm)[3], 3)
        tm = []
        for _ in range(7):
            t0 = time.perf_counter(); B.msm(d_s, n=max(1, n // 64)); tm.append((time.perf_counter() - t0) * 1e3)
        res[tag + "_msm_n/64_ms"] = round(sorted(tm)[3], 3)
    out["2^%d" % log2n] = res
lsa.set_table_threshold(0)
print(json.dumps(out))

#!/usr/bin/env python3
"""tools/copy_probe.py -- GPU box: host-buffer entry points whose cost is mostly the copies (NTT, batch_exp, normalise,
folds) at small and large sizes; run once per LSA_H2D mode."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import legosnark_amd as lsa
import oracle_lib as o

def med(f, reps=9):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 3)

lsa.init(0)
out = {"mode": os.environ.get("LSA_H2D", "auto")}
for log2n in (12, 16, 20):
    n = 1 << log2n
    sc, _ = o.random_scalars(n, seed=3)
    omega = o.fr_mont(o.fr_root_of_unity(log2n))
    row = {}
    if omega is not None:
        row["fr_ntt_ms"] = med(lambda: lsa.fr_ntt(sc.copy(), omega))
    g = o.arith_bases("g1", 3, 1, 1)[0]
    row["batch_exp_g1_ms"] = med(lambda: lsa.batch_exp("g1", g, sc), reps=5)
    pts = lsa.batch_exp("g1", g, sc[: min(n, 1 << 16)])
    row["normalize_g1_ms(<=2^16)"] = med(lambda: lsa.normalize("g1", pts), reps=5)
    r, _ = o.random_scalars(1, seed=4)
    row["fr_fold_ms"] = med(lambda: lsa.fr_fold(sc, r[0]))
    out["2^%d" % log2n] = row
print(json.dumps(out))

#!/usr/bin/env python3
"""Randomised stress of the wide MSM path on one resident table (run on the GPU box): random sizes
between 2^15 and the table size, scalar shapes that skew the bucket populations (small values, values
just below r, few distinct values, powers of two, runs of equal scalars), every result checked by the
known-discrete-log identity with bases (a + i b) G."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import legosnark_amd as lsa
import oracle_lib as o

R = o.R
lsa.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n_table = (1 << 20) + 777
a, b = 0xA11CE << 200 | 0xB0B, 0xC0FFEE << 100 | 0x5
bases = o.arith_bases("g1", a, b, n_table)
lsa.set_table_threshold(0)
B = lsa.Bases("g1", bases)
assert B.has_table()
G = o.generator("g1")


def scalars(kind, n):
    if kind == "uniform":
        return [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n)]
    if kind == "small":
        return [int(x) for x in rng.integers(0, 1 << int(rng.integers(1, 40)), size=n)]
    if kind == "near_r":
        return [R - 1 - int(x) for x in rng.integers(0, 1 << 30, size=n)]
    if kind == "few":
        vals = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(int(rng.integers(1, 6)))]
        return [vals[int(x)] for x in rng.integers(0, len(vals), size=n)]
    if kind == "pow2":
        return [1 << int(x) for x in rng.integers(0, 253, size=n)]
    if kind == "bits":
        return [int(x) for x in rng.integers(0, 2, size=n)]
    if kind == "bytes":
        return [int(x) for x in rng.integers(0, 256, size=n)]
    if kind == "runs":
        out, v = [], 0
        while len(out) < n:
            v = int.from_bytes(rng.bytes(32), "little") % R
            out += [v] * int(rng.integers(1, 5000))
        return out[:n]
    raise ValueError(kind)


bad = 0
kinds = ["uniform", "small", "near_r", "few", "pow2", "runs", "bits", "bytes"]
for c in range(cases):
    kind = kinds[c % len(kinds)]
    n = int(rng.integers(1 << 15, n_table + 1)) if c % 5 else n_table
    sc = scalars(kind, n)
    d_s = torch.from_numpy(o.fr_mont_array(sc).view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = B.msm(d_s, n=n)
    dt = time.perf_counter() - t0
    k = sum(s * (a + i * b) for i, s in enumerate(sc)) % R
    ok = o.g1_canonical_affine(got) == o.g1_canonical_affine(o.g1_mul(G, o.fr_mont(k)))
    bad += not ok
    print("%-8s n=%8d  %.2f ms  %s" % (kind, n, dt * 1e3, "ok" if ok else "MISMATCH"), flush=True)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)

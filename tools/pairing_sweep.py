#!/usr/bin/env python3
"""lsa_pairing_product wall time vs n for one forced Miller kernel (LSA_MILLER_KERNEL): the call a
verifier makes -- upload, Miller loops, product tree, one final exponentiation, download -- in a
loop, i.e. every Miller kernel starts after a nearly idle final exponentiation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
rng = synth.Xoshiro256ss(seed=9)
N = 1 << 14
ps = lsa.batch_exp("g1", curve.generator("g1"), rng.uniform_fr(N))
qs = lsa.batch_exp("g2", curve.generator("g2"), rng.uniform_fr(N))
print("kernel", os.environ.get("LSA_MILLER_KERNEL", "auto"))
for n in (1, 16, 256, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 16384):
    lsa.pairing_product(ps[:n], qs[:n])
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); lsa.pairing_product(ps[:n], qs[:n]); ts.append(time.perf_counter() - t0)
    print("%6d  min %7.3f  median %7.3f ms" % (n, min(ts) * 1e3, sorted(ts)[3] * 1e3), flush=True)

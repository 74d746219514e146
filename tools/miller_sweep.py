#!/usr/bin/env python3
"""Miller-loop batch time vs n for one forced kernel (LSA_MILLER_KERNEL = 1 wave, 4 g12, 2 g6,
3 one lane per pairing), host buffers (upload + kernel + download of n x 384 B); run once per
kernel on the GPU box to find the crossovers miller_device() encodes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
rng = synth.Xoshiro256ss(seed=9)
N = 1 << 15
ps = lsa.batch_exp("g1", curve.generator("g1"), rng.uniform_fr(N))
qs = lsa.batch_exp("g2", curve.generator("g2"), rng.uniform_fr(N))
print("kernel", os.environ.get("LSA_MILLER_KERNEL", "auto"))
for n in (256, 512, 1024, 1536, 2048, 3072, 4096, 5120, 6144, 8192, 12288, 16384, 32768):
    lsa.miller_loop(ps[:n], qs[:n])
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); lsa.miller_loop(ps[:n], qs[:n]); best = min(best, time.perf_counter() - t0)
    print("%6d %8.3f ms" % (n, best * 1e3), flush=True)

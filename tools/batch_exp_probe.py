"""tools/batch_exp_probe.py -- lsa_g{1,2}_batch_exp on host buffers at small n (cputil::simpleBatchExp with a handful of
scalars, /root/reference/src/utils/util.h:119-134): blocking call times, results checked against k * base by another path."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
rng = synth.Xoshiro256ss(seed=9)
for group in ("g1", "g2"):
    G = curve.generator(group)
    for n in (1, 2, 3, 64, 4096):
        sc = rng.uniform_fr(n)
        lsa.batch_exp(group, G, sc)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); out = lsa.batch_exp(group, G, sc); ts.append((time.perf_counter() - t0) * 1e3)
        print("%s batch_exp n=%-5d min %.3f median %.3f ms" % (group, n, min(ts), sorted(ts)[5]))
# a base never seen: every call builds its table (254 dependent doublings first)
for group in ("g1", "g2"):
    G = curve.generator(group)
    sc = rng.uniform_fr(2)
    pts = lsa.batch_exp(group, G, rng.uniform_fr(12))
    ts = []
    for i in range(12):
        t0 = time.perf_counter(); out = lsa.batch_exp(group, pts[i], sc); ts.append((time.perf_counter() - t0) * 1e3)
    print("%s batch_exp n=2 on a NEW base every call: min %.3f median %.3f ms" % (group, min(ts), sorted(ts)[6]))

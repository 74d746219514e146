import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
G2 = curve.generator("g2")
import torch
rng = synth.Xoshiro256ss(seed=9)
ks = rng.uniform_fr(64)
pts = lsa.batch_exp("g2", G2, torch.from_numpy(ks.view(np.int64)).to("cuda:0")).cpu().numpy().view(np.uint64)
for n in (1, 5):
    ts = []
    for i in range(12):
        q = pts[(i * n) % 50:(i * n) % 50 + n]
        t0 = time.perf_counter(); t = lsa.g2_precompute(q); ts.append((time.perf_counter() - t0) * 1e3)
    print("g2_precompute n=%d: min %.3f median %.3f ms" % (n, min(ts), sorted(ts)[6]))

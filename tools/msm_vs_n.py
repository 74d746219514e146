#!/usr/bin/env python3
"""MSM time vs n on one resident handle (CPPoly::prove's ladder shape: prefixes of g1s; CommScheme::commit's G2 half):
single blocking call (latency) and 8 back-to-back asynchronous calls (throughput), every result
checked against k*G through the fixed-base kernel.  Run on the GPU box.
  python tools/msm_vs_n.py [g1|g2] [log2 of the handle's size, default 20] [largest log2 n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
dev = torch.device("cuda:0")
GROUP = sys.argv[1] if len(sys.argv) > 1 else "g1"
WIDTH = 12 if GROUP == "g1" else 24
G1 = curve.generator(GROUP)
LOGN = int(sys.argv[2]) if len(sys.argv) > 2 else 20
KMAX = int(sys.argv[3]) if len(sys.argv) > 3 else LOGN
N = 1 << LOGN
rng = synth.Xoshiro256ss(seed=77)
x = synth.arith_fr_mont(rng.fr_int(), rng.fr_int(), N)
B = lsa.Bases(GROUP, lsa.batch_exp(GROUP, G1, torch.from_numpy(x.view(np.int64)).to(dev)), on_device=True)
s = rng.uniform_fr(N)
d_s = torch.from_numpy(s.view(np.int64)).to(dev)
outs = torch.zeros((8, WIDTH), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
print("%8s %12s %14s %8s" % ("log2 n", "latency_ms", "pipelined_ms", "check"))
print("# %s handle of 2^%d points, copies: %s" % (GROUP, LOGN, B.has_table()))
for k in range(0, KMAX + 1):
    n = 1 << k
    B.msm(d_s, n=n)
    lat = []
    for _ in range(3):
        t0 = time.perf_counter(); r = B.msm(d_s, n=n); lat.append(time.perf_counter() - t0)
    want = lsa.normalize(GROUP, lsa.batch_exp(GROUP, G1, curve.fr_mont(synth.fr_dot_mont(s[:n], x[:n])).reshape(1, 4)))[0]
    ok = np.array_equal(lsa.normalize(GROUP, r.reshape(1, WIDTH))[0], want)
    lsa.synchronize()
    t0 = time.perf_counter()
    for rep in range(3):
        for j in range(8):
            B.msm_async(d_s, outs[j], n=n)
    lsa.synchronize()
    thr = (time.perf_counter() - t0) / 24
    print("%8d %12.3f %14.3f %8s" % (k, sorted(lat)[1] * 1e3, thr * 1e3, "ok" if ok else "MISMATCH"), flush=True)

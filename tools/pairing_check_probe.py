"""tools/pairing_check_probe.py -- one pairing check the way CPpoly's verifier issues it (src/gadgets/poly.h:105-112:
two Miller loops over precomputed G2 values, one of them inverted, one final exponentiation; a fresh G1 pair per check,
the same two G2 tables every time), blocking, host buffers.  Prints wall time per call; run under
rocprofv3 --kernel-trace for the kernels behind it."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
G1, G2 = curve.generator("g1"), curve.generator("g2")
n = 64
rng = synth.Xoshiro256ss(seed=7)
import torch
ks = rng.uniform_fr(n)
pts = lsa.batch_exp("g1", G1, torch.from_numpy(ks.view(np.int64)).to("cuda:0")).cpu().numpy().view(np.uint64)
pts = lsa.normalize("g1", pts)
tab = lsa.g2_precompute(np.stack([G2, G2]))
off = np.array([0, 2], dtype=np.uint64)
fl = np.array([0, 1], dtype=np.uint8)
one = None
ts = []
for rep in range(3):
    for i in range(n):
        g1 = np.stack([pts[i], pts[i]])
        t0 = time.perf_counter()
        r = lsa.pairing_terms(g1, off, tables=tab, index=[0, 1], flags=fl, final_exp=True)
        ts.append((time.perf_counter() - t0) * 1e3)
        if one is None: one = r.copy()
        assert (r == one).all()          # e(P, Q) / e(P, Q) = 1 every time
print("last 64 calls, us:", " ".join("%d" % (t * 1e3) for t in ts[-64:]))
ts = sorted(ts[n:])
print("pairing check (2 terms over cached tables + final exponentiation), blocking: min %.3f median %.3f p90 %.3f ms" % (ts[0], ts[len(ts) // 2], ts[len(ts) * 9 // 10]))

#!/usr/bin/env python3
"""Is a lone pairing slow because an almost idle chip sits in a low clock state?  Times
lsa_pairing_product (1 pair and 4096 pairs, host buffers) alone and while an unrelated stream keeps
the other CUs busy with matrix products (torch.mm in a loop on torch's stream)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
rng = synth.Xoshiro256ss(seed=9)
N = 1 << 12
ps = lsa.batch_exp("g1", curve.generator("g1"), rng.uniform_fr(N))
qs = lsa.batch_exp("g2", curve.generator("g2"), rng.uniform_fr(N))


def timed(n, reps=8):
    lsa.pairing_product(ps[:n], qs[:n])
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); lsa.pairing_product(ps[:n], qs[:n]); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3


for n in (1, 4096):
    print("n=%d alone: min %.3f median %.3f ms" % ((n,) + timed(n)), flush=True)
a = torch.randn(4096, 4096, device="cuda:0", dtype=torch.float32)
b = torch.randn(4096, 4096, device="cuda:0", dtype=torch.float32)
stop = False


def burn():
    while not stop:
        for _ in range(4):
            torch.mm(a, b)
        torch.cuda.current_stream().synchronize()


th = threading.Thread(target=burn); th.start()
time.sleep(0.5)
for n in (1, 4096):
    print("n=%d beside a matmul loop: min %.3f median %.3f ms" % ((n,) + timed(n)), flush=True)
stop = True; th.join()

"""tools/final_exp_probe.py -- lone final exponentiations (GPU box): blocking calls on host buffers, checked against the
oracle-free identity f^r == 1 is too slow here, so the value is compared with the one-lane kernel's (n >= 16384 path is a
different code path: tower code per lane) through lsa_final_exponentiation on a batch."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
rng = np.random.default_rng(3)
G1, G2 = curve.generator("g1"), curve.generator("g2")
f = lsa.miller_loop(lsa.normalize("g1", G1.reshape(1, 12)), lsa.normalize("g2", G2.reshape(1, 24)))
for _ in range(3):
    lsa.final_exponentiation(f)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); r = lsa.final_exponentiation(f); ts.append((time.perf_counter() - t0) * 1e3)
print("one final exponentiation, blocking, host buffers: min %.3f median %.3f ms" % (min(ts), sorted(ts)[10]))

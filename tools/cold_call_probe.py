"""tools/cold_call_probe.py -- what the FIRST lsa_g1_msm of a process costs (GPU box): the wall-clock split of
lsa_msm_host_stats for the first three calls on a 2^20-point host vector, after lsa_init alone."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
t0 = time.perf_counter()
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
print("init_ms %.2f" % ((time.perf_counter() - t0) * 1e3))
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
G1 = curve.generator("g1")
bases = np.tile(G1, (n, 1))
rng = synth.Xoshiro256ss(seed=5)
for rep in range(4):
    sc = rng.uniform_fr(n)
    t = time.perf_counter()
    lsa.msm("g1", bases, sc)
    dt = (time.perf_counter() - t) * 1e3
    st = lsa.msm_host_stats()
    print("call %d: %.3f ms  h2d %.3f fp_wait %.3f prepare %.3f msm %.3f hit %d table %d" % (rep, dt, st["h2d_scalars_ms"], st["fingerprint_wait_ms"], st["bases_prepare_ms"], st["msm_ms"], st["cache_hit"], st["table"]))

"""tools/single_call_trace_g2.py [n] -- GPU box, under rocprofv3 --kernel-trace: isolated blocking G2 MSM calls on a resident
2^16-point handle (a pause between them), for tools/rocpd_summary.py timeline."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
dev = torch.device("cuda:0")
G2 = curve.generator("g2")
N = 1 << 16
rng = synth.Xoshiro256ss(seed=78)
x = synth.arith_fr_mont(rng.fr_int(), rng.fr_int(), N)
B = lsa.Bases("g2", lsa.batch_exp("g2", G2, torch.from_numpy(x.view(np.int64)).to(dev)), on_device=True)
s = rng.uniform_fr(N)
d_s = torch.from_numpy(s.view(np.int64)).to(dev)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for _ in range(30):                      # the handle's pre-shifted copies are built in the background after a few calls
    B.msm(d_s, n=n)
lsa.crs_cache_wait_tables()
for _ in range(4):
    B.msm(d_s, n=n)
    time.sleep(0.05)
t = []
for _ in range(5):
    t0 = time.perf_counter(); B.msm(d_s, n=n); t.append((time.perf_counter() - t0) * 1e3); time.sleep(0.05)
print("single call ms", sorted(t))

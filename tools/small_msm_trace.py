import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
dev = torch.device("cuda:0")
G1 = curve.generator("g1")
N = 1 << 20
rng = synth.Xoshiro256ss(seed=77)
x = synth.arith_fr_mont(rng.fr_int(), rng.fr_int(), N)
B = lsa.Bases("g1", lsa.batch_exp("g1", G1, torch.from_numpy(x.view(np.int64)).to(dev)), on_device=True)
s = rng.uniform_fr(N)
d_s = torch.from_numpy(s.view(np.int64)).to(dev)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B.msm(d_s, n=n)
for _ in range(20):
    B.msm(d_s, n=n)

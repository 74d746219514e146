"""Dev helper: per-stage MSM timing at n=2^k for g1 / g2 on a resident handle (inputs from
legosnark_amd.synth; result checked by the discrete-log identity through batch_exp)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import legosnark_amd as lsa
from legosnark_amd import curve, synth
lsa.init(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
group = sys.argv[2] if len(sys.argv) > 2 else "g1"
n = 1 << k
G = curve.generator(group)
rng = synth.Xoshiro256ss(seed=5)
x = synth.arith_fr_mont(rng.fr_int(), rng.fr_int(), n)
s = rng.uniform_fr(n)
dev = torch.device("cuda:0")
t = time.time(); B = lsa.Bases(group, lsa.batch_exp(group, G, torch.from_numpy(x.view(np.int64)).to(dev)), on_device=True); lsa.synchronize(); print("handle %.3fs, copies %d" % (time.time() - t, B.table_windows()))
d_s = torch.from_numpy(s.view(np.int64)).to(dev)
w = 12 if group == "g1" else 24
out = torch.zeros(w, dtype=torch.int64, device=dev)
r = B.msm(d_s)
want = lsa.normalize(group, lsa.batch_exp(group, G, curve.fr_mont(synth.fr_dot_mont(s, x)).reshape(1, 4)))[0]
print("check", np.array_equal(lsa.normalize(group, r.reshape(1, w))[0], want))
lsa.profile_enable(True)
for it in range(8):
    B.msm_async(d_s, out)
lsa.synchronize()
print({k2: round(v, 3) for k2, v in lsa.profile_last_msm().items()})
lsa.profile_enable(False)
t = time.time()
for it in range(8): B.msm_async(d_s, out)
lsa.synchronize()
print("pipelined avg %.3f ms -> %.3e pairs/s" % ((time.time() - t) / 8 * 1e3, n / ((time.time() - t) / 8)))

import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import legosnark_amd as lsa
import oracle_lib as o
R = o.R
lsa.init(0)
rng = np.random.default_rng(3)
n = 1 << 20
a, b = 0xA11CE << 200 | 0xB0B, 0xC0FFEE << 100 | 0x5
bases = o.arith_bases("g1", a, b, n)
lsa.set_table_threshold(0)
B = lsa.Bases("g1", bases)
kind = sys.argv[1]
if kind == "few":
    vals = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(5)]
    sc = [vals[int(x)] for x in rng.integers(0, 5, size=n)]
else:
    sc, v = [], 0
    while len(sc) < n:
        v = int.from_bytes(rng.bytes(32), "little") % R
        sc += [v] * int(rng.integers(1, 5000))
    sc = sc[:n]
d_s = torch.from_numpy(o.fr_mont_array(sc).view(np.int64)).to("cuda:0")
torch.cuda.synchronize()
B.msm(d_s)
for _ in range(5):
    t0 = time.perf_counter(); B.msm(d_s); print("%.2f ms" % ((time.perf_counter() - t0) * 1e3))

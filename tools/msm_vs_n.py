"""Dev helper: G1 MSM latency vs n (random bases and n copies of the generator)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import legosnark_amd as lsa
from legosnark_amd import curve
lsa.init(0)
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1)
def rfr(c):
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (c, 4), dtype=torch.int64, device=dev, generator=gen); t[:, 3] &= (1 << 60) - 1; return t.contiguous()
N = 1 << 20
Brand = lsa.Bases("g1", lsa.batch_exp("g1", curve.generator("g1"), rfr(N)), on_device=True)
Bgen = lsa.Bases("g1", torch.from_numpy(np.tile(curve.generator("g1").view(np.int64), (N, 1))).to(dev), on_device=True)
s = rfr(N); out = torch.zeros(12, dtype=torch.int64, device=dev)
for k in range(0, 21, 2):
    n = 1 << k
    res = []
    for B in (Brand, Bgen):
        B.msm_async(s[:n], out, n=n); lsa.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): B.msm_async(s[:n], out, n=n)
        lsa.synchronize(); res.append((time.perf_counter() - t0) / 5 * 1e3)
    lsa.profile_enable(True); Brand.msm_async(s[:n], out, n=n); st = lsa.profile_last_msm(); lsa.profile_enable(False)
    print("n=2^%-2d c=%2d random %.3f ms  generator-copies %.3f ms  stages %s" % (k, lsa.msm_window_bits(n), res[0], res[1], {a: round(b, 3) for a, b in st.items() if a in ("digits","scatter","accumulate","reduce","fold")}))

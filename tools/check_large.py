"""One-off: G1 MSM at n=2^k (default 24) checked by the discrete-log identity, with inputs
generated on the GPU: bases x_i*G (batch_exp), MSM(s, bases) == (sum s_i*x_i mod r)*G."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import legosnark_amd as lsa
from legosnark_amd import curve
import oracle_lib as o
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << k
lsa.init(0)
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(99)
def rfr(c):
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (c, 4), dtype=torch.int64, device=dev, generator=gen); t[:, 3] &= (1 << 60) - 1; return t.contiguous()
x = rfr(n); s = rfr(n)
t0 = time.time(); bases = lsa.batch_exp("g1", curve.generator("g1"), x); B = lsa.Bases("g1", bases, on_device=True); del bases
print("setup %.2fs" % (time.time() - t0))
out = torch.zeros(12, dtype=torch.int64, device=dev)
B.msm_async(s, out); lsa.synchronize()
t0 = time.perf_counter()
for _ in range(3): B.msm_async(s, out)
lsa.synchronize(); dt = (time.perf_counter() - t0) / 3
print("n=2^%d: %.3f ms per MSM, %.3e pairs/s" % (k, dt * 1e3, n / dt))
got = out.cpu().numpy().view(np.uint64)
# host check: sum over Montgomery representatives; canonical value of rep v is v*R^-1 mod r
xs = x.cpu().numpy().view(np.uint64); ss = s.cpu().numpy().view(np.uint64)
def ints(a):
    return (a[:, 0].astype(object) | (a[:, 1].astype(object) << 64) | (a[:, 2].astype(object) << 128) | (a[:, 3].astype(object) << 192))
t0 = time.time(); acc = int(np.dot(ints(xs), ints(ss))) ; print("host dot %.1fs" % (time.time() - t0))
Rinv = pow(1 << 256, -1, o.R)
kk = acc * Rinv * Rinv % o.R
want = o.g1_mul(o.generator("g1"), o.fr_mont(kk))
print("MATCH" if o.g1_canonical_affine(got) == o.g1_canonical_affine(want) else "MISMATCH")

"""Diagnostic: duration of the Miller-loop kernel in different neighbourhoods (back to back, after
a single-workgroup kernel, after host-side idle) -- run under rocprofv3 --kernel-trace and read
the per-dispatch durations from the trace."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import legosnark_amd as lsa, oracle_lib as o
from legosnark_amd import curve
lsa.init(0)
n = 4096
gen = torch.Generator(device="cuda:0").manual_seed(1)
def rfr(c):
    t = torch.randint(0, 1 << 62, (c, 4), dtype=torch.int64, device="cuda:0", generator=gen); t[:, 3] &= (1 << 60) - 1; return t
ps = lsa.batch_exp("g1", curve.generator("g1"), rfr(n)); qs = lsa.batch_exp("g2", curve.generator("g2"), rfr(n))
out = torch.empty((n, 48), dtype=torch.int64, device="cuda:0")
fe = torch.empty((1, 48), dtype=torch.int64, device="cuda:0")
L = lsa.lib()
def miller(): lsa._check(L.lsa_miller_loop(ps.data_ptr(), qs.data_ptr(), n, out.data_ptr(), 1))
def fexp(): lsa._check(L.lsa_final_exponentiation(out.data_ptr(), 1, fe.data_ptr(), 1))
for _ in range(2): miller(); fexp()
lsa.synchronize()
for _ in range(4): miller()                       # A: back to back
lsa.synchronize()
for _ in range(4): miller(); fexp()               # B: alternating with the one-element final exponentiation
lsa.synchronize()
for _ in range(4): miller(); lsa.synchronize(); time.sleep(0.003)   # C: host idle in between
for _ in range(4): miller(); lsa.synchronize(); time.sleep(0.05)    # D: long host idle in between
lsa.synchronize()
hp = ps.cpu().numpy().view(np.uint64); hq = qs.cpu().numpy().view(np.uint64)
for _ in range(3): lsa.miller_loop(hp, hq)                      # E: host-buffer entry point (H2D, kernel, D2H)
for _ in range(3): lsa.pairing_product(hp, hq)                  # F: host-buffer product

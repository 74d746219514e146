#!/usr/bin/env python3
"""tools/fused_experiment.py -- k_miller_fused (2^12 fresh pairs, one launch) timed with HIP events under the setting of
LSA_FUSED_EXPERIMENT in the environment: 0 = the kernel as shipped; 1 / 2 / 4 and their sums leave out the G2 side's
combine phase / the Fq12 chain / the G2 side's product phase (WRONG values: timing only).  What any restructuring of
those phases could save at most.

The switches were removed from the shipped kernel after round 5 (a nonzero value made every fresh-pair pairing product
wrong and poisoned the G2 table cache): to repeat the experiment check out commit 6534bcb (csrc/tmiller.h G2Pre::round,
csrc/pairing.hip miller_fused_device), whose numbers are profiles/r05_v1_fused_miller_experiments.txt.  On the current
tree this script times the kernel as shipped, whatever LSA_FUSED_EXPERIMENT says."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import legosnark_amd as lsa
from legosnark_amd import curve
lsa.init(0)
n = 1 << 12
gen = torch.Generator(device="cuda:0").manual_seed(1)
def rfr(c):
    t = torch.randint(0, 1 << 62, (c, 4), dtype=torch.int64, device="cuda:0", generator=gen); t[:, 3] &= (1 << 60) - 1; return t
ps = lsa.batch_exp("g1", curve.generator("g1"), rfr(n)); qs = lsa.batch_exp("g2", curve.generator("g2"), rfr(n))
if os.environ.get("FUSED_NORMALISED", "1") != "0":          # Z = 1 inputs, as bench.py's config 5 passes them (no inversion in the kernel's set-up)
    ps = torch.from_numpy(lsa.normalize("g1", ps.cpu().numpy().view(np.uint64)).view(np.int64)).to("cuda:0")
    qs = torch.from_numpy(lsa.normalize("g2", qs.cpu().numpy().view(np.uint64)).view(np.int64)).to("cuda:0")
out = torch.empty((n, 48), dtype=torch.int64, device="cuda:0")
L = lsa.lib()
def miller(): lsa._check(L.lsa_miller_loop(ps.data_ptr(), qs.data_ptr(), n, out.data_ptr(), 1))
for _ in range(3): miller()
lsa.synchronize()
ts = []
for _ in range(10):
    lsa.synchronize(); t0 = time.perf_counter(); miller(); lsa.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("normalised=%s LSA_FUSED_EXPERIMENT=%s  miller_loop(2^12 fresh pairs): median %.3f ms  min %.3f ms" % (os.environ.get("FUSED_NORMALISED", "1"), os.environ.get("LSA_FUSED_EXPERIMENT", "0"), sorted(ts)[5], min(ts)))

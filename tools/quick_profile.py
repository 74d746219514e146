"""Dev helper (uses the test oracle for inputs): per-stage MSM timing at n=2^k."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import legosnark_amd as lsa, oracle_lib as o
lsa.init(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
group = sys.argv[2] if len(sys.argv) > 2 else "g1"
n = 1 << k
t = time.time(); bases = o.arith_bases(group, 0x1F2E3D4C5B6A7988 << 64 | 0x123, 0x0FEDCBA987654321 << 32 | 0x77, n); print("bases gen %.1fs" % (time.time() - t))
t = time.time(); B = lsa.Bases(group, bases); print("bases upload+normalize %.3fs" % (time.time() - t))
rng = np.random.default_rng(1)
raw = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); raw[:, 3] &= np.uint64((1 << 60) - 1)
d_s = torch.from_numpy(raw.view(np.int64)).to("cuda:0"); torch.cuda.synchronize()
lsa.profile_enable(True)
for it in range(4):
    t = time.time(); r = B.msm(d_s); dt = time.time() - t
    print("iter %d wall %.3f ms" % (it, dt * 1e3), {k2: round(v, 3) for k2, v in lsa.profile_last_msm().items()})
lsa.profile_enable(False)
t = time.time()
for it in range(5): r = B.msm(d_s)
print("no-profile avg wall %.3f ms -> %.3e pairs/s" % ((time.time() - t) / 5 * 1e3, n / ((time.time() - t) / 5)))

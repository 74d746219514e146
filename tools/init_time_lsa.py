#!/usr/bin/env python3
"""tools/init_time_lsa.py -- where lsa_init's second goes: the library loaded without torch (LSA_NO_TORCH_PRELOAD=1), timed
against a bare HIP start-up (tools/init_time.py) in fresh processes; LSA_WARM=0 / LSA_WARM_MB select the warm-up."""
import os, sys, time
os.environ.setdefault("LSA_NO_TORCH_PRELOAD", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
import legosnark_amd as lsa
L = lsa.lib()
t1 = time.perf_counter()
lsa.init(0)
t2 = time.perf_counter()
print("LSA_WARM=%s LSA_WARM_MB=%s: load library %.1f ms, lsa_init %.1f ms" % (os.environ.get("LSA_WARM", "-"), os.environ.get("LSA_WARM_MB", "-"), (t1 - t0) * 1e3, (t2 - t1) * 1e3))

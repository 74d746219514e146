"""GPU: the two-lane G2 bucket accumulation (one Fq component per lane of a pair, csrc/fp29x2l.h, LSA_G2_PAIR=1) against
the oracle: uniform scalars, runs of equal points (every first addition of a bucket is a doubling), points at infinity
and negated digits, on a handle with pre-shifted copies (the wide path, where that kernel runs).  The switch is read
once per process, so the variant runs in its own interpreter."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import legosnark_amd as lsa
import oracle_lib as o
lsa.init(0)
lsa.crs_cache_configure(lsa.CRS_CACHE_FULL, 8 << 30)
lsa.crs_cache_table_after(1)
lsa.set_table_threshold(1024)
out = {}
for name, n in (("uniform", 6000), ("equal_points", 3000), ("with_infinity", 2500)):
    if name == "equal_points":
        bases = np.ascontiguousarray(np.repeat(o.arith_bases("g2", 5, 0, 1), n, axis=0))
    else:
        bases = np.ascontiguousarray(o.arith_bases("g2", 77, 3, n))
    if name == "with_infinity":
        bases[5] = 0; bases[6] = 0; bases[n - 1] = 0
    sc, _ = o.random_scalars(n, seed=n)
    if name == "equal_points":
        sc[: n // 2] = sc[0]                      # the same digit in every window for half of the pairs
    want = o.g2_canonical_affine(o.multi_exp("g2", bases, sc, mode="mixed"))
    for i in range(2):
        lsa.msm("g2", bases, sc)
    lsa.crs_cache_wait_tables()
    got = o.g2_canonical_affine(lsa.msm("g2", bases, sc))
    st = lsa.msm_host_stats()
    out[name] = {"ok": got == want, "table": st["table"]}
print("RESULT " + json.dumps(out))
"""


def test_pair_kernel_matches_the_oracle():
    env = dict(os.environ, LSA_G2_PAIR="1")
    r = subprocess.run([sys.executable, "-c", SNIPPET % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert set(res) == {"uniform", "equal_points", "with_infinity"}
    for name, v in res.items():
        assert v["ok"] and v["table"] == 1, (name, v)

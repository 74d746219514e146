"""GPU: the two bucket-sort pipelines of the wide MSM path must agree.

Large single MSMs take the partitioned sort (k_partition / k_fine_sort_part: population-balanced
segments staged through LDS); LSA_NO_PART=1 forces the older k_scatter_wide / k_fine_sort pair, which
segmented calls and n > 2^25 still use.  The switch is read once per process, so each variant runs
in its own interpreter on the same seeded inputs; both results are also checked against the
known-discrete-log identity MSM(s, (a + i b) G) = (sum s_i (a + i b)) G."""
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
import torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import legosnark_amd as lsa
import oracle_lib as o
lsa.init(0)
R = o.R
n_table = (1 << 19) + 64
a, b = 0x1F2E3D4C5B6A79881726354 << 30 | 7, 0x9E3779B97F4A7C15 << 8 | 3
bases = o.arith_bases("g1", a, b, n_table)
lsa.set_table_threshold(0)
B = lsa.Bases("g1", bases)
assert B.has_table()
rng = np.random.default_rng(2026)
out = {}
for name, n in (("uniform_2^17+5", (1 << 17) + 5), ("uniform_full", n_table), ("small_values", 1 << 18), ("one_value", (1 << 16) + 1)):
    if name.startswith("uniform"):
        sc = [int.from_bytes(rng.bytes(32), "little") %% R for _ in range(n)]
    elif name == "small_values":
        sc = [int(x) for x in rng.integers(0, 1 << 20, size=n)]
    else:
        sc = [0x0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F %% R] * n
    d_s = torch.from_numpy(o.fr_mont_array(sc).view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    got = B.msm(d_s, n=n)
    k = sum(s * (a + i * b) for i, s in enumerate(sc)) %% R
    want = o.g1_mul(o.generator("g1"), o.fr_mont(k))
    ca = o.g1_canonical_affine(got)
    out[name] = {"affine": None if ca is None else [str(ca[0]), str(ca[1])], "ok": ca == o.g1_canonical_affine(want)}
print("RESULT " + json.dumps(out))
"""


def run_variant(extra_env):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, "-c", SNIPPET % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_partitioned_and_scatter_sorts_agree():
    part = run_variant({})
    scat = run_variant({"LSA_NO_PART": "1"})
    assert set(part) == set(scat)
    for name in part:
        assert part[name]["ok"], "partitioned sort: " + name
        assert scat[name]["ok"], "scatter sort: " + name
        assert part[name]["affine"] == scat[name]["affine"], name

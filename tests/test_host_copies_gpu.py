"""GPU: host <-> device copies of caller buffers (csrc/capi.hip, upload_host / download_host).  By default copies of
32 KiB .. 16 MiB go through pinned slots owned by the library (several threads from two slots up), larger ones are left
to the runtime; LSA_H2D=staged / direct force one way for every size.  The switch is read once per process, so each
variant runs in its own interpreter on the same seeded inputs: the same MSM point (multi-slot upload), the same
batch_exp / normalise outputs (multi-slot download), byte for byte, and equal to the oracle."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import legosnark_amd as lsa
import oracle_lib as o
lsa.init(0)
out = {}
n = (1 << 18) + 77                                     # 8 MiB of scalars, 25 MB of points: several slots, a ragged tail
bases = np.ascontiguousarray(o.arith_bases("g1", 31, 5, n))
sc, _ = o.random_scalars(n, seed=9)
got = lsa.msm("g1", bases, sc)
out["msm"] = [str(x) for x in o.g1_canonical_affine(got)]
out["msm_ok"] = o.g1_canonical_affine(got) == o.g1_canonical_affine(o.multi_exp("g1", bases, sc, mode="mixed"))
m = 70001                                              # 6.7 MB of points back: several slots
pts = lsa.batch_exp("g1", o.arith_bases("g1", 3, 0, 1)[0], sc[:m])
out["batch_exp_sha"] = hashlib.sha256(np.ascontiguousarray(pts).tobytes()).hexdigest()
nz = lsa.normalize("g1", pts)
out["normalize_sha"] = hashlib.sha256(np.ascontiguousarray(nz).tobytes()).hexdigest()
k = 12345
out["spot_ok"] = o.g1_canonical_affine(pts[k]) == o.g1_canonical_affine(o.g1_mul(o.arith_bases("g1", 3, 0, 1)[0], sc[k]))
w = o.fr_mont(o.fr_root_of_unity(16))
a = lsa.fr_ntt(sc[: 1 << 16], w)                       # 2 MiB up and down
out["ntt_sha"] = hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
out["ntt_ok"] = bool(np.array_equal(a, o.fr_domain_transform(sc[: 1 << 16], w)))
print("RESULT " + json.dumps(out))
"""


def run_variant(mode):
    env = dict(os.environ)
    env.pop("LSA_H2D", None)
    if mode:
        env["LSA_H2D"] = mode
    r = subprocess.run([sys.executable, "-c", SNIPPET % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_staged_direct_and_default_copies_agree():
    res = {mode: run_variant(mode) for mode in ("", "staged", "direct")}
    for mode, v in res.items():
        assert v["msm_ok"] and v["spot_ok"] and v["ntt_ok"], (mode, v)
    assert res[""] == res["staged"] == res["direct"]

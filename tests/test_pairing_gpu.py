"""GPU parity tests for the pairing path (C-ABI): Miller-loop values and GT values are
canonical Fq12 elements, so they are compared byte-for-byte with the oracle and with the
golden vectors of the independent model."""
import random

import numpy as np
import pytest

import oracle_lib as o
from conftest import f12_dec, g1_dec, g2_dec

pytestmark = pytest.mark.gpu
P, R = o.P, o.R


def test_pairing_golden(lsa, golden):
    pr = golden["pairing"]
    g1, g2 = o.generator("g1"), o.generator("g2")
    e = lsa.pairing_product(g1, g2)
    assert o.fq12_to_model(e) == f12_dec(pr["e_g1_g2"])
    rng = random.Random(15)
    for t in pr["bilinear"]:
        Pp = o.g1_from_affine(g1_dec(t["P"]), rng.randrange(2, P))                         # un-normalised inputs
        Qq = o.g2_from_affine(g2_dec(t["Q"]), (rng.randrange(2, P), rng.randrange(P)))
        assert o.fq12_to_model(lsa.pairing_product(Pp, Qq)) == f12_dec(t["e"])
    pp = pr["planted_product"]
    Ps = o.g1_array([g1_dec(x) for x in pp["P"]])
    Qs = o.g2_array([g2_dec(x) for x in pp["Q"]])
    assert np.array_equal(lsa.pairing_product(Ps, Qs), o.fq12_one())
    p2 = pr["product2"]
    Ps = o.g1_array([g1_dec(x) for x in p2["P"]])
    Qs = o.g2_array([g2_dec(x) for x in p2["Q"]])
    assert o.fq12_to_model(lsa.pairing_product(Ps, Qs)) == f12_dec(p2["result"])


@pytest.mark.parametrize("n", [1, 2, 3, 9, 70])
def test_miller_and_final_exp_vs_oracle(lsa, n):
    ps = o.arith_bases("g1", 1000 + n, 77, n)          # un-normalised Jacobian
    qs = o.arith_bases("g2", 31337, 5 + n, n)
    f_gpu = lsa.miller_loop(ps, qs)
    f_cpu = o.miller_loop_batch(ps, qs)
    assert np.array_equal(f_gpu, f_cpu)                 # libff miller_loop values, bit-exact
    e_gpu = lsa.final_exponentiation(f_gpu)
    for i in range(0, n, max(1, n // 5)):
        assert np.array_equal(e_gpu[i], o.final_exponentiation(f_cpu[i]))
    assert np.array_equal(lsa.pairing_product(ps, qs), o.pairing_product(ps, qs))
    # double_miller_loop shape: product of Miller loops without the final exponentiation
    prod = f_cpu[0]
    for i in range(1, n):
        prod = o.fq12_mul(prod, f_cpu[i])
    assert np.array_equal(lsa.miller_loop_product(ps, qs), prod)


def test_pairing_identities_and_edge_cases(lsa):
    g1, g2 = o.generator("g1"), o.generator("g2")
    one = o.fq12_one()
    # empty product, final_exp(1) = 1
    assert np.array_equal(lsa.pairing_product(np.zeros((0, 12), np.uint64), np.zeros((0, 24), np.uint64)), one)
    assert np.array_equal(lsa.final_exponentiation(one)[0], one)
    # e(P,Q) e(-P,Q) = 1  (simple_pairing_check shape, globl.h:94-105)
    import ctypes as C
    neg = np.zeros(12, dtype=np.uint64)
    o.lib().og1_neg(o._p(neg), o._p(g1))
    assert np.array_equal(lsa.pairing_product(np.stack([g1, neg]), np.stack([g2, g2])), one)
    # bilinearity through the GPU only: e(aP, bQ) == e(abP, Q)
    a, b = 0x1234567890ABCDEF, 0xFEDCBA0987654321
    lhs = lsa.pairing_product(o.g1_mul(g1, o.fr_mont(a)), o.g2_mul(g2, o.fr_mont(b)))
    rhs = lsa.pairing_product(o.g1_mul(g1, o.fr_mont(a * b % R)), g2)
    assert np.array_equal(lhs, rhs)
    # points at infinity follow libff's (unguarded) to_affine_coordinates() convention
    inf1 = np.zeros(12, dtype=np.uint64); inf1[4:8] = o.fq_mont(1)
    assert np.array_equal(lsa.miller_loop(inf1, g2)[0], o.miller_loop_batch(inf1.reshape(1, 12), g2.reshape(1, 24))[0])


def test_batched_planted_product_256(lsa):
    """BASELINE.json configs[4] shape at a size the oracle checks in seconds: a planted
    relation sum alpha_j beta_j = 0 mod r makes prod e(P_j, Q_j) = 1."""
    n = 256
    rng = random.Random(99)
    al = [rng.randrange(1, R) for _ in range(n)]
    be = [rng.randrange(1, R) for _ in range(n - 1)]
    s = sum(x * y for x, y in zip(al, be)) % R
    be.append((-s) * pow(al[-1], -1, R) % R)
    ps = o.batch_exp("g1", o.generator("g1"), o.fr_mont_array(al))
    qs = o.batch_exp("g2", o.generator("g2"), o.fr_mont_array(be))
    assert np.array_equal(lsa.pairing_product(ps, qs), o.fq12_one())
    # break the relation -> not one
    qs2 = qs.copy(); qs2[0] = qs[1]
    assert not np.array_equal(lsa.pairing_product(ps, qs2), o.fq12_one())


@pytest.mark.parametrize("n", [0, 1, 8, 9, 77])
def test_fq12_product_vs_oracle(lsa, n):
    ps = o.arith_bases("g1", 5, 3, max(n, 1))[:n]
    qs = o.arith_bases("g2", 9, 2, max(n, 1))[:n]
    fs = o.miller_loop_batch(ps, qs) if n else np.zeros((0, 48), np.uint64)
    assert np.array_equal(lsa.fq12_product(fs), o.fq12_product(fs))


def test_sharded_pairing_product_two_ranks_on_one_gpu(lsa):
    """SURVEY.md section 8e, pairings: split the batch over ranks, all-gather the 384-byte
    partial products, multiply, one final exponentiation.  Both ranks run here in turn (the
    collective is replaced by a hand-over), the result must equal the unsplit GT value."""
    from legosnark_amd import sharded
    n = 13
    ps = o.arith_bases("g1", 21, 4, n)
    qs = o.arith_bases("g2", 8, 15, n)
    partials = []
    for r in range(2):
        lo, hi = sharded.shard_range(n, 2, r)
        partials.append(lsa.miller_loop_product(ps[lo:hi], qs[lo:hi]))

    class FakeDist:
        def get_backend(self):
            return "gloo"

        def all_gather(self, parts, t):
            import torch
            for i in range(2):
                parts[i].copy_(torch.from_numpy(partials[i].view(np.int64)))

    for r in range(2):
        job = sharded.make_gpu_sharded_pairing(lsa, 2, r, dist=FakeDist())
        assert np.array_equal(job.run(ps, qs), o.pairing_product(ps, qs))


@pytest.mark.parametrize("kernel", [3, 5, 6])
def test_every_miller_kernel_vs_oracle(kernel):
    """LSA_MILLER_KERNEL forces one Miller-loop path: 3 = one pairing per lane (miller.h, the family's fallback),
    5 = line tables (k_g2_precomp + the table kernels), 6 = the fused kernel (what fresh pairs take by default).  Each is
    run in its own process on 25 pairs (five full workgroups of five) with un-normalised inputs and an infinity, byte for
    byte against the oracle.  (The wavefront / six-lane / twelve-lane engines of rounds 1-2 were removed in round 5.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import legosnark_amd as lsa, oracle_lib as o\n"
        "lsa.init(0)\n"
        "ps = o.arith_bases('g1', 77, 5, 25); qs = o.arith_bases('g2', 99, 7, 25)\n"
        "ps[3] = 0\n"
        "ps[4] = o.generator('g1'); qs[4] = o.generator('g2')\n"
        "assert np.array_equal(lsa.miller_loop(ps, qs), o.miller_loop_batch(ps, qs))\n"
        "assert np.array_equal(lsa.pairing_product(ps, qs), o.pairing_product(ps, qs))\n"
        "print('OK')\n"
    ) % (root, os.path.join(root, "tests"))
    env = dict(os.environ, LSA_MILLER_KERNEL=str(kernel))
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("naf,lockstep", [("1", "0"), ("0", "0"), ("1", "1"), ("0", "1")])
def test_signed_digit_miller_loop_gives_the_same_gt(naf, lockstep):
    """Jobs whose values only leave through a final exponentiation run the fused kernel over the SIGNED digits of 6u + 2
    (88 table entries instead of 102: csrc/miller.h ate_naf_digit; LSA_MILLER_NAF=0 keeps libff's binary loop).  The
    Miller values differ from libff's by vertical lines, which the final exponent kills: every GT value -- one product, a
    verifier's segments, terms with -P, P at infinity -- byte for byte the oracle's either way; a workgroup that holds a G2
    point at infinity keeps the binary loop (libff's (0, 1) convention is not a point: only its own formulas define the
    result), so that case is bit-exact too.  Raw Miller values (no final exponentiation) always come from the binary loop."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import legosnark_amd as lsa, oracle_lib as o\n"
        "lsa.init(0)\n"
        "n = 43\n"
        "ps = o.arith_bases('g1', 1234, 11, n); qs = o.arith_bases('g2', 4321, 3, n)\n"
        "ps[3] = 0\n"
        "assert np.array_equal(lsa.pairing_product(ps, qs), o.pairing_product(ps, qs))\n"
        "assert np.array_equal(lsa.miller_loop_product(ps, qs), o.fq12_product(o.miller_loop_batch(ps, qs)))\n"
        "offs = np.array([0, 5, 5, 12, 30, n], dtype=np.uint64)\n"
        "got = lsa.pairing_product_segments(ps, qs, offs)\n"
        "for j in range(5):\n"
        "    lo, hi = int(offs[j]), int(offs[j + 1])\n"
        "    assert np.array_equal(got[j], o.pairing_product(ps[lo:hi], qs[lo:hi]) if hi > lo else o.fq12_one()), j\n"
        "qs2 = qs.copy(); qs2[17] = 0\n"
        "assert np.array_equal(lsa.pairing_product(ps, qs2), o.pairing_product(ps, qs2))\n"
        "ps3 = ps.copy(); ps3[5, 4:8] = o.fq_mont(12345)\n"          # P off the curve: libff's formulas still define a value
        "assert np.array_equal(lsa.pairing_product(ps3, qs), o.pairing_product(ps3, qs))\n"
        "qs3 = qs.copy(); qs3[9, 8:12] = o.fq_mont(777)\n"           # Q off the twist
        "assert np.array_equal(lsa.pairing_product(ps, qs3), o.pairing_product(ps, qs3))\n"
        "print('OK')\n"
    ) % (root, os.path.join(root, "tests"))
    env = dict(os.environ, LSA_MILLER_KERNEL="6", LSA_MILLER_NAF=naf, LSA_FUSED_LOCKSTEP=lockstep)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("helper", ["1", "0"])
def test_final_exponentiation_of_arbitrary_elements(helper):
    """lsa_final_exponentiation on elements that are NOT Miller values: random Fq12 elements, elements of the subfields
    Fq6 / Fq2 / Fq, one, sparse ones -- byte for byte the oracle's libff chain.  LSA_FE_HELPER=1 (default): the 256-lane
    kernel whose fourth wavefront replaces the chain's one inversion by a power of the norm (csrc/w12.h: w12_rows, HLP;
    tools/gen_fe_scalar_exponent.py); 0: the 192-lane kernel with the inversion.  Zero has no inverse: both kernels
    return zero for it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, random, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import legosnark_amd as lsa, oracle_lib as o\n"
        "lsa.init(0)\n"
        "rng = random.Random(99)\n"
        "def elt(mask):\n"
        "    f = np.zeros(48, dtype=np.uint64)\n"
        "    for c in range(12):\n"
        "        if (mask >> c) & 1: f[4 * c:4 * c + 4] = o.fq_mont(rng.randrange(o.P))\n"
        "    return f\n"
        "masks = [0xfff] * 12 + [0x03f, 0x003, 0x001, 0xfc0, 0x555, 0xaaa, 0x800, 0x041, 0x3cf]\n"
        "fs = np.stack([elt(m) for m in masks] + [o.fq12_one().reshape(48)])\n"
        "got = lsa.final_exponentiation(fs)\n"
        "for i in range(len(fs)):\n"
        "    assert np.array_equal(got[i].reshape(-1), o.final_exponentiation(fs[i]).reshape(-1)), (i, hex(masks[i]) if i < len(masks) else 'one')\n"
        "one = lsa.final_exponentiation(fs[:1])\n"              # a lone element: the same kernel, one workgroup
        "assert np.array_equal(one[0].reshape(-1), got[0].reshape(-1))\n"
        "z = lsa.final_exponentiation(np.zeros((1, 48), dtype=np.uint64))\n"
        "assert not z.any()\n"
        "print('OK')\n"
    ) % (root, os.path.join(root, "tests"))
    env = dict(os.environ, LSA_FE_HELPER=helper)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:]


def test_large_batch_takes_the_grouped_kernel_and_matches(lsa):
    """2000 pairs (above the one-wavefront-per-pairing range): product of the Miller values
    against the oracle on a sample, and the planted relation prod e(a_i G1, b_i G2) = 1 with
    sum a_i b_i = 0 mod r over the whole batch."""
    rng = random.Random(5)
    n = 2000
    a = [rng.randrange(1, R) for _ in range(n)]
    b = [rng.randrange(1, R) for _ in range(n - 1)]
    s = sum(x * y for x, y in zip(a, b)) % R
    b.append((-s) * pow(a[-1], -1, R) % R)
    ps = lsa.batch_exp("g1", o.generator("g1"), o.fr_mont_array(a))
    qs = lsa.batch_exp("g2", o.generator("g2"), o.fr_mont_array(b))
    f = lsa.miller_loop(ps, qs)
    for i in (0, 9, 10, 1234, n - 1):
        assert np.array_equal(f[i], o.miller_loop_batch(ps[i:i + 1], qs[i:i + 1])[0]), i
    assert np.array_equal(lsa.pairing_product(ps, qs), o.fq12_one())


def test_segmented_pairing_products_vs_oracle(lsa):
    """lsa_pairing_product_segments: a verifier's many short products (CPPoly::verify multiplies two
    or three Miller values per final exponentiation, src/gadgets/poly.h:105-122) in one pass, every
    segment byte for byte against the oracle; empty segments give one."""
    lens = [3, 2, 0, 1, 4, 0, 3, 7]
    n = sum(lens)
    ps = o.arith_bases("g1", 2024, 9, n)
    qs = o.arith_bases("g2", 77, 13, n)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    got = lsa.pairing_product_segments(ps, qs, offs)
    raw = lsa.pairing_product_segments(ps, qs, offs, final_exp=False)
    f = o.miller_loop_batch(ps, qs)
    for j, m in enumerate(lens):
        lo = int(offs[j])
        assert np.array_equal(raw[j], o.fq12_product(f[lo:lo + m])), j
        assert np.array_equal(got[j], o.pairing_product(ps[lo:lo + m], qs[lo:lo + m]) if m else o.fq12_one()), j

"""Public known answers (EIP-196 / EIP-197 precompile vectors, tests/golden/public_vectors.json)
and Miller-loop pins that do not pass through the hard part of the final exponentiation.

CPU half: the oracle (oracle/bn254.c) against the vectors.  GPU half (-m gpu): the HIP path
through the C-ABI against the same vectors.  The vectors were produced by other people's
implementations of this curve (see tests/golden/make_public_vectors.py), not by anything in
this repository."""
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib as o

HERE = os.path.dirname(os.path.abspath(__file__))
P, R = o.P, o.R


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(HERE, "golden", "public_vectors.json")) as f:
        return json.load(f)


def g1_from_hex(h):
    x, y = int(h[:64], 16), int(h[64:128], 16)
    return None if x == 0 and y == 0 else (x, y)


def g2_from_hex(h):
    xi, xr, yi, yr = (int(h[64 * i:64 * i + 64], 16) for i in range(4))
    return None if xi == xr == yi == yr == 0 else ((xr, xi), (yr, yi))


def pairs_from_hex(inp):
    ps, qs = [], []
    for j in range(len(inp) // 384):
        c = inp[384 * j:384 * j + 384]
        ps.append(g1_from_hex(c[:128]))
        qs.append(g2_from_hex(c[128:]))
    return o.g1_array(ps).reshape(-1, 12), o.g2_array(qs).reshape(-1, 24)


def f12_dec(f):
    return [(int(c[0], 16), int(c[1], 16)) for c in f]


def easy_part(f_limbs):
    """(p^6-1)(p^2+1)-th power of a libff-layout Fq12, in the model's representation: t = conj(f) / f, then
    frobenius^2(t) * t -- with the C oracle (its Frobenius map is pinned to the generic p-th power in
    tests/test_oracle_golden.py), so that no Python model is needed where these tests run."""
    t = o.fq12_mul(o.fq12_unitary_inverse(f_limbs), o.fq12_inverse(f_limbs))
    return o.fq12_to_model(o.fq12_mul(o.fq12_frobenius(t, 2), t))


# ------------------------------------------------------------------ CPU: the oracle
def test_oracle_ec_add(vec):
    for e in vec["ec_add"]:
        a, b = g1_from_hex(e["input"][:128]), g1_from_hex(e["input"][128:256])
        want = g1_from_hex(e["expected"])
        A, B = o.g1_from_affine(a, 5), o.g1_from_affine(b)
        assert o.g1_canonical_affine(o.g1_add(A, B)) == want, e["name"]
        if b is not None:
            assert o.g1_canonical_affine(o.g1_mixed_add(A, B)) == want, e["name"]
        ones = o.fr_mont_array([1, 1])
        assert o.g1_canonical_affine(o.multi_exp("g1", np.stack([A, B]), ones, mode="mixed")) == want, e["name"]
        assert o.g1_canonical_affine(o.multi_exp("g1", np.stack([A, B]), ones, mode="inner")) == want, e["name"]


def test_oracle_ec_mul(vec):
    for e in vec["ec_mul"]:
        a, k = g1_from_hex(e["input"][:128]), int(e["input"][128:192], 16)
        want = g1_from_hex(e["expected"])
        A = o.g1_from_affine(a, 7)
        km = o.fr_mont(k % R)                      # the point has order r
        assert o.g1_canonical_affine(o.g1_mul(A, km)) == want, e["name"]
        assert o.g1_canonical_affine(o.multi_exp("g1", A.reshape(1, 12), km.reshape(1, 4), mode="inner")) == want, e["name"]
        assert o.g1_canonical_affine(o.batch_exp("g1", A, km.reshape(1, 4))[0]) == want, e["name"]


def test_oracle_ec_pairing(vec):
    for e in vec["ec_pairing"]:
        ps, qs = pairs_from_hex(e["input"])
        is_one = np.array_equal(o.pairing_product(ps, qs), o.fq12_one())
        assert int(is_one) == e["expected"], e["name"]


def test_oracle_miller_values_up_to_subfield_factor(vec):
    """oracle Miller value / textbook Miller value lies in a proper subfield of Fp12: equal
    (p^6-1)(p^2+1)-th powers.  Pins the projective line scaling of the restated libff loop
    independently of the hard part of the final exponentiation."""
    for pin in vec["miller_pins"]:
        ps, qs = pairs_from_hex(pin["pair"])
        f = o.miller_loop_batch(ps, qs)[0]
        assert easy_part(f) == f12_dec(pin["easy"]), pin["from"]


# ------------------------------------------------------------------ GPU: the HIP path
@pytest.mark.gpu
def test_gpu_ec_add(lsa, vec):
    for e in vec["ec_add"]:
        a, b = g1_from_hex(e["input"][:128]), g1_from_hex(e["input"][128:256])
        want = g1_from_hex(e["expected"])
        pts = np.stack([o.g1_from_affine(a, 5), o.g1_from_affine(b)])
        got = lsa.msm("g1", pts, o.fr_mont_array([1, 1]))      # the "ones" path of multi_exp_with_mixed_addition
        assert o.g1_canonical_affine(got) == want, e["name"]
        import torch
        d = torch.from_numpy(pts.view(np.int64)).to("cuda:0")
        out = torch.zeros(12, dtype=torch.int64, device="cuda:0")
        lsa.sum_async("g1", d, 2, out)
        lsa.synchronize()
        assert o.g1_canonical_affine(out.cpu().numpy().view(np.uint64)) == want, e["name"]


@pytest.mark.gpu
def test_gpu_ec_mul(lsa, vec):
    pts, ks, wants = [], [], []
    for e in vec["ec_mul"]:
        a, k = g1_from_hex(e["input"][:128]), int(e["input"][128:192], 16)
        want = g1_from_hex(e["expected"])
        A, km = o.g1_from_affine(a, 7), o.fr_mont(k % R)
        assert o.g1_canonical_affine(lsa.msm("g1", A.reshape(1, 12), km.reshape(1, 4))) == want, e["name"]
        assert o.g1_canonical_affine(lsa.batch_exp("g1", A, km.reshape(1, 4))[0]) == want, e["name"]
        pts.append(A); ks.append(km); wants.append(want)
    out = lsa.scalar_mul_batch(np.stack(pts), np.stack(ks))
    for i, w in enumerate(wants):
        assert o.g1_canonical_affine(out[i]) == w, i


@pytest.mark.gpu
def test_gpu_ec_pairing(lsa, vec):
    for e in vec["ec_pairing"]:
        ps, qs = pairs_from_hex(e["input"])
        is_one = np.array_equal(lsa.pairing_product(ps, qs), o.fq12_one())
        assert int(is_one) == e["expected"], e["name"]


@pytest.mark.gpu
def test_gpu_miller_values_up_to_subfield_factor(lsa, vec):
    for pin in vec["miller_pins"]:
        ps, qs = pairs_from_hex(pin["pair"])
        f = lsa.miller_loop(ps, qs)[0]
        assert easy_part(f) == f12_dec(pin["easy"]), pin["from"]
    # every Miller kernel the batch size can select (wave / g12 / g6 / one-lane): replicate the
    # pinned pairs up to each kernel's range and check a sample of each batch
    pins = vec["miller_pins"]
    base_p = np.concatenate([pairs_from_hex(p["pair"])[0] for p in pins])
    base_q = np.concatenate([pairs_from_hex(p["pair"])[1] for p in pins])
    for n in (2000, 6000, 70000):
        reps = (n + len(pins) - 1) // len(pins)
        f = lsa.miller_loop(np.tile(base_p, (reps, 1))[:n], np.tile(base_q, (reps, 1))[:n])
        for i in (0, n // 2 + 3, n - 1):
            assert easy_part(f[i]) == f12_dec(pins[i % len(pins)]["easy"]), (n, i)

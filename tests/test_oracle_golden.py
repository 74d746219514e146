"""CPU suite: pins the C restatement (oracle/) against the golden vectors produced by
the independent big-int model, the public alt_bn128 known answers and algebraic
identities.  No GPU needed."""
import random

import numpy as np
import pytest

import oracle_lib as o
from conftest import f12_dec, g1_dec, g2_dec

P, R = o.P, o.R


def test_field_montgomery_roundtrip(golden):
    import ctypes as C
    for e in golden["field"]:
        which = 0 if e["field"] == "fq" else 1
        x = int(e["x"], 16)
        lim = o.int_to_limbs(x)
        out = np.zeros(4, dtype=np.uint64)
        o.lib().ofp_from_canonical(o._p(out), o._p(lim), C.c_int(which))
        assert o.limbs_to_int(out) == int(e["mont"], 16)
        back = np.zeros(4, dtype=np.uint64)
        o.lib().ofp_to_canonical(o._p(back), o._p(out), C.c_int(which))
        assert o.limbs_to_int(back) == x


def test_field_mul_inv(golden):
    import ctypes as C
    for e in golden["field_mul"]:
        which = 0 if e["field"] == "fq" else 1
        m = P if which == 0 else R
        a, b = int(e["a"], 16), int(e["b"], 16)
        am = o.int_to_limbs(a * o.MONT % m)
        bm = o.int_to_limbs(b * o.MONT % m)
        r = np.zeros(4, dtype=np.uint64)
        o.lib().ofp_mul(o._p(r), o._p(am), o._p(bm), C.c_int(which))
        assert o.limbs_to_int(r) == int(e["ab"], 16) * o.MONT % m
        o.lib().ofp_inv(o._p(r), o._p(am), C.c_int(which))
        assert o.limbs_to_int(r) == int(e["a_inv"], 16) * o.MONT % m


def test_public_known_answers(golden):
    # 2*G1 (EIP-196 test vector, quoted in SURVEY.md section 4)
    two = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
           9918110051302171585080402603319702774565515993150576347155970296011118125764)
    assert g1_dec(golden["kat"]["g1_two"]) == two
    g = o.generator("g1")
    assert o.g1_canonical_affine(o.g1_dbl(g)) == two
    assert o.g1_canonical_affine(o.g1_add(g, g)) == two
    # r*G = O in both groups (r = 0 mod r, so multiply by r-1 and add G)
    rm1 = o.fr_mont(R - 1)
    assert o.g1_canonical_affine(o.g1_add(o.g1_mul(g, rm1), g)) is None
    g2 = o.generator("g2")
    assert o.g2_canonical_affine(g2) == g2_dec(golden["kat"]["g2_gen"])
    assert o.g2_canonical_affine(o.g2_add(o.g2_mul(g2, rm1), g2)) is None
    assert o.lib().og1_is_well_formed(o._p(g)) == 1
    assert o.lib().og2_is_well_formed(o._p(g2)) == 1


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_group_law_cases(golden, group):
    dec = g1_dec if group == "g1" else g2_dec
    mk = o.g1_from_affine if group == "g1" else o.g2_from_affine
    add = o.g1_add if group == "g1" else o.g2_add
    madd = o.g1_mixed_add if group == "g1" else o.g2_mixed_add
    canon = o.g1_canonical_affine if group == "g1" else o.g2_canonical_affine
    rng = random.Random(7)
    for e in golden[group + "_add"]:
        a, b, s = dec(e["a"]), dec(e["b"]), dec(e["sum"])
        za = rng.randrange(2, P) if group == "g1" else (rng.randrange(2, P), rng.randrange(P))
        zb = rng.randrange(2, P) if group == "g1" else (rng.randrange(2, P), rng.randrange(P))
        assert canon(add(mk(a), mk(b))) == s, e["name"]
        assert canon(add(mk(a, za) if a else mk(a), mk(b, zb) if b else mk(b))) == s, e["name"]
        assert canon(madd(mk(a, za) if a else mk(a), mk(b))) == s, e["name"]


def test_scalar_mul(golden):
    g1, g2 = o.generator("g1"), o.generator("g2")
    for e in golden["scalar_mul"]:
        k = o.fr_mont(int(e["k"], 16))
        assert o.g1_canonical_affine(o.g1_mul(g1, k)) == g1_dec(e["g1"])
        assert o.g2_canonical_affine(o.g2_mul(g2, k)) == g2_dec(e["g2"])


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_golden(golden, group):
    dec = g1_dec if group == "g1" else g2_dec
    arr = o.g1_array if group == "g1" else o.g2_array
    canon = o.g1_canonical_affine if group == "g1" else o.g2_canonical_affine
    rng = random.Random(11)
    for e in golden[group + "_msm"]:
        pts = [dec(p) for p in e["bases"]]
        sc = o.fr_mont_array([int(s, 16) for s in e["scalars"]])
        want = dec(e["result"])
        if group == "g1":
            zs = [rng.randrange(1, P) for _ in pts]
        else:
            zs = [(rng.randrange(1, P), rng.randrange(P)) for _ in pts]
        bases = arr(pts, zs)
        for mode, chunks, threads in (("inner", 1, 0), ("multi_exp", 1, 0), ("multi_exp", 3, 0),
                                      ("mixed", 1, 0), ("mixed", 4, 2)):
            got = canon(o.multi_exp(group, bases, sc, chunks=chunks, threads=threads, mode=mode))
            assert got == want, (e["name"], mode, chunks)


def test_bdlo12_window_rule(golden):
    for n, c in golden["bdlo12_window"]:
        assert o.lib().oracle_bdlo12_window(n) == c
    # values quoted in SURVEY.md section 7 step 1
    for n, c in ((1, 2), (2, 3), (3, 4), (8, 4), (1 << 10, 9), (1025, 10), (1 << 20, 16), (1 << 24, 18)):
        assert o.lib().oracle_bdlo12_window(n) == c


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_batch_exp_matches_scalar_mul(group):
    g = o.generator(group)
    sc, ints = o.random_scalars(9, seed=5)
    sc[0] = o.fr_mont(0)
    sc[1] = o.fr_mont(1)
    mul = o.g1_mul if group == "g1" else o.g2_mul
    canon = o.g1_canonical_affine if group == "g1" else o.g2_canonical_affine
    for window in (None, 1, 5):
        out = o.batch_exp(group, g, sc, window=window)
        for i in range(len(sc)):
            assert canon(out[i]) == canon(mul(g, sc[i]))


def test_msm_known_discrete_log_medium():
    """n = 4096 with bases (a + i b) G: result must be (sum s_i (a + i b)) G."""
    n = 4096
    a, b = 0x1234567 << 100 | 5, 0xDEADBEEF << 64 | 9
    sc, ints = o.random_scalars(n, seed=3)
    for group in ("g1", "g2"):
        bases = o.arith_bases(group, a, b, n)
        got = o.multi_exp(group, bases, sc, chunks=8, threads=4, mode="mixed")
        k = sum(s * (a + i * b) for i, s in enumerate(ints)) % R
        mul = o.g1_mul if group == "g1" else o.g2_mul
        canon = o.g1_canonical_affine if group == "g1" else o.g2_canonical_affine
        assert canon(got) == canon(mul(o.generator(group), o.fr_mont(k)))


def test_final_exponentiation_of_arbitrary_elements_vs_model():
    """The oracle's final exponentiation (libff's chain: easy part with its inversion, three exponentiations by z) on elements
    that are NOT Miller values -- random Fq12 elements, elements of Fq6 / Fq2 / Fq, sparse ones -- against the independent model's
    plain power f^(libff's exponent) (oracle/pymodel: square-and-multiply on schoolbook Fp12 arithmetic).  This is the checker of
    the GPU kernel that replaces the chain's inversion by a power of the norm (csrc/w12.h: w12_final_exponentiation_h)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "pymodel"))
    import bn254_model as M
    rng = random.Random(2026)
    for mask in (0x3f, 0x3f, 0x3f, 0x15, 0x01, 0x2a, 0x09, 0x20):       # which of the six Fp2 coefficients are non-zero
        g = [(rng.randrange(P), rng.randrange(P)) if (mask >> i) & 1 else (0, 0) for i in range(6)]
        if mask == 0x01:
            g[0] = (rng.randrange(1, P), 0)                              # an element of Fq
        assert o.fq12_to_model(o.final_exponentiation(o.fq12_from_model(g))) == M.final_exponentiation(g), hex(mask)


def test_pairing_golden(golden):
    pr = golden["pairing"]
    g1, g2 = o.generator("g1"), o.generator("g2")
    e = o.reduced_pairing(g1, g2)
    assert o.fq12_to_model(e) == f12_dec(pr["e_g1_g2"])
    one = o.fq12_one()
    assert o.fq12_to_model(o.final_exponentiation(one)) == f12_dec(pr["final_exp_of_one"])
    assert np.array_equal(o.final_exponentiation(one), one)
    rng = random.Random(5)
    for t in pr["bilinear"]:
        Pp = o.g1_from_affine(g1_dec(t["P"]), rng.randrange(2, P))
        Qq = o.g2_from_affine(g2_dec(t["Q"]), (rng.randrange(2, P), rng.randrange(P)))
        assert o.fq12_to_model(o.reduced_pairing(Pp, Qq)) == f12_dec(t["e"])
    pp = pr["planted_product"]
    Ps = o.g1_array([g1_dec(x) for x in pp["P"]])
    Qs = o.g2_array([g2_dec(x) for x in pp["Q"]])
    assert np.array_equal(o.pairing_product(Ps, Qs), one)
    p2 = pr["product2"]
    Ps = o.g1_array([g1_dec(x) for x in p2["P"]])
    Qs = o.g2_array([g2_dec(x) for x in p2["Q"]])
    assert o.fq12_to_model(o.pairing_product(Ps, Qs)) == f12_dec(p2["result"])


def test_pairing_identities():
    import ctypes as C
    g1, g2 = o.generator("g1"), o.generator("g2")
    a, b = 123456789, 987654321
    ea = o.reduced_pairing(o.g1_mul(g1, o.fr_mont(a)), o.g2_mul(g2, o.fr_mont(b)))
    eb = o.reduced_pairing(o.g1_mul(g1, o.fr_mont(a * b % R)), g2)
    assert np.array_equal(ea, eb)
    # e(P,Q) * e(-P,Q) == 1
    neg = np.zeros(12, dtype=np.uint64)
    o.lib().og1_neg(o._p(neg), o._p(g1))
    prod = o.pairing_product(np.stack([g1, neg]), np.stack([g2, g2]))
    assert np.array_equal(prod, o.fq12_one())
    # Frobenius == generic p-th power; double_miller_loop == product of miller loops
    f = o.miller_loop_batch(g1.reshape(1, 12), g2.reshape(1, 24))[0]
    fr = np.zeros(48, dtype=np.uint64)
    fp = np.zeros(48, dtype=np.uint64)
    o.lib().ofq12_frobenius(o._p(fr), o._p(f), C.c_uint(1))
    o.lib().ofq12_pow_p(o._p(fp), o._p(f))
    assert np.array_equal(fr, fp)
    inv = np.zeros(48, dtype=np.uint64)
    o.lib().ofq12_inverse(o._p(inv), o._p(f))
    assert np.array_equal(o.fq12_mul(f, inv), o.fq12_one())


def test_oracle_mtxmultiexp_matches_plain_scalar_muls():
    """sparsemexpG's three branches (zero entry, generator entry folded in Fr, multi_exp of the
    rest) against the plain definition sum_e exps[row(e)] * vals[e] with libff scalar*point."""
    import numpy as np
    rng = np.random.default_rng(3)
    pool = o.arith_bases("g1", 77, 13, 16)
    gen = o.generator("g1")
    nrows = 4
    k, _ = o.random_scalars(nrows, seed=21)
    vals, rows, col_ptr = [], [], [0]
    for j in range(12):
        for _ in range(j % 5):
            t = rng.integers(0, 6)
            vals.append(np.zeros(12, dtype=np.uint64) if t == 0 else gen if t == 1 else pool[rng.integers(0, 16)])
            rows.append(int(rng.integers(0, nrows)))
        col_ptr.append(len(vals))
    vals = np.array(vals, dtype=np.uint64)
    got = o.mtxmultiexp(vals, rows, col_ptr, k)
    terms = o.g1_mul_batch(vals, k[np.array(rows, dtype=np.int64)])
    for j in range(12):
        acc = np.zeros(12, dtype=np.uint64)
        for e in range(col_ptr[j], col_ptr[j + 1]):
            acc = o.g1_add(acc, terms[e])
        assert o.g1_canonical_affine(got[j]) == o.g1_canonical_affine(acc)


def _fr_ints(a):
    import numpy as np
    Rinv = pow(o.MONT, -1, o.R)
    return [o.limbs_to_int(x) * Rinv % o.R for x in np.asarray(a, dtype=np.uint64).reshape(-1, 4)]


def test_oracle_fr_vector_loops_match_bigint_formulas():
    """The Fr loops restated from poly.h:51-67, polytools.h:207-234 and mle.h:199-210 against
    the same formulas evaluated with Python integers (independent of the C field code)."""
    R = o.R
    for d in (0, 1, 2, 5):
        N = 1 << d
        v_m, v = o.random_scalars(N, seed=100 + d)
        r_m, r = (o.random_scalars(d, seed=200 + d) if d else (o.fr_mont_array([]).reshape(0, 4), []))
        v, r = [int(x) for x in v], [int(x) for x in r]
        # CPpoly witness recursion
        tmp, w = list(v), [0] * N
        start = 0
        for i in range(d):
            bound = 1 << (d - i - 1)
            for p in range(bound):
                a, b = tmp[2 * p], tmp[2 * p + 1]
                w[start + p] = (b - a) % R
                tmp[p] = (-a * (r[i] - 1) + b * r[i]) % R
            start += bound
        assert _fr_ints(o.fr_cppoly_witness(v_m, r_m)) == w
        # multilinear extension: sum_p v[p] prod_i (bit_i(p) ? r_i : 1 - r_i)
        want = 0
        for p in range(N):
            term = v[p]
            for i in range(d):
                term = term * (r[i] if (p >> i) & 1 else 1 - r[i]) % R
            want = (want + term) % R
        assert _fr_ints(o.fr_eval_mle(v_m, r_m)) == [want]
        if d:
            half = N // 2
            got = _fr_ints(o.fr_push_randomness(v_m, r_m[0]))
            assert got == [(v[p] * (1 - r[0]) + v[p + half] * r[0]) % R for p in range(half)]


def test_oracle_sumcheck_round_matches_bigint_formula():
    """make_new_h_poly restated with polynomial products per p, against
    h_j = ((1-rho) + (2rho-1)X) * pre * sum_p suff[p] prod_t (v0 + (v1-v0) X) in Python integers."""
    R = o.R
    for m, half, beta in ((2, 8, True), (1, 4, True), (3, 5, True), (2, 6, False)):
        tabs_m, tabs = [], []
        for t in range(m):
            a_m, a = o.random_scalars(2 * half, seed=1000 + 10 * m + t)
            tabs_m.append(a_m); tabs.append([int(x) for x in a])
        s_m, s = o.random_scalars(half, seed=77 + m)
        pr_m, pr = o.random_scalars(2, seed=78 + m)
        S = [0] * (m + 1)
        for p in range(half):
            q = [int(s[p]) if beta else 1]
            for t in range(m):
                v0, dv = tabs[t][p], (tabs[t][p + half] - tabs[t][p]) % R
                nq = [0] * (len(q) + 1)
                for i, c in enumerate(q):
                    nq[i] = (nq[i] + c * v0) % R
                    nq[i + 1] = (nq[i + 1] + c * dv) % R
                q = nq
            S = [(x + y) % R for x, y in zip(S, q)]
        if beta:
            pre, rho = int(pr[0]), int(pr[1])
            e0, e1 = (1 - rho) * pre % R, (2 * rho - 1) * pre % R
            want = [((S[i] * e0 if i <= m else 0) + (S[i - 1] * e1 if i >= 1 else 0)) % R for i in range(m + 2)]
            got = o.fr_sumcheck_round(tabs_m, suff=s_m, pre=pr_m[0], rho_j=pr_m[1])
        else:
            want = S
            got = o.fr_sumcheck_round(tabs_m)
        assert _fr_ints(got) == want
    old_m, old = o.random_scalars(10, seed=5)
    k_m, k = o.random_scalars(1, seed=6)
    assert _fr_ints(o.fr_scale_upper(old_m, k_m[0])) == [int(old[5 + p]) * int(k[0]) % R for p in range(5)]
    # DPBeta::compute_eq_tbl (mle.h:93-105), the loop as the reference wrote it, in Python integers.  Its doubling step reads
    # dst[p >> 1] (not dst[p & (2^j - 1)]), so every factor ends up selected by the TOP bit of p: the table holds
    # prod (1 - r_j) in its lower half and prod r_j in its upper half.  That is what DPBeta::precomputeAll consumes and what a
    # drop-in has to return; the closed form is asserted beside the literal loop.
    for d in (1, 2, 3, 5):
        r_m, r = o.random_scalars(d, seed=70 + d)
        dst = [(1 - int(r[0])) % R, int(r[0])]
        for j in range(1, d):
            dst = [((int(r[j]) if p >= (1 << j) else (1 - int(r[j]))) * dst[p >> 1]) % R for p in range(1 << (j + 1))]
        assert _fr_ints(o.fr_eq_table(r_m)) == dst
        lo = hi = 1
        for x in r:
            lo, hi = lo * (1 - int(x)) % R, hi * int(x) % R
        assert dst == [lo] * (1 << (d - 1)) + [hi] * (1 << (d - 1))


def test_oracle_radix2_fft_is_the_dft():
    """The restated _basic_radix2_FFT against the definition a[k] = sum_i a[i] w^(ik) in Python
    integers; iFFT / coset variants by their defining identities."""
    R = o.R
    for log_n in (1, 2, 3, 5):
        n = 1 << log_n
        w = o.fr_root_of_unity(log_n)
        assert pow(w, n, R) == 1 and pow(w, n // 2, R) == R - 1
        a_m, a = o.random_scalars(n, seed=300 + log_n)
        a = [int(x) for x in a]
        w_m = o.fr_mont(w)
        g = 5
        got = _fr_ints(o.fr_domain_transform(a_m, w_m))
        assert got == [sum(a[i] * pow(w, i * k, R) for i in range(n)) % R for k in range(n)]
        back = _fr_ints(o.fr_domain_transform(o.fr_domain_transform(a_m, w_m), w_m, inverse=True))
        assert back == a
        cos = _fr_ints(o.fr_domain_transform(a_m, w_m, coset=o.fr_mont(g)))
        assert cos == [sum(a[i] * pow(g, i, R) * pow(w, i * k, R) for i in range(n)) % R for k in range(n)]
        back = _fr_ints(o.fr_domain_transform(o.fr_domain_transform(a_m, w_m, coset=o.fr_mont(g)), w_m, inverse=True, coset=o.fr_mont(g)))
        assert back == a


def test_oracle_step_domain_is_evaluation_at_its_points():
    """The restated step_radix2_domain (m = 2^b + 2^s) against its definition in Python integers: FFT gives the values of
    the polynomial at omega^(2k), k < 2^b, then at omega sigma^j, j < 2^s (libfqfft's get_domain_element order); iFFT and
    the coset variants by their defining identities."""
    R = o.R
    for big_log, small_log in ((1, 0), (2, 0), (2, 1), (3, 1), (4, 3), (5, 0), (5, 2)):
        big, small = 1 << big_log, 1 << small_log
        m = big + small
        w = o.fr_root_of_unity(big_log + 1)
        sigma = pow(w, 2 * big // small, R)
        assert sigma == o.fr_root_of_unity(small_log)                       # get_root_of_unity(small_m)
        pts = [pow(w, 2 * k, R) for k in range(big)] + [w * pow(sigma, j, R) % R for j in range(small)]
        assert len(set(pts)) == m
        a_m, a = o.random_scalars(m, seed=900 + 10 * big_log + small_log)
        a = [int(x) for x in a]
        w_m, g = o.fr_mont(w), 5
        ev = lambda cs, x: sum(c * pow(x, i, R) for i, c in enumerate(cs)) % R
        got = _fr_ints(o.fr_step_domain_transform(a_m, big_log, small_log, w_m))
        assert got == [ev(a, x) for x in pts], (big_log, small_log)
        back = o.fr_step_domain_transform(o.fr_step_domain_transform(a_m, big_log, small_log, w_m), big_log, small_log, w_m, inverse=True)
        assert _fr_ints(back) == a
        cos = o.fr_step_domain_transform(a_m, big_log, small_log, w_m, coset=o.fr_mont(g))
        assert _fr_ints(cos) == [ev(a, g * x % R) for x in pts]
        back = o.fr_step_domain_transform(cos, big_log, small_log, w_m, inverse=True, coset=o.fr_mont(g))
        assert _fr_ints(back) == a

"""GPU parity tests for the Fr-vector kernels through the
C-ABI: Fr values are canonical Montgomery residues, so outputs are compared byte for byte
with the oracle's line-by-line restatement of the reference loops."""
import numpy as np
import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d", [0, 1, 2, 3, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19, 20, 21, 22])
def test_cppoly_witness_vs_oracle(lsa, d):
    """d <= 9: every round in the one-workgroup tail kernel; from 10 on passes of k_fold_pairs_fused: tiles of 4 inputs
    (d = 10: two rounds in registers, no tree), of 16 ... 2048 (the tree in LDS, one and two values per lane), of 4096 from
    d = 20 on; two passes at 21 and 22 (the second with tiles of 4 and 16)."""
    v, _ = o.random_scalars(1 << d, seed=300 + d)
    r, _ = o.random_scalars(max(d, 1), seed=400 + d)
    r = r[:d]
    got = lsa.cppoly_witness(v, r)
    want = o.fr_cppoly_witness(v, r)
    assert np.array_equal(got, want)
    assert not got[-1].any()                       # value-initialised tail entry (poly.h:52)


@pytest.mark.parametrize("d", [0, 1, 2, 7, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 21, 22])
def test_eval_mle_vs_oracle(lsa, d):
    v, _ = o.random_scalars(1 << d, seed=500 + d)
    r, _ = o.random_scalars(max(d, 1), seed=600 + d)
    r = r[:d]
    assert np.array_equal(lsa.eval_mle(v, r), o.fr_eval_mle(v, r))


@pytest.mark.parametrize("switch", ["LSA_FR_GRAPHS=0"])
def test_fold_schedules_behind_their_switches_give_the_oracles_bytes(switch):
    """Without the cached hipGraph (plain launches): the same witness coefficients and the same evalMLE value as the oracle
    at d = 16 and 17, called twice, from a child process (the switch is read once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import legosnark_amd as lsa, oracle_lib as o\n"
        "lsa.init(0)\n"
        "for d in (16, 17):\n"
        "    v, _ = o.random_scalars(1 << d, seed=900 + d); r, _ = o.random_scalars(d, seed=950 + d)\n"
        "    ww, we = o.fr_cppoly_witness(v, r), o.fr_eval_mle(v, r)\n"
        "    for rep in range(2):\n"
        "        assert np.array_equal(lsa.cppoly_witness(v, r), ww), (d, rep)\n"
        "        assert np.array_equal(lsa.eval_mle(v, r), we), (d, rep)\n"
        "print('OK')\n"
    ) % (root, os.path.join(root, "tests"))
    k, val = switch.split("=")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{k: val}), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:]


def test_eval_mle_special_points(lsa):
    """r in {0,1}^d selects one table entry; r = 0 and r = 1 exercise the (1-r) and r factors."""
    d = 6
    v, _ = o.random_scalars(1 << d, seed=77)
    for idx in (0, 1, 37, 63):
        r = np.array([o.fr_mont((idx >> i) & 1) for i in range(d)], dtype=np.uint64)
        assert np.array_equal(lsa.eval_mle(v, r), v[idx])


@pytest.mark.parametrize("half", [1, 2, 100, 4096])
def test_push_randomness_vs_oracle(lsa, half):
    old, _ = o.random_scalars(2 * half, seed=700 + half)
    r, _ = o.random_scalars(1, seed=701)
    assert np.array_equal(lsa.fr_fold(old, r[0]), o.fr_push_randomness(old, r[0]))


def test_cppoly_prove_on_device_end_to_end(lsa):
    """CPPoly::prove (poly.h:45-91) with nothing leaving the device between the fold and the
    MSM ladder: w = witness coefficients (device), pf.witness[i] = MSM(g1s, w[start_i ..])."""
    import torch
    d = 10
    N = 1 << d
    v, _ = o.random_scalars(N, seed=801)
    r, _ = o.random_scalars(d, seed=802)
    bases = o.arith_bases("g1", 5150, 31, N)
    B = lsa.Bases("g1", bases)
    d_v = torch.from_numpy(v.view(np.int64)).to("cuda:0")
    d_r = torch.from_numpy(r.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    d_w = lsa.cppoly_witness(d_v, d_r)
    outs = torch.zeros((d, 12), dtype=torch.int64, device="cuda:0")
    start = 0
    for i in range(d):
        m = 1 << (d - i - 1)
        B.msm_async(d_w[start:start + m], outs[i], n=m)
        start += m
    lsa.synchronize()
    w = o.fr_cppoly_witness(v, r)
    assert np.array_equal(d_w.cpu().numpy().view(np.uint64), w)
    got = outs.cpu().numpy().view(np.uint64)
    start = 0
    for i in range(d):
        m = 1 << (d - i - 1)
        want = o.multi_exp("g1", bases[:m], w[start:start + m], mode="mixed")
        assert o.g1_canonical_affine(got[i]) == o.g1_canonical_affine(want)
        start += m
    B.close()


@pytest.mark.parametrize("m,half,beta,suff", [(2, 1, True, True), (2, 7, True, True), (2, 5000, True, True), (2, 300, True, False),
                                              (1, 64, True, True), (3, 1000, True, True), (4, 33, False, False), (2, 70000, False, False),
                                              (2, (1 << 18) + 777, True, True), (2, 5 * 262144 + 3, True, True)])
def test_sumcheck_round_vs_oracle(lsa, m, half, beta, suff):
    """(The last two shapes give the lanes of the two-table kernel 1 ... 2 and 5 ... 6 indices each: full groups of four
    indices sharing a reduction and a remainder of one or two.)"""
    tabs = [o.random_scalars(2 * half, seed=2000 + 7 * m + t)[0] for t in range(m)]
    s = o.random_scalars(half, seed=31 + m)[0] if suff else None
    pr = o.random_scalars(2, seed=32 + m)[0]
    kw = dict(suff=s, pre=pr[0], rho_j=pr[1]) if beta else {}
    got = lsa.sumcheck_round(tabs, **kw)
    want = o.fr_sumcheck_round(tabs, **kw)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kernel", ["reduced", "wide"])
def test_sumcheck_three_tables_both_kernels(kernel):
    """Three tables: the default kernel reduces every product (k_sumcheck_partial3r, three wavefronts per SIMD, 768
    workgroups), LSA_SC3=wide keeps the one with four shared reductions per four indices (k_sumcheck_partial<3>).  Both
    against the oracle: with and without the suffix table and the beta factor, a ragged size, and 3.4 M indices -- seventeen
    and eighteen per lane, so that the lane's lazy sums are brought back below 2r (every sixteen additions) and the shared
    reductions meet a remainder of one and two."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import legosnark_amd as lsa, oracle_lib as o\n"
        "lsa.init(0)\n"
        "rng = np.random.default_rng(5000)\n"
        "def residues(n):\n"                                  # any 256-bit value below r is a Montgomery residue: top limb below 2^60
        "    a = rng.integers(0, 1 << 64, (n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1); return a\n"
        "for half, beta, suff in ((1, True, True), (1000, True, False), (777, False, False), (70001, True, True), (3400000 + 5, True, True)):\n"
        "    tabs = [residues(2 * half) for t in range(3)]\n"
        "    s = residues(half) if suff else None\n"
        "    pr = o.random_scalars(2, seed=78)[0]\n"
        "    kw = dict(suff=s, pre=pr[0], rho_j=pr[1]) if beta else {}\n"
        "    assert np.array_equal(lsa.sumcheck_round(tabs, **kw), o.fr_sumcheck_round(tabs, **kw)), (half, beta, suff)\n"
        "print('OK')\n"
    ) % (root, os.path.join(root, "tests"))
    env = dict(os.environ, LSA_SC3=kernel)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("d", [1, 2, 3, 7, 8, 9, 13, 16])
def test_eq_table_both_variants(lsa, d):
    """DPBeta::compute_eq_tbl (mle.h:93-105).  Variant 0 = the reference's loop as written (restated literally in the
    oracle: the doubling step reads dst[p >> 1]); variant 1 = the eq monomials, whose dot product with any v is
    evalMLE(v, r) (polytools.h:207-234 builds the same table)."""
    r, _ = o.random_scalars(d, seed=900 + d)
    assert np.array_equal(lsa.fr_eq_table(r, 0), o.fr_eq_table(r))
    eq1 = lsa.fr_eq_table(r, 1)
    v, _ = o.random_scalars(1 << d, seed=950 + d)
    R = o.R
    rinv = pow(o.MONT, -1, R)
    fr_int = lambda x: o.limbs_to_int(x) * rinv % R
    assert o.fr_dot(v, eq1) == fr_int(o.fr_eval_mle(v, r))
    if d <= 8:
        ri = [fr_int(x) for x in r]
        want = []
        for p in range(1 << d):
            acc = 1
            for j in range(d):
                acc = acc * (ri[j] if (p >> j) & 1 else 1 - ri[j]) % R
            want.append(acc)
        assert [fr_int(x) for x in eq1] == want


def test_eq_table_rejects_bad_arguments(lsa):
    r, _ = o.random_scalars(3, seed=1)
    with pytest.raises(lsa.LsaError):
        lsa.fr_eq_table(r, 2)                      # variants are 0 (the reference's loop) and 1 (the eq monomials)
    with pytest.raises(lsa.LsaError):
        lsa.fr_eq_table(r[:0], 0)                  # the reference's loop writes dst[1]: d >= 1


def test_sumcheck_prover_inner_loop_on_device(lsa):
    """The per-round device work of CPSumcheck::prove at d = 8 (two MLE tables + the beta suffix
    table, all resident): round polynomial, then pushRandomness on every table and the suffix
    update -- against the same sequence in the oracle."""
    import torch
    d = 8
    N = 1 << d
    a, _ = o.random_scalars(N, seed=41)
    b, _ = o.random_scalars(N, seed=42)
    suff, _ = o.random_scalars(N // 2, seed=43)
    rs, _ = o.random_scalars(d, seed=44)
    ks, _ = o.random_scalars(d, seed=45)
    pre, _ = o.random_scalars(d, seed=46)
    rho, _ = o.random_scalars(d, seed=47)
    to_dev = lambda x: torch.from_numpy(x.view(np.int64).copy()).to("cuda:0")
    d_a, d_b, d_s, d_r = to_dev(a), to_dev(b), to_dev(suff), to_dev(rs)
    torch.cuda.synchronize()
    ha, hb, hs = a.copy(), b.copy(), suff.copy()
    for j in range(d):
        half = 1 << (d - j - 1)
        use_suff = j + 1 <= d - 1
        got = lsa.sumcheck_round([d_a[:2 * half], d_b[:2 * half]], suff=d_s[:half] if use_suff else None, pre=pre[j], rho_j=rho[j])
        want = o.fr_sumcheck_round([ha[:2 * half], hb[:2 * half]], suff=hs[:half] if use_suff else None, pre=pre[j], rho_j=rho[j])
        assert np.array_equal(got, want), j
        for dv, hv in ((d_a, ha), (d_b, hb)):
            lsa._check(lsa.lib().lsa_fr_fold(dv.data_ptr(), half, d_r[j].data_ptr(), dv.data_ptr(), 1))
            hv[:half] = o.fr_push_randomness(hv[:2 * half], rs[j])
        if half >= 2:
            lsa._check(lsa.lib().lsa_fr_scale_upper(d_s.data_ptr(), half // 2, lsa._host_ptr(ks[j]), d_s.data_ptr(), 1))
            hs[:half // 2] = o.fr_scale_upper(hs[:half], ks[j])
    lsa.synchronize()
    assert np.array_equal(d_a[:1].cpu().numpy().view(np.uint64), ha[:1])
    assert np.array_equal(d_b[:1].cpu().numpy().view(np.uint64), hb[:1])


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 9, 10, 11, 12, 13, 15, 16, 17, 18, 20])
def test_ntt_vs_oracle(lsa, log_n):
    """libfqfft FFT / iFFT / cosetFFT / icosetFFT over Fr, byte for byte against the restated
    _basic_radix2_FFT: one pass (up to the 1024-element LDS tile), two passes (up to 2^16), three passes."""
    n = 1 << log_n
    a, _ = o.random_scalars(n, seed=3000 + log_n)
    w = o.fr_mont(o.fr_root_of_unity(log_n))
    g = o.fr_mont(o.FR_GENERATOR)
    for inverse in (False, True):
        for coset in (None, g):
            got = lsa.fr_ntt(a, w, inverse=inverse, coset=coset)
            want = o.fr_domain_transform(a, w, inverse=inverse, coset=coset) if log_n else a
            assert np.array_equal(got, want), (inverse, coset is not None)


@pytest.mark.parametrize("log_n", [10, 13, 16, 19])
def test_ntt_extreme_values(lsa, log_n):
    """Inputs that push the lazy sums of the butterflies to their bounds (csrc/ntt_core.h: the first stage of a stage pair adds
    without carries, values grow by up to 6r per pair): every element r - 1, alternating 0 / r - 1, blocks of r - 1 and of
    small values -- all four modes against the oracle."""
    n = 1 << log_n
    top = o.fr_mont(o.R - 1)
    rng = np.random.default_rng(log_n)
    pats = []
    a = np.tile(top, (n, 1)); pats.append(a)
    b = np.zeros((n, 4), dtype=np.uint64); b[::2] = top; pats.append(b)
    c = np.tile(top, (n, 1)); c[n // 2:] = o.fr_mont(1); c[rng.integers(0, n, 8)] = o.fr_mont(2); pats.append(c)
    d = rng.integers(0, 1 << 64, (n, 4), dtype=np.uint64); d[:, 3] = np.uint64(0x30644E72E131A028)      # top limb just below r's: values close to r
    pats.append(d)
    w = o.fr_mont(o.fr_root_of_unity(log_n))
    g = o.fr_mont(o.FR_GENERATOR)
    for k, a in enumerate(pats):
        for inverse, coset in ((False, None), (True, g)) if k else ((False, None), (True, None), (False, g), (True, g)):
            got = lsa.fr_ntt(a, w, inverse=inverse, coset=coset)
            assert np.array_equal(got, o.fr_domain_transform(a, w, inverse=inverse, coset=coset)), (k, inverse, coset is not None)


def test_ntt_round_trip_full_size_on_device(lsa):
    """n = 2^20 on a device-resident vector: icosetFFT(cosetFFT(a)) == a, and the transform of a
    delta at position 1 is the geometric sequence omega^k (checked at sampled positions)."""
    import torch
    log_n = 20
    n = 1 << log_n
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    d_a = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda:0", generator=gen)
    d_a[:, 3] &= (1 << 60) - 1
    keep = d_a.clone()
    wi = o.fr_root_of_unity(log_n)
    w, g = o.fr_mont(wi), o.fr_mont(o.FR_GENERATOR)
    lsa.fr_ntt(d_a, w, coset=g)
    assert not torch.equal(d_a, keep)
    lsa.fr_ntt(d_a, w, inverse=True, coset=g)
    lsa.synchronize()
    assert torch.equal(d_a, keep)
    delta = torch.zeros((n, 4), dtype=torch.int64, device="cuda:0")
    delta[1] = torch.from_numpy(o.fr_mont(1).view(np.int64))
    lsa.fr_ntt(delta, w)
    lsa.synchronize()
    host = delta.cpu().numpy().view(np.uint64)
    for k in (0, 1, 2, 1023, 1024, 65537, n - 1):
        assert np.array_equal(host[k], o.fr_mont(pow(wi, k, o.R))), k


def test_ntt_domains_alternate_and_tables_are_reused(lsa):
    """The per-domain twiddle tables are cached (four domains, four coset generators): transforms over five domains in
    turn, forward and inverse, with two coset generators, keep giving the oracle's bytes when their tables have been
    evicted and rebuilt in between; a host buffer and a device buffer give the same result."""
    import torch
    g5, g7 = o.fr_mont(o.FR_GENERATOR), o.fr_mont(7)
    data = {}
    for log_n in (6, 11, 12, 14, 17):
        a, _ = o.random_scalars(1 << log_n, seed=4000 + log_n)
        data[log_n] = (a, o.fr_mont(o.fr_root_of_unity(log_n)))
    for rep in range(2):
        for log_n, (a, w) in data.items():
            for inverse in (False, True):
                for coset in (None, g5, g7):
                    got = lsa.fr_ntt(a, w, inverse=inverse, coset=coset)
                    assert np.array_equal(got, o.fr_domain_transform(a, w, inverse=inverse, coset=coset)), (rep, log_n, inverse)
    a, w = data[14]
    d_a = torch.from_numpy(a.view(np.int64)).to("cuda:0")
    lsa.fr_ntt(d_a, w, coset=g7)
    lsa.synchronize()
    assert np.array_equal(d_a.cpu().numpy().view(np.uint64), lsa.fr_ntt(a, w, coset=g7))


@pytest.mark.parametrize("big_log,small_log", [(1, 0), (2, 1), (3, 0), (5, 2), (9, 8), (10, 0), (10, 9), (11, 3), (12, 10), (14, 13), (16, 5), (17, 16), (18, 12)])
def test_step_domain_vs_oracle(lsa, big_log, small_log):
    """libfqfft step_radix2_domain (the domain of every size that is not a power of two) FFT / iFFT / cosetFFT / icosetFFT,
    byte for byte against its restatement in the oracle: sub-transforms of one, two and three passes, small_m = 1, and
    small_m = big_m / 2."""
    m = (1 << big_log) + (1 << small_log)
    a, _ = o.random_scalars(m, seed=5000 + 32 * big_log + small_log)
    w = o.fr_mont(o.fr_root_of_unity(big_log + 1))
    g = o.fr_mont(o.FR_GENERATOR)
    for inverse in (False, True):
        for coset in (None, g):
            got = lsa.fr_ntt_step(a, big_log, small_log, w, inverse=inverse, coset=coset)
            want = o.fr_step_domain_transform(a, big_log, small_log, w, inverse=inverse, coset=coset)
            assert np.array_equal(got, want), (inverse, coset is not None)


def test_step_domain_round_trip_and_spot_values_at_full_size_on_device(lsa):
    """m = 2^22 + 2^21 (three quarters of 2^23) on the device: icosetFFT(cosetFFT(a)) = a, and FFT's values at sampled domain
    points equal Horner's at omega^(2k) / omega sigma^j for a sparse polynomial."""
    import torch
    big_log, small_log = 22, 21
    big, small = 1 << big_log, 1 << small_log
    m = big + small
    a, _ = o.random_scalars(m, seed=77)
    w_int = o.fr_root_of_unity(big_log + 1)
    w, g = o.fr_mont(w_int), o.fr_mont(o.FR_GENERATOR)
    d_a = torch.from_numpy(a.view(np.int64).copy()).to("cuda:0")
    lsa.fr_ntt_step(d_a, big_log, small_log, w, coset=g)
    assert not np.array_equal(d_a.cpu().numpy().view(np.uint64), a)
    lsa.fr_ntt_step(d_a, big_log, small_log, w, inverse=True, coset=g)
    assert np.array_equal(d_a.cpu().numpy().view(np.uint64), a)
    # a(x) = 3 + 5 x^7 + 11 x^(big + 12345) + x^(m - 1)
    R = o.R
    terms = {0: 3, 7: 5, big + 12345: 11, m - 1: 1}
    sp = np.zeros((m, 4), dtype=np.uint64)
    for i, c in terms.items():
        sp[i] = o.fr_mont(c)
    got = lsa.fr_ntt_step(sp, big_log, small_log, w)
    sigma = pow(w_int, 2 * big // small, R)
    for idx in (0, 1, 2, big // 2 + 3, big - 1, big, big + 1, big + small // 3, m - 1):
        x = pow(w_int, 2 * idx, R) if idx < big else w_int * pow(sigma, idx - big, R) % R
        want = sum(c * pow(x, i, R) for i, c in terms.items()) % R
        assert np.array_equal(got[idx], o.fr_mont(want).reshape(4)), idx


def test_step_domain_arguments_are_checked(lsa):
    a, _ = o.random_scalars(3, seed=1)
    w = o.fr_mont(o.fr_root_of_unity(2))
    with pytest.raises(Exception, match="big_log"):
        lsa._check(lsa.lib().lsa_fr_ntt_step(lsa._host_ptr(a), 28, 0, lsa._host_ptr(w), 0, None, 0))  # omega would be a 2^29-th root of unity
    with pytest.raises(Exception, match="big_log"):
        lsa._check(lsa.lib().lsa_fr_ntt_step(lsa._host_ptr(a), 1, 1, lsa._host_ptr(w), 0, None, 0))   # small_m = big_m: a power of two

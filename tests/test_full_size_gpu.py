"""Full-size, result-checked GPU tests for every BASELINE.json configuration that fits one GPU
("full-size configs are not result-checked").  Every test checks a RESULT -- by the
known-discrete-log identity (an O(n) computation in Fr that touches no elliptic-curve code, done
on the host by the oracle's Fr arithmetic) or against oracle samples -- none is timing-only.

  (i)   G2 MSM, n = 2^20                                   (configs[2], G2 half of CPpoly commit)
  (ii)  n = 2^20 copies of the generator, G1 and G2, uniform scalars and u[i] = i
        (CommScheme's actual bases, src/prototools/commit.h:134-138; src/examples/hadamard.cc:130-135)
  (iii) CPpoly d = 20: witness recursion + the 39-MSM ladder on the device, every ladder result
        (src/gadgets/poly.h:55-86)
  (iv)  2^12 pairings with a planted relation: product == 1, plus sampled Miller values vs oracle
        (configs[4])
  (v)   CPlink prover shape N = 2^24 (n = N + 2 of a 2N + 2 CRS with N trailing infinities) on
        one GPU, and the same input through the two-rank sharded path at 2^23 per rank
        (configs[3] minus the other seven GPUs; src/gadgets/subspace.cc:78-85)
"""
import os

import numpy as np
import pytest

import oracle_lib as o
from legosnark_amd import curve, sharded, synth

pytestmark = pytest.mark.gpu
R = o.R


def canon(group, pt):
    return o.g1_canonical_affine(pt) if group == "g1" else o.g2_canonical_affine(pt)


def k_times_gen(group, k):
    mul = o.g1_mul if group == "g1" else o.g2_mul
    return canon(group, mul(o.generator(group), o.fr_mont(k % R)))


def to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")


def to_host(t):
    return t.cpu().numpy().view(np.uint64)


def test_g2_msm_2pow20_by_identity(lsa):
    n = 1 << 20
    rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0x62)
    a, b = rng.fr_int(), rng.fr_int()
    x = synth.arith_fr_mont(a, b, n)
    s = rng.uniform_fr(n)
    bases = lsa.batch_exp("g2", curve.generator("g2"), to_dev(x))     # un-normalised Jacobian on the device
    B = lsa.Bases("g2", bases, on_device=True)
    del bases
    got = B.msm(to_dev(s))
    assert canon("g2", got) == k_times_gen("g2", o.fr_dot(s, x))
    # a second run over a sub-range with small scalars (31-bit, src/examples/matrixsc.cc:50-53)
    m = n - 12345
    small = synth.small_fr_mont(rng.u64(m) >> np.uint64(33))
    got = B.msm(to_dev(small), n=m, first=100)
    assert canon("g2", got) == k_times_gen("g2", o.fr_dot(small, x[100:100 + m]))
    B.close()


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_generator_copies_2pow20(lsa, group):
    """The bases CommScheme::keygen really produces: n copies of the generator.  Every first
    collision in a bucket is P + P, every later one the same point again."""
    import torch
    n = 1 << 20
    g = curve.generator(group)
    bases = torch.from_numpy(g.view(np.int64)).to("cuda:0").repeat(n, 1).contiguous()
    B = lsa.Bases(group, bases, on_device=True)
    del bases
    rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0xC0)
    s = rng.uniform_fr(n)
    assert canon(group, B.msm(to_dev(s))) == k_times_gen(group, synth.fr_sum_mont(s))
    u = synth.small_fr_mont(np.arange(n))                              # u[i] = i
    assert canon(group, B.msm(to_dev(u))) == k_times_gen(group, n * (n - 1) // 2)
    u2 = synth.small_fr_mont(np.arange(n, dtype=np.uint64) ** 2)       # u[i]^2
    assert canon(group, B.msm(to_dev(u2))) == k_times_gen(group, (n - 1) * n * (2 * n - 1) // 6)
    B.close()


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_pipelined_calls_of_2pow18_pairs_agree_with_blocking_ones(lsa, group):
    """Queued calls above the compact pipeline's range (2^17 pairs) reduce their 2^19 buckets with a lane-private first level
    (k_reduce2_lane, msm.hip) where a blocking call uses the quad level and the bit trees: both
    tails on the same inputs, and the known-discrete-log identity on top."""
    import torch
    n = 1 << 18
    rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0x17)
    a, b = rng.fr_int(), rng.fr_int()
    x = synth.arith_fr_mont(a, b, n)
    bases = lsa.batch_exp(group, curve.generator(group), to_dev(x))
    B = lsa.Bases(group, bases, on_device=True)
    del bases
    sizes = [n, n - 4097, (1 << 17) + 1, n - 1]
    scs = [rng.uniform_fr(m) for m in sizes]
    dev = [to_dev(s) for s in scs]
    words = 12 if group == "g1" else 24
    outs = torch.zeros((2 * len(sizes), words), dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    for rep in range(2):
        for i, m in enumerate(sizes):
            B.msm_async(dev[i], outs[rep * len(sizes) + i], n=m)
    lsa.synchronize()
    got = to_host(outs)
    for i, m in enumerate(sizes):
        want = canon(group, B.msm(dev[i], n=m))
        assert canon(group, got[i]) == want, (i, m)
        assert canon(group, got[len(sizes) + i]) == want, (i, m)
        assert want == k_times_gen(group, o.fr_dot(scs[i], x[:m])), (i, m)
    B.close()


def test_cppoly_d20_commit_and_ladder(lsa):
    import torch
    d = 20
    n = 1 << d
    rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0x20)
    v = rng.uniform_fr(n)
    r = rng.uniform_fr(d)
    g1 = torch.from_numpy(curve.generator("g1").view(np.int64)).to("cuda:0").repeat(n, 1).contiguous()
    g2 = torch.from_numpy(curve.generator("g2").view(np.int64)).to("cuda:0").repeat(n, 1).contiguous()
    B1 = lsa.Bases("g1", g1, on_device=True)
    B2 = lsa.Bases("g2", g2, on_device=True)
    del g1, g2
    d_v, d_r = to_dev(v), to_dev(r)
    # commitPoly (poly.h:30-32 -> commit.h:154-155)
    vsum = synth.fr_sum_mont(v)
    assert canon("g1", B1.msm(d_v)) == k_times_gen("g1", vsum)
    assert canon("g2", B2.msm(d_v)) == k_times_gen("g2", vsum)
    c1 = torch.zeros(12, dtype=torch.int64, device="cuda:0")
    c2 = torch.zeros(24, dtype=torch.int64, device="cuda:0")
    lsa.commit_async(B1, B2, d_v, c1, c2)                        # the pair with one shared scalar sort
    lsa.synchronize()
    assert canon("g1", to_host(c1)) == k_times_gen("g1", vsum)
    assert canon("g2", to_host(c2)) == k_times_gen("g2", vsum)
    # prove: witness recursion on the device, bit-exact vs the oracle's restatement of poly.h:55-67
    d_w = lsa.cppoly_witness(d_v, d_r)
    lsa.synchronize()
    w = to_host(d_w)
    assert np.array_equal(w, o.fr_cppoly_witness(v, r))
    # ladder (poly.h:77-86): witness[i] over w[start .. start + 2^(d-1-i)), again for i >= 1
    outs = torch.zeros((2 * d, 12), dtype=torch.int64, device="cuda:0")
    start, calls = 0, []
    for i in range(d):
        m = 1 << (d - 1 - i)
        B1.msm_async(d_w[start:start + m], outs[2 * i], n=m)
        calls.append((2 * i, start, m))
        if i:
            B1.msm_async(d_w[start:start + m], outs[2 * i + 1], n=m)
            calls.append((2 * i + 1, start, m))
        start += m
    lsa.synchronize()
    res = to_host(outs)
    assert len(calls) == 39
    for slot, st, m in calls:
        assert canon("g1", res[slot]) == k_times_gen("g1", synth.fr_sum_mont(w[st:st + m])), (slot, st, m)
    # the same ladder as an integrator would issue it: rungs of up to 2^15 pairs in ONE segmented
    # call over consecutive slices of w (witnessa[i] is the same sum as witness[i]), the longer
    # rungs one call each
    if B1.has_table():
        starts = np.concatenate([[0], np.cumsum([1 << (d - 1 - i) for i in range(d)])])
        first_small = next(i for i in range(d) if (1 << (d - 1 - i)) <= (1 << 15))
        seg_outs = torch.zeros((d - first_small, 12), dtype=torch.int64, device="cuda:0")
        big_outs = torch.zeros((first_small, 12), dtype=torch.int64, device="cuda:0")
        for i in range(first_small):
            B1.msm_async(d_w[int(starts[i]):int(starts[i + 1])], big_outs[i], n=1 << (d - 1 - i))
        B1.msm_segments_async(d_w, starts[first_small:], seg_outs)
        lsa.synchronize()
        both = np.concatenate([to_host(big_outs), to_host(seg_outs)])
        for i in range(d):
            assert canon("g1", both[i]) == canon("g1", res[2 * i]), i
    B1.close(); B2.close()


def test_pairing_product_2pow12_planted(lsa):
    n = 1 << 12
    rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0x12)
    al = rng.uniform_fr(n)
    be = rng.uniform_fr(n)
    # plant sum_j alpha_j * beta_j == 0 through the last beta
    rinv = pow(o.MONT, -1, R)
    a_last = synth.limbs_to_int(al[-1]) * rinv % R
    partial = o.fr_dot(al[:-1], be[:-1])
    be[-1] = o.fr_mont((-partial) * pow(a_last, -1, R) % R)
    assert o.fr_dot(al, be) == 0
    ps = lsa.batch_exp("g1", curve.generator("g1"), al)                # host buffers, un-normalised Jacobian
    qs = lsa.batch_exp("g2", curve.generator("g2"), be)
    assert np.array_equal(lsa.pairing_product(ps, qs), o.fq12_one())
    # an unplanted batch must not be one
    qs2 = qs.copy()
    qs2[7] = qs[8]
    assert not np.array_equal(lsa.pairing_product(ps, qs2), o.fq12_one())
    # sampled Miller values of the 2^12 batch vs the oracle, byte for byte
    f = lsa.miller_loop(ps, qs)
    idx = [0, 1, 511, 2047, 2048, 3000, 4094, 4095]
    want = o.miller_loop_batch(ps[idx], qs[idx])
    assert np.array_equal(f[idx], want)
    # product of all Miller values == the library's product entry point; final exp of it == 1
    prod = lsa.fq12_product(f)
    assert np.array_equal(prod, lsa.miller_loop_product(ps, qs))
    assert np.array_equal(lsa.final_exponentiation(prod)[0], o.fq12_one())


def test_cplink_prover_shape_2pow24_and_two_rank_split(lsa):
    import torch
    N = 1 << 24
    n = N + 2
    rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0x24)
    a, b = rng.fr_int(), rng.fr_int()
    x = synth.arith_fr_mont(a, b, n)
    # CRS P: N + 2 meaningful points, then N points at infinity (cplink.cc:21-22 + empty columns)
    crs = torch.zeros((2 * N + 2, 12), dtype=torch.int64, device="cuda:0")
    lsa.batch_exp("g1", curve.generator("g1"), to_dev(x), out=crs[:n])
    crs[n:, 4:8] = torch.from_numpy(curve.fq_mont(1).view(np.int64)).to("cuda:0")       # libff zero() = (0, 1, 0)
    # witness w = (0, rF, u) (cplink.cc:107-108, commit.h:152): raw 252-bit limbs are valid
    # Montgomery representatives (< r), generated on the device
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(2024)
    d_w = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda:0", generator=gen)
    d_w[:, 3] &= (1 << 60) - 1
    d_w[0] = 0
    d_w = d_w.contiguous()
    torch.cuda.synchronize()
    w = to_host(d_w)
    want = k_times_gen("g1", o.fr_dot(w, x))
    B = lsa.Bases("g1", crs, on_device=True)
    assert B.n == 2 * N + 2
    got = B.msm(d_w, n=n)                                              # min(|P|, |w|) = N + 2 (globl.h:66)
    assert canon("g1", got) == want
    # two ranks, libff chunk split of [0, n): rank 1's partial computed beforehand on its range,
    # rank 0 runs the sharded step (side stream, gather buffer, fold) with a stand-in collective
    lo0, hi0 = sharded.shard_range(n, 2, 0)
    lo1, hi1 = sharded.shard_range(n, 2, 1)
    assert (lo0, hi0, lo1, hi1) == (0, n // 2, n // 2, n) and hi0 - lo0 >= 1 << 23
    peer = B.msm(d_w[lo1:hi1], n=hi1 - lo1, first=lo1)
    # short MSMs over the same handle: 24 copies and 11-bit digits here (tables of 6 * 2^20 points and more: 1024 buckets,
    # eleven weight bits in the compact pipeline of msm_compact.hip), prefixes and an offset range
    if B.has_table():
        for first, m in ((0, 1), (0, 300), (0, 4096), (0, 1 << 16), (12345, 5000)):
            got_s = B.msm(d_w[first:first + m], n=m, first=first)
            assert canon("g1", got_s) == k_times_gen("g1", o.fr_dot(w[first:first + m], x[first:first + m])), (first, m)
    B.close()
    d_peer = to_dev(peer)

    class FakeDist:
        def all_gather_into_tensor(self, out, inp):
            out.view(2, 12)[0].copy_(inp)
            out.view(2, 12)[1].copy_(d_peer)

    B0 = lsa.Bases("g1", crs[lo0:hi0], on_device=True)
    del crs
    job = sharded.make_gpu_sharded(lsa, "g1", B0, 2, 0, dist=FakeDist())
    res = job.run(d_w[lo0:hi0])
    assert canon("g1", job.result_host(res)) == want
    # (the handle of 2^23 + 1 points carries 24 copies -- the full one above is too large for 30-bit entries)
    assert B0.has_table() == (os.environ.get("LSA_PRECOMPUTE", "1")[:1] != "0")
    if B0.has_table():
        for first, m in ((0, 1), (0, 300), (0, 4096), (0, 1 << 16), (12345, 5000)):
            got_s = B0.msm(d_w[first:first + m], n=m, first=first)
            assert canon("g1", got_s) == k_times_gen("g1", o.fr_dot(w[first:first + m], x[first:first + m])), (first, m)
    B0.close()


@pytest.mark.parametrize("log_n", [22, 25])
def test_ntt_beyond_2pow20_round_trip_and_delta(lsa, log_n):
    """The three-pass NTT at sizes the oracle is too slow for (2^22: passes of 8 + 7 + 7 stages; 2^25: 9 + 8 + 8, tiles of
    two columns): icosetFFT(cosetFFT(a)) == a, iFFT(FFT(a)) == a, the transform of the delta at position 1 is omega^k at
    sampled positions, and linearity FFT(a) + FFT(delta) == FFT(a + delta) at those positions."""
    import torch
    n = 1 << log_n
    gen = torch.Generator(device="cuda:0").manual_seed(100 + log_n)
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
    lsa.fr_ntt(d_a, w)
    lsa.synchronize()
    fa = d_a.clone()
    lsa.fr_ntt(d_a, w, inverse=True)
    lsa.synchronize()
    assert torch.equal(d_a, keep)
    delta = torch.zeros((n, 4), dtype=torch.int64, device="cuda:0")
    delta[1] = torch.from_numpy(o.fr_mont(1).view(np.int64))
    both = keep.clone()
    a1 = o.limbs_to_int(keep[1].cpu().numpy().view(np.uint64)) * pow(o.MONT, -1, o.R) % o.R
    both[1] = torch.from_numpy(o.fr_mont((a1 + 1) % o.R).view(np.int64))
    lsa.fr_ntt(delta, w)
    lsa.fr_ntt(both, w)
    lsa.synchronize()
    hd, hf, hb = (t.cpu().numpy().view(np.uint64) for t in (delta, fa, both))
    minv = pow(o.MONT, -1, o.R)
    for k in (0, 1, 2, 1023, 1024, 65537, (1 << 21) + 5, n - 1):
        assert np.array_equal(hd[k], o.fr_mont(pow(wi, k, o.R))), k
        s = (o.limbs_to_int(hf[k]) + o.limbs_to_int(hd[k])) * minv % o.R
        assert o.limbs_to_int(hb[k]) * minv % o.R == s, k

#!/usr/bin/env python3
"""tests/fuzz_parity.py [seconds] [seed] [--big] -- randomised differential run of the C-ABI against the oracle (GPU box).

The parametrised tests under tests/ pin chosen shapes; this draws shapes and contents at random for a time budget and
compares every result with the oracle's restatement of the reference algorithm: MSMs (G1 / G2; bases with random Z,
points at infinity, repeated bases; scalars of 254 / 128 / 64 / 31 / 16 bits with 0, 1 and r - 1 planted), batch_exp,
batched scalar multiplication, pairing products with conjugated terms and several segments, final exponentiations of
arbitrary Fq12 elements, the radix-2 and the step
NTT in all four modes, the witness recursion, evalMLE, pushRandomness and the sumcheck round polynomial.  One JSON line
per operation kind at the end (cases, failures, the seeds of failures); exit status 1 on any mismatch.  A script, not a
pytest module (its run time is a budget, not a property): the oracle is the checker here as in the tests beside it."""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import legosnark_amd as lsa  # noqa: E402
import oracle_lib as o  # noqa: E402

R, P = o.R, o.P


def canon(group, pt):
    return o.g1_canonical_affine(pt) if group == "g1" else o.g2_canonical_affine(pt)


def scalars(rng, n):
    bits = rng.choice([254, 254, 254, 128, 64, 31, 16])
    sc, ints = o.random_scalars(n, seed=rng.randrange(1 << 30), bits=bits)
    for _ in range(min(n, rng.choice([0, 0, 1, 3, n // 3 + 1]))):
        i = rng.randrange(n)
        sc[i] = o.fr_mont(rng.choice([0, 1, R - 1, 2]))
    return sc


def bases(rng, group, n):
    a, b = rng.randrange(1, R), rng.randrange(R)
    pts = o.arith_bases(group, a, b, n)                     # un-normalised Jacobian
    w = pts.shape[1]
    for _ in range(min(n, rng.choice([0, 0, 1, 2, n // 4 + 1]))):
        i = rng.randrange(n)
        if rng.random() < 0.5:
            pts[i, 2 * w // 3:] = 0                         # Z = 0: the point at infinity
        else:
            pts[i] = pts[rng.randrange(n)]                  # a repeated base
    return pts


BIG = "--big" in sys.argv            # larger MSMs (the general pipeline beyond the compact one's range, the CRS cache's prefix tables)


def case_msm(rng):
    group = rng.choice(["g1", "g1", "g2"])
    top = (19 if group == "g1" else 16) if BIG else (16 if group == "g1" else 13)
    n = rng.choice([rng.randrange(0, 40), rng.randrange(40, 3000), 1 << rng.randrange(5, top), (1 << rng.randrange(5, top)) + rng.randrange(-3, 4)])
    n = max(n, 0)
    pts, sc = bases(rng, group, n) if n else np.zeros((0, 12 if group == "g1" else 24), dtype=np.uint64), scalars(rng, n) if n else np.zeros((0, 4), dtype=np.uint64)
    want = canon(group, o.multi_exp(group, pts, sc, chunks=1, mode="mixed")) if n else None
    got = canon(group, lsa.msm(group, pts, sc))
    return got == want, "%s n=%d" % (group, n)


def case_batch_exp(rng):
    group = rng.choice(["g1", "g2"])
    n = rng.choice([1, rng.randrange(1, 200), rng.randrange(200, 3000)])
    base = o.arith_bases(group, rng.randrange(1, R), 0, 1)[0]       # an un-normalised Jacobian point
    sc = scalars(rng, n)
    want = o.batch_exp(group, base, sc)
    got = lsa.batch_exp(group, base, sc)
    ok = all(canon(group, got[i]) == canon(group, want[i]) for i in range(n))
    return ok, "%s n=%d" % (group, n)


def case_smul(rng):
    n = rng.choice([1, rng.randrange(1, 100), rng.randrange(100, 1500)])
    pts, sc = bases(rng, "g1", n), scalars(rng, n)
    want = o.g1_mul_batch(pts, sc)
    got = lsa.scalar_mul_batch(pts, sc)
    return all(canon("g1", got[i]) == canon("g1", want[i]) for i in range(n)), "n=%d" % n


def case_pairing(rng):
    nseg = rng.choice([1, 1, 2, 3])
    sizes = [rng.randrange(1, 6) for _ in range(nseg)]
    n = sum(sizes)
    g1, g2 = bases(rng, "g1", n), bases(rng, "g2", n)
    flags = np.array([rng.randrange(2) for _ in range(n)], dtype=np.uint8)
    final = rng.random() < 0.7
    off = np.cumsum([0] + sizes).astype(np.uint64)
    fs = o.miller_loop_batch(g1, g2)
    want = []
    for j in range(nseg):
        acc = o.fq12_one()
        for i in range(int(off[j]), int(off[j + 1])):
            acc = o.fq12_mul(acc, o.fq12_unitary_inverse(fs[i]) if flags[i] else fs[i])
        want.append(o.final_exponentiation(acc) if final else acc)
    got = lsa.pairing_terms(g1, off, g2=g2, flags=flags, final_exp=final)
    return all(np.array_equal(got[j], want[j]) for j in range(nseg)), "segments=%s final=%d" % (sizes, final)


def case_ntt(rng):
    log_n = rng.randrange(0, 15)
    a, _ = o.random_scalars(1 << log_n, seed=rng.randrange(1 << 30))
    w = o.fr_mont(o.fr_root_of_unity(log_n))
    inverse, coset = rng.random() < 0.5, (o.fr_mont(rng.randrange(2, R)) if rng.random() < 0.5 else None)
    got = lsa.fr_ntt(a, w, inverse=inverse, coset=coset)
    want = o.fr_domain_transform(a, w, inverse=inverse, coset=coset) if log_n else a
    return np.array_equal(got, want), "log_n=%d inverse=%d coset=%d" % (log_n, inverse, coset is not None)


def case_ntt_step(rng):
    big = rng.randrange(1, 14)
    small = rng.randrange(0, big)
    m = (1 << big) + (1 << small)
    a, _ = o.random_scalars(m, seed=rng.randrange(1 << 30))
    w = o.fr_mont(o.fr_root_of_unity(big + 1))
    inverse, coset = rng.random() < 0.5, (o.fr_mont(rng.randrange(2, R)) if rng.random() < 0.5 else None)
    got = lsa.fr_ntt_step(a, big, small, w, inverse=inverse, coset=coset)
    want = o.fr_step_domain_transform(a, big, small, w, inverse=inverse, coset=coset)
    return np.array_equal(got, want), "2^%d+2^%d inverse=%d coset=%d" % (big, small, inverse, coset is not None)


def case_fold(rng):
    d = rng.choice([rng.randrange(0, 15), rng.randrange(0, 15), rng.randrange(15, 18)])      # (evalMLE takes another route from d = 16 on)
    v, _ = o.random_scalars(1 << d, seed=rng.randrange(1 << 30))
    r, _ = o.random_scalars(max(d, 1), seed=rng.randrange(1 << 30))
    r = r[:d]
    for i in range(d):
        if rng.random() < 0.1:
            r[i] = o.fr_mont(rng.choice([0, 1, R - 1]))
    kind = rng.choice(["witness", "eval_mle", "push"])
    if kind == "witness":
        return np.array_equal(lsa.cppoly_witness(v, r), o.fr_cppoly_witness(v, r)), "witness d=%d" % d
    if kind == "eval_mle":
        return np.array_equal(lsa.eval_mle(v, r), o.fr_eval_mle(v, r)), "eval_mle d=%d" % d
    if d == 0:
        return True, "push d=0"
    return np.array_equal(lsa.fr_fold(v, r[0]), o.fr_push_randomness(v, r[0])), "push d=%d" % d


def case_sumcheck(rng):
    m = rng.randrange(1, 5)
    half = rng.choice([1, rng.randrange(1, 50), rng.randrange(50, 6000)])
    tabs = [o.random_scalars(2 * half, seed=rng.randrange(1 << 30))[0] for _ in range(m)]
    for t in tabs:                                             # the values at the edges of the kernels' lazy bounds
        for _ in range(rng.choice([0, 0, 2, half])):
            t[rng.randrange(2 * half)] = o.fr_mont(rng.choice([0, 1, R - 1, R - 2, (R - 1) // 2]))
    beta = rng.random() < 0.6
    suff = o.random_scalars(half, seed=rng.randrange(1 << 30))[0] if (beta and rng.random() < 0.8) else None
    pre = o.random_scalars(1, seed=rng.randrange(1 << 30))[0][0] if beta else None
    rho = o.random_scalars(1, seed=rng.randrange(1 << 30))[0][0] if beta else None
    got = lsa.sumcheck_round(tabs, suff=suff, pre=pre, rho_j=rho)
    want = o.fr_sumcheck_round(tabs, suff=suff, pre=pre, rho_j=rho)
    return np.array_equal(got, want), "m=%d half=%d beta=%d suff=%d" % (m, half, beta, suff is not None)


def case_eq_table(rng):
    d = rng.randrange(1, 15)
    r, _ = o.random_scalars(d, seed=rng.randrange(1 << 30))
    for i in range(d):
        if rng.random() < 0.15:
            r[i] = o.fr_mont(rng.choice([0, 1, R - 1]))
    ok = np.array_equal(lsa.fr_eq_table(r, 0), o.fr_eq_table(r))          # DPBeta::compute_eq_tbl as the reference's loop computes it
    v, _ = o.random_scalars(1 << d, seed=rng.randrange(1 << 30))
    rinv = pow(o.MONT, -1, R)
    ok = ok and o.fr_dot(v, lsa.fr_eq_table(r, 1)) == o.limbs_to_int(o.fr_eval_mle(v, r)) * rinv % R   # the eq monomials: <v, eq> = evalMLE(v, r)
    return ok, "eq_table d=%d" % d


def case_final_exp(rng):
    """lsa_final_exponentiation on arbitrary Fq12 elements (not Miller values): random components, random subsets of them zero"""
    n = rng.choice([1, 1, 2, 5, 17])
    fs = np.zeros((n, 48), dtype=np.uint64)
    for i in range(n):
        mask = rng.choice([0xfff, 0xfff, 0xfff, rng.randrange(1, 0x1000)])
        for c in range(12):
            if (mask >> c) & 1:
                fs[i, 4 * c:4 * c + 4] = o.fq_mont(rng.randrange(P))
    got = lsa.final_exponentiation(fs)
    ok = all(np.array_equal(got[i].reshape(-1), o.final_exponentiation(fs[i]).reshape(-1)) for i in range(n))
    return ok, "final_exp n=%d" % n


CASES = [("final_exp", case_final_exp, 1), ("eq_table", case_eq_table, 1), ("msm", case_msm, 5), ("batch_exp", case_batch_exp, 2), ("scalar_mul_batch", case_smul, 2), ("pairing_terms", case_pairing, 3),
         ("ntt", case_ntt, 2), ("ntt_step", case_ntt_step, 2), ("fr_fold", case_fold, 3), ("sumcheck_round", case_sumcheck, 2)]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    budget = float(args[0]) if len(args) > 0 else 300.0
    seed0 = int(args[1]) if len(args) > 1 else 20261003
    lsa.init(0)
    stats = {name: {"cases": 0, "failures": 0, "failed": []} for name, _, _ in CASES}
    pick = [c for c in CASES for _ in range(c[2])]
    master = random.Random(seed0)
    t0 = time.time()
    k = 0
    while time.time() - t0 < budget:
        name, fn, _ = master.choice(pick)
        seed = master.randrange(1 << 40)
        try:
            ok, what = fn(random.Random(seed))
        except Exception as e:                                  # an error is a failure too, with its seed
            ok, what = False, "raised %r" % (e,)
        st = stats[name]
        st["cases"] += 1
        if not ok:
            st["failures"] += 1
            st["failed"].append({"seed": seed, "what": what})
            print("MISMATCH", name, seed, what, flush=True)
        k += 1
    total_fail = sum(s["failures"] for s in stats.values())
    for name, st in stats.items():
        print(json.dumps({"op": name, **st}))
    print(json.dumps({"fuzz_parity": {"seconds": round(time.time() - t0, 1), "seed": seed0, "cases": k, "failures": total_fail}}))
    return 1 if total_fail else 0


if __name__ == "__main__":
    sys.exit(main())

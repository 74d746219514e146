#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs that are parity-test cases rather
than the headline bench line (run on the GPU box; prints one JSON object per config).
EVERY time printed belongs to a result that was checked in the same run -- by the
known-discrete-log identity (k*G recomputed by the fixed-base batch_exp kernel, a different
code path from the MSM) or, for pairings, by a planted product that must equal one.  A config
whose check fails prints {"config": ..., "error": "result check failed"} and no time.

  cplink_prover   SubspaceSnark::prove shape (subspace.cc:78-85): MSM over N+2 pairs, w[0]=0
  g2_msm          alt_bn128 G2 MSM
  cppoly          CPpoly d-variable commit + prove ladder (poly.h:30-32,76-88): G1+G2 MSM of
                  2^d, then G1 MSMs of 2^(d-1-i) (twice for i>=1), bases = copies of the generator
  pairing         batched Miller loops + one final exponentiation over 2^k pairs (planted)
  cphad_verify    the CPhad verifier's pairing shape at d: 9d+1 Miller loops, 3d+2 final exps
Inputs follow legosnark_amd/synth.py."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# Field multiplications of one libff-shape Miller loop + their share of a final exponentiation,
# counted from the formulas in csrc/miller.h / tower.h (Fq2 product = 3 Fq products, Fq2
# square = 2): doubling step 6 Fq2 sqr + 4 Fq2 mul (point) + Fq12 sqr (12 Fq2 mul) +
# mul_by_024 (13 Fq2 mul) + 2 Fq2-by-Fq scalings; addition step 2 sqr + 11 mul (point) +
# mul_by_024; 64 doublings, 23 additions + 2 Frobenius additions.
FQ_PER_FQ2_MUL, FQ_PER_FQ2_SQR = 3, 2
MILLER_DBL = 6 * FQ_PER_FQ2_SQR + (4 + 12 + 13) * FQ_PER_FQ2_MUL + 4
MILLER_ADD = 2 * FQ_PER_FQ2_SQR + (11 + 13) * FQ_PER_FQ2_MUL + 4
MILLER_FQ_MULS = 64 * MILLER_DBL + 25 * MILLER_ADD
FMUL_PEAK_G = 175.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--d", type=int, default=20)
    ap.add_argument("--log2pairs", type=int, default=12)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import legosnark_amd as lsa
    from legosnark_amd import curve, synth
    dev = torch.device("cuda:0")
    lsa.init(0)
    G1, G2 = curve.generator("g1"), curve.generator("g2")
    rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0xBC)

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)

    def host(t):
        return t.cpu().numpy().view(np.uint64)

    def affine(group, pts):
        w = 12 if group == "g1" else 24
        return lsa.normalize(group, np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, w))

    def k_times_gen(group, ks):
        sc = np.stack([curve.fr_mont(k) for k in ks])
        return affine(group, lsa.batch_exp(group, G1 if group == "g1" else G2, sc))

    def timed(fn, reps):
        fn(); lsa.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        lsa.synchronize(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def emit(cfg, ok, fields):
        if not ok:
            print(json.dumps({"config": cfg, "error": "result check failed -- no time reported"}), flush=True)
            return
        d = {"config": cfg, "result_checked": True}
        d.update(fields)
        print(json.dumps(d), flush=True)

    want = set(args.only.split(",")) if args.only else None

    def on(name):
        return want is None or name in want

    if on("cplink_prover"):
        N = 1 << args.log2n
        a, b = rng.fr_int(), rng.fr_int()
        x = synth.arith_fr_mont(a, b, N + 2)
        w = rng.uniform_fr(N + 2)
        w[0] = 0
        P = lsa.Bases("g1", lsa.batch_exp("g1", G1, to_dev(x)), on_device=True)
        d_w = to_dev(w)
        out = torch.zeros(12, dtype=torch.int64, device=dev)
        ms = timed(lambda: P.msm_async(d_w, out), args.reps)
        ok = np.array_equal(affine("g1", host(out))[0], k_times_gen("g1", [synth.fr_dot_mont(w, x)])[0])
        emit("CPlink prover (SubspaceSnark::prove MSM), N=2^%d" % args.log2n, ok, {"pairs": N + 2, "ms": ms})
        P.close()

    if on("g2_msm"):
        n = 1 << args.log2n
        a, b = rng.fr_int(), rng.fr_int()
        x = synth.arith_fr_mont(a, b, n)
        s = rng.uniform_fr(n)
        Q = lsa.Bases("g2", lsa.batch_exp("g2", G2, to_dev(x)), on_device=True)
        d_s = to_dev(s)
        out = torch.zeros(24, dtype=torch.int64, device=dev)
        ms = timed(lambda: Q.msm_async(d_s, out), max(1, args.reps // 2))
        ok = np.array_equal(affine("g2", host(out))[0], k_times_gen("g2", [synth.fr_dot_mont(s, x)])[0])
        emit("G2 MSM n=2^%d" % args.log2n, ok, {"ms": ms, "pairs_per_s": n / ms * 1e3})
        Q.close()

    if on("cppoly"):
        d = args.d
        n = 1 << d
        g1 = torch.from_numpy(G1.view(np.int64)).to(dev).repeat(n, 1).contiguous()
        g2 = torch.from_numpy(G2.view(np.int64)).to(dev).repeat(n, 1).contiguous()
        B1 = lsa.Bases("g1", g1, on_device=True)
        B2 = lsa.Bases("g2", g2, on_device=True)
        del g1, g2
        v = rng.uniform_fr(n)
        s = to_dev(v)
        o1 = torch.zeros(12, dtype=torch.int64, device=dev)
        o2 = torch.zeros(24, dtype=torch.int64, device=dev)

        def commit():                                # commit.h:154-155: one shared scalar sort
            lsa.commit_async(B1, B2, s, o1, o2)

        def commit_two_calls():
            B1.msm_async(s, o1)
            B2.msm_async(s, o2)

        r = to_dev(rng.uniform_fr(d))
        w = torch.empty_like(s)
        outs = torch.zeros((2 * d, 12), dtype=torch.int64, device=dev)

        def fold():
            lsa.cppoly_witness(s, r, out=w)          # poly.h:55-67 on the device

        def ladder():                                # poly.h:76-88, scalars = slices of w
            start = 0
            for i in range(d):
                m = 1 << (d - 1 - i)
                B1.msm_async(w[start:start + m], outs[2 * i], n=m)
                if i:
                    B1.msm_async(w[start:start + m], outs[2 * i + 1], n=m)
                start += m

        # the ladder as an integrator issues it: rungs of up to 2^15 pairs in ONE segmented call
        # (consecutive slices of w; witnessa[i] is the same sum as witness[i] and is not recomputed),
        # the longer rungs one call each
        starts = np.concatenate([[0], np.cumsum([1 << (d - 1 - i) for i in range(d)])]).astype(np.uint64)
        first_small = next(i for i in range(d) if (1 << (d - 1 - i)) <= (1 << 15))
        seg_outs = torch.zeros((d, 12), dtype=torch.int64, device=dev)

        def ladder_segmented():
            for i in range(first_small):
                B1.msm_async(w[int(starts[i]):int(starts[i + 1])], seg_outs[i], n=1 << (d - 1 - i))
            B1.msm_segments_async(w, starts[first_small:], seg_outs[first_small:])

        def prove():
            fold()
            ladder_segmented() if B1.has_table() else ladder()

        ms_c2 = timed(commit_two_calls, max(1, args.reps // 2))
        ms_c = timed(commit, max(1, args.reps // 2))
        ms_f = timed(fold, max(1, args.reps // 2))
        ms_l = timed(ladder, max(1, args.reps // 2))
        ms_ls = timed(ladder_segmented, max(1, args.reps // 2)) if B1.has_table() else None
        ms_p = timed(prove, max(1, args.reps // 2))
        vs = synth.fr_sum_mont(v)
        ok = np.array_equal(affine("g1", host(o1))[0], k_times_gen("g1", [vs])[0])
        ok = ok and np.array_equal(affine("g2", host(o2))[0], k_times_gen("g2", [vs])[0])
        wh = host(w)
        ks, slots, start = [], [], 0
        for i in range(d):
            m = 1 << (d - 1 - i)
            k = synth.fr_sum_mont(wh[start:start + m])
            ks.append(k); slots.append(2 * i)
            if i:
                ks.append(k); slots.append(2 * i + 1)
            start += m
        want_pts = k_times_gen("g1", ks)
        ok = ok and np.array_equal(affine("g1", host(outs)[slots]), want_pts)
        if B1.has_table():
            seg_slots = [sl // 2 for sl in slots]
            ok = ok and np.array_equal(affine("g1", host(seg_outs)[seg_slots]), want_pts)
        emit("CPpoly d=%d" % d, ok, {"commit_ms": ms_c, "commit_as_two_msm_calls_ms": ms_c2, "prove_fold_ms": ms_f, "prove_msm_ladder_39_calls_ms": ms_l,
                                    "prove_msm_ladder_segmented_ms": ms_ls,
                                    "prove_total_ms": ms_p, "prove_pairs": (n - 1) + (n // 2 - 1), "msms_checked": len(ks) + 2})
        B1.close(); B2.close()

    fq12_one = np.zeros(48, dtype=np.uint64)
    fq12_one[0:4] = curve.fq_mont(1)

    def planted(n):
        al, be = rng.uniform_fr(n), rng.uniform_fr(n)
        rinv = pow(curve.MONT, -1, curve.R)
        a_last = synth.limbs_to_int(al[-1]) * rinv % curve.R
        be[-1] = curve.fr_mont((-synth.fr_dot_mont(al[:-1], be[:-1])) * pow(a_last, -1, curve.R) % curve.R)
        return lsa.batch_exp("g1", G1, al), lsa.batch_exp("g2", G2, be)

    if on("pairing"):
        n = 1 << args.log2pairs
        ps, qs = planted(n)
        res = lsa.pairing_product(ps, qs)
        best = 1e9
        for _ in range(max(2, args.reps)):
            t0 = time.perf_counter()
            res = lsa.pairing_product(ps, qs)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        gf = n * MILLER_FQ_MULS / (best * 1e-3) / 1e9
        emit("pairing product, 2^%d Miller loops + 1 final exp (host buffers, planted product == 1)" % args.log2pairs,
             np.array_equal(res, fq12_one),
             {"ms": best, "pairings_per_s": n / best * 1e3,
              "roofline": {"bound": "valu", "achieved": gf, "peak": FMUL_PEAK_G, "unit": "G field-mults/s", "frac": gf / FMUL_PEAK_G,
                           "fq_mults_per_miller_loop": MILLER_FQ_MULS,
                           "note": "whole call (upload + Miller loops + product tree + final exp) against the measured "
                                   "chip-wide ceiling of the 9x29-bit Montgomery product"}})

    if on("cphad_verify"):
        # CPhad verify at d: 9d+1 Miller loops in 3d+2 products, each followed by a
        # final exponentiation; every product planted to one
        d = args.d
        sizes = [3] * (3 * d + 1) + [4]              # 9d+7 >= 9d+1 loops in 3d+2 products
        groups = [planted(m) for m in sizes]
        ps = np.concatenate([g[0] for g in groups]); qs = np.concatenate([g[1] for g in groups])
        edges = np.cumsum([0] + sizes)

        def run():
            return lsa.pairing_product_segments(ps, qs, edges)
        res = run()
        best = 1e9
        for _ in range(max(2, args.reps)):
            t0 = time.perf_counter()
            res = run()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        emit("CPhad verifier pairing shape d=%d: %d Miller loops, %d final exponentiations (lsa_pairing_product_segments, host buffers)" % (d, len(ps), len(sizes)),
             all(np.array_equal(r_, fq12_one) for r_ in res), {"ms": best})


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Verifier-side timings, every one belonging to a checked result (run on the GPU box).  The shapes are the
reference's own pairing call patterns; what is resident and what is fresh is stated per line, because that
is what decides the cost on this library (a Q seen before costs the Fq12 chain only):

  miller_cached     one miller_loop on a precomputed / resident Q  (subspace.cc:152-166, lipmaa.cc:194-200)
  miller_fresh      one miller_loop on a Q never seen (table cache off)
  check2            simple_pairing_check (globl.h:94-105): final_exp(e(a1,a2) * e(b1,b2)^-1) as ONE product
  cppoly_verify     CPPoly::verify at d (poly.h:92-123): d-1 checks of two loops on the fixed generator, then one
                    product of d+1 loops whose Q are pts[i]*g2 -- FRESH per proof -- and one final exponentiation
  cphad_shape       62 products / 187 loops / 62 final exponentiations, Q resident | Q fresh
  product           one product of 2^k fresh pairs (BASELINE config 5)
G1 points are normalised (Z = 1: MSM outputs are) unless --jacobian; G2 points likewise."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--d", type=int, default=20)
    ap.add_argument("--log2pairs", type=int, default=12)
    ap.add_argument("--jacobian", action="store_true", help="un-normalised inputs (one field inversion per point on the device)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import numpy as np
    import legosnark_amd as lsa
    from legosnark_amd import curve, synth
    lsa.init(0)
    G1, G2 = curve.generator("g1"), curve.generator("g2")
    rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0x5A)
    one = np.zeros(48, dtype=np.uint64)
    one[0:4] = curve.fq_mont(1)
    rinv = pow(curve.MONT, -1, curve.R)

    def pts(group, scalars_mont):
        p = lsa.batch_exp(group, G1 if group == "g1" else G2, scalars_mont)
        return p if args.jacobian else lsa.normalize(group, p)

    def fr_int(limbs):
        return synth.limbs_to_int(limbs) * rinv % curve.R

    def best_of(fn, reps=None, before=None):
        fn()
        t = []
        for _ in range(reps or args.reps):
            if before:
                before()
            t0 = time.perf_counter()
            r = fn()
            t.append((time.perf_counter() - t0) * 1e3)
        t.sort()
        return r, t[0], t[len(t) // 2]

    def emit(name, ok, ms_min, ms_med, **kw):
        d = {"shape": name}
        if not ok:
            d["error"] = "result check failed -- no time reported"
        else:
            d.update({"result_checked": True, "ms": round(ms_min, 4), "ms_median": round(ms_med, 4)})
            d.update(kw)
        print(json.dumps(d), flush=True)

    want = set(args.only.split(",")) if args.only else None

    def on(n):
        return want is None or n in want

    # ---- lone Miller loops
    if on("miller"):
        a, b = rng.uniform_fr(1), rng.uniform_fr(1)
        P, Q = pts("g1", a), pts("g2", b)
        tab = lsa.g2_precompute(Q)
        ref = lsa.miller_loop(P, Q)
        lsa.g2_table_cache(4096)
        r, mn, md = best_of(lambda: lsa.miller_loop_precomp(P, tab))
        emit("miller_cached: 1 miller_loop on a precomputed Q (host buffers, blocking)", np.array_equal(r, ref), mn, md)
        r, mn, md = best_of(lambda: lsa.miller_loop(P, Q))
        emit("miller_cached: 1 miller_loop on a point seen before (table resident)", np.array_equal(r, ref), mn, md)
        lsa.g2_table_cache(0)
        r, mn, md = best_of(lambda: lsa.miller_loop(P, Q))
        emit("miller_fresh: 1 miller_loop on a point never seen (table cache off)", np.array_equal(r, ref), mn, md)
        r, mn, md = best_of(lambda: lsa.g2_precompute(Q))
        emit("precompute_G2: 1 point -> libff G2_precomp bytes", np.array_equal(r, tab), mn, md)
        lsa.g2_table_cache(4096)

    # ---- simple_pairing_check: e(a*b G1, G2) == e(a G1, b G2)
    if on("check2"):
        a, b = rng.uniform_fr(1), rng.uniform_fr(1)
        ab = np.array([curve.fr_mont(fr_int(a[0]) * fr_int(b[0]) % curve.R)])
        ps = np.concatenate([pts("g1", ab), pts("g1", a)])
        qs = np.concatenate([pts("g2", np.array([curve.fr_mont(1)])), pts("g2", b)])
        off = np.array([0, 2], dtype=np.uint64)
        r, mn, md = best_of(lambda: lsa.pairing_terms(ps, off, g2=qs, flags=[0, 1]))
        emit("check2: final_exp(e(a1,a2) * e(b1,b2)^-1) as one product, both Q resident", np.array_equal(r[0], one), mn, md)

    # ---- CPPoly::verify at d
    if on("cppoly_verify"):
        d = args.d
        g2 = pts("g2", np.array([curve.fr_mont(1)]))[0]
        # checks i = 1 .. d-1: e(w_i, g2) * e(wa_i, g2)^-1 with wa_i = w_i (the reference uses g^a = g, poly.h:98)
        w = pts("g1", rng.uniform_fr(d))
        # last product: prod_i e(w_i, r_i g2) * e(c, g2)^-1 == 1 with c = (sum w_i r_i) G1
        wk = rng.uniform_fr(d)
        wl = pts("g1", wk)
        state = {}

        def fresh():
            rr = rng.uniform_fr(d)
            state["qs"] = pts("g2", rr)
            state["c"] = pts("g1", np.array([curve.fr_mont(synth.fr_dot_mont(wk, rr))]))

        def run():
            ps = np.concatenate([np.repeat(w[1:], 2, axis=0), wl, state["c"]])
            qs = np.concatenate([np.repeat(g2[None], 2 * (d - 1), axis=0), state["qs"], g2[None]])
            flags = np.array([0, 1] * (d - 1) + [0] * d + [1], dtype=np.uint8)
            off = np.array(list(range(0, 2 * (d - 1) + 1, 2)) + [2 * (d - 1) + d + 1], dtype=np.uint64)
            return lsa.pairing_terms(ps, off, g2=qs, flags=flags)
        fresh()
        r, mn, md = best_of(run, before=fresh)
        emit("cppoly_verify d=%d: %d loops, %d final exps in ONE call; %d Q fresh per proof, the generator resident" % (d, 3 * d - 1, d, d),
             all(np.array_equal(x, one) for x in r), mn, md, fresh_q=d)

    # ---- the CPhad verifier shape: 62 products of 3 (one of 4) loops
    if on("cphad_shape"):
        sizes = [3] * 61 + [4]
        P, Q = [], []
        for m in sizes:
            al, be = rng.uniform_fr(m), rng.uniform_fr(m)
            be[-1] = curve.fr_mont((-synth.fr_dot_mont(al[:-1], be[:-1])) * pow(fr_int(al[-1]), -1, curve.R) % curve.R)
            P.append(pts("g1", al)); Q.append(pts("g2", be))
        ps, qs = np.concatenate(P), np.concatenate(Q)
        off = np.cumsum([0] + sizes).astype(np.uint64)
        lsa.g2_table_cache(4096)
        r, mn, md = best_of(lambda: lsa.pairing_product_segments(ps, qs, off))
        emit("cphad_shape: 187 loops in 62 products + 62 final exps, every Q resident", all(np.array_equal(x, one) for x in r), mn, md)
        lsa.g2_table_cache(0)
        r, mn, md = best_of(lambda: lsa.pairing_product_segments(ps, qs, off))
        emit("cphad_shape: 187 loops in 62 products + 62 final exps, every Q fresh (table cache off)", all(np.array_equal(x, one) for x in r), mn, md)
        lsa.g2_table_cache(4096)

    # ---- one big product of fresh pairs
    if on("product"):
        n = 1 << args.log2pairs
        al, be = rng.uniform_fr(n), rng.uniform_fr(n)
        be[-1] = curve.fr_mont((-synth.fr_dot_mont(al[:-1], be[:-1])) * pow(fr_int(al[-1]), -1, curve.R) % curve.R)
        ps, qs = pts("g1", al), pts("g2", be)
        r, mn, md = best_of(lambda: lsa.pairing_product(ps, qs), reps=5)
        emit("product: 2^%d fresh pairs, 1 final exp (host buffers, planted == 1)" % args.log2pairs, np.array_equal(r, one), mn, md, pairs=n)


if __name__ == "__main__":
    main()

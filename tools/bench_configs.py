#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs that are parity-test cases rather
than the headline bench line (run on the GPU box; prints one JSON object per config):

  cplink_prover   SubspaceSnark::prove shape (subspace.cc:78-85): MSM over N+2 pairs, w[0]=0
  g2_msm          alt_bn128 G2 MSM
  cppoly          CPpoly d-variable commit + prove ladder (poly.h:30-32,76-88): G1+G2 MSM of
                  2^d, then G1 MSMs of 2^(d-1-i) (twice for i>=1), bases = copies of the generator
  pairing         batched Miller loops + one final exponentiation over 2^k pairs
Inputs are generated on the GPU with the library's own batch_exp."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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
    from legosnark_amd import curve
    dev = torch.device("cuda:0")
    lsa.init(0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(12345)

    def random_fr(count):
        t = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device=dev, generator=gen)
        t[:, 3] &= (1 << 60) - 1
        return t.contiguous()

    def timed(fn, reps):
        fn(); lsa.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        lsa.synchronize(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    want = set(args.only.split(",")) if args.only else None

    def on(name):
        return want is None or name in want

    if on("cplink_prover"):
        N = 1 << args.log2n
        P = lsa.Bases("g1", lsa.batch_exp("g1", curve.generator("g1"), random_fr(N + 2)), on_device=True)
        w = random_fr(N + 2)
        w[0] = 0
        out = torch.zeros(12, dtype=torch.int64, device=dev)
        ms = timed(lambda: P.msm_async(w, out), args.reps)
        print(json.dumps({"config": "CPlink prover (SubspaceSnark::prove MSM), N=2^%d" % args.log2n, "pairs": N + 2, "ms": ms}), flush=True)
        P.close()

    if on("g2_msm"):
        n = 1 << args.log2n
        Q = lsa.Bases("g2", lsa.batch_exp("g2", curve.generator("g2"), random_fr(n)), on_device=True)
        s = random_fr(n)
        out = torch.zeros(24, dtype=torch.int64, device=dev)
        ms = timed(lambda: Q.msm_async(s, out), max(1, args.reps // 2))
        print(json.dumps({"config": "G2 MSM n=2^%d" % args.log2n, "ms": ms, "pairs_per_s": n / ms * 1e3}), flush=True)
        Q.close()

    if on("cppoly"):
        d = args.d
        n = 1 << d
        g1 = torch.from_numpy(np.tile(curve.generator("g1").view(np.int64), (n, 1))).to(dev)
        g2 = torch.from_numpy(np.tile(curve.generator("g2").view(np.int64), (n, 1))).to(dev)
        B1 = lsa.Bases("g1", g1, on_device=True)
        B2 = lsa.Bases("g2", g2, on_device=True)
        s = random_fr(n)
        o1 = torch.zeros(12, dtype=torch.int64, device=dev)
        o2 = torch.zeros(24, dtype=torch.int64, device=dev)

        def commit():
            B1.msm_async(s, o1)
            B2.msm_async(s, o2)

        r = random_fr(d)
        w = torch.empty_like(s)

        def fold():
            lsa.cppoly_witness(s, r, out=w)          # poly.h:55-67 on the device

        def ladder():                                # poly.h:76-88, scalars = slices of w
            start = 0
            for i in range(d):
                m = 1 << (d - 1 - i)
                B1.msm_async(w[start:start + m], o1, n=m)
                if i:
                    B1.msm_async(w[start:start + m], o1, n=m)
                start += m

        def prove():
            fold()
            ladder()

        ms_c = timed(commit, max(1, args.reps // 2))
        ms_f = timed(fold, max(1, args.reps // 2))
        ms_l = timed(ladder, max(1, args.reps // 2))
        ms_p = timed(prove, max(1, args.reps // 2))
        print(json.dumps({"config": "CPpoly d=%d" % d, "commit_ms": ms_c, "prove_fold_ms": ms_f, "prove_msm_ladder_ms": ms_l,
                          "prove_total_ms": ms_p, "prove_pairs": (n - 1) + (n // 2 - 1)}), flush=True)
        B1.close(); B2.close()

    if on("pairing"):
        n = 1 << args.log2pairs
        ps = lsa.batch_exp("g1", curve.generator("g1"), random_fr(n)).cpu().numpy().view(np.uint64)
        qs = lsa.batch_exp("g2", curve.generator("g2"), random_fr(n)).cpu().numpy().view(np.uint64)
        t0 = time.perf_counter()
        lsa.pairing_product(ps, qs)
        ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        lsa.pairing_product(ps, qs)
        ms = min(ms, (time.perf_counter() - t0) * 1e3)
        print(json.dumps({"config": "pairing product, 2^%d Miller loops + 1 final exp (host buffers)" % args.log2pairs,
                          "ms": ms, "pairings_per_s": n / ms * 1e3}), flush=True)


if __name__ == "__main__":
    main()

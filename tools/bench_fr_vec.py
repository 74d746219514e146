#!/usr/bin/env python3
"""Times the Fr-vector kernels on device-resident vectors: the CPpoly witness recursion and
evalMLE at d = 20 and d = 24, with the algorithmic HBM bytes of the first (largest) round, and
the oracle's loops on one host core beside them."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import legosnark_amd as lsa  # noqa: E402
import oracle_lib as o  # noqa: E402


def main():
    lsa.init(0)
    for d in (20, 24):
        N = 1 << d
        g = torch.Generator(device="cuda:0").manual_seed(d)
        d_v = torch.randint(0, 1 << 62, (N, 4), dtype=torch.int64, device="cuda:0", generator=g)
        d_v[:, 3] &= (1 << 60) - 1            # any value < r is a valid Montgomery residue
        r, _ = o.random_scalars(d, seed=9)
        d_r = torch.from_numpy(r.view(np.int64)).to("cuda:0")
        d_w = torch.empty_like(d_v)
        d_out = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        for name, fn in (("cppoly_witness", lambda: lsa.cppoly_witness(d_v, d_r, out=d_w)),
                         ("eval_mle", lambda: lsa.eval_mle_device(d_v, d_r, d_out))):
            fn()
            lsa.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                fn()
            lsa.synchronize()
            dt = (time.perf_counter() - t0) / reps
            # algorithmic: v read once + w written once (64 N) / the table read once (32 N); moved: the same + 1 / 4096 of it
            # (twelve rounds per pass; two rounds per launch moved 85 N / 53 N, every round through memory 128 N / 96 N)
            alg = (64 if name == "cppoly_witness" else 32) * N
            moved = alg + alg // 4096 * 2
            print(json.dumps({"op": name, "d": d, "ms": round(dt * 1e3, 4), "moved_GBps": round(moved / dt / 1e9, 1), "frac_of_8TBps_moved": round(moved / dt / 8e12, 4),
                              "frac_of_8TBps_algorithmic": round(alg / dt / 8e12, 4)}))
        if d == 20:
            v = d_v.cpu().numpy().view(np.uint64)
            t0 = time.perf_counter()
            w = o.fr_cppoly_witness(v, r)
            t1 = time.perf_counter()
            e = o.fr_eval_mle(v, r)
            t2 = time.perf_counter()
            ok = np.array_equal(w, d_w.cpu().numpy().view(np.uint64)) and np.array_equal(e, d_out.cpu().numpy().view(np.uint64))
            print(json.dumps({"op": "oracle loops, 1 core", "d": d, "cppoly_witness_ms": (t1 - t0) * 1e3,
                              "eval_mle_ms": (t2 - t1) * 1e3, "matches_gpu": bool(ok)}))


def ntt():
    """lsa_fr_ntt on device-resident vectors (libfqfft's FFT / icosetFFT as /root/reference/src/gadgets/lipmaa.cc:102-168
    calls them): ms per transform with the domain's tables cached, 64 B per element of algorithmic traffic (read once,
    written once), field products per second against the 175 G/s ceiling of the 29-bit-limb product."""
    lsa.init(0)
    g = o.fr_mont(o.FR_GENERATOR)
    for log_n in (10, 12, 16, 18, 20, 22, 24):
        n = 1 << log_n
        gen = torch.Generator(device="cuda:0").manual_seed(log_n)
        d_a = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda:0", generator=gen)
        d_a[:, 3] &= (1 << 60) - 1
        w = o.fr_mont(o.fr_root_of_unity(log_n))
        for name, kw in (("FFT", {}), ("icosetFFT", {"inverse": True, "coset": g})):
            lsa.fr_ntt(d_a, w, **kw)
            lsa.synchronize()
            reps = 20 if log_n <= 20 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                lsa.fr_ntt(d_a, w, **kw)
            lsa.synchronize()
            dt = (time.perf_counter() - t0) / reps
            passes = 1 if log_n <= 10 else (2 if log_n <= 16 else 3)
            mults = n * (log_n / 2 - 0.5 * passes + 2 * (passes - 1) + 1 + (1 if kw else 0))
            print(json.dumps({"op": "ntt " + name, "log_n": log_n, "ms": round(dt * 1e3, 4), "passes": passes,
                              "algorithmic_GBps": round(64 * n / dt / 1e9, 1), "frac_of_8TBps": round(64 * n / dt / 8e12, 4),
                              "moved_GBps": round(64 * n * passes / dt / 1e9, 1), "G_field_mults_per_s": round(mults / dt / 1e9, 1)}), flush=True)
    # libfqfft's step radix-2 domain (sizes that are not powers of two): m = 2^b + 2^s
    for big_log, small_log in ((10, 9), (16, 12), (20, 19), (20, 10), (22, 21), (23, 22)):
        m = (1 << big_log) + (1 << small_log)
        gen = torch.Generator(device="cuda:0").manual_seed(100 + big_log)
        d_a = torch.randint(0, 1 << 62, (m, 4), dtype=torch.int64, device="cuda:0", generator=gen)
        d_a[:, 3] &= (1 << 60) - 1
        w = o.fr_mont(o.fr_root_of_unity(big_log + 1))
        for name, kw in (("FFT", {}), ("icosetFFT", {"inverse": True, "coset": g})):
            lsa.fr_ntt_step(d_a, big_log, small_log, w, **kw)
            lsa.synchronize()
            reps = 20 if big_log <= 20 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                lsa.fr_ntt_step(d_a, big_log, small_log, w, **kw)
            lsa.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(json.dumps({"op": "step domain " + name, "m": "2^%d + 2^%d" % (big_log, small_log), "ms": round(dt * 1e3, 4),
                              "algorithmic_GBps": round(64 * m / dt / 1e9, 1), "frac_of_8TBps": round(64 * m / dt / 8e12, 4)}), flush=True)


def sumcheck():
    """lsa_fr_sumcheck_round on device-resident tables (CPSumcheck::make_new_h_poly, /root/reference/src/gadgets/sumcheck.h:85-106):
    ms per round polynomial at half = 2^23 (round 0 of a prover at d = 24) and 2^19, two and three MLE tables with the beta
    factor; 32 (2 m + 1) bytes per index of algorithmic traffic."""
    lsa.init(0)
    pre, rho = o.fr_mont(12345), o.fr_mont(67890)
    for log_half in (19, 23):
        half = 1 << log_half
        gen = torch.Generator(device="cuda:0").manual_seed(log_half)
        def table(n):
            t = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda:0", generator=gen)
            t[:, 3] &= (1 << 60) - 1
            return t
        for m in (2, 3):
            tabs = [table(2 * half) for _ in range(m)]
            suff = table(half)
            lsa.sumcheck_round(tabs, suff=suff, pre=pre, rho_j=rho)
            ts = []
            for _ in range(11):                              # blocking calls: the median of eleven (one hiccup of the box does not move it)
                t0 = time.perf_counter()
                lsa.sumcheck_round(tabs, suff=suff, pre=pre, rho_j=rho)
                ts.append(time.perf_counter() - t0)
            dt = sorted(ts)[5]
            byt = 32 * (2 * m + 1) * half
            print(json.dumps({"op": "sumcheck_round", "m": m, "log_half": log_half, "ms": round(dt * 1e3, 4), "algorithmic_GBps": round(byt / dt / 1e9, 1),
                              "frac_of_8TBps": round(byt / dt / 8e12, 4)}), flush=True)
            del tabs, suff


def single_rounds():
    """One round of DPMle::pushRandomness (lsa_fr_fold: 64 B in + 32 B out per output element) and of the DPBeta suffix
    update (lsa_fr_scale_upper: 32 B in + 32 B out) on device-resident tables: the per-round work of a resident sumcheck
    prover beside lsa_fr_sumcheck_round -- plain HBM streams, one product per output element."""
    lsa.init(0)
    L = lsa.lib()
    for d in (20, 24):
        n, half = 1 << d, 1 << (d - 1)
        gen = torch.Generator(device="cuda:0").manual_seed(d)
        t = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda:0", generator=gen)
        t[:, 3] &= (1 << 60) - 1
        r = torch.from_numpy(o.fr_mont(12345).view(np.int64).copy()).to("cuda:0")
        k = o.fr_mont(777)
        cur = torch.empty((half, 4), dtype=torch.int64, device="cuda:0")
        for name, byt, fn in (("fr_fold (pushRandomness)", 96 * half, lambda: lsa._check(L.lsa_fr_fold(t.data_ptr(), half, r.data_ptr(), cur.data_ptr(), 1))),
                              ("fr_scale_upper (DPBeta suffix)", 64 * half, lambda: lsa._check(L.lsa_fr_scale_upper(t.data_ptr(), half, lsa._host_ptr(k), cur.data_ptr(), 1)))):
            fn()
            lsa.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            lsa.synchronize()
            dt = (time.perf_counter() - t0) / 10
            print(json.dumps({"op": name, "log_half": d - 1, "ms": round(dt * 1e3, 4), "algorithmic_GBps": round(byt / dt / 1e9, 1),
                              "frac_of_8TBps": round(byt / dt / 8e12, 4)}), flush=True)


def prover_loop():
    """The Fr work of a whole sumcheck prover over two MLE tables and the beta factor, tables resident on the device
    (CPSumcheck::prove's loop, /root/reference/src/gadgets/sumcheck.cc:60-92 with sumcheck.h:85-106, mle.h:46-53,199-210):
    per round j the round polynomial (its coefficients come back to the host: the prover hashes / commits them), then
    pushRandomness on both tables and the suffix update -- d rounds, the vectors halving.  The reference runs these loops
    on the host field (`##had_sc (Sumcheck) Prove` of hadamard d also holds its commitments)."""
    lsa.init(0)
    L = lsa.lib()
    for d in (20, 24):
        n = 1 << d
        gen = torch.Generator(device="cuda:0").manual_seed(1000 + d)
        def table(m):
            t = torch.randint(0, 1 << 62, (m, 4), dtype=torch.int64, device="cuda:0", generator=gen)
            t[:, 3] &= (1 << 60) - 1
            return t
        a0, b0, s0 = table(n), table(n), table(n // 2)
        rs, _ = o.random_scalars(d, seed=7)
        ks, _ = o.random_scalars(d, seed=8)
        pre, rho = o.fr_mont(12345), o.fr_mont(67890)
        d_r = torch.from_numpy(rs.view(np.int64).copy()).to("cuda:0")
        ts = []
        for rep in range(4):
            a, b, sf = a0.clone(), b0.clone(), s0.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j in range(d):
                half = 1 << (d - j - 1)
                use_suff = j + 1 <= d - 1
                lsa.sumcheck_round([a[:2 * half], b[:2 * half]], suff=sf[:half] if use_suff else None, pre=pre, rho_j=rho)
                for tv in (a, b):
                    lsa._check(L.lsa_fr_fold(tv.data_ptr(), half, d_r[j].data_ptr(), tv.data_ptr(), 1))
                if half >= 2:
                    lsa._check(L.lsa_fr_scale_upper(sf.data_ptr(), half // 2, lsa._host_ptr(ks[j]), sf.data_ptr(), 1))
            lsa.synchronize()
            ts.append(time.perf_counter() - t0)
        print(json.dumps({"op": "resident sumcheck prover loop, two tables + beta", "d": d, "rounds": d, "ms": round(sorted(ts)[1] * 1e3, 3),
                          "ms_runs": [round(x * 1e3, 3) for x in ts]}), flush=True)
        del a0, b0, s0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "prover":
        prover_loop()
    elif len(sys.argv) > 1 and sys.argv[1] == "single":
        single_rounds()
    elif len(sys.argv) > 1 and sys.argv[1] == "sumcheck":
        sumcheck()
    elif len(sys.argv) > 1 and sys.argv[1] == "ntt":
        ntt()
    else:
        main()

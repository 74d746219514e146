"""Measurements of the BASELINE.json configs that are not the headline bench line, shared by bench.py (its
`configs` block) and tools/bench_configs.py.  EVERY time returned belongs to a result that was checked in the same
run -- by the known-discrete-log identity (k*G recomputed by the fixed-base batch_exp kernel, a different code path
from the MSM) or, for pairings, by planted products that must equal one.  A config whose check fails returns
{"config": ..., "error": ...} and no time.

  cplink_prover   SubspaceSnark::prove shape (subspace.cc:78-85): MSM over N+2 pairs, w[0] = 0
  cplink_2pow24   the same at N = 2^24 on ONE GPU (BASELINE configs[3]'s workload without its 8-way split): 1.5 GiB of bases,
                  24 pre-shifted copies (24 GiB), checked by the known-discrete-log identity
  g2_msm          alt_bn128 G2 MSM
  cppoly          CPpoly d-variable commit + prove ladder (poly.h:30-32,76-88)
  fr_fold         CPpoly witness recursion + evalMLE at d = 24 (HBM-bound Fr streams)
  ntt             radix-2 NTT of 2^24 Fr values
  pairing         ONE product of 2^k Miller loops over pairs never seen + one final exponentiation (BASELINE configs[4])
  cphad_verify    the CPhad verifier's pairing shape: 187 Miller loops in 62 products, 62 final exponentiations; with
                  every Q resident (a verifier's keys) and with every Q fresh
Inputs follow legosnark_amd/synth.py.  Field-multiplication counts are libff-shape operation counts
(Fq2 product = 3 Fq products, Fq2 square = 2), stated per line; the ceiling is the microbenchmarked chip-wide rate of
the 9 x 29-bit Montgomery product (profiles/r01_ubench_field_mul.txt)."""
import time

FMUL_PEAK_G = 175.0
HBM_PEAK_GBS = 8000.0
FQ_PER_FQ2_MUL, FQ_PER_FQ2_SQR = 3, 2
# one Miller loop as libff runs it: 64 doubling steps, 36 + 2 addition steps.  G2 side of a doubling step: 4 Fq2
# products + 6 squares; of an addition step: 11 + 2; f side: Fq12 square (12 Fq2 products) + mul_by_024 (13), resp.
# mul_by_024; 4 Fq products scale each line by (px, py)
MILLER_G2_DBL = 4 * FQ_PER_FQ2_MUL + 6 * FQ_PER_FQ2_SQR
MILLER_G2_ADD = 11 * FQ_PER_FQ2_MUL + 2 * FQ_PER_FQ2_SQR
MILLER_F_DBL = (12 + 13) * FQ_PER_FQ2_MUL + 4
MILLER_F_ADD = 13 * FQ_PER_FQ2_MUL + 4
MILLER_FQ_MULS = 64 * (MILLER_G2_DBL + MILLER_F_DBL) + 38 * (MILLER_G2_ADD + MILLER_F_ADD)      # 9 632: the G2 arithmetic included
MILLER_FQ_MULS_RESIDENT = 64 * MILLER_F_DBL + 38 * MILLER_F_ADD                               # 6 690: over a resident line table
# final exponentiation: 3 exp-by-z of 62 cyclotomic squarings (9 Fq2 squares) + ~19 products (18 Fq2 products), ~20 more
# products, one inversion (~100 Fq products), 5 Frobenius maps
FINAL_EXP_FQ_MULS = 3 * (62 * 9 * FQ_PER_FQ2_SQR + 19 * 18 * FQ_PER_FQ2_MUL) + 20 * 18 * FQ_PER_FQ2_MUL + 100 + 5 * 15
G2_MSM_FQ_MULS_PER_ADD = 10 * FQ_PER_FQ2_MUL      # XYZZ mixed addition over Fq2: 8 products + 2 squares, counted as 10 products


def measure(lsa, torch, np, dev, log2n=20, d=20, log2pairs=12, reps=5, only=None):
    """Runs the selected configs on `dev` and returns a list of dicts (one per line)."""
    from legosnark_amd import curve, synth
    G1, G2 = curve.generator("g1"), curve.generator("g2")
    rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0xBC)
    rinv = pow(curve.MONT, -1, curve.R)
    out = []

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

    def timed(fn, n):
        fn(); lsa.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        lsa.synchronize(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def best_of(fn, n, before=None):
        fn()
        ts = []
        for _ in range(n):
            if before:
                before()
            t0 = time.perf_counter()
            r = fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return r, min(ts)

    def emit(cfg, ok, fields):
        if not ok:
            out.append({"config": cfg, "error": "result check failed -- no time reported"})
            return
        dct = {"config": cfg, "result_checked": True}
        dct.update(fields)
        out.append(dct)

    def valu(fq_mults, ms):
        g = fq_mults / (ms * 1e-3) / 1e9
        return {"achieved_Gfmul_s": round(g, 3), "peak_Gfmul_s": FMUL_PEAK_G, "frac": round(g / FMUL_PEAK_G, 4), "fq_mults": int(fq_mults)}

    def on(name):
        return only is None or name in only

    half = max(1, reps // 2)
    if on("cplink_prover"):
        N = 1 << log2n
        a, b = rng.fr_int(), rng.fr_int()
        x = synth.arith_fr_mont(a, b, N + 2)
        w = rng.uniform_fr(N + 2)
        w[0] = 0
        P = lsa.Bases("g1", lsa.batch_exp("g1", G1, to_dev(x)), on_device=True)
        d_w = to_dev(w)
        res = torch.zeros(12, dtype=torch.int64, device=dev)
        ms = timed(lambda: P.msm_async(d_w, res), reps)
        ok = np.array_equal(affine("g1", host(res))[0], k_times_gen("g1", [synth.fr_dot_mont(w, x)])[0])
        fm = P.field_mults_per_pair(N + 2)
        emit("CPlink prover (SubspaceSnark::prove MSM), N=2^%d, resident CRS and witness" % log2n, ok,
             {"pairs": N + 2, "ms": ms, "algorithmic_bytes": 96 * (N + 2), "valu": valu((N + 2) * fm, ms)})
        P.close()

    if on("cplink_2pow24"):
        # BASELINE configs[3] is this MSM sharded over 8 GPUs; one GPU holds all of it (288 GB): the one-GPU stand-in.
        # w = (0, rF, u) (cplink.cc:107-108): any value below r is a valid Montgomery residue, drawn on the device.
        N = 1 << 24
        n = N + 2
        a, b = rng.fr_int(), rng.fr_int()
        x = synth.arith_fr_mont(a, b, n)
        t0 = time.perf_counter()
        P = lsa.Bases("g1", lsa.batch_exp("g1", G1, to_dev(x)), on_device=True)
        lsa.synchronize()
        setup_s = time.perf_counter() - t0
        del x
        gen = torch.Generator(device=dev).manual_seed(2024)
        d_w = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=gen)
        d_w[:, 3] &= (1 << 60) - 1
        d_w[0] = 0
        d_w = d_w.contiguous()
        res = torch.zeros(12, dtype=torch.int64, device=dev)
        ms = timed(lambda: P.msm_async(d_w, res), max(3, half))
        ts = []
        for _ in range(3):
            lsa.synchronize()
            t0 = time.perf_counter()
            P.msm_async(d_w, res)
            lsa.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        # sum_i w_i (a + i b) = a T1 + b T2 with T1 = sum w_i, T2 = sum i w_i, exactly, in blocks of 4096 rows of 32-bit words
        wh = host(d_w)
        w32 = wh.view(np.uint32).reshape(n, 8)
        blk = 4096
        nb = (n + blk - 1) // blk
        t1 = t2 = 0
        il = (np.arange(n, dtype=np.uint64) % blk)
        edges = np.arange(0, n, blk)
        for k in range(8):
            col = w32[:, k].astype(np.uint64)
            sb = np.add.reduceat(col, edges)                       # per block: sum of words (< 2^44)
            lb = np.add.reduceat(col * il, edges)                  # per block: sum of (row in block) * word (< 2^56)
            sk = sum(int(v) for v in sb)
            lk = sum(int(v) for v in lb) + sum(int(j) * blk * int(v) for j, v in enumerate(sb))
            t1 += sk << (32 * k)
            t2 += lk << (32 * k)
        del wh, w32, il
        dot = (a * t1 + b * t2) * rinv % curve.R
        ok = np.array_equal(affine("g1", host(res))[0], k_times_gen("g1", [dot])[0])
        fm = P.field_mults_per_pair(n)
        emit("CPlink prover (SubspaceSnark::prove MSM) N=2^24+2 on ONE GPU, resident CRS and witness (configs[3]'s workload, unsharded)", ok,
             {"pairs": n, "ms": ms, "blocking_ms": min(ts), "pairs_per_s": n / ms * 1e3, "algorithmic_bytes": 96 * n,
              "hbm_frac_algorithmic": round(96 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "valu": valu(n * fm, ms),
              "table_bytes": int(P.table_windows()) * 64 * n, "pre_shifted_copies": int(P.table_windows()), "handle_setup_s": round(setup_s, 2),
              "checked_by": "known-discrete-log identity: sum_i w_i (a + i b) * G recomputed by the fixed-base kernel"})
        P.close()
        del d_w
        torch.cuda.empty_cache()

    if on("g2_msm"):
        n = 1 << log2n
        a, b = rng.fr_int(), rng.fr_int()
        x = synth.arith_fr_mont(a, b, n)
        s = rng.uniform_fr(n)
        Q = lsa.Bases("g2", lsa.batch_exp("g2", G2, to_dev(x)), on_device=True)
        d_s = to_dev(s)
        res = torch.zeros(24, dtype=torch.int64, device=dev)
        ms = timed(lambda: Q.msm_async(d_s, res), half)
        ok = np.array_equal(affine("g2", host(res))[0], k_times_gen("g2", [synth.fr_dot_mont(s, x)])[0])
        adds = Q.field_mults_per_pair(n) // 10 if Q.has_table() else 16
        emit("G2 MSM n=2^%d, resident bases" % log2n, ok,
             {"ms": ms, "pairs_per_s": n / ms * 1e3, "algorithmic_bytes": 160 * n, "hbm_frac_algorithmic": round(160 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
              "valu": valu(n * adds * G2_MSM_FQ_MULS_PER_ADD, ms), "bucket_additions_per_pair": adds})
        Q.close()

    if on("cppoly"):
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

        # the ladder as an integrator issues it: rungs of up to 2^15 pairs in ONE segmented call (consecutive slices
        # of w; witnessa[i] is the same sum as witness[i] and is not recomputed), the longer rungs one call each
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

        ms_c2 = timed(commit_two_calls, half)
        ms_c = timed(commit, half)
        ms_f = timed(fold, half)
        ms_l = timed(ladder, half)
        ms_ls = timed(ladder_segmented, half) if B1.has_table() else None
        ms_p = timed(prove, half)
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
        fm1 = B1.field_mults_per_pair(n)
        adds2 = B2.field_mults_per_pair(n) // 10 if B2.has_table() else 16
        emit("CPpoly d=%d: commit (G1 + G2 MSM of 2^%d, one shared sort) and prove (witness recursion + MSM ladder)" % (d, d), ok,
             {"commit_ms": ms_c, "commit_as_two_msm_calls_ms": ms_c2, "prove_fold_ms": ms_f, "prove_msm_ladder_39_calls_ms": ms_l,
              "prove_msm_ladder_segmented_ms": ms_ls, "prove_total_ms": ms_p, "prove_pairs": (n - 1) + (n // 2 - 1), "msms_checked": len(ks) + 2,
              "commit_algorithmic_bytes": (96 + 160) * n, "commit_valu": valu(n * (fm1 + adds2 * G2_MSM_FQ_MULS_PER_ADD), ms_c)})
        B1.close(); B2.close()

    # ---- the Fr streams around the MSMs: HBM-bound kernels
    if on("fr_fold"):
        dd = 24
        n = 1 << dd
        gen = torch.Generator(device=dev).manual_seed(24)
        d_v = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device=dev, generator=gen)
        d_v[:, 3] &= (1 << 60) - 1                     # any value below r is a valid Montgomery residue
        r = rng.uniform_fr(dd)
        d_r = to_dev(r)
        d_w = torch.empty_like(d_v)
        d_out = torch.zeros(4, dtype=torch.int64, device=dev)
        ms_w = timed(lambda: lsa.cppoly_witness(d_v, d_r, out=d_w), half)
        ms_e = timed(lambda: lsa.eval_mle_device(d_v, d_r, d_out), half)
        val = host(d_out).copy()
        # checks: evalMLE at the corners of the cube is an entry of v; at the random point it equals the value the
        # round-by-round fold (another kernel: DPMle::pushRandomness, lsa_fr_fold) ends with
        ok = True
        for bit, idx in ((0, 0), (1, n - 1)):
            corner = to_dev(np.stack([curve.fr_mont(bit)] * dd))
            lsa.eval_mle_device(d_v, corner, d_out)
            lsa.synchronize()
            ok = ok and np.array_equal(host(d_out), host(d_v[idx]))
        cur = d_v.clone()
        size = n
        for j in range(dd):
            size //= 2
            lsa._check(lsa.lib().lsa_fr_fold(cur.data_ptr(), size, d_r[j].data_ptr(), cur.data_ptr(), 1))
        lsa.synchronize()
        fold_end = host(cur[0]).copy()
        # (the two kernels fold the variables in opposite orders: the fold of the reversed point must agree)
        cur = d_v.clone()
        size = n
        d_rr = to_dev(r[::-1].copy())
        for j in range(dd):
            size //= 2
            lsa._check(lsa.lib().lsa_fr_fold(cur.data_ptr(), size, d_rr[j].data_ptr(), cur.data_ptr(), 1))
        lsa.synchronize()
        ok = ok and (np.array_equal(val, fold_end) or np.array_equal(val, host(cur[0])))
        # a resident sumcheck prover's first round at d = 24: the round polynomial over two MLE tables with the beta factor
        # (make_new_h_poly, sumcheck.h:85-106) and one pushRandomness (mle.h:199-210) -- the second table is d_w's storage
        hh = n // 2
        pre_m, rho_m = curve.fr_mont(12345), curve.fr_mont(67890)
        d_w.copy_(d_v.flip(0))
        ms_sc = timed(lambda: lsa.sumcheck_round([d_v, d_w], suff=d_w[:hh], pre=pre_m, rho_j=rho_m), half)
        scr = torch.empty((hh, 4), dtype=torch.int64, device=dev)
        ms_pr = timed(lambda: lsa._check(lsa.lib().lsa_fr_fold(d_v.data_ptr(), hh, d_r[0].data_ptr(), scr.data_ptr(), 1)), half)
        del scr
        # algorithmic bytes: v read once, w written once (witness); the table read once (evalMLE).  Bytes moved: what the
        # launches read and write -- twelve rounds per pass (fr_vec.hip: k_fold_pairs_fused), so a pass over n_in inputs
        # moves 64 n_in (witness) / 32 n_in (evalMLE) bytes + n_in / 4096 outputs, and the passes' inputs are N, N / 4096
        bw_alg, be_alg = 64 * n, 32 * n
        bw, be = 64 * n + (64 + 32) * (n >> 12), 32 * n + (32 + 32) * (n >> 12)
        emit("Fr fold d=24: CPpoly witness recursion (poly.h:55-67) and evalMLE (polytools.h:207-234) on a resident vector of 2^24", ok,
             {"witness_ms": ms_w, "eval_mle_ms": ms_e, "algorithmic_bytes": {"witness": bw_alg, "eval_mle": be_alg}, "bytes_moved": {"witness": bw, "eval_mle": be},
              "hbm": {"bound": "hbm", "peak_GBps": HBM_PEAK_GBS,
                      "witness_moved_GBps": round(bw / ms_w / 1e6, 1), "witness_frac_moved": round(bw / ms_w / 1e6 / HBM_PEAK_GBS, 4),
                      "witness_frac_algorithmic": round(bw_alg / ms_w / 1e6 / HBM_PEAK_GBS, 4),
                      "eval_mle_moved_GBps": round(be / ms_e / 1e6, 1), "eval_mle_frac_moved": round(be / ms_e / 1e6 / HBM_PEAK_GBS, 4),
                      "eval_mle_frac_algorithmic": round(be_alg / ms_e / 1e6 / HBM_PEAK_GBS, 4)},
              "resident_prover_round_0": {"sumcheck_round_ms": ms_sc, "sumcheck_round_algorithmic_bytes": 160 * hh,
                                          "sumcheck_round_frac_of_hbm": round(160 * hh / ms_sc / 1e6 / HBM_PEAK_GBS, 4),
                                          "push_randomness_ms": ms_pr, "push_randomness_algorithmic_bytes": 96 * hh,
                                          "push_randomness_frac_of_hbm": round(96 * hh / ms_pr / 1e6 / HBM_PEAK_GBS, 4),
                                          "note": "lsa_fr_sumcheck_round over two tables of 2^24 with the beta factor (a blocking call: the coefficients "
                                                  "come back to the host), lsa_fr_fold of one table: the per-round work of a prover whose tables stay on the device"},
              "note": "all d rounds of each recursion: one product per output element on 29-bit limbs, twelve rounds per pass (two in "
                      "registers, ten as a tree in LDS), the last rounds in one workgroup, the launch sequence replayed as a hipGraph; "
                      "with two rounds per launch (start of round 5) the same recursions moved 85 N / 53 N bytes, with every round "
                      "through memory (round 4) 128 N / 96 N"})
        del d_v, d_w, cur

    if on("ntt"):
        ln = 24
        n = 1 << ln
        gen = torch.Generator(device=dev).manual_seed(42)
        d_a = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device=dev, generator=gen)
        d_a[:, 3] &= (1 << 60) - 1
        keep = d_a.clone()
        wroot = pow(5, (curve.R - 1) >> 28, curve.R)
        for _ in range(28 - ln):
            wroot = wroot * wroot % curve.R
        w = curve.fr_mont(wroot)
        g5 = curve.fr_mont(5)                          # FieldT::multiplicative_generator (lipmaa.cc:138)
        # (a transform is 2.5 ms: a lone warm-up call does not bring an idle chip's clocks up -- 2.9 ms for the two timed calls
        # that followed it, 2.44 for the same call in a loop -- so a few more go first and at least five are timed)
        for _ in range(4):
            lsa.fr_ntt(d_a, w)
        ntt_reps = max(half, 5)
        ms_f = timed(lambda: lsa.fr_ntt(d_a, w), ntt_reps)
        d_a.copy_(keep)
        # checks: icosetFFT(cosetFFT(a)) == a; FFT of the delta at 1 is omega^k (spot values)
        lsa.fr_ntt(d_a, w, coset=g5)
        changed = not torch.equal(d_a, keep)
        lsa.fr_ntt(d_a, w, inverse=True, coset=g5)
        lsa.synchronize()
        ok = changed and torch.equal(d_a, keep)
        ms_ic = timed(lambda: lsa.fr_ntt(d_a, w, inverse=True, coset=g5), ntt_reps)
        delta = torch.zeros((n, 4), dtype=torch.int64, device=dev)
        delta[1] = torch.from_numpy(curve.fr_mont(1).view(np.int64))
        lsa.fr_ntt(delta, w)
        lsa.synchronize()
        hd = host(delta)
        for k in (0, 1, 2, 1023, 1024, 65537, n - 1):
            ok = ok and np.array_equal(hd[k], curve.fr_mont(pow(wroot, k, curve.R)))
        # libfqfft's step radix-2 domain (what get_evaluation_domain returns for sizes that are not powers of two): m = 2^23 + 2^22
        sb, ss = ln - 1, ln - 2
        sm = (1 << sb) + (1 << ss)
        d_s = keep[:sm].clone()
        wstep = wroot                                   # omega of the step domain: a primitive 2^(sb + 1) = 2^24-th root of unity
        w_s = curve.fr_mont(wstep)
        for _ in range(2):
            lsa.fr_ntt_step(d_s, sb, ss, w_s)
        ms_sf = timed(lambda: lsa.fr_ntt_step(d_s, sb, ss, w_s), ntt_reps)
        d_s.copy_(keep[:sm])
        lsa.fr_ntt_step(d_s, sb, ss, w_s, coset=g5)
        lsa.fr_ntt_step(d_s, sb, ss, w_s, inverse=True, coset=g5)
        lsa.synchronize()
        ok = ok and torch.equal(d_s, keep[:sm])
        ms_sic = timed(lambda: lsa.fr_ntt_step(d_s, sb, ss, w_s, inverse=True, coset=g5), ntt_reps)
        del d_s
        mults = n * (ln / 2 - 1.5 + 1 + 1 + 1)         # butterflies less the three trivial first stages, pass-1 twiddle (one table), pass-2 twiddle, final scale
        emit("NTT 2^24 over Fr (libfqfft FFT / icosetFFT, lipmaa.cc:102-168) on a resident vector, three passes", ok,
             {"fft_ms": ms_f, "icoset_fft_ms": ms_ic, "algorithmic_bytes": 64 * n, "passes_over_the_data": 3,
              "step_domain_2^23+2^22": {"fft_ms": ms_sf, "icoset_fft_ms": ms_sic, "algorithmic_bytes": 64 * sm,
                                        "frac_of_hbm_algorithmic": round(64 * sm / ms_sf / 1e6 / HBM_PEAK_GBS, 4), "round_trip_checked": True},
              "hbm": {"algorithmic_GBps": round(64 * n / ms_f / 1e6, 1), "moved_GBps": round(3 * 64 * n / ms_f / 1e6, 1), "peak_GBps": HBM_PEAK_GBS,
                      "frac_algorithmic": round(64 * n / ms_f / 1e6 / HBM_PEAK_GBS, 4)},
              "valu": valu(mults, ms_f),
              "note": "VALU-bound, not HBM-bound: %.1f field products per element against 64 B of traffic -- at the 175 G/s product "
                      "ceiling the transform cannot beat %.2f ms, where three passes at 8 TB/s would take %.2f ms" % (
                          mults / n, mults / FMUL_PEAK_G / 1e6, 3 * 64 * n / HBM_PEAK_GBS / 1e6)})
        del d_a, keep, delta

    fq12_one = np.zeros(48, dtype=np.uint64)
    fq12_one[0:4] = curve.fq_mont(1)

    def fr_int(limbs):
        return synth.limbs_to_int(limbs) * rinv % curve.R

    def planted(n):
        """n pairs (alpha_i G1, beta_i G2), normalised, with sum alpha_i beta_i = 0: the product of their pairings is one."""
        al, be = rng.uniform_fr(n), rng.uniform_fr(n)
        be[-1] = curve.fr_mont((-synth.fr_dot_mont(al[:-1], be[:-1])) * pow(fr_int(al[-1]), -1, curve.R) % curve.R)
        return lsa.normalize("g1", lsa.batch_exp("g1", G1, al)), lsa.normalize("g2", lsa.batch_exp("g2", G2, be))

    if on("pairing"):
        n = 1 << log2pairs
        ps, qs = planted(n)
        res, ms = best_of(lambda: lsa.pairing_product(ps, qs), max(2, reps))
        emit("pairing product: 2^%d Miller loops on pairs never seen + 1 final exponentiation (host buffers, planted == 1)" % log2pairs,
             np.array_equal(res, fq12_one),
             {"ms": ms, "pairings_per_s": n / ms * 1e3, "algorithmic_bytes": 192 * n + 384,
              "valu": valu(n * MILLER_FQ_MULS + FINAL_EXP_FQ_MULS, ms),
              "note": "whole call (upload + Miller loops + product tree + final exponentiation); more than 1024 pairs bypass the "
                      "line-table cache, so every call redoes the G2 arithmetic"})

    if on("cphad_verify"):
        # CPhad verify at d = 20: 187 Miller loops in 62 products, each followed by a final exponentiation
        sizes = [3] * 61 + [4]
        groups = [planted(m) for m in sizes]
        ps = np.concatenate([g[0] for g in groups]); qs = np.concatenate([g[1] for g in groups])
        edges = np.cumsum([0] + sizes).astype(np.uint64)
        nl, nf = len(ps), len(sizes)
        lsa.g2_table_cache(4096)
        res, ms = best_of(lambda: lsa.pairing_product_segments(ps, qs, edges), max(2, reps))
        emit("CPhad verifier shape: %d Miller loops in %d products + %d final exponentiations, every Q resident (keys)" % (nl, nf, nf),
             all(np.array_equal(r_, fq12_one) for r_ in res),
             {"ms": ms, "algorithmic_bytes": 192 * nl + 384 * nf, "valu": valu(nl * MILLER_FQ_MULS_RESIDENT + nf * FINAL_EXP_FQ_MULS, ms)})
        lsa.g2_table_cache(0)
        res, ms = best_of(lambda: lsa.pairing_product_segments(ps, qs, edges), max(2, reps))
        emit("CPhad verifier shape: %d Miller loops in %d products + %d final exponentiations, every Q fresh (table cache off)" % (nl, nf, nf),
             all(np.array_equal(r_, fq12_one) for r_ in res),
             {"ms": ms, "algorithmic_bytes": 192 * nl + 384 * nf, "valu": valu(nl * MILLER_FQ_MULS + nf * FINAL_EXP_FQ_MULS, ms)})
        lsa.g2_table_cache(4096)
        # one Miller loop and one whole check, as the reference's verifiers issue them (globl.h:94-105)
        P1, Q1 = ps[:1], qs[:1]
        ref = lsa.miller_loop(P1, Q1)
        res, ms = best_of(lambda: lsa.miller_loop(P1, Q1), max(3, reps))
        emit("one miller_loop on a resident Q (host buffers, blocking)", np.array_equal(res, ref), {"ms": ms, "valu": valu(MILLER_FQ_MULS_RESIDENT, ms)})
        p2, q2 = planted(2)
        off2 = np.array([0, 2], dtype=np.uint64)
        res, ms = best_of(lambda: lsa.pairing_terms(p2, off2, g2=q2), max(3, reps))
        emit("one pairing check of two terms (simple_pairing_check as one product), both Q resident", np.array_equal(res[0], fq12_one),
             {"ms": ms, "valu": valu(2 * MILLER_FQ_MULS_RESIDENT + FINAL_EXP_FQ_MULS, ms)})
    return out

#!/usr/bin/env python3
"""tools/cold_msm.py -- the FIRST multiExpMA of a process (GPU box): lsa_g1_msm / lsa_g2_msm on pageable host vectors
the library has never seen, which is the only call pattern the reference has (one SubspaceSnark::prove per process,
/root/reference/src/gadgets/subspace.cc:78-85; CommScheme::commit, src/prototools/commit.h:149-158).

Parent: runs `--runs` FRESH child processes per setting (environment overrides after `--env`, e.g.
`--env LSA_H2D=direct --env LSA_H2D_THREADS=6`), prints one JSON line per setting with every run's wall time and the
library's own split (lsa_msm_host_stats), median and p90.
Child (`--child`): lsa_init, the CPlink-prover inputs of bench.py (P = (a + i b) G as N + 2 Jacobian points followed by N
points at infinity, w = (0, rF, u)), built on the device and copied to fresh host arrays; ONE cold call, checked against
the identity; then the same call twice more (resident bases, plain layout)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def thp():
    try:
        return open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
    except OSError:
        return None


def host_array(np, shape, dtype, small_pages):
    """A zeroed host array; small_pages: an anonymous mapping with MADV_NOHUGEPAGE set BEFORE it is touched -- what a
    std::vector on the glibc heap looks like under transparent_hugepage=madvise / never (numpy asks for huge pages)."""
    if not small_pages:
        return np.zeros(shape, dtype=dtype)
    import ctypes, mmap
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    m = mmap.mmap(-1, nbytes)
    addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
    rc = ctypes.CDLL(None, use_errno=True).madvise(ctypes.c_void_p(addr), ctypes.c_size_t(nbytes), 15)   # MADV_NOHUGEPAGE
    assert rc == 0, "madvise failed"
    return np.frombuffer(m, dtype=dtype).reshape(shape)


def child(args):
    import numpy as np
    t0 = time.perf_counter()
    import legosnark_amd as lsa
    from legosnark_amd import curve, synth
    lsa.init(0)
    init_ms = (time.perf_counter() - t0) * 1e3
    group = args.group
    N = 1 << args.log2n
    npl = N + 2
    rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0xC9 ^ args.seed)
    a, b = rng.fr_int(), rng.fr_int()
    x = synth.arith_fr_mont(a, b, npl)
    w_vec = rng.uniform_fr(npl)
    w_vec[0] = 0
    G = curve.generator(group)
    width = 12 if group == "g1" else 24
    pts = lsa.batch_exp(group, G, x)
    P_host = host_array(np, (2 * N + 2, width), np.uint64, args.small_pages)
    if args.small_pages:
        w_small = host_array(np, w_vec.shape, np.uint64, True)
        w_small[...] = w_vec
        w_vec = w_small
    P_host[:npl] = np.asarray(pts).view(np.uint64).reshape(npl, width)
    one = curve.fq_mont(1)
    if group == "g1":
        P_host[npl:, 4:8] = one
    else:
        P_host[npl:, 8:12] = one
    keep_alive = [pts]        # (nothing large is freed before the timed call: unmapping a buffer the HIP runtime has pinned stalls the next submission)
    if args.free_temporaries:                                   # ... unless asked to: the download's destination goes back to the OS right here
        keep_alive.clear()
        del pts
    want = lsa.batch_exp(group, G, curve.fr_mont(synth.fr_dot_mont(w_vec, x)).reshape(1, 4))
    want_aff = lsa.normalize(group, want)
    lsa.synchronize()
    time.sleep(0.05)
    runs = []
    for i in range(3):
        w_i = w_vec if i == 0 else w_vec.copy()                 # later calls: a fresh scalar vector, the bases resident
        t_ = time.perf_counter()
        r_ = lsa.msm(group, P_host, w_i)
        dt = (time.perf_counter() - t_) * 1e3
        st = lsa.msm_host_stats()
        ok = bool(np.array_equal(lsa.normalize(group, r_.reshape(1, width)), want_aff))
        runs.append({"ms": round(dt, 3), "ok": ok, **{k: round(float(st[k]), 3) for k in ("h2d_scalars_ms", "fingerprint_wait_ms", "bases_prepare_ms", "msm_ms")},
                     "cache_hit": int(st["cache_hit"])})
    print(json.dumps({"cold_child": {"group": group, "log2n": args.log2n, "init_ms": round(init_ms, 1), "thp": thp(), "small_pages": bool(args.small_pages), "cold": runs[0], "second": runs[1], "third": runs[2]}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--group", default="g1")
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--free-temporaries", action="store_true", help="free the 96 / 192-MiB download buffer right before the timed call")
    ap.add_argument("--small-pages", action="store_true", help="host vectors on 4-KiB pages (MADV_NOHUGEPAGE)")
    ap.add_argument("--env", action="append", default=[], help="KEY=VALUE for the children (one setting); repeatable")
    ap.add_argument("--settings", action="append", default=[], help="a whole setting 'K=V,K=V' (repeatable): one line each")
    args = ap.parse_args()
    if args.child:
        return child(args)
    settings = [dict(kv.split("=", 1) for kv in s.split(",") if kv) for s in args.settings] or [dict(kv.split("=", 1) for kv in args.env)]
    for setting in settings:
        outs = []
        for r in range(args.runs):
            env = dict(os.environ, **setting)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--group", args.group, "--log2n", str(args.log2n), "--seed", str(r)] + (["--small-pages"] if args.small_pages else []) + (["--free-temporaries"] if args.free_temporaries else []),
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith('{"cold_child"')]
            if p.returncode != 0 or not line:
                outs.append({"error": (p.stderr or p.stdout)[-400:]})
                continue
            outs.append(json.loads(line[-1])["cold_child"])
        good = [o for o in outs if "cold" in o]
        cold = sorted(o["cold"]["ms"] for o in good)
        summary = {"setting": setting, "group": args.group, "log2n": args.log2n, "runs": len(outs), "small_pages": bool(args.small_pages), "free_temporaries": bool(args.free_temporaries), "thp": good[0]["thp"] if good else None,
                   "cold_ms_runs": [o["cold"]["ms"] for o in good], "import_and_lsa_init_ms": [o["init_ms"] for o in good],
                   "cold_ms_median": cold[len(cold) // 2] if cold else None,
                   "cold_ms_p90": cold[min(len(cold) - 1, int(0.9 * len(cold)))] if cold else None,
                   "all_ok": all(o["cold"]["ok"] and o["second"]["ok"] and o["third"]["ok"] for o in good) and len(good) == len(outs),
                   "cold_split": [o["cold"] for o in good], "second_ms": [o["second"]["ms"] for o in good], "third_ms": [o["third"]["ms"] for o in good],
                   "errors": [o["error"] for o in outs if "error" in o]}
        print(json.dumps({"cold_msm": summary}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- alt_bn128 G1 Pippenger MSM throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--total-log2n T]

One "step" = one multi-scalar multiplication through the C-ABI with bases (prepared affine,
64 B) and scalars (Montgomery Fr, 32 B) already resident in HBM.

* default (weak scaling): n = 2^20 point-scalar pairs per GPU (BASELINE.json configs[1]:
  "alt_bn128 G1 Pippenger MSM, n=2^20 random scalars").  For N > 1 (launched by
  torch.distributed.run, one rank per GPU) rank r owns the r-th contiguous 2^20-pair range of an
  N*2^20 MSM and the 96-byte Jacobian partials are combined with ONE RCCL all-gather + fold per
  step, issued by the library itself (lsa_msm_run_sharded_async, csrc/comm.hip).
* --total-log2n T (strong scaling, BASELINE.json configs[3] at T=24): ONE CPlink-prover MSM over
  n = 2^T + 2 pairs (w[0] = 0, src/gadgets/subspace.cc:78-85) split across the N ranks by
  libff's chunk rule (lsa_shard_range).  Runs with N > 1 also measure it after the weak loop
  and report it as "cplink_sharded" in the same JSON line.

Inputs follow SURVEY.md section 8(d): scalars uniform in [0, r) from splitmix64-seeded
xoshiro256** (seed 0x4C45474F534E4152); bases P_i = (a + i*b)*G1, handed to the library as
un-normalised Jacobian points (random Z) and, in a second run, normalised (Z = 1).  The result
of the timed loop's last step is checked by the known-discrete-log identity.

Prints ONE JSON line (rank 0) with the throughput, a `roofline` object for the dominant kernel
(bucket accumulation; 96 algorithmic bytes per pair against the HBM peak, plus the field-
multiplication rate against the measured integer ceiling, which is what really bounds it), a
`cpu_baseline` object (the oracle = libff-algorithm restatement, timed on this host) and a `configs`
block: the other BASELINE.json configs (G2 MSM 2^20, CPpoly d=20, the 2^12-pairing product, the CPhad
verifier shape, the CPlink prover MSM), each result-checked, each with its time, algorithmic bytes and
field-multiplication rate (legosnark_amd/benchcfg.py).  `unchanged_reference_binary`: the reference's own
`hadamard 20` example (unchanged source, built against the shim; a child process) with ITS timers
(`##had_sc ... Prove/Verify`, src/examples/hadamard.cc:98-105) and the time it spent inside the library.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks (one child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, started before this process
touches torch or the GPU) and relays rank 0's line; the line's n_gpus must equal N or the run fails.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
ALG_BYTES_PER_PAIR = 96        # 64 B affine point + 32 B scalar, read once
# The path is integer-VALU bound, so the roofline object also carries the field-multiplication
# rate of the dominant kernel against the measured chip-wide ceiling of the 29-bit-limb
# Montgomery product (tools/ubench_int.hip: 175 G mults/s at >= 4 waves/SIMD).
FMUL_PEAK_G = 175.0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_traffic(kernel_substr="k_accumulate", exclude="heavy", timeout_s=150, extra_args=("--no-configs",), expect_us=None):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC counters, collected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc
    passes (they do not fit one pass), each over a short child run of this same script; both
    counters are KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so
    bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (the factor is calibrated on streaming reads,
    not on this kernel's 64-byte gathers: the raw sum is reported next to it).  Returns a dict or
    None when rocprofv3 is unavailable / fails -- the bench line then carries traffic = null."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="lsa_pmc_")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [exe, "--pmc", ctr, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-host-path", "--no-pmc"] + list(extra_args)
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            db = sqlite3.connect(dbs[0])
            rows = db.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
                              "where counter_name = ? group by kernel_name", (ctr,)).fetchall()
            rows = [x for x in rows if kernel_substr in x[0] and exclude not in x[0]]
            if not rows:
                return None
            rows.sort(key=lambda x: -x[1])
            vals[ctr] = {"kib_mean": rows[0][2], "launches": rows[0][1], "kernel_us": rows[0][3] / 1e3, "kernel": rows[0][0][:60]}
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    f, w = vals["FETCH_SIZE"]["kib_mean"], vals["WRITE_SIZE"]["kib_mean"]
    return {"bytes_per_launch": 2 * f * 1024 + w * 1024, "raw_bytes_per_launch": (f + w) * 1024,
            "fetch_size_kib": f, "write_size_kib": w, "launches_sampled": [vals["FETCH_SIZE"]["launches"], vals["WRITE_SIZE"]["launches"]],
            "kernel_us_under_pmc": [vals["FETCH_SIZE"]["kernel_us"], vals["WRITE_SIZE"]["kernel_us"]],
            "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate child runs of bench.py; "
                      "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 half-count correction, calibrated on streaming reads)"}


def reference_binary_run(d=20):
    """The reference's UNCHANGED `hadamard` example (built against the shim by __graft_entry__.build() where
    /root/reference exists; build/ travels with the snapshot) run as a child process at d variables, with the
    reference's own timers (##had_sc ... Prove/Verify: N micros, src/examples/hadamard.cc:98-105) and the shim's
    accounting of the time spent inside the library (LSA_SHIM_STATS=1).  What a LegoSNARK user sees end to end."""
    import re
    import subprocess
    # the binary the reference's OWN CMakeLists.txt builds (its flags: OPT_FLAGS "-O3 -ggdb", -Wall -Wextra -Wfatal-errors,
    # /root/reference/CMakeLists.txt:27-33,54-64) against the drop-in packages; the Makefile build (-O3 -w) if that is absent
    rel = os.path.join("build", "reference_cmake", "src", "examples", "hadamard")
    how = "the reference's CMakeLists.txt, default configuration (-O3 -ggdb, MULTICORE=OFF)"
    if not os.path.exists(os.path.join(ROOT, rel)):
        rel, how = os.path.join("build", "reference", "hadamard"), "legosnark_amd/shim/Makefile (-O3)"
    exe = os.path.join(ROOT, rel)
    if not os.path.exists(exe):
        return {"program": rel, "status": "absent (built only where the reference's sources are)"}
    env = dict(os.environ, LSA_SHIM_STATS="1")
    t0 = time.perf_counter()
    try:
        r = subprocess.run([exe, str(d)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        return {"program": "%s %d" % (rel, d), "status": "timeout"}
    wall = time.perf_counter() - t0
    res = {"program": "%s %d (unchanged source, libff-compatible shim)" % (rel, d), "built_by": how, "status": "rc=%d" % r.returncode, "wall_s": round(wall, 3),
           "timers_ms": {}}
    for m in re.finditer(r"^##(\S+) (.*?): ([0-9.e+]+) micros", r.stdout, re.M):
        res["timers_ms"]["%s %s" % (m.group(1), m.group(2))] = round(float(m.group(3)) / 1e3, 3)
    # the reference's -DMULTICORE=ON configuration (-fopenmp -DMULTICORE=1: chunks = threads, lipmaa.cc's OpenMP loops)
    mc = os.path.join(ROOT, "build", "reference_cmake_mc", "src", "examples", "hadamard")
    if os.path.exists(mc):
        try:
            nthr = str(min(8, len(os.sched_getaffinity(0))))
            m_ = subprocess.run([mc, str(d)], env=dict(os.environ, OMP_NUM_THREADS=nthr), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            tm = {"%s %s" % (x.group(1), x.group(2)): round(float(x.group(3)) / 1e3, 3) for x in re.finditer(r"^##(\S+) (.*?): ([0-9.e+]+) micros", m_.stdout, re.M)}
            res["multicore_build"] = {"program": "build/reference_cmake_mc/src/examples/hadamard %d (-DMULTICORE=ON)" % d, "rc": m_.returncode, "omp_threads": int(nthr),
                                      "nchunks_printed": sorted(set(re.findall(r"NCHUNKS : (\d+)", m_.stdout))),
                                      "timers_ms": {k: v for k, v in tm.items() if "TOTAL" in k or "lipmaa" in k}}
        except Exception as e:
            res["multicore_build"] = {"error": str(e)[:200]}
    # rc 0 says little (the reference's verifiers print nothing when a check fails): the verifier side of
    # the same build is checked beside it -- legosnark_amd/shim/checks/pairing_check.cc runs the reference's own
    # simple_pairing_check on true and false statements and its unchanged CPPoly::verify, deferred and call by call
    chk = os.path.join(ROOT, "build", "reference", "pairing_check")
    if os.path.exists(chk):
        try:
            c = subprocess.run([chk, "12"], env=dict(os.environ, LSA_SEED="11"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
            last = json.loads(c.stdout.strip().splitlines()[-1]) if c.stdout.strip() else {}
            res["verifier_check"] = {"program": "build/reference/pairing_check 12", "rc": c.returncode, "failures": last.get("failures"),
                                     "cppoly_verify": last.get("cppoly_verify"), "cppoly_verify_ms": last.get("cppoly_verify_ms")}
        except Exception as e:                               # a broken check is reported, not hidden
            res["verifier_check"] = {"program": "build/reference/pairing_check 12", "error": str(e)[:200]}
    # the same prover with its vectors resident on the device (legosnark_amd/shim/checks/resident_prover_check.cc): CPHad's
    # prove written against the C-ABI, fed the inputs and the random stream of the unchanged CPHad::prove, every proof
    # element compared, the resident proof accepted by the reference's verifier
    rp = os.path.join(ROOT, "build", "reference", "resident_prover_check")
    if os.path.exists(rp):
        try:
            c = subprocess.run([rp, str(d), "squares"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            js = [l for l in c.stdout.splitlines() if l.startswith('{"resident_prover"')]
            res["resident_prover"] = dict(json.loads(js[-1])["resident_prover"], rc=c.returncode) if js else {"rc": c.returncode, "error": (c.stderr or c.stdout)[-300:]}
        except Exception as e:
            res["resident_prover"] = {"error": str(e)[:200]}
    for line in r.stderr.splitlines():
        if line.startswith('{"lsa_shim_stats"'):
            st = json.loads(line)["lsa_shim_stats"]
            res["inside_library_ms"] = st["inside_ms"]
            res["process_ms"] = st["process_ms"]
            res["library_calls"] = {k: v for k, v in st.items() if isinstance(v, dict) and v.get("calls")}
    return res


def configs_summary(configs, ref_run):
    """One short entry per BASELINE / SURVEY 8(f) config: [ms, fraction, of what].  valu = field products/s over the measured
    175 G/s ceiling, hbm = algorithmic bytes/s over 8 TB/s."""
    def r3(x):
        return None if x is None else round(float(x), 3)
    out = {}
    for c in configs:
        name = c["config"]
        if "error" in c:
            out[name[:24]] = "check failed"
        elif name.startswith("G2 MSM"):
            out["g2_msm_2^20"] = [r3(c["ms"]), c["valu"]["frac"], "valu"]
        elif name.startswith("CPpoly"):
            out["cppoly_d20_commit"] = [r3(c["commit_ms"]), c["commit_valu"]["frac"], "valu"]
            out["cppoly_d20_prove"] = [r3(c["prove_total_ms"]), None, "fold %.3f + ladder %.3f ms" % (c["prove_fold_ms"], c["prove_msm_ladder_segmented_ms"] or c["prove_msm_ladder_39_calls_ms"])]
        elif name.startswith("Fr fold"):
            out["fr_witness_d24"] = [r3(c["witness_ms"]), c["hbm"]["witness_frac_algorithmic"], "hbm"]
            out["fr_eval_mle_d24"] = [r3(c["eval_mle_ms"]), c["hbm"]["eval_mle_frac_algorithmic"], "hbm"]
            rr = c["resident_prover_round_0"]
            out["fr_sumcheck_round_2^23"] = [r3(rr["sumcheck_round_ms"]), rr["sumcheck_round_frac_of_hbm"], "hbm"]
            out["fr_push_randomness_2^23"] = [r3(rr["push_randomness_ms"]), rr["push_randomness_frac_of_hbm"], "hbm"]
        elif name.startswith("NTT"):
            out["ntt_2^24"] = [r3(c["fft_ms"]), c["valu"]["frac"], "valu"]
            out["ntt_step_2^23+2^22"] = [r3(c["step_domain_2^23+2^22"]["fft_ms"]), c["step_domain_2^23+2^22"]["frac_of_hbm_algorithmic"], "hbm"]
        elif name.startswith("pairing product"):
            out["pairing_2^12_fresh+1_final_exp"] = [r3(c["ms"]), c["valu"]["frac"], "valu"]
        elif name.startswith("CPhad verifier shape") and "resident" in name:
            out["cphad_verify_resident_Q"] = [r3(c["ms"]), c["valu"]["frac"], "valu"]
        elif name.startswith("CPhad verifier shape"):
            out["cphad_verify_fresh_Q"] = [r3(c["ms"]), c["valu"]["frac"], "valu"]
        elif name.startswith("one miller_loop"):
            out["one_miller_loop"] = [r3(c["ms"]), c["valu"]["frac"], "valu"]
        elif name.startswith("one pairing check"):
            out["one_pairing_check"] = [r3(c["ms"]), c["valu"]["frac"], "valu"]
        elif "N=2^24+2" in name:
            out["cplink_prover_2^24+2_one_gpu"] = [r3(c["ms"]), c["valu"]["frac"], "valu", {"pairs_per_s": round(c["pairs_per_s"]), "table_GB": round(c["table_bytes"] / 1e9, 1)}]
    if ref_run:
        tm = ref_run.get("timers_ms", {})
        out["unchanged_hadamard_20"] = {"prove_ms": tm.get("had_sc TOTAL Prove"), "verify_ms": tm.get("had_sc TOTAL Verify"), "inside_library_ms": ref_run.get("inside_library_ms")}
        rp = ref_run.get("resident_prover")
        if rp:
            out["resident_prover_d20"] = {k: rp.get(k) for k in ("prove_ms", "reference_prove_ms", "proof_equal", "reference_verifier_accepts")}
    out["format"] = "[ms, fraction, valu = of 175 G field products/s | hbm = algorithmic bytes over 8 TB/s]; every entry result-checked in this run"
    return out


def being_profiled():
    """True under rocprofv3 / rocprof (their preloaded tool library): no nested profiler runs then."""
    env = os.environ
    return any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in env) or "rocprof" in env.get("LD_PRELOAD", "") or \
        "rocprofiler" in env.get("LD_PRELOAD", "") or "rocprofiler" in env.get("HSA_TOOLS_LIB", "")


def launch_ranks(args, argv):
    """--gpus N > 1 without a launcher: start N ranks as ONE child process tree (torch.distributed.run), relay
    rank 0's JSON line, fail unless it reports n_gpus == N.  Runs before this process imports torch or touches the
    GPU; the parent never initialises a device (no exec from a GPU-initialised process either)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    if args.dry_launch:
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if r.returncode != 0 or line is None:
        sys.stderr.write("bench.py: the %d-rank run failed (exit %d)\n%s\n" % (args.gpus, r.returncode, r.stdout[-2000:]))
        return r.returncode or 1
    got = json.loads(line).get("n_gpus")
    if got != args.gpus:
        sys.stderr.write("bench.py: asked for %d GPUs, the ranks report n_gpus = %r\n" % (args.gpus, got))
        return 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--total-log2n", type=int, default=0,
                    help="strong scaling: one CPlink-prover MSM over 2^T + 2 pairs split across the ranks")
    ap.add_argument("--cpu-sample-log2", type=int, default=20,
                    help="pairs of the same workload timed on the host CPU (2^k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (the other BASELINE.json configs)")
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1: print the launch command and exit")
    ap.add_argument("--only-configs", default="", help="comma-separated names of legosnark_amd.benchcfg configs: run only those (the PMC child "
                                                       "passes behind a config's counter traffic use this) and print their lines")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, [a for a in sys.argv[1:] if a != "--dry-launch"]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("WORLD_SIZE (%s) != --gpus (%d): this run would not measure what it was asked to" % (os.environ.get("WORLD_SIZE", "unset"), args.gpus))

    if args.only_configs:
        import numpy as np
        import torch
        import legosnark_amd as lsa
        from legosnark_amd import benchcfg
        lsa.init(0)
        torch.cuda.set_device(0)
        print(json.dumps({"configs": benchcfg.measure(lsa, torch, np, torch.device("cuda", 0), log2n=args.log2n, d=20, log2pairs=12, reps=3,
                                                      only=set(args.only_configs.split(",")))}), flush=True)
        return

    # roofline.traffic: two short profiled child runs of this script, before this process
    # initialises the GPU (or even imports torch)
    traffic = None
    pairing_traffic = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_pmc and not args.total_log2n and args.log2n == 20 and not being_profiled():
        traffic = pmc_traffic()
        if not args.no_configs:
            # the same two counter passes for the dominant kernel of BASELINE configs[4] (2^12 fresh pairs: k_miller_fused)
            pairing_traffic = pmc_traffic(kernel_substr="k_miller_fused", exclude="\x00", extra_args=("--only-configs", "pairing"))

    # The FIRST multiExpMA of a process -- all the reference ever makes per key (one prove per process,
    # src/gadgets/subspace.cc:78-85): five fresh child processes (tools/cold_msm.py), each its own lsa_init, the CPlink
    # prover's P and w on pageable host memory, every result checked by the identity.  Run BEFORE this process creates
    # its GPU context: a prover has the device to itself (beside a second live context every submission of the child
    # waits for the hardware scheduler: +1.5 ms, measured).
    cold_runs = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_host_path and not args.total_log2n and not being_profiled():
        try:
            import subprocess
            cm = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cold_msm.py"), "--runs", "5", "--log2n", str(args.log2n)],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
            cl = [json.loads(l)["cold_msm"] for l in cm.stdout.splitlines() if l.startswith('{"cold_msm"')]
            if cl:
                c0 = cl[0]
                cold_runs = {
                    "cold_ms_runs": c0["cold_ms_runs"], "cold_ms_median": c0["cold_ms_median"], "cold_ms_p90": c0["cold_ms_p90"],
                    "cold_runs_all_checked": c0["all_ok"],
                    "cold_split_of_the_median_run": sorted(c0["cold_split"], key=lambda t: t["ms"])[len(c0["cold_split"]) // 2] if c0["cold_split"] else None,
                    "cold_second_call_ms": c0["second_ms"], "import_and_lsa_init_ms": c0["import_and_lsa_init_ms"], "transparent_hugepage": c0["thp"],
                    "cold_note": "cold_ms_runs: the first lsa_g1_msm of five fresh processes that have the GPU to themselves (tools/cold_msm.py); cold_ms: "
                                 "the same call inside this long-lived process right after its cache was cleared (1.8 GB of copies freed)",
                }
            else:
                cold_runs = {"cold_ms_runs": None, "cold_error": (cm.stderr or cm.stdout)[-300:]}
        except Exception as e:                               # reported, not hidden
            cold_runs = {"cold_ms_runs": None, "cold_error": str(e)[:300]}

    # ... and the reference's unchanged example binaries, for the same reason before this process has a GPU context
    ref_run = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_configs and not args.total_log2n and args.log2n == 20 and not being_profiled():
        ref_run = reference_binary_run()

    import numpy as np
    import torch
    import legosnark_amd as lsa
    from legosnark_amd import curve, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # debug hooks to exercise the multi-process path on a one-GPU box: every rank on cuda:0 and
    # gloo collectives (RCCL refuses two ranks on one device); default: one rank per GPU, RCCL
    if os.environ.get("LSA_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("LSA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lsa.init(local_rank)
    dist = None
    if world > 1 and backend != "nccl":
        import torch.distributed as dist
        dist.init_process_group(backend)
        comm_kind = "torch"
    elif world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        # the library's own RCCL communicator: rank 0 creates the id, torch broadcasts the 128 bytes
        idt = torch.zeros(lsa.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(lsa.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        comm_kind = "capi"
        try:
            if os.environ.get("LSA_BENCH_COMM", "capi") != "capi":
                raise lsa.LsaError("torch.distributed exchange requested")
            lsa.comm_init(rank, world, bytes(idt.cpu().numpy().tobytes()))
            ok = 1
        except lsa.LsaError as e:
            sys.stderr.write("rank %d: C-ABI communicator unavailable (%s)\n" % (rank, e))
            ok = 0
        # every rank must take the same path
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            comm_kind = "torch"              # round-1 path: torch.distributed all-gather + lsa_g1_sum_on
            if ok:
                lsa.comm_destroy()

    if world == 1:
        comm_kind = "none"
    G1 = curve.generator("g1")

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        lsa.synchronize()
        torch.cuda.synchronize()

    def affine_of(pt_host):
        return lsa.normalize("g1", np.ascontiguousarray(pt_host, dtype=np.uint64).reshape(1, 12))[0]

    def k_times_g(k):
        return affine_of(lsa.batch_exp("g1", G1, curve.fr_mont(k).reshape(1, 4))[0])

    def sum_over_ranks(k):
        if world == 1:
            return k
        parts = [None] * world
        dist.all_gather_object(parts, int(k))
        return sum(parts) % curve.R

    class Workload:
        """Rank-local slice [lo, hi) of an MSM over n_total pairs: bases P_i = (a + i*b)*G,
        scalars uniform in [0, r); `cplink` zeroes the first scalar (w[0] = 0, commit.h:152)."""

        def __init__(self, n_total, lo, hi, cplink=False):
            rng = synth.Xoshiro256ss(seed=synth.SEED)           # the same a, b on every rank
            self.a, self.b = rng.fr_int(), rng.fr_int()
            srng = synth.Xoshiro256ss(seed=synth.SEED + 1 + rank)
            self.n = hi - lo
            self.x = synth.arith_fr_mont(self.a + lo * self.b, self.b, self.n)
            self.s = srng.uniform_fr(self.n)
            if cplink and lo == 0 and self.n:
                self.s[0] = 0
            t0 = time.perf_counter()
            self.bases_jac = lsa.batch_exp("g1", G1, to_dev(self.x))      # un-normalised Jacobian (random Z)
            self.B = lsa.Bases("g1", self.bases_jac, on_device=True)
            lsa.synchronize()
            self.setup_s = time.perf_counter() - t0
            self.d_s = to_dev(self.s)
            self.outs = torch.zeros((4, 12), dtype=torch.int64, device=dev)
            self.calls = 0
            self.job = None
            if world > 1 and comm_kind == "torch":
                from legosnark_amd import sharded
                self.job = sharded.make_gpu_sharded(lsa, "g1", self.B, world, rank, dist=dist)
            torch.cuda.synchronize()

        def step(self):
            if self.job is not None:
                return self.job.run(self.d_s)
            out = self.outs[self.calls % 4]
            self.calls += 1
            if world > 1:
                self.B.msm_sharded_async(self.d_s, out)
            else:
                self.B.msm_async(self.d_s, out)
            return out

        def finish(self):
            if self.job is not None:
                self.job.side.synchronize()
            elif world > 1:
                lsa.comm_join()
            lsa.synchronize()

        def expected(self):
            return k_times_g(sum_over_ranks(synth.fr_dot_mont(self.s, self.x)))

        def timed(self, steps, warmup, profile=False):
            for _ in range(warmup):
                self.step()
            self.finish()
            barrier()
            if profile:
                lsa.profile_enable(True)     # per-stage HIP events on the library stream; no host sync
            t0 = time.perf_counter()
            for _ in range(steps):
                res = self.step()
            self.finish()
            barrier()
            elapsed = time.perf_counter() - t0
            stages = lsa.profile_last_msm() if profile else None
            if profile:
                lsa.profile_enable(False)
            if world > 1:
                tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                elapsed = float(tmax.item())
            got = affine_of(res.cpu().numpy().view(np.uint64))
            return elapsed, stages, bool(np.array_equal(got, self.expected()))

    strong = args.total_log2n > 0
    if strong:
        n_total = (1 << args.total_log2n) + 2
        lo, hi = lsa.shard_range(n_total, world, rank)
        wl = Workload(n_total, lo, hi, cplink=True)
    else:
        n = 1 << args.log2n
        n_total = world * n
        wl = Workload(n_total, rank * n, (rank + 1) * n)
    elapsed, stages, checked = wl.timed(args.steps, args.warmup, profile=True)
    value = n_total * args.steps / elapsed
    n_local = wl.n
    fm_per_pair = wl.B.field_mults_per_pair(wl.n)
    table_windows = wl.B.table_windows()

    # single-call latency (no overlap with a following call), for the record
    lat = []
    for _ in range(5):
        barrier()
        tl = time.perf_counter()
        wl.step()
        wl.finish()
        lat.append(time.perf_counter() - tl)
    latency_ms = sorted(lat)[len(lat) // 2] * 1e3

    extra = {}
    if world == 1 and not strong:
        # second run of SURVEY 8(d): the same bases normalised (Z = 1) must give the same point
        z1 = lsa.normalize("g1", wl.bases_jac.cpu().numpy().view(np.uint64))
        B1 = lsa.Bases("g1", z1)
        r_z1 = affine_of(B1.msm(wl.d_s))
        r_rz = affine_of(wl.B.msm(wl.d_s))
        extra["normalised_bases_run_matches"] = bool(np.array_equal(r_z1, r_rz))
        B1.close()
        del z1

    # "CPlink prover ms" (the second half of BASELINE.json's metric): SubspaceSnark::prove
    # (src/gadgets/subspace.cc:78-85) is ONE multiExpMA over the N+2 meaningful CRS points with
    # w = (0, rF, u) (src/examples/cplink.cc:107-108).
    cplink_ms = None
    host_path = None
    if world == 1 and not strong:
        N = 1 << args.log2n
        npl = N + 2
        rng = synth.Xoshiro256ss(seed=synth.SEED ^ 0xC9)
        a, b = rng.fr_int(), rng.fr_int()
        x = synth.arith_fr_mont(a, b, npl)
        w_vec = rng.uniform_fr(npl)
        w_vec[0] = 0
        want = k_times_g(synth.fr_dot_mont(w_vec, x))
        crs_dev = lsa.batch_exp("g1", G1, to_dev(x))
        # (i) resident CRS handle + device-resident witness: what a prover integrated through
        #     lsa_g1_bases_create / lsa_msm_run sees
        crs = lsa.Bases("g1", crs_dev, on_device=True)
        d_w = to_dev(w_vec)
        torch.cuda.synchronize()
        got = affine_of(crs.msm(d_w))
        tl = []
        for _ in range(5):
            t_ = time.perf_counter()
            crs.msm(d_w)
            tl.append(time.perf_counter() - t_)
        cplink_ms = sorted(tl)[len(tl) // 2] * 1e3
        extra["cplink_prover_checked"] = bool(np.array_equal(got, want))
        crs.close()
        if not args.no_host_path:
            # (ii) exactly what the unchanged reference delivers: multiExpMA(crs->P, w) with host
            #      std::vectors (src/utils/globl.h:74-77) -> lsa_g1_msm on pageable host memory.
            #      P has 2N+2 entries, the trailing N are infinity; n = min(|P|, |w|) = N+2.
            P_host = np.zeros((2 * N + 2, 12), dtype=np.uint64)
            P_host[:npl] = crs_dev.cpu().numpy().view(np.uint64)
            P_host[npl:, 4:8] = curve.fq_mont(1)
            del crs_dev
            lsa.crs_cache_clear()
            runs = []
            first_table = None
            for i in range(80):
                t_ = time.perf_counter()
                r_ = lsa.msm("g1", P_host, w_vec)
                dt = (time.perf_counter() - t_) * 1e3
                st = lsa.msm_host_stats()
                runs.append((dt, st, bool(np.array_equal(affine_of(r_), want))))
                if st["table"] and first_table is None:
                    first_table = i
                if first_table is not None and i >= first_table + 5:
                    break
            keys = ("cache_hit", "table", "table_building", "h2d_scalars_ms", "fingerprint_wait_ms", "bases_prepare_ms", "msm_ms")
            building = sorted(t[0] for t in runs[1:first_table]) if first_table else []
            warm_runs = runs[first_table:] if first_table is not None else runs[2:]
            warm = sorted(warm_runs, key=lambda t: t[0])[len(warm_runs) // 2]
            host_path = {
                "call": "lsa_g1_msm(host P[2N+2], host w[N+2]) = multiExpMA(crs->P, w), N=2^%d, pageable memory" % args.log2n,
                "cold_ms": runs[0][0], "second_ms": runs[1][0],
                "ms_on_the_plain_layout": building[len(building) // 2] if building else None,
                "ms_max_while_the_copies_are_built": max(t[0] for t in runs[1:first_table] if t[1]["table_building"]) if first_table and any(t[1]["table_building"] for t in runs[1:first_table]) else None,
                "calls_until_table": first_table, "warm_ms": warm[0],
                "resident_bytes_with_copies": lsa.crs_cache_stats()["resident_bytes"],
                "cold": {k: runs[0][1][k] for k in keys}, "second": {k: runs[1][1][k] for k in keys}, "warm": {k: warm[1][k] for k in keys},
                "all_results_checked": all(t[2] for t in runs), "calls": len(runs),
                "note": "cold = upload + normalise + MSM; second = resident, plain layout; warm = copies resident (built in the background "
                        "from the 23rd hit on), every byte of P re-fingerprinted while w uploads",
            }
            lsa.crs_cache_clear()
            del P_host
            if cold_runs is not None:
                host_path.update(cold_runs)

    # BASELINE.json configs[3] shape whenever there is more than one rank: one 2^24+2 CPlink MSM
    cplink_sharded = None
    if world > 1 and not strong:
        T = int(os.environ.get("LSA_BENCH_TOTAL_LOG2N", "24"))
        nt = (1 << T) + 2
        lo, hi = lsa.shard_range(nt, world, rank)
        wl.B.close()
        del wl
        w4 = Workload(nt, lo, hi, cplink=True)
        el4, _, ok4 = w4.timed(10, 2)
        cplink_sharded = {"workload": "CPlink prover MSM n=2^%d+2 split over %d GPUs (lsa_shard_range), 1 RCCL all-gather of 96-B partials" % (T, world),
                          "ms_per_prove": el4 / 10 * 1e3, "pairs_per_s": nt * 10 / el4, "result_checked": ok4, "steps": 10}

    out = None
    if rank == 0:
        acc_ms = stages["accumulate"]
        fmuls_per_pair = fm_per_pair
        if traffic:
            # the counter sample must be the launches this line is about: the same kernel at the same size (its
            # duration under the serialising PMC passes within 15 % of the live HIP-event figure) -- else null
            us = traffic["kernel_us_under_pmc"]
            traffic["kernel_us_live"] = acc_ms * 1e3
            if acc_ms <= 0 or any(abs(u - acc_ms * 1e3) > 0.15 * acc_ms * 1e3 for u in us):
                traffic["rejected"] = "kernel duration under PMC differs from the live figure by more than 15 %: not the same launches"
            else:
                traffic["ratio_to_algorithmic_bytes"] = {"corrected": traffic["bytes_per_launch"] / (n_local * ALG_BYTES_PER_PAIR),
                                                         "raw": traffic["raw_bytes_per_launch"] / (n_local * ALG_BYTES_PER_PAIR)}
        achieved = n_local * ALG_BYTES_PER_PAIR / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        gf = n_local * fmuls_per_pair / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        if strong:
            metric = "alt_bn128 G1 MSM point-scalar pairs/s, CPlink prover n=2^%d+2" % args.total_log2n
            workload = "CPlink prover at n=2^%d (+2), MSM sharded across %d GPU(s), RCCL all-gather of Jacobian partials" % (args.total_log2n, world)
        else:
            metric = "alt_bn128 G1 MSM point-scalar pairs/s at n=2^%d" % args.log2n
            workload = "alt_bn128 G1 Pippenger MSM, n=2^%d random scalars per GPU" % args.log2n
        out = {
            "metric": metric,
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u32 limbs (254-bit Montgomery integers)", "data": "synthetic",
            "config": {"workload": workload,
                       "inputs": "scalars uniform in [0,r), xoshiro256** seed 0x4C45474F534E4152; bases (a+i*b)*G1, un-normalised Jacobian",
                       "digits_per_scalar": fm_per_pair // 10 if table_windows else None,
                       "pre_shifted_copies": table_windows if table_windows else None,
                       "pipeline": ("wide windows over %d pre-shifted copies of the resident bases: %d bucket additions per pair, one shared bucket space"
                                    % (table_windows, fm_per_pair // 10)) if table_windows else "signed 16-bit windows + GLV",
                       "sharding": ("index ranges (libff chunk split), 1 RCCL all-gather of 96-B partials via "
                                    + ("lsa_msm_run_sharded_async (C-ABI, csrc/comm.hip)" if comm_kind == "capi" else "torch.distributed + lsa_g1_sum_on"))
                       if world > 1 else "single GPU"},
            "result_checked_by_identity": checked,
            "roofline": {"kernel": "k_accumulate<CurveG1>", "bound": "hbm", "limited_by": "integer VALU (see valu)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic["bytes_per_launch"] if traffic and "rejected" not in traffic else None, "traffic_detail": traffic,
                         "algorithmic_bytes_per_launch": n_local * ALG_BYTES_PER_PAIR,
                         "kernel_ms": acc_ms, "calls_averaged": stages["calls"],
                         "valu": {"achieved_Gfmul_s": gf, "peak_Gfmul_s": FMUL_PEAK_G, "frac": gf / FMUL_PEAK_G,
                                  "field_mults_per_pair": fmuls_per_pair,
                                  "note": "254-bit field multiplications/s in k_accumulate vs the microbenchmarked "
                                          "ceiling of the 9x29-bit Montgomery product on this chip"}},
            "stage_ms": {k: round(v, 4) for k, v in stages.items() if k != "calls"},
            "single_call_latency_ms": latency_ms,
            "cplink_prover_ms": cplink_ms,
            "cplink_prover_host_path_ms": host_path,
            "cplink_sharded": cplink_sharded,
            "host": {"cpu_model": cpu_model(), "nproc": os.cpu_count()},
            "pipelining": "the tail (reduce+fold) of step i runs on an internal stream and overlaps the front of "
                          "step i+1; stage_ms are measured under that overlap; LSA_NO_OVERLAP=1 serialises",
        }
        out.update(extra)
        if not args.no_configs and world == 1 and not strong and args.log2n == 20:
            # the other BASELINE.json configs, each result-checked (a failed check carries "error" and no time)
            from legosnark_amd import benchcfg
            out["configs"] = benchcfg.measure(lsa, torch, np, dev, log2n=20, d=20, log2pairs=12, reps=5,
                                              only={"g2_msm", "cppoly", "pairing", "cphad_verify", "fr_fold", "ntt", "cplink_2pow24"})
            for c in out["configs"]:
                if c["config"].startswith("pairing product") and "error" not in c:
                    # 192 B in per pair; 384 B out per workgroup of five pairs (the partial products the tree continues from)
                    c["traffic"] = pairing_traffic["bytes_per_launch"] if pairing_traffic else None
                    c["traffic_detail"] = pairing_traffic
            if any("error" in c for c in out["configs"]):
                raise SystemExit("bench.py: a config's result check failed: %s" % [c for c in out["configs"] if "error" in c])
        if ref_run is not None:
            out["unchanged_reference_binary"] = ref_run
            vc = out["unchanged_reference_binary"].get("verifier_check")
            if vc and (vc.get("rc") != 0 or vc.get("failures") != 0 or "error" in vc):
                raise SystemExit("bench.py: the verifier check beside the unchanged reference binary failed: %s" % vc)
            rp = ref_run.get("resident_prover")
            if rp and (rp.get("rc") != 0 or rp.get("proof_equal") is not True):
                raise SystemExit("bench.py: the resident prover's proof differs from the unchanged reference prover's: %s" % rp)
        # every config once more in a few hundred bytes, close to the END of the line: whoever keeps only the tail of this
        # process's output still sees one time and one fraction per config (BENCH_r05's tail began inside the pairing entry)
        out["configs_summary"] = configs_summary(out.get("configs") or [], ref_run)
        if not args.no_cpu_baseline and world == 1 and not strong:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as o   # the checker, timed as the reported CPU baseline
            ns = 1 << min(args.cpu_sample_log2, args.log2n)
            hb = wl.bases_jac[:ns].cpu().numpy().view(np.uint64)
            hs = wl.s[:ns]
            tc = time.perf_counter()
            ref = o.multi_exp("g1", hb, hs, chunks=1, threads=0, mode="mixed")
            tc = time.perf_counter() - tc
            got = wl.B.msm(wl.d_s[:ns], n=ns)
            same = o.g1_canonical_affine(ref) == o.g1_canonical_affine(got)
            out["cpu_baseline"] = {
                "value": ns / tc, "unit": "pairs/s", "cores": 1, "kind": "port",
                "sample": "first 2^%d pairs of the same workload, libff-algorithm restatement "
                          "(multi_exp_with_mixed_addition<BDLO12>, chunks=1, gcc -O3), %d host cores present (%s)"
                          % (min(args.cpu_sample_log2, args.log2n), os.cpu_count(), cpu_model()),
                "seconds": tc, "gpu_result_matches_cpu": bool(same),
            }
            # libff's MULTICORE build: chunks = threads (src/utils/globl.h:67-69), here every host core
            # the process may use
            try:
                nthr = len(os.sched_getaffinity(0))
            except AttributeError:
                nthr = os.cpu_count() or 1
            if nthr > 1:
                tm = time.perf_counter()
                refm = o.multi_exp("g1", hb, hs, chunks=nthr, threads=nthr, mode="mixed")
                tm = time.perf_counter() - tm
                out["cpu_baseline"]["multicore"] = {
                    "value": ns / tm, "unit": "pairs/s", "cores": nthr, "seconds": tm,
                    "sample": "same pairs, chunks = threads = %d (libff MULTICORE chunking)" % nthr,
                    "matches_single_core": o.g1_canonical_affine(refm) == o.g1_canonical_affine(ref),
                }
        # One line, the long blocks FIRST: whoever keeps only the tail of this process's stdout still sees the metric, the
        # roofline object, the host-path split and the CPU baseline (round 3's driver record lost those to truncation).
        tail_keys = ("cplink_prover_host_path_ms", "stage_ms", "cpu_baseline", "roofline", "configs_summary", "config", "metric", "value", "unit", "n_gpus", "steps",
                     "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "result_checked_by_identity",
                     "single_call_latency_ms", "cplink_prover_ms")
        ordered = {k: v for k, v in out.items() if k not in tail_keys}
        ordered.update({k: out[k] for k in tail_keys if k in out})
        print(json.dumps(ordered), flush=True)
    if world > 1:
        dist.barrier()
        if comm_kind == "capi":
            lsa.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

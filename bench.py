#!/usr/bin/env python3
"""bench.py -- alt_bn128 G1 Pippenger MSM throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one multi-scalar multiplication over n = 2^20 point-scalar pairs per GPU
(BASELINE.json configs[1]: "alt_bn128 G1 Pippenger MSM, n=2^20 random scalars"), through
the C-ABI (lsa_msm_run_async), with bases (affine, 64 B) and scalars (Montgomery Fr, 32 B)
already resident in HBM.  For N > 1 (launched by torch.distributed.run, one rank per GPU)
every rank runs its own 2^20-pair slice of an N*2^20 MSM and the 96-byte Jacobian partials
are combined with one RCCL all-gather + fold per step (weak scaling).

Prints ONE JSON line (rank 0) with the throughput, a `roofline` object for the dominant
kernel (bucket accumulation; HBM-bound accounting of 96 algorithmic bytes per pair) and a
`cpu_baseline` object (the oracle = libff-algorithm restatement, timed on this host).
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
FMULS_PER_PAIR = 16 * 10 + 8   # 16 mixed adds (8M+2S) + 8 beta-multiplies (GLV) per pair


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--cpu-sample-log2", type=int, default=20,
                    help="pairs of the same workload timed on the host CPU (2^k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import legosnark_amd as lsa
    from legosnark_amd import curve, sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # debug hooks to exercise the multi-process path on a one-GPU box: every rank on cuda:0 and
    # a gloo collective (default: one rank per GPU, backend "nccl" = RCCL over xGMI)
    if os.environ.get("LSA_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("LSA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lsa.init(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n = 1 << args.log2n
    # ---- synthetic workload, generated on the GPU by the product's own batch_exp kernel:
    # bases P_i = x_i * G1 for random x_i (un-normalised Jacobian, libff layout), then
    # normalised to affine by lsa_g1_bases_create; scalars are random Montgomery
    # representatives < 2^252 < r, i.e. uniformly spread field elements.
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x4C45474F + rank)

    def random_fr(count):
        t = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device=dev, generator=gen)
        t[:, 3] &= (1 << 60) - 1
        return t.contiguous()

    x = random_fr(n)
    bases_jac = lsa.batch_exp("g1", curve.generator("g1"), x)
    B = lsa.Bases("g1", bases_jac, on_device=True)
    d_scalars = random_fr(n)
    torch.cuda.synchronize()
    job = sharded.make_gpu_sharded(lsa, "g1", B, world, rank, dist=dist)

    def barrier():
        if world > 1:
            dist.barrier()
        lsa.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        job.run(d_scalars)
    barrier()
    lsa.profile_enable(True)     # per-stage HIP events on the library stream; no host sync
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = job.run(d_scalars)
    barrier()
    t1 = time.perf_counter()
    stages = lsa.profile_last_msm()
    lsa.profile_enable(False)
    elapsed = t1 - t0
    # single-call latency (no overlap with a following call), for the record
    lat = []
    for _ in range(5):
        barrier()
        tl = time.perf_counter()
        job.run(d_scalars)
        lsa.synchronize()
        lat.append(time.perf_counter() - tl)
    latency_ms = sorted(lat)[len(lat) // 2] * 1e3
    # the second half of BASELINE.json's metric, "CPlink prover ms": SubspaceSnark::prove
    # (src/gadgets/subspace.cc:78-85) is ONE multiExpMA over the N+2 meaningful CRS points with
    # w = (0, rF, u) (src/examples/cplink.cc:107-108); timed as a blocking lsa_msm_run call on a
    # resident CRS, host result included.
    cplink_ms = None
    if world == 1:
        npl = n + 2
        crs = lsa.Bases("g1", lsa.batch_exp("g1", curve.generator("g1"), random_fr(npl)), on_device=True)
        w_vec = random_fr(npl)
        w_vec[0] = 0
        torch.cuda.synchronize()
        crs.msm(w_vec)
        tl = []
        for _ in range(5):
            t_ = time.perf_counter()
            crs.msm(w_vec)
            tl.append(time.perf_counter() - t_)
        cplink_ms = sorted(tl)[len(tl) // 2] * 1e3
        crs.close()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    value = world * n * args.steps / elapsed

    out = None
    if rank == 0:
        acc_ms = stages["accumulate"]
        achieved = n * ALG_BYTES_PER_PAIR / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_accumulate.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "alt_bn128 G1 MSM point-scalar pairs/s at n=2^%d" % args.log2n,
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (254-bit Montgomery integers)", "data": "synthetic",
            "config": {"workload": "alt_bn128 G1 Pippenger MSM, n=2^%d random scalars per GPU" % args.log2n,
                       "window_bits": lsa.msm_window_bits(n), "glv": True, "sharding": "index ranges, 1 RCCL all-gather of 96-B partials"
                       if world > 1 else "single GPU"},
            "roofline": {"kernel": "k_accumulate<CurveG1>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": acc_ms, "calls_averaged": stages["calls"],
                         "valu": {"achieved_Gfmul_s": n * FMULS_PER_PAIR / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0,
                                  "peak_Gfmul_s": FMUL_PEAK_G,
                                  "frac": (n * FMULS_PER_PAIR / (acc_ms * 1e-3) / 1e9 / FMUL_PEAK_G) if acc_ms > 0 else 0.0,
                                  "note": "254-bit field multiplications/s in k_accumulate vs the microbenchmarked "
                                          "ceiling of the 9x29-bit Montgomery product on this chip"}},
            "stage_ms": {k: round(v, 4) for k, v in stages.items() if k not in ("calls", "reserved")},
            "single_call_latency_ms": latency_ms,
            "cplink_prover_ms": cplink_ms,
            "pipelining": "the tail (reduce+fold) of step i runs on an internal stream and overlaps the front of "
                          "step i+1; stage_ms are measured under that overlap; LSA_NO_OVERLAP=1 serialises",
        }
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as o   # the checker, timed as the reported CPU baseline
            ns = 1 << min(args.cpu_sample_log2, args.log2n)
            hb = bases_jac[:ns].cpu().numpy().view(np.uint64)
            hs = d_scalars[:ns].cpu().numpy().view(np.uint64)
            tc = time.perf_counter()
            ref = o.multi_exp("g1", hb, hs, chunks=1, threads=0, mode="mixed")
            tc = time.perf_counter() - tc
            got = B.msm(d_scalars[:ns], n=ns)
            same = o.g1_canonical_affine(ref) == o.g1_canonical_affine(got)
            out["cpu_baseline"] = {
                "value": ns / tc, "unit": "pairs/s", "cores": 1, "kind": "port",
                "sample": "first 2^%d pairs of the same workload, libff-algorithm restatement "
                          "(multi_exp_with_mixed_addition<BDLO12>, chunks=1, gcc -O3), %d host cores present"
                          % (min(args.cpu_sample_log2, args.log2n), os.cpu_count()),
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
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

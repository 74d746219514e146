"""CPU suite, world_size 2 over gloo: the multi-GPU MSM host logic (libff chunk split,
one all-gather of the 96-byte partials, fold in rank order) with the local compute
replaced by the oracle stand-in.  The HIP library is not involved (no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as o
from legosnark_amd import sharded


def test_shard_range_matches_libff_chunking():
    # libff multi_exp: one = total/chunks, last chunk takes the remainder; total<chunks -> no split
    assert [sharded.shard_range(10, 3, r) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
    assert [sharded.shard_range(1 << 20, 8, r)[1] - sharded.shard_range(1 << 20, 8, r)[0] for r in range(8)] == [1 << 17] * 8
    assert [sharded.shard_range(2, 4, r) for r in range(4)] == [(0, 2), (2, 2), (2, 2), (2, 2)]
    assert sharded.shard_range(7, 1, 0) == (0, 7)
    cover = []
    for r in range(5):
        lo, hi = sharded.shard_range(1026, 5, r)
        cover += list(range(lo, hi))
    assert cover == list(range(1026))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, group, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = 12 if group == "g1" else 24
        bases = o.arith_bases(group, 1234567, 7654321, n)
        sc, _ = o.random_scalars(n, seed=99)
        lo, hi = sharded.shard_range(n, world, rank)

        def local_msm(d_scalars, d_out):
            part = o.multi_exp(group, bases[lo:hi], d_scalars.numpy().view(np.uint64).reshape(-1, 4), mode="mixed")
            d_out.copy_(torch.from_numpy(part.view(np.int64)))

        def fold(gathered, cnt, d_total, stream=None):
            add = o.g1_add if group == "g1" else o.g2_add
            acc = gathered[0].numpy().view(np.uint64).copy()
            for i in range(1, cnt):
                acc = add(acc, gathered[i].numpy().view(np.uint64).copy())
            d_total.copy_(torch.from_numpy(acc.view(np.int64)))

        job = sharded.ShardedMSM(group, world, rank, local_msm, fold, torch.device("cpu"), dist=dist)
        res = job.run(torch.from_numpy(sc[lo:hi].view(np.int64)))
        got = res.numpy().view(np.uint64).copy()
        want = o.multi_exp(group, bases, sc, mode="mixed")
        canon = o.g1_canonical_affine if group == "g1" else o.g2_canonical_affine
        q.put((rank, canon(got) == canon(want)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("group,n", [("g1", 1001), ("g2", 130), ("g1", 1)])
def test_sharded_msm_world2_gloo(group, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, group, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = sorted(q.get(timeout=5) for _ in range(2))
    assert results == [(0, True), (1, True)]


def _pairing_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ps = o.arith_bases("g1", 11, 5, n)
        qs = o.arith_bases("g2", 7, 3, n)
        job = sharded.ShardedPairingProduct(
            world, rank,
            local_product=lambda a, b: o.fq12_product(o.miller_loop_batch(a, b)),
            product=o.fq12_product, final_exp=o.final_exponentiation, dist=dist)
        got = job.run(ps, qs)
        want = o.pairing_product(ps, qs)
        q.put((rank, bool((got == want).all())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 1])
def test_sharded_pairing_product_world2_gloo(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pairing_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = sorted(q.get(timeout=5) for _ in range(2))
    assert results == [(0, True), (1, True)]

"""The C-ABI multi-GPU step (legosnark_amd/csrc/comm.hip): host-side logic and linkage on CPU,
a one-rank communicator on the GPU (RCCL refuses two ranks on one device, so world > 1 runs only
on the driver's multi-GPU node; the partition / gather / fold logic for world = 2 is covered by
tests/test_sharded_cpu.py under gloo and by the two-rank stand-in tests)."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

import legosnark_amd
from legosnark_amd import sharded


def test_shard_range_is_libffs_chunk_split():
    rng = random.Random(5)
    cases = [(0, 1, 0), (5, 8, 0), (5, 8, 7), (8, 8, 3), ((1 << 24) + 2, 8, 7), ((1 << 24) + 2, 8, 0)]
    cases += [(rng.randrange(0, 1 << 26), w, r) for w in (1, 2, 3, 4, 8) for r in range(w) for _ in range(3)]
    for n, world, rank in cases:
        assert legosnark_amd.shard_range(n, world, rank) == sharded.shard_range(n, world, rank), (n, world, rank)
    # the ranges tile [0, n)
    for n, world in ((1 << 24) + 2, 8), (1000, 7), (3, 8):
        edges = [legosnark_amd.shard_range(n, world, r) for r in range(world)]
        covered = sorted(x for lo, hi in edges for x in ((lo, hi),) if hi > lo)
        assert covered[0][0] == 0 and covered[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


def test_library_links_rccl_and_rccl_has_the_collectives():
    out = subprocess.check_output(["readelf", "-d", legosnark_amd.LIB_PATH], text=True)
    assert "librccl.so" in out, "liblegosnark_amd.so must link RCCL (comm.hip)"
    rccl = ctypes.CDLL("librccl.so.1") if os.path.exists("/opt/rocm/lib/librccl.so.1") else ctypes.CDLL("librccl.so")
    for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclCommDestroy", "ncclGetErrorString"):
        assert hasattr(rccl, sym), sym


def test_exactly_one_rccl_is_mapped():
    """torch's bundled librccl.so and the library's librccl.so.1 dependency share a SONAME; whichever is loaded first
    must serve both (legosnark_amd.lib() loads torch first and refuses to continue with two)."""
    import torch  # noqa: F401
    paths = legosnark_amd.rccl_paths()
    assert len(paths) == 1, paths


def test_comm_entry_points_fail_loudly_without_device_or_communicator():
    if legosnark_amd.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(legosnark_amd.LsaError):
        legosnark_amd.comm_init(0, 1, b"\0" * 128)
    with pytest.raises(legosnark_amd.LsaError):
        legosnark_amd.comm_join()
    assert legosnark_amd.comm_world() == 1 and legosnark_amd.comm_rank() == 0


@pytest.mark.gpu
def test_one_rank_communicator_matches_single_gpu_entry_points(lsa, tmp_path):
    import torch
    import oracle_lib as o
    with pytest.raises(legosnark_amd.LsaError):
        lsa.comm_join()                                       # no communicator yet
    lsa.comm_init(0, 1, lsa.comm_unique_id())
    try:
        assert (lsa.comm_world(), lsa.comm_rank()) == (1, 0)
        n = 6000
        bases = o.arith_bases("g1", 5, 7, n)
        sc, _ = o.random_scalars(n, seed=1)
        want = o.g1_canonical_affine(o.multi_exp("g1", bases, sc, mode="mixed"))
        B = lsa.Bases("g1", bases)
        d_s = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        assert o.g1_canonical_affine(B.msm_sharded(d_s)) == want
        outs = torch.zeros((6, 12), dtype=torch.int64, device="cuda:0")
        for i in range(6):                                    # more calls than rotating buffer sets
            B.msm_sharded_async(d_s, outs[i])
        lsa.comm_join()
        lsa.synchronize()
        for i in range(6):
            assert o.g1_canonical_affine(outs[i].cpu().numpy().view(np.uint64)) == want
        assert o.g1_canonical_affine(lsa.msm_sharded("g1", bases, sc)) == want
        q = o.arith_bases("g2", 3, 11, 5)
        p = o.arith_bases("g1", 9, 2, 5)
        assert np.array_equal(lsa.pairing_product_sharded(p, q), lsa.pairing_product(p, q))
        B.close()
    finally:
        lsa.comm_destroy()
    # file bootstrap (what a C++ SPMD prover without its own transport uses)
    lsa.comm_init_file(0, 1, str(tmp_path / "lsa_comm.id"))
    assert lsa.comm_world() == 1
    lsa.comm_destroy()


_LOOPBACK_BODY = r"""
import numpy as np, torch, sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import legosnark_amd as lsa
import oracle_lib as o
assert lsa.LIB_PATH.endswith("liblegosnark_amd_loopback.so")
lsa.init(0)
W = %(world)d
lsa.comm_init(0, 1, lsa.comm_unique_id())
assert lsa.comm_world() == W
n = 5000
bases = o.arith_bases("g1", 12, 5, n)
sc, _ = o.random_scalars(n, seed=21)
one = o.multi_exp("g1", bases, sc, mode="mixed")
want = o.g1_canonical_affine(o.g1_mul(one, o.fr_mont(W)))
B = lsa.Bases("g1", bases)
d_s = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
outs = torch.zeros((9, 12), dtype=torch.int64, device="cuda:0")
for i in range(9):                                    # more than twice the rotating buffer sets
    B.msm_sharded_async(d_s, outs[i])
lsa.comm_join()
lsa.synchronize()
for i in range(9):
    assert o.g1_canonical_affine(outs[i].cpu().numpy().view(np.uint64)) == want, i
assert o.g1_canonical_affine(B.msm_sharded(d_s)) == want
assert o.g1_canonical_affine(lsa.msm_sharded("g1", bases, sc)) == want
q = o.arith_bases("g2", 3, 11, 4)
p = o.arith_bases("g1", 9, 2, 4)
e1 = lsa.pairing_product(p, q)
eW = e1
for _ in range(W - 1):
    eW = o.fq12_mul(eW, e1)
assert np.array_equal(lsa.pairing_product_sharded(p, q), eW)
# G2: the 192-byte partials through the same exchange step
b2 = o.arith_bases("g2", 7, 3, 700)
s2, _ = o.random_scalars(700, seed=5)
one2 = o.multi_exp("g2", b2, s2, mode="mixed")
want2 = o.g2_canonical_affine(o.g2_mul(one2, o.fr_mont(W)))
assert o.g2_canonical_affine(lsa.msm_sharded("g2", b2, s2)) == want2
B2 = lsa.Bases("g2", b2)
assert o.g2_canonical_affine(B2.msm_sharded(torch.from_numpy(s2.view(np.int64)).to("cuda:0"))) == want2
B2.close()
B.close()
lsa.comm_destroy()
print("LOOPBACK OK")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world", [3, 8])
def test_multi_rank_step_on_one_gpu_with_loopback_peers(world):
    """The TEST build of the library (liblegosnark_amd_loopback.so: comm.hip compiled with LSA_COMM_TEST_LOOPBACK;
    the product library has no such hook) with LSA_COMM_LOOPBACK=W: a one-rank communicator that behaves as rank 0
    of W ranks whose peers contribute this rank's own partial (W device copies instead of ncclAllGather).  Runs the
    whole world > 1 code path of csrc/comm.hip -- side stream, rotating buffer sets, events, k_sum_points, the
    host-vector entry point, the sharded pairing product -- on one GPU; results are W times / the W-th power of
    the local ones.  In its own process: the variant is chosen when the library is loaded."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LSA_LIB_VARIANT="loopback", LSA_COMM_LOOPBACK=str(world))     # 8: BASELINE.json configs[3]'s node
    body = _LOOPBACK_BODY % {"tests": os.path.join(root, "tests"), "root": root, "world": world}
    r = subprocess.run([sys.executable, "-c", body], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "LOOPBACK OK" in r.stdout, r.stdout[-3000:]


def test_product_library_has_no_loopback_hook():
    blob = open(os.path.join(os.path.dirname(legosnark_amd.LIB_PATH), "liblegosnark_amd.so"), "rb").read()
    assert b"LSA_COMM_LOOPBACK" not in blob

"""bench.py must never under-report: `--gpus N` launches its own N ranks when nothing else has (one child process
tree, started before the parent touches torch or the GPU) and fails unless the line it relays says n_gpus == N;
a WORLD_SIZE that disagrees with --gpus is an error in both directions.  CPU-only checks of that launcher."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_dry_launch_shows_one_rank_per_gpu():
    r = run(["--gpus", "8", "--steps", "7", "--warmup", "2", "--dry-launch"])
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "8", "--steps", "7", "--warmup", "2"]          # the ranks get the same arguments, minus --dry-launch


def test_world_size_that_disagrees_with_gpus_is_an_error():
    r = run(["--gpus", "4"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    r = run(["--gpus", "1"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_a_failed_multi_rank_run_is_a_failed_bench():
    """No GPU here: the two ranks exit non-zero, and so must the launcher -- never a one-GPU number in its place."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU box: covered by the driver's own multi-GPU run")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-pmc", "--no-cpu-baseline", "--no-host-path", "--no-configs"])
    assert r.returncode != 0
    assert '"metric"' not in r.stdout


def test_config_4_launch_and_its_shards():
    """BASELINE.json configs[3]: ONE CPlink-prover MSM over 2^24 + 2 pairs on the 8 GPUs of a node.  The launcher hands
    every rank `--total-log2n 24`, and lsa_shard_range (libff's chunk rule, /root/reference/src/utils/globl.h:67-77:
    n / chunks each, the last takes the remainder) tiles [0, 2^24 + 2) exactly, in rank order, like the Python mirror."""
    r = run(["--gpus", "8", "--total-log2n", "24", "--dry-launch"])
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "8", "--total-log2n", "24"]
    import legosnark_amd as lsa
    from legosnark_amd import sharded
    n = (1 << 24) + 2
    for world in (8, 2, 1, 7):
        edges = [lsa.shard_range(n, world, rank) for rank in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        for rank in range(world):
            assert edges[rank] == sharded.shard_range(n, world, rank)
            if rank:
                assert edges[rank][0] == edges[rank - 1][1]
            if rank < world - 1:
                assert edges[rank][1] - edges[rank][0] == n // world
    # fewer pairs than ranks: libff runs the inner loop unchunked -- rank 0 owns everything
    assert [lsa.shard_range(5, 8, rank) for rank in range(8)] == [(0, 5)] + [(5, 5)] * 7

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

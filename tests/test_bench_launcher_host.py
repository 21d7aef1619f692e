"""bench.py's rank launching, without a GPU: `--gpus N` alone starts N rank processes with the
environment torch.distributed expects; under torch.distributed.run it starts none of its own."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _lines(cmd, env=None):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]


def _clean_env():
    return {k: v for k, v in os.environ.items()
            if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}


def test_gpus_flag_starts_that_many_ranks():
    got = _lines([sys.executable, BENCH, "--gpus", "3", "--rank-info"], _clean_env())
    assert sorted(g["rank"] for g in got) == [0, 1, 2]
    assert all(g["world"] == 3 and g["local_rank"] == g["rank"] for g in got)
    assert len({g["pid"] for g in got}) == 3 and len({g["master"] for g in got}) == 1
    assert got[0]["master"].startswith("127.0.0.1:")


def test_one_gpu_is_one_process():
    got = _lines([sys.executable, BENCH, "--gpus", "1", "--rank-info"], _clean_env())
    assert got == [dict(got[0], rank=0, local_rank=0, world=1)] and got[0]["master"] == "None:None"


def test_under_torchrun_no_second_launch():
    # the port is free when it is picked and may be taken by the time torchrun binds it: another port then
    got, err = None, ""
    for _ in range(3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--rank-info"],
                           cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            got = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
            break
        err = r.stderr[-2000:]
    assert got is not None, err
    assert sorted(g["rank"] for g in got) == [0, 1] and all(g["world"] == 2 for g in got)


def test_failed_rank_fails_the_launch():
    """No GPU here: every rank exits with an error and the parent reports failure."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a machine without a GPU")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--streams", "4", "--cpu-streams", "0"], cwd=ROOT,
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no GPU visible" in r.stderr

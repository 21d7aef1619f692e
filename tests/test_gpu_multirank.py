"""The multi-rank branch of bench.py on real hardware.  With one device, two ranks share it and
the decoded-frame records travel over gloo; with two or more devices the same branch runs one
rank per device over RCCL.  Rank 0 checks inside bench.py that the gathered records are every
rank's frames in stream order; here the printed line is checked."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--cpu-streams", "0", *flags], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_device_gather_over_gloo():
    line = _bench("--gpus", "2", "--streams", "48", "--backend", "gloo", "--share-devices")
    assert line["n_gpus"] == 2 and line["launcher"] == "bench.py --gpus" and line["scaling"] == "weak"
    assert line["gather"]["records"] == 96 and line["gather"]["verified_in_stream_order"] is True
    assert line["config"]["streams_per_gpu"] == 48
    # 2 ranks x 48 streams x 433 windows of 7055 samples per step
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 2 * 48 * 433 * 7055 / 44100.0) < 1e-6 * line["value"]


def test_two_ranks_on_two_devices_gather_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    line = _bench("--gpus", "2", "--streams", "256")
    assert line["n_gpus"] == 2 and line["gather"]["verified_in_stream_order"] is True
    assert line["gather"]["ranks_share_devices"] is False and "RCCL" in line["gather"]["backend"]


RCCL_ONE_RANK = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from wavenet_amd import sharding as SH
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
g = torch.Generator().manual_seed(3)
packed = torch.randint(0, 256, (1000, SH.RECORD_BYTES), dtype=torch.uint8, generator=g).to(dev)
out = SH.gather_records(packed, 1000, dst=0, even_alone=True)
assert out.is_cuda and torch.equal(out, packed)
cls = torch.randint(-300, 300, (22503,), dtype=torch.int16, generator=g).to(dev)
# the all-gather of class words goes through the same backend on a world of one
bufs = [torch.empty_like(cls.view(torch.uint8))]
dist.all_gather(bufs, cls.view(torch.uint8))
assert torch.equal(bufs[0].view(torch.int16), cls)
assert SH.all_reduce_max(1.25, dev, even_alone=True) == 1.25
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
"""


def test_rccl_collectives_on_a_communicator_of_one(tmp_path):
    """No second GPU on this box, but RCCL itself can be exercised: a communicator of one rank on
    the MI355X runs the all-gather of 140-byte records, the all-gather of class words and the
    max-reduce of the step time on device tensors, through the functions bench.py uses."""
    script = tmp_path / "one_rank.py"
    script.write_text(RCCL_ONE_RANK % ROOT)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_bench_gathers_through_rccl_with_one_rank():
    """bench.py's own distributed branch (process group on RCCL, warm-up collectives, gather inside
    the timed steps, max-reduce of the time, check of the gathered records) with a world of one."""
    line = _bench("--gpus", "1", "--streams", "64", "--dist-alone", "--no-extras")
    assert line["n_gpus"] == 1 and "RCCL" in line["gather"]["backend"]
    assert line["gather"]["records"] == 64 and line["gather"]["verified_in_stream_order"] is True

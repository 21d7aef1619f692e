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

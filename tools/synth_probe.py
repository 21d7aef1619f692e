#!/usr/bin/env python3
"""Times the synthesis launch of the benchmark workload (10 000 streams, 122 GB of float32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_amd import modem as M  # noqa: E402

m = M.Modem(M.esquema_b())
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
stride = M.pad4(m.esquema.max_frame_samples(126))
g = torch.Generator().manual_seed(1)
fields = torch.zeros((S, 16), dtype=torch.uint8)
fields[:, 0] = 1
fields[:, 13] = 3
fields[:, 14] = 107
pay = torch.randint(0, 256, (S, 112), dtype=torch.uint8, generator=g)
frames, lens = m.build_frames(fields.cuda(), pay.cuda(), 128)
audio = torch.empty((S, stride), dtype=torch.float32, device="cuda")
m.encode_streams(frames, lens, stride, audio=audio)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
for a, b in ev:
    a.record()
    m.encode_streams(frames, lens, stride, audio=audio)
    b.record()
torch.cuda.synchronize()
ms = min(a.elapsed_time(b) for a, b in ev)
print("synth %d streams: %.2f ms, %.0f GB/s (env %s)" % (S, ms, S * stride * 4 / ms / 1e6, {k: v for k, v in os.environ.items() if k.startswith("WN_SYNTH")}), flush=True)
print("checksum", float(audio[::997, ::1013].double().sum()), flush=True)

#!/usr/bin/env python3
"""Does the time of the synthesis launch depend on where the audio buffer starts?  One allocation,
the same 10 000-stream workload written at several offsets into it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_amd import modem as M  # noqa: E402

m = M.Modem(M.esquema_b())
S = 10000
stride = M.pad4(m.esquema.max_frame_samples(126))
g = torch.Generator().manual_seed(1)
fields = torch.zeros((S, 16), dtype=torch.uint8)
fields[:, 0] = 1
fields[:, 13] = 3
fields[:, 14] = 107
pay = torch.randint(0, 256, (S, 112), dtype=torch.uint8, generator=g)
frames, lens = m.build_frames(fields.cuda(), pay.cuda(), 128)
big = torch.empty(S * stride + (64 << 20), dtype=torch.float32, device="cuda")
print("base address %#x" % big.data_ptr(), flush=True)
for rep in range(2):
    for off_kb in (0, 4, 64, 256, 1024, 2048, 3072, 8192, 65536, 100000):
        off = off_kb * 256
        audio = big[off:off + S * stride].view(S, stride)
        m.encode_streams(frames, lens, stride, audio=audio)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in ev:
            a.record()
            m.encode_streams(frames, lens, stride, audio=audio)
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        print("offset %7d KB: %.2f %.2f %.2f ms" % (off_kb, ms[0], ms[1], ms[2]), flush=True)

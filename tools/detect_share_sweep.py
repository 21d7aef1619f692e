#!/usr/bin/env python3
"""Detection time of the benchmark launch against the share of the pilot's energy that sends a
window the one-tile way (WN_DETECT_SHARE), all on one buffer in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_amd import modem as M  # noqa: E402

S = 10000
m0 = M.Modem(M.esquema_b())
stride = M.pad4(m0.esquema.max_frame_samples(126))
nwin = stride // 7055
g = torch.Generator().manual_seed(1)
fields = torch.zeros((S, 16), dtype=torch.uint8)
fields[:, 0] = 1
fields[:, 13] = 3
fields[:, 14] = 107
pay = torch.randint(0, 256, (S, 112), dtype=torch.uint8, generator=g)
frames, lens = m0.build_frames(fields.cuda(), pay.cuda(), 128)
audio = torch.empty((S, stride), dtype=torch.float32, device="cuda")
m0.encode_streams(frames, lens, stride, audio=audio)
for rep in range(2):
    for share in (0.7, 0.6, 0.65, 0.75, 0.8, 0.7):
        os.environ["WN_DETECT_SHARE"] = str(share)
        m = M.Modem(M.esquema_b())
        m.detect(audio, S, stride, nwin)
        torch.cuda.synchronize()
        m.detect_stats()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in ev:
            a.record()
            m.detect(audio, S, stride, nwin)
            b.record()
        torch.cuda.synchronize()
        st = m.detect_stats(columns=True)
        print("share %.2f: %.2f ms (min of 3), windows done twice %.2f %%, stage-1 products %.1f, stage-2 products %.1f per window"
              % (share, min(a.elapsed_time(b) for a, b in ev), 100.0 * st[5] / st[0], st[4] / st[0], st[3] / st[0]), flush=True)

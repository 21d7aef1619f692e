#!/usr/bin/env python3
"""Two 122 GB audio buffers in one process: does the synthesis (and detection) time depend on
which one is used?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_amd import modem as M  # noqa: E402

m = M.Modem(M.esquema_b())
S = 10000
stride = M.pad4(m.esquema.max_frame_samples(126))
nwin = stride // 7055
g = torch.Generator().manual_seed(1)
fields = torch.zeros((S, 16), dtype=torch.uint8)
fields[:, 0] = 1
fields[:, 13] = 3
fields[:, 14] = 107
pay = torch.randint(0, 256, (S, 112), dtype=torch.uint8, generator=g)
frames, lens = m.build_frames(fields.cuda(), pay.cuda(), 128)
bufs = [torch.empty((S, stride), dtype=torch.float32, device="cuda") for _ in range(2)]
print("free after both: %.1f GB" % (torch.cuda.mem_get_info()[0] / 1e9), ["%#x" % b.data_ptr() for b in bufs], flush=True)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)


for rep in range(2):
    for i, audio in enumerate(bufs):
        s = timed(lambda: m.encode_streams(frames, lens, stride, audio=audio))
        d = timed(lambda: m.detect(audio, S, stride, nwin))
        print("buffer %d: synth %.2f ms, detect %.2f ms" % (i, s, d), flush=True)

#!/usr/bin/env python3
"""Does an HBM-write-bound kernel on a second stream overlap the MFMA-bound detection kernel?
Times detection of 5000 streams, a fill of a second 61 GB buffer, both back to back on one
stream and both at once on two streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_amd import modem as M  # noqa: E402

m = M.Modem(M.esquema_b())
S = 5000
stride = M.pad4(m.esquema.max_frame_samples(126))
nwin = stride // 7055
g = torch.Generator().manual_seed(1)
fields = torch.zeros((S, 16), dtype=torch.uint8)
fields[:, 0] = 1
fields[:, 13] = 3
fields[:, 14] = 107
pay = torch.randint(0, 256, (S, 112), dtype=torch.uint8, generator=g)
frames, lens = m.build_frames(fields.cuda(), pay.cuda(), 128)
audio = torch.empty((S, stride), dtype=torch.float32, device="cuda")
other = torch.empty((S, stride), dtype=torch.float32, device="cuda")
m.encode_streams(frames, lens, stride, audio=audio)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def wall(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        m.detect(audio, S, stride, nwin)
    with torch.cuda.stream(s2):
        kind()
    cur.wait_stream(s1)
    cur.wait_stream(s2)


for name, kind in (("fill", lambda: other.fill_(1.0)),
                   ("synth", lambda: m.encode_streams(frames, lens, stride, audio=other))):
    d = wall(lambda: m.detect(audio, S, stride, nwin))
    f = wall(kind)
    seq = wall(lambda: (m.detect(audio, S, stride, nwin), kind()))
    par = wall(both)
    print("%s: detect %.2f ms, %s %.2f ms, back to back %.2f ms, two streams %.2f ms" % (name, d, name, f, seq, par), flush=True)

# who pays when both run at once: per-stream durations, either launch order
for order in ("detect first", "fill first"):
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)

    def run_detect():
        with torch.cuda.stream(s1):
            e[0].record()
            m.detect(audio, S, stride, nwin)
            e[1].record()

    def run_fill():
        with torch.cuda.stream(s2):
            e[2].record()
            other.fill_(1.0)
            e[3].record()
    if order == "detect first":
        run_detect(); run_fill()
    else:
        run_fill(); run_detect()
    torch.cuda.synchronize()
    print("%s: detect %.2f ms, fill %.2f ms, fill starts %.2f ms after detect, ends %.2f ms after detect starts"
          % (order, e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3]), e[0].elapsed_time(e[2]), e[0].elapsed_time(e[3])), flush=True)

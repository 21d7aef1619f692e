#!/usr/bin/env python3
"""Detection of one half of the streams on one HIP stream while the other half is synthesised on
another: one after the other and at the same time, with the ordinary synthesis kernel and with
the small-footprint one (which fits beside the detection waves on a CU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_amd import modem as M  # noqa: E402

m = M.Modem(M.esquema_b())
S = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
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
ref = torch.empty_like(other)
m.encode_streams(frames, lens, stride, audio=ref)
m.encode_streams(frames, lens, stride, audio=other, beside=True)
torch.cuda.synchronize()
assert torch.equal(ref, other), "the small-footprint kernel writes the same samples"
del ref
lay = m.layout(frames, lens)
dst = torch.arange(S, dtype=torch.int64, device="cuda") * stride
fill = torch.full((S,), stride, dtype=torch.int64, device="cuda")
s_det, s_syn = torch.cuda.Stream(), torch.cuda.Stream()


def synth(beside):
    with torch.cuda.stream(s_syn):
        m.synth(lay, dst, other, stride, fill, beside=beside)


def detect():
    with torch.cuda.stream(s_det):
        m.detect(audio, S, stride, nwin)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = (1e9, 0.0, 0.0)
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        s_det.wait_event(a)
        s_syn.wait_event(a)
        fn()
        e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e1.record(s_det)
        e2.record(s_syn)
        torch.cuda.current_stream().wait_event(e1)
        torch.cuda.current_stream().wait_event(e2)
        b.record()
        torch.cuda.synchronize()
        best = min(best, (a.elapsed_time(b), a.elapsed_time(e1), a.elapsed_time(e2)))
    return best


print("streams %d, %.1f GB each side" % (S, audio.numel() * 4 / 1e9), flush=True)
print("detect alone                 %.2f ms" % timed(detect)[0], flush=True)
for beside in (False, True):
    name = "small-footprint synthesis" if beside else "ordinary synthesis       "
    alone = timed(lambda: synth(beside))[0]
    both = timed(lambda: (detect(), synth(beside)))
    print("%s alone %.2f ms; with detection: all %.2f ms, detection done at %.2f, synthesis done at %.2f"
          % ((name, alone) + both), flush=True)

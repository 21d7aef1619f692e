"""Per-phase shader-clock counts of the detection kernel's window loop (wave 0 of workgroup 0).

Run with WN_DETECT_PROFILE=1 in the environment: the library then launches the instrumented
build of the kernel, which leaves the counts in the first entries of the bin output.
"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from wavenet_amd import modem as M  # noqa: E402

assert int(os.environ.get("WN_DETECT_PROFILE", "0")), "set WN_DETECT_PROFILE=1"
m = M.Modem(M.esquema_b())
S = 2000
stride = M.pad4(m.esquema.max_frame_samples(126))
nwin = stride // 7055
audio = 0.3 * torch.randn((S, stride), device="cuda")
for i in range(3):
    cls, mag, kbin = m.detect(audio, S, stride, nwin)
torch.cuda.synchronize()
v = kbin.view(-1)[:9].tolist()
nw = (S * nwin + 1023) // 1024
names = ["dma wait", "stage 1", "stores + next dma", "stage 2 + arg-max", "merge + decision"]
for n, c in zip(names, v):
    print("%-20s %8.0f clocks/window" % (n, c * 256.0 / nw))
print("%-20s %8.0f clocks/window" % ("total", v[7] * 256.0 / nw))
print("shader clock %.3f GHz" % (v[7] / v[8] * 0.1))

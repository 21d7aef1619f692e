"""Per-phase shader-clock counts of the detection kernel's window loop (wave 0 of workgroup 0),
on noise (no column tile of stage 2 can be left out) and on clean frames (most can).

Run with WN_DETECT_PROFILE=1 in the environment: the library then launches the instrumented
build of the kernel, which leaves the counts in the first entries of the bin output.
"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from wavenet_amd import modem as M  # noqa: E402

assert int(os.environ.get("WN_DETECT_PROFILE", "0")), "set WN_DETECT_PROFILE=1 (every phase) or 2 (clocks per window and way only: less disturbed)"
m = M.Modem(M.esquema_b())
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
stride = M.pad4(m.esquema.max_frame_samples(126))
nwin = stride // 7055
g = torch.Generator().manual_seed(1)
fields = torch.zeros((S, 16), dtype=torch.uint8)
fields[:, 0] = 1
fields[:, 1:13] = torch.randint(0, 256, (S, 12), dtype=torch.uint8, generator=g)
fields[:, 13] = 3
fields[:, 14] = 107
frames, lens = m.build_frames(fields.cuda(), torch.randint(0, 256, (S, 112), dtype=torch.uint8, generator=g).cuda(), 128)
clean, _ = m.encode_streams(frames, lens, stride)
names = ["dma wait", "pilot + tile energies", "stores + next dma", "stage 2 + arg-max", "decision", "window energy",
         "rest of stage 1"]
for what, audio in (("noise", 0.3 * torch.randn((S, stride), device="cuda")), ("clean frames", clean)):
    for i in range(3):
        m.detect_stats()
        cls, mag, kbin = m.detect(audio, S, stride, nwin)
    torch.cuda.synchronize()
    v = kbin.view(-1)[:9].tolist()
    nw = (S * nwin + 1023) // 1024
    print(what)
    for n, c in zip(names, v):
        print("  %-22s %8.0f clocks/window" % (n, c * 256.0 / nw))
    print("  %-22s %8.0f clocks/window" % ("total", v[7] * 256.0 / nw))
    extra = kbin.view(-1)[9:13].tolist()
    print("  of the pilot: sample reads %.0f, folds %.0f, matrix instructions %.0f clocks/window"
          % tuple(c * 256.0 / nw for c in extra[:3]))
    print("  %-22s %8.0f clocks/window (chunk sums, next DMA, decision of the windows decided that way)"
          % ("separator's bin alone", extra[3] * 256.0 / nw))
    ways = kbin.view(-1)[13:29].tolist()
    for i, name in enumerate(("one column tile", "long way, weakest tile left out", "long way, every tile",
                              "one tile, then the long way", "the long way twice", "sixteen chosen columns",
                              "chosen columns, then the long way", "the separator's bin alone")):
        if ways[i]:
            print("  %-34s %6d windows (%.1f %%), %8.0f clocks each"
                  % (name, ways[i], 100.0 * ways[i] / sum(ways[:8]), ways[8 + i] * 256.0 / ways[i]))
    print("  shader clock %.3f GHz" % (v[7] / v[8] * 0.1))
    t = mag.view(-1)[16:16 + 1024].double() * 1e-5   # ms
    print("  waves ran %.3f / %.3f / %.3f ms (min / mean / max): the launch waits %.2f %% for its slowest wave"
          % (t.min().item(), t.mean().item(), t.max().item(), 100.0 * (t.max().item() / t.mean().item() - 1.0)))
    wg = t.view(-1, 4)[:256]
    print("  mean by workgroup index mod 8 (ms): " + " ".join("%.3f" % wg[i::8].mean().item() for i in range(8)))
    print("  spread inside a workgroup (max - min of its four waves, mean over workgroups): %.3f ms; slowest workgroups: %s"
          % ((wg.max(1).values - wg.min(1).values).mean().item(),
             ", ".join("%d (%.3f)" % (i, wg[i].max().item()) for i in wg.max(1).values.argsort(descending=True)[:6].tolist())))

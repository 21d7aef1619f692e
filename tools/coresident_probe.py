#!/usr/bin/env python3
"""Detection of one half of the streams on one HIP stream, plain stores over as many bytes on
another: one after the other, and at the same time."""
import ctypes as C
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
from wavenet_amd import modem as M  # noqa: E402

lib = C.CDLL(os.path.join(here, "coresident_probe.so"))
lib.store_launch.argtypes = [C.c_void_p, C.c_ulonglong, C.c_uint, C.c_uint, C.c_int, C.c_void_p]

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
torch.cuda.synchronize()
nbytes = other.numel() * 4
nbytes -= nbytes % (192 * 1024)
s_det, s_sto = torch.cuda.Stream(), torch.cuda.Stream()


def stores(grid, lds):
    rc = lib.store_launch(C.c_void_p(other.data_ptr()), nbytes, 192 * 1024, grid, lds, C.c_void_p(s_sto.cuda_stream))
    assert rc == 0, rc


def detect():
    with torch.cuda.stream(s_det):
        m.detect(audio, S, stride, nwin)


def timed(fn, reps=3, each=False):
    """Best total time of fn over reps; with each=True also when each stream finished."""
    fn()
    torch.cuda.synchronize()
    best = (1e9, 0.0, 0.0)
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        s_det.wait_event(a)
        s_sto.wait_event(a)
        fn()
        e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e1.record(s_det)
        e2.record(s_sto)
        torch.cuda.current_stream().wait_event(e1)
        torch.cuda.current_stream().wait_event(e2)
        b.record()
        torch.cuda.synchronize()
        t = (a.elapsed_time(b), a.elapsed_time(e1), a.elapsed_time(e2))
        best = min(best, t)
    return best if each else best[0]


items = nbytes // (192 * 1024)
print("streams %d, %.1f GB each side" % (S, nbytes / 1e9), flush=True)
print("detect alone        %.2f ms" % timed(detect), flush=True)
for lds in (0, 24 * 1024):
    for grid in (256, 512, 1024, items // 8):
        alone = timed(lambda: stores(grid, lds))
        both = timed(lambda: (detect(), stores(grid, lds)), each=True)
        print("stores lds %5d grid %6d: alone %.2f ms (%.0f GB/s); detect then stores: all %.2f ms, detect done at %.2f, stores done at %.2f"
              % ((lds, grid, alone, nbytes / alone / 1e6) + both), flush=True)

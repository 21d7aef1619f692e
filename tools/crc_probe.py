#!/usr/bin/env python3
"""What bounds the checksum kernel: 10 M rows of stride 128 with every payload length equal
(0, 45, 107) and with mixed lengths.  Equal time for L = 107 and mixed lengths means the
longest row of a wave sets the pace (LDS table reads), not the bytes moved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_amd import modem as M  # noqa: E402

m = M.Modem(M.esquema_b())
n = 10_000_000
g = torch.Generator(device="cuda").manual_seed(5)
rows = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)


def run(plen, what):
    rows[:, 14] = plen.to(torch.uint8)
    nbytes = (plen + 19).to(torch.int32)
    m.parse_frames(rows, nbytes)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record()
        m.parse_frames(rows, nbytes)
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    print("%-14s min %.3f ms median %.3f ms, %.0f GB/s of 64-byte sectors" % (
        what, t[0], t[5], float(((nbytes + 63) // 64 * 64 + 12).sum()) / t[5] / 1e6), flush=True)


for L in (0, 45, 107):
    run(torch.full((n,), L, dtype=torch.int32, device="cuda"), "all L = %d" % L)
run(torch.randint(0, 108, (n,), dtype=torch.int32, device="cuda", generator=g), "L uniform")
run(torch.sort(torch.randint(0, 108, (n,), dtype=torch.int32, device="cuda", generator=g))[0], "L sorted")

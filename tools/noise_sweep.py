#!/usr/bin/env python3
"""What detection costs when the channel is not clean: the round trip's audio (one max-payload
frame per stream) with white noise of several strengths added, detection timed on each, the
receive state machine and the checksum run on the result.  One JSON line per noise level."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_amd import modem as M  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = M.Modem(M.esquema_b())
stride = M.pad4(m.esquema.max_frame_samples(126))
nwin = stride // 7055
g = torch.Generator().manual_seed(1)
fields = torch.zeros((S, 16), dtype=torch.uint8)
fields[:, 0] = 1
fields[:, 1:13] = torch.randint(0, 256, (S, 12), dtype=torch.uint8, generator=g)
fields[:, 13] = 3
fields[:, 14] = 107
frames, lens = m.build_frames(fields.cuda(), torch.randint(0, 256, (S, 112), dtype=torch.uint8, generator=g).cuda(), 128)
audio = torch.empty((S, stride), dtype=torch.float32, device="cuda")


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


for sigma in (0.0, 0.03, 0.1, 0.3, 0.6, 1.0, 2.0):
    m.encode_streams(frames, lens, stride, audio=audio)
    if sigma:
        m.awgn(audio, sigma, 7)
    m.detect_stats()
    ms = timed(lambda: m.detect(audio, S, stride, nwin))
    w, top2, full, cells, units, redone, whole, two, chosen, sep_alone, tone_alone = m.detect_stats(paths=True)
    cls, _, _ = m.detect(audio, S, stride, nwin)
    by, ln, _, _, _, _ = m.slice(cls, m.timeout_windows(123.0), 1, 128)
    rec = m.parse_frames(by.view(S, 128), ln.view(-1)).cpu().numpy().view(M.REC_DTYPE).reshape(-1)
    exact = int((torch.eq(by.view(S, 128)[:, :126], frames[:, :126]).all(dim=1) & (ln.view(-1) == 126)).sum())
    snr = None if sigma == 0 else 10 * np.log10(0.5 * 0.975 ** 2 / sigma ** 2)
    print(json.dumps({"sigma": sigma, "snr_dB_against_a_tone": snr, "streams": S, "detect_ms": ms,
                      "ms_per_million_windows": ms / (S * nwin) * 1e6,
                      "stage1_products_per_window": units / w, "stage2_products_per_window": cells / w,
                      "long_way_passes_per_window": whole / w, "of_them_on_two_tiles": two / w,
                      "windows_done_twice": redone / w, "float64_candidates_per_window": top2 / w,
                      "float64_spectra_per_window": full / w, "frames_exact": exact,
                      "frames_with_valid_checksum": int(((rec["status"] == 0) & (rec["crc_ok"] == 1)).sum())}), flush=True)

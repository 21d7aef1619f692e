#!/usr/bin/env python3
"""Per-kernel measurements behind the roofline figures: one JSON line per kernel.

  synth   K1  10k-stream synthesis (HBM-write-bound, 4 B/sample)
  sync    K3  ping correlator over the same audio (HBM-read-bound, 4 B/sample)
  detect  K2  full-spectrum detection (f32 MFMA-bound)
  crc     K4  10 M frames, payload 0..107, 1 % corrupted, stride 128 (HBM-bound)
  long    C3  one 1-hour capture with AWGN: sync search + detection + slicing
  protoA      the prototype scheme (2205-sample windows): synthesis + detection of 127-byte messages

Usage: python tools/kernel_bench.py [--streams N] [--frames N] [--only name,...] [--reps R]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_amd import modem as M  # noqa: E402

WIN = 7055


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in ev]
    return float(np.mean(t)), float(np.min(t))


def frames_for(m, n, seed=1):
    g = torch.Generator().manual_seed(seed)
    fields = torch.zeros((n, 16), dtype=torch.uint8)
    fields[:, 0] = 1
    fields[:, 1:13] = torch.randint(0, 256, (n, 12), dtype=torch.uint8, generator=g)
    fields[:, 13] = 3
    fields[:, 14] = 107
    pay = torch.randint(0, 256, (n, 112), dtype=torch.uint8, generator=g)
    return m.build_frames(fields.cuda(), pay.cuda(), 128)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=10000)
    ap.add_argument("--frames", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="synth,sync,detect,crc,long,protoA")
    ap.add_argument("--messages", type=int, default=20000, help="messages of the prototype scheme (1.13 MB of audio each)")
    args = ap.parse_args()
    only = set(args.only.split(","))
    m = M.Modem(M.esquema_b())
    class Out(list):
        def append(self, o):
            print(json.dumps(o), flush=True)
    out = Out()
    S = args.streams
    stride = M.pad4(m.esquema.max_frame_samples(126))
    nwin = stride // WIN

    if only & {"synth", "sync", "detect"}:
        frames, lens = frames_for(m, S)
        audio = torch.empty((S, stride), dtype=torch.float32, device="cuda")
        lay = m.layout(frames, lens)
        dst = torch.arange(S, dtype=torch.int64, device="cuda") * stride
        fill = torch.full((S,), stride, dtype=torch.int64, device="cuda")
        mean, best = timed(lambda: m.synth(lay, dst, audio, stride, fill), args.reps)
        by = S * stride * 4
        if "synth" in only:
            out.append({"kernel": "synth_kernel<float, false>", "bound": "hbm", "ms": mean, "ms_min": best,
                        "algorithmic_bytes": by, "achieved_GBps": by / mean / 1e6, "peak_GBps": 8000,
                        "frac": by / mean / 1e6 / 8000, "units": "%d streams x %d samples" % (S, stride)})
        if "sync" in only:
            mean, best = timed(lambda: m.sync_scan(audio, S, stride, nwin), args.reps)
            by = S * nwin * (WIN * 4 + 9)
            pe, we, vd = m.sync_scan(audio, S, stride, nwin)
            out.append({"kernel": "sync_kernel", "bound": "hbm", "ms": mean, "ms_min": best, "algorithmic_bytes": by,
                        "achieved_GBps": by / mean / 1e6, "peak_GBps": 8000, "frac": by / mean / 1e6 / 8000,
                        "units": "%d windows" % (S * nwin),
                        "verdicts": {str(k): int((vd == k).sum()) for k in (0, 1, 2)}})
        if "detect" in only:
            m.detect_stats()
            mean, best = timed(lambda: m.detect(audio, S, stride, nwin), args.reps)
            st = m.detect_stats(columns=True)
            n1, n2 = M.host_dft_plan(m.esquema)
            k2 = n2 // 2 + 1
            h1 = n1 // 2 + 1
            ta = (h1 + 15) // 16
            cells = st[3] / max(st[0], 1)           # (row tile, column) products of the second stage, per window
            units = st[4] / max(st[0], 1)           # (row tile, real/imaginary, column) products of the first stage
            flop = (2.0 * n1 * k2 * k2 * 2 * units / (4 * ta * k2) + 2.0 * h1 * h1 * 4 * k2 * cells / (ta * k2)) * S * nwin
            out.append({"kernel": "detect_kernel<83,85>+refine", "bound": "mfma", "ms": mean, "ms_min": best,
                        "algorithmic_flop": flop, "achieved_TFLOPs": flop / mean / 1e9, "peak_TFLOPs": 157.3,
                        "frac": flop / mean / 1e9 / 157.3, "hbm_GBps": S * nwin * WIN * 4 / mean / 1e6,
                        "units": "%d windows" % (S * nwin), "stage2_tile_columns_computed_per_window": cells, "stage2_tile_columns_per_window": ta * k2,
                        "stage1_tile_columns_computed_per_window": units, "stage1_tile_columns_per_window": 4 * ta * k2,
                        "refined": st[:3]})
        del audio

    if "crc" in only:
        n = args.frames
        g = torch.Generator(device="cuda").manual_seed(5)
        rows = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)
        plen = torch.randint(0, 108, (n,), dtype=torch.int32, device="cuda", generator=g)
        rows[:, 14] = plen.to(torch.uint8)
        nbytes = plen + 19
        crc = m.crc32(rows, plen + 15)                      # checksums of header+payload
        idx = (15 + plen).to(torch.int64)
        ar = torch.arange(n, device="cuda")
        for k in range(4):                                  # store them big-endian after the payload
            rows[ar, idx + k] = ((crc >> (8 * (3 - k))) & 0xFF).to(torch.uint8)
        bad = torch.rand(n, device="cuda", generator=g) < 0.01
        pos = (torch.rand(n, device="cuda", generator=g) * nbytes).to(torch.int64)
        rows[ar[bad], pos[bad]] ^= 0x10
        rec = m.parse_frames(rows, nbytes)
        r = rec.cpu().numpy().view(M.REC_DTYPE).reshape(-1)
        failed = (r["status"] != 0) | (r["crc_ok"] == 0)      # a flipped length byte can also end in a parse error
        assert np.array_equal(failed, bad.cpu().numpy()), "every corrupted frame and only those must fail"
        mean, best = timed(lambda: m.parse_frames(rows, nbytes), args.reps)
        by = int(nbytes.sum().item()) + 12 * n
        out.append({"kernel": "crc_rows_kernel<parse+validate>", "bound": "hbm", "ms": mean, "ms_min": best,
                    "algorithmic_bytes": by, "achieved_GBps": by / mean / 1e6, "peak_GBps": 8000,
                    "frac": by / mean / 1e6 / 8000, "frames_per_s": n / mean * 1e3,
                    "row_traffic_GBps": n * 140 / mean / 1e6, "units": "%d frames, stride 128" % n,
                    "invalid_found": int(failed.sum())})
        # the lane-parallel kernel (per-distance tables, eight lanes per row, DPP combine) on the same rows
        os.environ["WN_CRC_COMBINE"] = "1"
        mc = M.Modem(M.esquema_b())
        del os.environ["WN_CRC_COMBINE"]
        assert torch.equal(mc.parse_frames(rows, nbytes), rec), "both checksum kernels give the same records"
        mean, best = timed(lambda: mc.parse_frames(rows, nbytes), args.reps)
        out.append({"kernel": "crc_combine_kernel<parse+validate>", "bound": "hbm", "ms": mean, "ms_min": best,
                    "algorithmic_bytes": by, "achieved_GBps": by / mean / 1e6, "peak_GBps": 8000,
                    "frac": by / mean / 1e6 / 8000, "frames_per_s": n / mean * 1e3,
                    "row_traffic_GBps": n * 140 / mean / 1e6, "units": "%d frames, stride 128" % n,
                    "records_equal_the_row_kernel": True})
        del rows

    if "long" in only:
        # one hour of capture: back-to-back max-payload frames, each after a random 0..30000-sample gap, AWGN 0.1
        total = 158_760_000
        nf = 51
        frames, lens = frames_for(m, nf, seed=9)
        lay = m.layout(frames, lens)
        g = torch.Generator().manual_seed(3)
        gaps = torch.randint(0, 30001, (nf,), generator=g, dtype=torch.int64)
        ns = lay[3].cpu()
        dst = torch.cumsum(gaps + torch.cat([torch.zeros(1, dtype=torch.int64), ns[:-1]]), 0)
        assert int(dst[-1] + ns[-1]) <= total
        audio = torch.zeros(M.pad4(total), dtype=torch.float32, device="cuda")
        m.synth(lay, dst.cuda(), audio, stride)
        m.awgn(audio, 0.1, 77)
        nw = total // WIN

        def run():
            cls, mag, kbin = m.detect(audio, 1, audio.numel(), nw)
            return m.slice(cls, m.timeout_windows(123.0), nf + 4, 128)
        mean, best = timed(run, args.reps)
        by, ln, start, end, reason, ncalls = run()
        k = int(ncalls[0])
        got = [bytes(by[0, j, :int(ln[0, j])].cpu().numpy()) for j in range(k)]
        want = [bytes(frames[j, :126].cpu().numpy()) for j in range(nf)]
        ok = [x for x in got if len(x) == 126]
        out.append({"kernel": "C3 long capture: detect + slice", "ms": mean, "ms_min": best,
                    "audio_s_per_s": total / 44100 / mean * 1e3, "windows": nw, "calls": k,
                    "frames_recovered": len(ok), "frames_exact": int(sum(a == b for a, b in zip(ok, want))),
                    "frames_sent": nf, "refined": m.detect_stats()})

    if "protoA" in only:
        ma = M.Modem(M.esquema_a())
        n, msg = args.messages, 127
        g = torch.Generator().manual_seed(4)
        rows = torch.randint(0, 128, (n, 128), dtype=torch.uint8, generator=g).cuda()
        lens_a = torch.full((n,), msg, dtype=torch.int32, device="cuda")
        win_a = ma.esquema.win_n
        stride_a = M.pad4((msg + 1) * win_a)
        audio_a = torch.empty((n, stride_a), dtype=torch.float32, device="cuda")
        mean_s, _ = timed(lambda: ma.encode_streams(rows, lens_a, stride_a, audio=audio_a), args.reps)
        nw = stride_a // win_a
        ma.detect_stats()
        mean_d, _ = timed(lambda: ma.detect(audio_a, n, stride_a, nw), args.reps)
        st = ma.detect_stats(columns=True)
        cls, _, _ = ma.detect(audio_a, n, stride_a, nw)
        ok = bool(((cls[:, :msg] & 0x1FF).to(torch.uint8) == rows[:, :msg]).all())
        n1, n2 = M.host_dft_plan(ma.esquema)
        h1, h2 = n1 // 2 + 1, n2 // 2 + 1
        ta = (h1 + 15) // 16
        flop = (2.0 * n1 * h2 * h2 * 2 * (st[4] / max(st[0], 1)) / (4 * ta * h2)
                + 2.0 * h1 * h1 * 4 * h2 * (st[3] / max(st[0], 1)) / (ta * h2)) * n * nw
        out.append({"kernel": "prototype scheme: synth / detect_kernel<45,49>", "messages": n, "synth_ms": mean_s,
                    "synth_GBps": n * stride_a * 4 / mean_s / 1e6, "detect_ms": mean_d,
                    "detect_GBps": n * nw * (win_a * 4 + 10) / mean_d / 1e6, "detect_TFLOPs": flop / mean_d / 1e9, "detect_frac_f32_mfma": flop / mean_d / 1e9 / 157.3,
                    "windows": n * nw, "stage2_tile_columns_computed_per_window": st[3] / max(st[0], 1), "stage2_tile_columns_per_window": ta * h2,
                    "bytes_exact": ok})


if __name__ == "__main__":
    main()

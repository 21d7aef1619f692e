#!/usr/bin/env python3
"""Detection against NumPy's float64 FFT on many random windows (both schemes).

usage: python tools/fuzz_detect.py [windows_per_scheme] [seed]

Windows are drawn like tests/test_gpu_parity.py::test_detect_random_mixtures (up to four tones
of log-uniform strengths on arbitrary frequencies, DC, noise from nothing to louder than the
tones, partial silence, any scale) plus pairs of scheme tones meeting at a random point and, for a
scheme with a separator, the separator (slightly detuned at times) giving way to another tone
anywhere from the start of the window to its end, at any ratio of loudness.  A
window counts as a mismatch when the kernel's bin differs from np.argmax(|fft|[:N//2]) although
the two largest float64 magnitudes differ by more than 1e-9 relative."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_amd import modem as M  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for name, esq, n in (("B", M.esquema_b(), 7055), ("A", M.esquema_a(), 2205)):
    m = M.Modem(esq)
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    tones = [esq.base_freq + esq.freq_step * b for b in range(esq.max_byte + 1)] + [esq.eof_freq]
    if esq.sep_freq:
        tones += [esq.sep_freq, esq.ping_freq]
    bad = unclear = done = 0
    m.detect_stats()
    while done < count:
        k = min(2000, count - done)
        x = np.zeros((k, n))
        for i in range(k):
            kind = rng.random()
            if esq.sep_freq and kind >= 0.85:
                f = esq.sep_freq * (1.0 + (rng.uniform(-0.015, 0.015) if rng.random() < 0.3 else 0.0))
                g = float(rng.choice(tones))
                cut = int(n * min(1.0, rng.uniform(0.0, 1.4)))
                x[i] = np.where(t < cut, np.sin(2 * np.pi * f * (t + rng.integers(0, 5000)) / 44100),
                                10 ** rng.uniform(-1, 0.5) * np.sin(2 * np.pi * g * (t - cut) / 44100))
                if rng.random() < 0.5:
                    x[i] = x[i][::-1]
                x[i] += 10 ** rng.uniform(-5, -0.3) * rng.standard_normal(n)
            elif kind < 0.3:
                f, g = float(rng.choice(tones)), float(rng.choice(tones))
                cut = int(rng.integers(0, n + 1))
                x[i] = np.where(t < cut, np.sin(2 * np.pi * f * (t + rng.integers(0, 5000)) / 44100),
                                np.sin(2 * np.pi * g * (t - cut) / 44100))
                x[i] += 10 ** rng.uniform(-5, -0.5) * rng.standard_normal(n)
            else:
                for _ in range(int(rng.integers(0, 5))):
                    x[i] += 10 ** rng.uniform(-3, 0) * np.sin(2 * np.pi * rng.uniform(0, 22050) * t / 44100 + rng.uniform(0, 6.28))
                if rng.random() < 0.3:
                    x[i] += rng.uniform(-0.5, 0.5)
                if rng.random() < 0.7:
                    x[i] += 10 ** rng.uniform(-5, 0) * rng.standard_normal(n)
                if rng.random() < 0.2:
                    a, b = sorted(rng.integers(0, n, 2))
                    x[i, a:b] = 0.0
            x[i] *= 10 ** rng.uniform(-2, 1)
        x32 = x.astype(np.float32)
        spec = np.abs(np.fft.fft(x32.astype(np.float64), axis=1))[:, :n // 2]
        want = spec.argmax(axis=1)
        top2 = np.sort(spec, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0] > 1e-9 * top2[:, 1]) | (top2[:, 1] == 0)   # all zeros: np.argmax gives 0
        pad = np.zeros(M.pad4(k * n), dtype=np.float32)
        pad[:k * n] = x32.reshape(-1)
        _, mag, kbin = m.detect(torch.from_numpy(pad).cuda(), 1, len(pad), k)
        got = kbin.cpu().numpy()[0]
        bad += int(((got != want) & clear).sum())
        unclear += int((~clear).sum())
        ok = clear & (spec.max(axis=1) > 0)
        rel = np.abs(mag.cpu().numpy()[0][ok] - spec.max(axis=1)[ok]) / spec.max(axis=1)[ok]
        assert rel.max() <= 1e-4, rel.max()
        done += k
    st = m.detect_stats(paths=True)
    print("scheme %s: %d windows, %d mismatches, %d with float64 ties left out; float64 candidates %d, float64 spectra %d, "
          "stage-2 products per window %.1f, windows done twice %d, on sixteen chosen columns %d, from the separator's bin alone %d, from another scheme tone's bin alone %d"
          % (name, done, bad, unclear, st[1], st[2], st[3] / max(st[0], 1), st[5], st[8], st[9], st[10]), flush=True)
    assert bad == 0

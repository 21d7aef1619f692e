"""The library from several threads, the way SoundProtocol uses the modem (daemon threads,
WaveNetProtocols.py:303-310, 321-327): in a FRESH process two threads make their first calls of
every kernel at the same moment, each on its own stream, sharing one context; results must equal
the ones a single thread gets afterwards."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import sys, threading
    import numpy as np, torch
    sys.path.insert(0, %r)
    from wavenet_amd import modem as M

    WIN = 7055
    shared = M.Modem(M.esquema_b())          # one context for both threads; no kernel has run yet
    own = [None, None]                       # and one context made inside each thread, at the same moment
    go = threading.Barrier(2)
    out, err = [None, None], [None, None]

    def inputs(seed):
        g = torch.Generator().manual_seed(seed)
        fields = torch.zeros((6, 16), dtype=torch.uint8)
        fields[:, 0] = 1
        fields[:, 1:13] = torch.randint(0, 256, (6, 12), dtype=torch.uint8, generator=g)
        fields[:, 13] = 3
        fields[:, 14] = torch.randint(0, 20, (6,), dtype=torch.uint8, generator=g)
        pay = torch.randint(0, 256, (6, 112), dtype=torch.uint8, generator=g)
        return fields.cuda(), pay.cuda()

    def work(m, seed):
        fields, pay = inputs(seed)
        frames, lens = m.build_frames(fields, pay, 128)
        crc = m.crc32(frames, lens - 4)
        stride = M.pad4(m.esquema.max_frame_samples(39) + WIN)
        nwin = stride // WIN
        audio, _ = m.encode_streams(frames, lens, stride)
        m.awgn(audio, 0.05, seed)
        pe, we, vd = m.sync_scan(audio, 6, stride, nwin)
        d = m.decode_streams(audio, 6, stride, nwin, 123.0)
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (frames, lens, crc, audio, vd, d["cls"], d["bin"], d["mag"], d["bytes"],
                                          d["lens"], d["rec"])]

    def thread(i):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                go.wait()
                own[i] = M.Modem(M.esquema_b())
                a = work(shared, 100 + i)
                b = work(own[i], 200 + i)
                out[i] = (a, b)
        except BaseException as e:      # noqa: BLE001
            err[i] = repr(e)

    ts = [threading.Thread(target=thread, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert err == [None, None], err
    for i in range(2):
        for got, seed in zip(out[i], (100 + i, 200 + i)):
            want = work(shared, seed)           # single thread, default stream
            assert all(np.array_equal(g, w) for g, w in zip(got, want)), (i, seed)
            assert (got[9].ravel() == got[1]).all()           # and every frame came back whole
    print("threads ok")
''') % ROOT


def test_first_calls_from_two_threads_at_once():
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "threads ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])

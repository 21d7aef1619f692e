"""wn_detect_windows_noalloc against wn_detect_windows, through the C ABI: two calls in a row on
the same GPU, the second with a smaller input (so the workspace still holds the first call's
flags and magnitudes), then the size limit."""
import ctypes as C

import numpy as np
import pytest
import torch

from wavenet_amd import _lib as L
from wavenet_amd import modem as M

pytestmark = pytest.mark.gpu
WIN = 7055


def _capture(n_streams, n_windows, seed):
    """Streams of tones, near-ties, noise, digital silence and very quiet windows: every path of
    the float64 pass gets work (flags in the workspace between the two kernels)."""
    rng = np.random.default_rng(seed)
    t = np.arange(WIN)
    stride = M.pad4(n_windows * WIN + 5)
    a = np.zeros((n_streams, stride), dtype=np.float32)
    for s in range(n_streams):
        for w in range(n_windows):
            kind = (s * n_windows + w) % 6
            if kind == 0:
                x = 0.45 * np.sin(2 * np.pi * (350 + 70 * int(rng.integers(0, 256))) * t / 44100)
            elif kind == 1:
                k1, k2 = rng.integers(60, 3000, 2)
                x = 0.3 * (np.cos(2 * np.pi * k1 * t / WIN) + (1 + 1e-7) * np.cos(2 * np.pi * k2 * t / WIN + 0.3))
            elif kind == 2:
                x = 0.2 * rng.standard_normal(WIN)
            elif kind == 3:
                x = np.zeros(WIN)
            elif kind == 4:
                x = 1e-22 * rng.standard_normal(WIN)
            else:
                x = 0.4 * np.cos(2 * np.pi * (WIN // 2) * t / WIN) + 1e-5 * rng.standard_normal(WIN)
            a[s, 1 + w * WIN:1 + (w + 1) * WIN] = x
    return torch.from_numpy(a).cuda(), stride


def test_no_alloc_detection_equals_the_allocating_one():
    m = M.Modem(M.esquema_b())
    for n_streams, n_windows, seed in ((7, 12, 1), (3, 5, 2)):      # the second call is smaller
        audio, stride = _capture(n_streams, n_windows, seed)
        want, _, _ = m.detect(audio, n_streams, stride, n_windows, first_sample=1)
        got = torch.full((n_streams, n_windows), -1, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        # free bytes of the device as the driver counts them: an allocation by the library (hipMalloc,
        # hipMallocAsync, a pool growing) shows here, which torch's own counters would not see
        free_before = torch.cuda.mem_get_info()[0]
        torch_before = torch.cuda.memory_reserved()
        rc = L.lib.wn_detect_windows_noalloc(m._ctx, C.c_void_p(audio.data_ptr()), audio.numel(), n_streams, stride, 1,
                                             n_windows, C.c_void_p(got.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == L.WN_OK, L.lib.wn_last_error()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before and torch.cuda.memory_reserved() == torch_before
        assert torch.equal(got, want)
        assert len(set(want.cpu().numpy().ravel().tolist())) > 3   # the capture was not trivial


def test_no_alloc_detection_under_graph_capture():
    """The use the entry point is for: captured once in a HIP graph (where any allocation, and any
    synchronising call, fails the capture), replayed on two different captures of audio, each replay
    equal to wn_detect_windows on the same audio."""
    m = M.Modem(M.esquema_b())
    n_streams, n_windows = 5, 9
    first, stride = _capture(n_streams, n_windows, 11)
    second, _ = _capture(n_streams, n_windows, 12)
    audio = first.clone()
    got = torch.full((n_streams, n_windows), -1, dtype=torch.int16, device="cuda")
    m.detect_classes_into(audio, n_streams, stride, n_windows, got, first_sample=1)   # code objects loaded before the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = L.lib.wn_detect_windows_noalloc(m._ctx, C.c_void_p(audio.data_ptr()), audio.numel(), n_streams, stride, 1,
                                             n_windows, C.c_void_p(got.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == L.WN_OK, L.lib.wn_last_error()
    for src in (second, first):
        audio.copy_(src)
        got.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        want, _, _ = m.detect(src, n_streams, stride, n_windows, first_sample=1)
        assert torch.equal(got, want)
    assert not torch.equal(m.detect(first, n_streams, stride, n_windows, first_sample=1)[0],
                           m.detect(second, n_streams, stride, n_windows, first_sample=1)[0])


def test_no_alloc_calls_on_two_streams_follow_each_other():
    """Calls enqueued on different streams with nothing between them share the one workspace: the
    library makes each wait for the one before it, so both results equal wn_detect_windows."""
    m = M.Modem(M.esquema_b())
    shapes = ((40, 30, 21), (9, 11, 22), (33, 25, 23), (2, 3, 24))
    caps = [_capture(s, w, seed) for s, w, seed in shapes]
    wants = [m.detect(a, s, stride, w, first_sample=1)[0] for (s, w, _), (a, stride) in zip(shapes, caps)]
    gots = [torch.full((s, w), -1, dtype=torch.int16, device="cuda") for s, w, _ in shapes]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for i, ((s, w, _), (a, stride)) in enumerate(zip(shapes, caps)):
        rc = L.lib.wn_detect_windows_noalloc(m._ctx, C.c_void_p(a.data_ptr()), a.numel(), s, stride, 1, w,
                                             C.c_void_p(gots[i].data_ptr()), C.c_void_p(streams[i % 2].cuda_stream))
        assert rc == L.WN_OK, L.lib.wn_last_error()
    torch.cuda.synchronize()
    for got, want in zip(gots, wants):
        assert torch.equal(got, want)


def test_no_alloc_detection_refuses_more_than_its_workspace_holds():
    m = M.Modem(M.esquema_b())
    cls = torch.zeros(4, dtype=torch.int16, device="cuda")
    audio = torch.zeros(8, dtype=torch.float32, device="cuda")
    for n_streams, n_windows in ((1, L.DETECT_NOALLOC_MAX_WINDOWS + 1), (2049, 512), (1 << 40, 1 << 40)):
        rc = L.lib.wn_detect_windows_noalloc(m._ctx, C.c_void_p(audio.data_ptr()), audio.numel(), n_streams,
                                             n_windows * WIN, 0, n_windows, C.c_void_p(cls.data_ptr()), None)
        assert rc == L.WN_ERR_ARG and b"workspace" in L.lib.wn_last_error()
    # at the limit the size check passes and the next check (the audio is too short) answers
    rc = L.lib.wn_detect_windows_noalloc(m._ctx, C.c_void_p(audio.data_ptr()), audio.numel(), 1,
                                         L.DETECT_NOALLOC_MAX_WINDOWS * WIN, 0, L.DETECT_NOALLOC_MAX_WINDOWS,
                                         C.c_void_p(cls.data_ptr()), None)
    assert rc == L.WN_ERR_ARG and b"audio_len" in L.lib.wn_last_error()

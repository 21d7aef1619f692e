"""Batched layer-1 modem on the GPU: torch tensors in, HIP kernels through the C ABI, torch
tensors out.  torch only owns the device memory and the stream; all arithmetic of the
hot path happens in libwavenet_hip.so.

Reference being replaced: DispositivoWaveNET/dispositivo_wavenet/c1_shared.py (scheme B)
and DispositivoWaveNET/capa1.py (scheme A); see each method.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib as L


@dataclass(frozen=True)
class Esquema:
    """Tone plan and timing of one modem variant (c1_shared.py:15-44 / capa1.py:12-18)."""
    nombre: str
    sample_rate: int
    base_freq: int
    freq_step: int
    max_byte: int
    tol_check: int
    sep_freq: int
    ping_freq: int
    eof_freq: int
    tone_n: int
    sep_n: int
    sep_short_n: int
    ping_n: int
    eof_n: int
    win_n: int
    amp: float
    gain: float

    def c_struct(self) -> L.WnScheme:
        s = L.WnScheme()
        for name, _ in L.WnScheme._fields_:
            setattr(s, name, getattr(self, name))
        return s

    def max_frame_samples(self, n_bytes: int, variant: int = L.EMIT_LIVE) -> int:
        """Upper bound of the samples a frame of n_bytes occupies (no repeated bytes)."""
        return int(L.lib.wn_host_frame_samples(C.byref(self.c_struct()), bytes(range(256)) * 2, n_bytes, variant))


def esquema_b() -> Esquema:
    """The live modem: the durations are the reference's own float expressions
    (c1_shared.py:16-17, 323, 390, 401, 420, 428)."""
    sr, byte_d, sil_d = 44100, 0.24, 0.30
    return Esquema("B", sr, 350, 70, 255, 1, 200, 250, 300,
                   tone_n=int(sr * byte_d), sep_n=int(sr * sil_d), sep_short_n=int(sr * (sil_d * 4 / 5)),
                   ping_n=int(sr * (byte_d * 2)), eof_n=int(sr * (byte_d * 2)), win_n=int(sr * (byte_d - 0.08)),
                   amp=0.5, gain=1.95)


def esquema_a() -> Esquema:
    """The prototype of capa1.py:12-18, 33-46: one 50 ms tone per byte, then 900 Hz."""
    sr = 44100
    n = int(sr * 0.05)
    return Esquema("A", sr, 500, 20, 127, 0, 0, 0, 900, tone_n=n, sep_n=0, sep_short_n=0, ping_n=0,
                   eof_n=n, win_n=n, amp=0.5, gain=1.0)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pad4(n: int) -> int:
    return (n + 3) & ~3


REC_DTYPE = np.dtype([("crc", "<u4"), ("status", "u1"), ("crc_ok", "u1"), ("tipo", "u1"), ("length", "u1")])


class Modem:
    """One scheme on one GPU.  Methods enqueue kernels on torch's current stream."""

    def __init__(self, esquema: Esquema | None = None, device: int | None = None):
        if not torch.cuda.is_available():
            raise RuntimeError("wavenet_amd needs a GPU: the modem runs as HIP kernels only")
        self.esquema = esquema or esquema_b()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            s = self.esquema.c_struct()
            L.check(L.lib.wn_scheme_init(C.byref(s), C.byref(self._ctx)), "wn_scheme_init")

    def __del__(self):
        ctx = getattr(self, "_ctx", None)
        lib = getattr(L, "lib", None)     # module globals may already be gone at interpreter exit
        if ctx and lib is not None:
            lib.wn_ctx_destroy(ctx)
            self._ctx = None

    # ------------------------------------------------------------------ K4
    def crc32(self, msgs: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
        """calcular_checksum (c1_shared.py:178-187) over rows; returns int64 CRC values."""
        assert msgs.dtype == torch.uint8 and msgs.dim() == 2 and msgs.is_contiguous()
        assert lens.dtype == torch.int32 and lens.numel() == msgs.shape[0]
        out = torch.empty(msgs.shape[0], dtype=torch.int32, device=self.device)
        L.check(L.lib.wn_crc32_batch(self._ctx, _ptr(msgs), _ptr(lens), msgs.shape[1], msgs.shape[0], _ptr(out),
                                     _stream()), "wn_crc32_batch")
        return out.to(torch.int64) & 0xFFFFFFFF

    def build_frames(self, fields: torch.Tensor, payloads: torch.Tensor, frame_stride: int = 128):
        """Trama(...)._build (c1_shared.py:105-111).  fields [n,16] u8 = version, mac_o, mac_d,
        tipo, length, 0; payloads [n,ps] u8.  Returns frames [n,frame_stride] u8, lens [n] i32."""
        n = fields.shape[0]
        assert fields.dtype == torch.uint8 and fields.shape[1] == 16 and fields.is_contiguous()
        assert payloads.dtype == torch.uint8 and payloads.is_contiguous() and payloads.shape[0] == n
        frames = torch.empty((n, frame_stride), dtype=torch.uint8, device=self.device)
        lens = torch.empty(n, dtype=torch.int32, device=self.device)
        L.check(L.lib.wn_frame_build_batch(self._ctx, _ptr(fields), _ptr(payloads), payloads.shape[1], n, _ptr(frames),
                                           frame_stride, _ptr(lens), _stream()), "wn_frame_build_batch")
        return frames, lens

    def parse_frames(self, frames: torch.Tensor, nbytes: torch.Tensor) -> torch.Tensor:
        """Trama(bytes_trama)._parse + get_checksum_valido (c1_shared.py:86-103, 119-125).
        Returns [n,8] u8 records (view as REC_DTYPE on the host)."""
        assert frames.dtype == torch.uint8 and frames.dim() == 2 and frames.is_contiguous()
        assert nbytes.dtype == torch.int32 and nbytes.numel() == frames.shape[0]
        out = torch.empty((frames.shape[0], 8), dtype=torch.uint8, device=self.device)
        L.check(L.lib.wn_frame_parse_validate_batch(self._ctx, _ptr(frames), _ptr(nbytes), frames.shape[1],
                                                    frames.shape[0], _ptr(out), _stream()),
                "wn_frame_parse_validate_batch")
        return out

    @staticmethod
    def records_to_numpy(rec: torch.Tensor) -> np.ndarray:
        return rec.cpu().numpy().view(REC_DTYPE).reshape(-1)

    # ------------------------------------------------------------------ K1
    def layout(self, frames: torch.Tensor, lens: torch.Tensor, variant: int = L.EMIT_LIVE):
        """Tone schedule of every frame (the loop of emitir_trama, c1_shared.py:420-433)."""
        n, stride = frames.shape
        seg_cap = 2 * stride + 4
        seg_start = torch.empty((n, seg_cap + 1), dtype=torch.int32, device=self.device)
        seg_freq = torch.empty((n, seg_cap), dtype=torch.int16, device=self.device)
        n_segs = torch.empty(n, dtype=torch.int32, device=self.device)
        n_samples = torch.empty(n, dtype=torch.int64, device=self.device)
        L.check(L.lib.wn_synth_layout(self._ctx, _ptr(frames), _ptr(lens), stride, n, variant, seg_cap,
                                      _ptr(seg_start), _ptr(seg_freq), _ptr(n_segs), _ptr(n_samples), _stream()),
                "wn_synth_layout")
        return seg_start, seg_freq, n_segs, n_samples

    def synth(self, layout, dst_off: torch.Tensor, audio: torch.Tensor, slot_len: int,
              fill_to: torch.Tensor | None = None, beside: bool = False):
        """transmite_freq / transmitir_silencio for every segment (c1_shared.py:386-406) into
        `audio` (float32 or int16, 1-D or [streams, stride] contiguous).  `beside`: the
        small-footprint kernel, which finds room on CUs that a detection launch on another
        stream occupies (the same samples)."""
        seg_start, seg_freq, n_segs, _ = layout
        fmt = {torch.float32: L.FMT_F32, torch.int16: L.FMT_I16}[audio.dtype] | (L.SYNTH_BESIDE if beside else 0)
        assert audio.is_contiguous() and dst_off.dtype == torch.int64
        L.check(L.lib.wn_synth_batch(self._ctx, _ptr(seg_start), _ptr(seg_freq), _ptr(n_segs), seg_freq.shape[1],
                                     n_segs.numel(), _ptr(dst_off), _ptr(fill_to), slot_len, _ptr(audio),
                                     audio.numel(), fmt, _stream()), "wn_synth_batch")

    def encode_streams(self, frames: torch.Tensor, lens: torch.Tensor, stream_stride: int | None = None,
                       variant: int = L.EMIT_LIVE, dtype=torch.float32, audio: torch.Tensor | None = None,
                       beside: bool = False):
        """emitir_trama (c1_shared.py:415-434) for one frame per stream: returns audio
        [n, stream_stride], each stream zero-filled after its frame."""
        n = frames.shape[0]
        if stream_stride is None:
            stream_stride = pad4(self.esquema.max_frame_samples(frames.shape[1], variant))
        if audio is None:
            audio = torch.empty((n, stream_stride), dtype=dtype, device=self.device)
        lay = self.layout(frames, lens, variant)
        dst = torch.arange(n, dtype=torch.int64, device=self.device) * stream_stride
        fill = torch.full((n,), stream_stride, dtype=torch.int64, device=self.device)
        self.synth(lay, dst, audio, stream_stride, fill, beside=beside)
        return audio, lay[3]

    def awgn(self, audio: torch.Tensor, sigma: float, seed: int):
        """Synthetic channel noise, in place."""
        assert audio.dtype == torch.float32 and audio.is_contiguous()
        L.check(L.lib.wn_channel_awgn(self._ctx, _ptr(audio), audio.numel(), float(sigma), int(seed), _stream()),
                "wn_channel_awgn")

    # ------------------------------------------------------------------ K2 / K3
    def detect(self, audio: torch.Tensor, n_streams: int, stream_stride: int, n_windows: int, first_sample: int = 0):
        """detectar_frecuencia + the class of its result for every window
        (c1_shared.py:286-299, 331-341).  Returns cls [S,W] i16 (class words), mag [S,W] f32, bin [S,W] i32."""
        assert audio.dtype == torch.float32 and audio.is_contiguous()
        cls = torch.empty((n_streams, n_windows), dtype=torch.int16, device=self.device)
        mag = torch.empty((n_streams, n_windows), dtype=torch.float32, device=self.device)
        kbin = torch.empty((n_streams, n_windows), dtype=torch.int32, device=self.device)
        L.check(L.lib.wn_detect_windows(self._ctx, _ptr(audio), audio.numel(), n_streams, stream_stride, first_sample,
                                        n_windows, _ptr(cls), _ptr(mag), _ptr(kbin), _stream()), "wn_detect_windows")
        return cls, mag, kbin

    def detect_classes_into(self, audio: torch.Tensor, n_streams: int, stream_stride: int, n_windows: int,
                            cls: torch.Tensor, first_sample: int = 0):
        """The class words of detect() written into a tensor the caller already holds, through the
        entry point that allocates nothing (for graph capture and serving loops); at most
        DETECT_NOALLOC_MAX_WINDOWS windows per call, calls on one GPU must not overlap."""
        assert audio.dtype == torch.float32 and audio.is_contiguous()
        assert cls.dtype == torch.int16 and cls.is_contiguous() and cls.numel() >= n_streams * n_windows
        L.check(L.lib.wn_detect_windows_noalloc(self._ctx, _ptr(audio), audio.numel(), n_streams, stream_stride,
                                                first_sample, n_windows, _ptr(cls), _stream()),
                "wn_detect_windows_noalloc")
        return cls

    def set_tie_band(self, rel: float):
        L.check(L.lib.wn_detect_set_tie_band(self._ctx, float(rel)), "wn_detect_set_tie_band")

    def detect_stats(self, columns: bool = False, paths: bool = False):
        """[windows, two-bin float64 re-evaluations, whole-spectrum ones] since the last call; with
        `columns` also the spectrum columns whose second stage was computed, the (row tile,
        column) products of the first stage that were, and the windows that were done twice; with
        `paths` also the passes the long way (first stage of every column tile, or of all but the
        weakest one), of them those that left the weakest column tile out, the windows that took
        one tile of sixteen chosen columns, the windows decided from the separator's bin alone and
        those decided from the bin of another tone of the scheme alone."""
        out = (C.c_int64 * 12)()
        L.check(L.lib.wn_detect_stats(self._ctx, out, _stream()), "wn_detect_stats")
        return list(out)[:11 if paths else (6 if columns else 3)]

    def sync_scan(self, audio: torch.Tensor, n_streams: int, stream_stride: int, n_windows: int, first_sample: int = 0):
        """Ping correlator (the test of escuchar_ping, c1_shared.py:380-382) per window:
        ping energy, window energy, verdict (1 ping / 0 not / 2 undecided)."""
        assert audio.dtype == torch.float32 and audio.is_contiguous()
        pe = torch.empty((n_streams, n_windows), dtype=torch.float32, device=self.device)
        we = torch.empty((n_streams, n_windows), dtype=torch.float32, device=self.device)
        vd = torch.empty((n_streams, n_windows), dtype=torch.uint8, device=self.device)
        L.check(L.lib.wn_sync_scan(self._ctx, _ptr(audio), audio.numel(), n_streams, stream_stride, first_sample,
                                   n_windows, _ptr(pe), _ptr(we), _ptr(vd), _stream()), "wn_sync_scan")
        return pe, we, vd

    # ------------------------------------------------------------------ receive state machine
    def timeout_windows(self, timeout: float) -> int:
        """Windows one listen call consumes before `time.time() - start > timeout` holds
        (c1_shared.py:319, 334) when the clock is samples consumed / sample rate."""
        e = self.esquema
        w = int(timeout * e.sample_rate // e.win_n)
        while w * e.win_n / e.sample_rate <= timeout:
            w += 1
        return max(w, 1)

    def slice(self, cls: torch.Tensor, timeout_windows: int, max_calls: int = 1, frame_cap: int = 128):
        """The loop of escuchar_y_retornar_trama (c1_shared.py:312-351) over class words."""
        n_streams, n_windows = cls.shape
        assert cls.dtype == torch.int16 and cls.is_contiguous()
        by = torch.zeros((n_streams, max_calls, frame_cap), dtype=torch.uint8, device=self.device)
        # the five integer results in one zeroed allocation (one fill instead of five)
        k = n_streams * max_calls
        meta = torch.zeros(4 * k + n_streams, dtype=torch.int32, device=self.device)
        lens, start, end, reason = (meta[i * k:(i + 1) * k].view(n_streams, max_calls) for i in range(4))
        n_calls = meta[4 * k:]
        L.check(L.lib.wn_slice_frames(self._ctx, _ptr(cls), n_streams, n_windows, timeout_windows, max_calls,
                                      frame_cap, _ptr(by), _ptr(lens), _ptr(start), _ptr(end), _ptr(reason),
                                      _ptr(n_calls), _stream()), "wn_slice_frames")
        return by, lens, start, end, reason, n_calls

    def decode_streams(self, audio: torch.Tensor, n_streams: int, stream_stride: int, n_windows: int,
                       timeout: float, max_calls: int = 1, frame_cap: int = 128, first_sample: int = 0):
        """escuchar_y_retornar_trama + Trama(bytes)._parse + get_checksum_valido for every
        stream: detection, state machine, parse/validate.  Returns a dict of device tensors."""
        cls, mag, kbin = self.detect(audio, n_streams, stream_stride, n_windows, first_sample)
        by, lens, start, end, reason, n_calls = self.slice(cls, self.timeout_windows(timeout), max_calls, frame_cap)
        rec = self.parse_frames(by.view(n_streams * max_calls, frame_cap), lens.view(-1))
        return dict(cls=cls, mag=mag, bin=kbin, bytes=by, lens=lens, start=start, end=end, reason=reason,
                    n_calls=n_calls, rec=rec.view(n_streams, max_calls, 8))

    def selftest_mfma(self) -> np.ndarray:
        out = torch.zeros(256, dtype=torch.float32, device=self.device)
        L.check(L.lib.wn_selftest_mfma(_ptr(out), _stream()), "wn_selftest_mfma")
        return out.cpu().numpy().reshape(64, 4)


# host-only helpers (usable without a GPU)
def host_class_lut(esq: Esquema) -> np.ndarray:
    out = np.zeros(esq.win_n // 2, dtype=np.uint16)
    n = L.lib.wn_host_class_lut(C.byref(esq.c_struct()), out.ctypes.data_as(C.POINTER(C.c_uint16)), len(out))
    if n < 0:
        raise L.WnError(n, "wn_host_class_lut")
    return out


def host_dft_plan(esq: Esquema):
    a, b = C.c_int32(), C.c_int32()
    L.check(L.lib.wn_host_dft_plan(C.byref(esq.c_struct()), C.byref(a), C.byref(b)), "wn_host_dft_plan")
    return a.value, b.value


def host_sine_tables(esq: Esquema):
    q = esq.sample_rate // 4 + 1
    f = np.zeros(q, dtype=np.float32)
    h = np.zeros(q, dtype=np.int16)
    n = L.lib.wn_host_sine_tables(C.byref(esq.c_struct()), f.ctypes.data_as(C.POINTER(C.c_float)),
                                  h.ctypes.data_as(C.POINTER(C.c_int16)), q)
    if n < 0:
        raise L.WnError(n, "wn_host_sine_tables")
    return f, h


def host_frame_samples(esq: Esquema, frame: bytes, variant: int = L.EMIT_LIVE) -> int:
    return int(L.lib.wn_host_frame_samples(C.byref(esq.c_struct()), bytes(frame), len(frame), variant))

"""CPU restatement of the reference's layer-1 acoustic modem.  TEST INFRASTRUCTURE ONLY.

This module is the parity checker for the HIP kernels: tests, the smoke check and the
CPU-baseline leg of the benchmark may import it; no product code does.  It follows the
reference line by line, with the same NumPy / zlib calls the reference makes, so that
its results are the reference's results on the same input.  It is pinned against vectors
produced by importing the reference itself (tests/golden/make_golden.py).

Cited reference files (relative to the reference tree):
  B = DispositivoWaveNET/dispositivo_wavenet/c1_shared.py   ("scheme B", the live modem)
  A = DispositivoWaveNET/capa1.py                           ("scheme A", the prototype)

The reference talks to a sound card through sounddevice.play/rec and reads wall time.
Here a capture is a 1-D array, `rec` reads it sequentially (silence past its end) and the
clock is samples-consumed / sample-rate, which is what the golden generator's stub does
to the reference as well.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass

import numpy as np

# ---------------------------------------------------------------------------------------
# Constants (B:15-44, A:12-18)
# ---------------------------------------------------------------------------------------
SAMPLE_RATE = 44100
BYTE_DURATION = 0.24
SILENCE_DURATION = 0.30
BASE_FREQ = 350
SILENCE_FREQ = 200
FREQ_STEP = 70
PING_FREQ = 250
FREQ_EOF = 300

TIPO_PING = 0x01
TIPO_TRAMA_ARCHIVO_INFO = 0x02
TIPO_TRAMA_ARCHIVO = 0x03
TIPO_TRAMA_FINAL_ARCHIVO = 0x04
TIPO_ERROR = 0x05
TIPO_OK = 0x06
VERSION = 1

A_SAMPLE_RATE = 44100
A_BYTE_DURATION = 0.05
A_BASE_FREQ = 500
A_FREQ_STEP = 20
A_EOF_FREQ = 900


@dataclass(frozen=True)
class Scheme:
    """Tone plan and timing of one modem variant, in the units the C-ABI uses."""
    name: str
    sample_rate: int
    base_freq: int
    freq_step: int
    max_byte: int          # highest byte value freq_to_byte accepts
    tol_check: bool        # B:161 distance test present / A:26-30 absent
    sep_freq: int          # 0 = the scheme has no separators
    ping_freq: int         # 0 = none
    eof_freq: int
    tone_n: int
    sep_n: int
    sep_short_n: int
    ping_n: int
    eof_n: int
    win_n: int
    amp: float             # first factor applied to sin (0.5 in both schemes)
    gain: float            # second factor (1.95 in B:392, absent = 1.0 in A)


def scheme_b() -> Scheme:
    sr = SAMPLE_RATE
    return Scheme(
        name="B", sample_rate=sr, base_freq=BASE_FREQ, freq_step=FREQ_STEP, max_byte=255,
        tol_check=True, sep_freq=SILENCE_FREQ, ping_freq=PING_FREQ, eof_freq=FREQ_EOF,
        tone_n=int(sr * BYTE_DURATION),                      # B:390
        sep_n=int(sr * SILENCE_DURATION),                    # B:401
        sep_short_n=int(sr * (SILENCE_DURATION * 4 / 5)),    # B:428
        ping_n=int(sr * (BYTE_DURATION * 2)),                # B:420
        eof_n=int(sr * (BYTE_DURATION * 2)),                 # B:433
        win_n=int(sr * (BYTE_DURATION - 0.08)),              # B:323
        amp=0.5, gain=1.95)


def scheme_a() -> Scheme:
    sr = A_SAMPLE_RATE
    n = int(sr * A_BYTE_DURATION)                            # A:34
    return Scheme(
        name="A", sample_rate=sr, base_freq=A_BASE_FREQ, freq_step=A_FREQ_STEP, max_byte=127,
        tol_check=False, sep_freq=0, ping_freq=0, eof_freq=A_EOF_FREQ,
        tone_n=n, sep_n=0, sep_short_n=0, ping_n=0, eof_n=n, win_n=n, amp=0.5, gain=1.0)


# ---------------------------------------------------------------------------------------
# Byte <-> frequency (B:146-163, A:20-30)
# ---------------------------------------------------------------------------------------
def byte_to_freq(byte: int, sch: Scheme) -> int:
    return sch.base_freq + byte * sch.freq_step


def freq_to_byte(freq: float, sch: Scheme):
    byte = int(round((freq - sch.base_freq) / sch.freq_step))     # Python round: half to even
    if 0 <= byte <= sch.max_byte:
        if not sch.tol_check:
            return byte
        freq_calc = sch.base_freq + byte * sch.freq_step
        if abs(freq - freq_calc) < (sch.freq_step / 2):
            return byte
    return None


# ---------------------------------------------------------------------------------------
# Checksum and frames (B:57-125, 178-280; A:80-91)
# ---------------------------------------------------------------------------------------
def crc32_be(data: bytes) -> bytes:
    """B:186-187: zlib CRC-32, four bytes, most significant first."""
    return zlib.crc32(bytes(data)).to_bytes(4, byteorder="big")


def build_frame(version: int, mac_o: bytes, mac_d: bytes, tipo: int, payload: bytes) -> bytes:
    """B:105-111 / A:86-91: [version][mac_o 6][mac_d 6][tipo][len][payload][crc32 BE]."""
    header = bytes([version]) + bytes(mac_o) + bytes(mac_d) + bytes([tipo, len(payload)])
    return header + bytes(payload) + crc32_be(header + bytes(payload))


class FrameError(ValueError):
    pass


@dataclass
class ParsedFrame:
    version: int
    mac_o: bytes
    mac_d: bytes
    tipo: int
    length: int
    payload: bytes
    checksum: bytes
    crc_ok: bool


# status codes shared with the C-ABI's parse kernel
ST_OK, ST_EMPTY, ST_SHORT_HEADER, ST_SHORT_BODY = 0, 1, 2, 3
PARSE_MESSAGES = {
    ST_EMPTY: "Faltan argumentos para crear la trama",                                # B:72-77
    ST_SHORT_HEADER: "Trama muy corta",                                              # B:90-91
    ST_SHORT_BODY: "La trama no contiene suficiente longitud según el campo 'length'",  # B:99-100
}


def parse_status(raw: bytes) -> int:
    if len(raw) == 0:
        return ST_EMPTY
    if len(raw) < 15:
        return ST_SHORT_HEADER
    if len(raw) < 15 + raw[14] + 4:
        return ST_SHORT_BODY
    return ST_OK


def parse_frame(raw: bytes) -> ParsedFrame:
    """B:72-103 and 119-125; trailing bytes beyond 15+len+4 are ignored."""
    raw = bytes(raw)
    st = parse_status(raw)
    if st != ST_OK:
        raise FrameError(PARSE_MESSAGES[st])
    length = raw[14]
    payload = raw[15:15 + length]
    checksum = raw[15 + length:15 + length + 4]
    return ParsedFrame(raw[0], raw[1:7], raw[7:13], raw[13], length, payload, checksum,
                       checksum == crc32_be(raw[:15 + length]))


def info_frame(mac_o: bytes, mac_d: bytes, count: int, name: str) -> bytes:
    """B:208-231."""
    return build_frame(VERSION, mac_o, mac_d, TIPO_TRAMA_ARCHIVO_INFO,
                       count.to_bytes(4, byteorder="big") + name.encode("utf-8"))


def decode_info_payload(payload: bytes):
    """B:233-249."""
    return int.from_bytes(payload[:4], byteorder="big"), bytes(payload[4:]).decode("utf-8")


def ok_frame(mac_o: bytes, mac_d: bytes, checksum: bytes) -> bytes:
    """B:266-280."""
    return build_frame(VERSION, mac_o, mac_d, TIPO_OK, checksum)


def frames_from_bytes(data: bytes, mac_o: bytes, mac_d: bytes) -> list:
    """B:494-513 (and B:473-492): 107-byte chunks, type ARCHIVO."""
    return [build_frame(VERSION, mac_o, mac_d, TIPO_TRAMA_ARCHIVO, data[i:i + 107])
            for i in range(0, len(data), 107)]


def string_transfer_frames(data: bytes, mac_o: bytes, mac_d: bytes) -> list:
    """The frames enviar_string_por_sonido puts on air, in order
    (c1_communication.py:247-266): info frame (count, "str"), data frames, the last one
    re-typed FINAL and rebuilt."""
    frames = frames_from_bytes(data, mac_o, mac_d)
    out = [info_frame(mac_o, mac_d, len(frames), "str")]
    for i, fr in enumerate(frames):
        if i == len(frames) - 1:
            p = parse_frame(fr)
            fr = build_frame(p.version, p.mac_o, p.mac_d, TIPO_TRAMA_FINAL_ARCHIVO, p.payload)
        out.append(fr)
    return out


# ---------------------------------------------------------------------------------------
# Synthesis (B:386-434, 440-471; A:33-46, 163-173)
# ---------------------------------------------------------------------------------------
def tone(freq: float, n: int, duration: float, sch: Scheme) -> np.ndarray:
    """B:390-393 / A:34-35.  `duration` is the float the reference passes to linspace."""
    t = np.linspace(0, duration, n, False)
    x = sch.amp * np.sin(2 * np.pi * freq * t)
    if sch.gain != 1.0:
        x = x * sch.gain
        x = np.clip(x, -1.0, 1.0)
    return x


LIVE, WAV, = 0, 1   # emission variants of scheme B; scheme A has one


def schedule(frame: bytes, sch: Scheme, variant: int = LIVE) -> list:
    """Tone schedule of one frame: list of (freq_hz, n_samples, duration_float)."""
    seg = []
    if sch.name == "A":                                           # A:40-46
        for b in frame:
            seg.append((byte_to_freq(b, sch), sch.tone_n, A_BYTE_DURATION))
        seg.append((sch.eof_freq, sch.eof_n, A_BYTE_DURATION))
        return seg
    if variant == LIVE:                                           # B:415-434
        seg.append((sch.ping_freq, sch.ping_n, BYTE_DURATION * 2))
        prev = None
        for b in frame:
            if prev != b:
                seg.append((sch.sep_freq, sch.sep_n, SILENCE_DURATION))
            else:
                seg.append((sch.sep_freq, sch.sep_short_n, SILENCE_DURATION * 4 / 5))
            seg.append((byte_to_freq(b, sch), sch.tone_n, BYTE_DURATION))
            prev = b
        seg.append((sch.sep_freq, sch.sep_n, SILENCE_DURATION))
        seg.append((sch.eof_freq, sch.eof_n, BYTE_DURATION * 2))
        return seg
    # WAV (B:445-466): ping tone of separator length, separator after every tone, short EOF
    seg.append((sch.ping_freq, sch.sep_n, SILENCE_DURATION))
    seg.append((sch.sep_freq, sch.sep_n, SILENCE_DURATION))
    for b in frame:
        seg.append((byte_to_freq(b, sch), sch.tone_n, BYTE_DURATION))
        seg.append((sch.sep_freq, sch.sep_n, SILENCE_DURATION))
    seg.append((sch.eof_freq, sch.tone_n, BYTE_DURATION))
    seg.append((sch.sep_freq, sch.sep_n, SILENCE_DURATION))
    return seg


def emit_frame(frame: bytes, sch: Scheme) -> np.ndarray:
    """Audio emitir_trama hands to the sound card, concatenated (float64)."""
    return np.concatenate([tone(f, n, d, sch) for f, n, d in schedule(frame, sch, LIVE)])


def wav_int16(frame: bytes, sch: Scheme) -> np.ndarray:
    """B:445-467 / A:163-173: amplitude 0.5, no gain, np.int16(x*32767) (truncation)."""
    plain = Scheme(**{**sch.__dict__, "gain": 1.0})
    if sch.name == "A":
        segs = [(byte_to_freq(b, sch), sch.tone_n, A_BYTE_DURATION) for b in frame]   # A:166-169: no EOF
    else:
        segs = schedule(frame, sch, WAV)
    x = np.concatenate([tone(f, n, d, plain) for f, n, d in segs])
    return np.int16(x * 32767)


# ---------------------------------------------------------------------------------------
# Detection (B:286-299 / A:54-59) and per-bin classes
# ---------------------------------------------------------------------------------------
def detect_bin(audio: np.ndarray):
    """Index and magnitude of the dominant bin among [0, N//2), first maximum wins."""
    mags = np.abs(np.fft.fft(audio))
    k = int(np.argmax(mags[:len(mags) // 2]))
    return k, float(mags[k])


def detect_freq(audio: np.ndarray, sample_rate: int) -> float:
    k, _ = detect_bin(audio)
    return abs(np.fft.fftfreq(len(audio), 1 / sample_rate)[k])


# class word per bin: low 9 bits = byte value or 256 (no byte); bit 9 = ping; bit 10 = EOF
CLS_NONE = 256
CLS_PING = 1 << 9
CLS_EOF = 1 << 10


def bin_classes(n: int, sch: Scheme) -> np.ndarray:
    """What the receive loop would decide for a window whose dominant bin is k (B:331-341).
    Scheme A has no receive loop (A:48-52 is a stub); its only decode primitive is
    freq_to_byte (A:26-30), so its classes carry no ping/EOF flag and its 900 Hz end tone
    reads as byte 20, as it does in the reference."""
    freqs = np.abs(np.fft.fftfreq(n, 1 / sch.sample_rate))
    out = np.zeros(n // 2, dtype=np.uint16)
    for k in range(n // 2):
        f = float(freqs[k])
        b = freq_to_byte(f, sch)
        c = CLS_NONE if b is None else b
        if sch.name == "B":
            if abs(f - sch.ping_freq) < 10:
                c |= CLS_PING
            if abs(f - sch.eof_freq) < 10:
                c |= CLS_EOF
        out[k] = c
    return out


# ---------------------------------------------------------------------------------------
# Receive loop over a capture (B:301-384)
# ---------------------------------------------------------------------------------------
R_EOF, R_TIMEOUT = 0, 1      # why a listen call ended


class Capture:
    """Sequential reader over a 1-D capture; silence after its end; clock = cursor / sr."""

    def __init__(self, audio: np.ndarray, sample_rate: int = SAMPLE_RATE, cursor: int = 0):
        self.audio = np.asarray(audio, dtype=np.float64)
        self.sr = sample_rate
        self.cursor = cursor

    def now(self) -> float:
        return self.cursor / self.sr

    def rec(self, n: int) -> np.ndarray:
        a = self.audio[self.cursor:self.cursor + n]
        if len(a) < n:
            a = np.concatenate([a, np.zeros(n - len(a))])
        self.cursor += n
        return a

    def exhausted(self) -> bool:
        return self.cursor >= len(self.audio)


def listen_frame(cap: Capture, timeout: float, sch: Scheme, stop_at_end: bool = True):
    """escuchar_y_retornar_trama (B:301-355) up to, not including, Trama(bytes).
    Returns (list of byte values, reason, windows consumed).  With stop_at_end the loop
    also ends once the capture is exhausted: past that point every window is silence,
    which can add no byte, so the bytes are those the reference returns at its timeout."""
    received = []
    expect_silence = True
    heard_ping = False
    start = cap.now()
    used = 0
    while True:
        if cap.now() - start > timeout:
            return received, R_TIMEOUT, used
        if stop_at_end and cap.exhausted():
            return received, R_TIMEOUT, used
        audio = cap.rec(sch.win_n)
        used += 1
        freq = detect_freq(audio, sch.sample_rate)
        if not heard_ping:
            if abs(freq - sch.ping_freq) < 10:
                heard_ping = True
                start = cap.now()
            continue
        if abs(freq - sch.eof_freq) < 10:
            return received, R_EOF, used
        byte = freq_to_byte(freq, sch)
        if byte is not None:
            if expect_silence:
                continue
            received.append(byte)
            expect_silence = True
        else:
            expect_silence = False


def listen_ping(cap: Capture, timeout: float, sch: Scheme, stop_at_end: bool = True) -> bool:
    """escuchar_ping (B:357-384)."""
    start = cap.now()
    while True:
        if cap.now() - start > timeout:
            return False
        if stop_at_end and cap.exhausted():
            return False
        audio = cap.rec(sch.win_n)
        freq = detect_freq(audio, sch.sample_rate)
        if abs(freq - sch.ping_freq) < 10:
            return True


def window_bins(audio: np.ndarray, win_n: int):
    """Dominant bin and magnitude of every whole window of a capture (vector form of
    detect_bin; np.fft.fft along the last axis gives the same values row by row)."""
    nwin = len(audio) // win_n
    x = np.asarray(audio[:nwin * win_n], dtype=np.float64).reshape(nwin, win_n)
    mags = np.abs(np.fft.fft(x, axis=1))[:, :win_n // 2]
    k = np.argmax(mags, axis=1)
    return k.astype(np.int32), mags[np.arange(nwin), k]


def slice_classes(cls: np.ndarray, timeout_windows: int, max_frames: int):
    """The receive state machine run over a class sequence, call after call, the way
    repeated escuchar_y_retornar_trama calls consume one capture: each call starts at the
    window after the previous call's last window.  `timeout_windows` = number of windows
    a call may consume before its clock test fails (counted again from the ping window).
    Returns a list of (start_window, end_window_exclusive, bytes, reason)."""
    out = []
    w = 0
    n = len(cls)
    while w < n and len(out) < max_frames:
        start_w = w
        received = []
        expect_silence = True
        heard = False
        t0 = w
        reason = R_TIMEOUT
        while True:
            if w - t0 >= timeout_windows or w >= n:
                break
            c = int(cls[w])
            w += 1
            if not heard:
                if c & CLS_PING:
                    heard = True
                    t0 = w
                continue
            if c & CLS_EOF:
                reason = R_EOF
                break
            b = c & 0x1FF
            if b != CLS_NONE:
                if expect_silence:
                    continue
                received.append(b)
                expect_silence = True
            else:
                expect_silence = False
        out.append((start_w, w, bytes(received), reason))
    return out


def timeout_to_windows(timeout: float, sch: Scheme) -> int:
    """Smallest w with w*win_n/sr > timeout: a call checks its clock before each window,
    so it consumes exactly that many windows when nothing ends it earlier."""
    w = int(timeout * sch.sample_rate // sch.win_n)
    while w * sch.win_n / sch.sample_rate <= timeout:
        w += 1
    return w

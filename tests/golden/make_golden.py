#!/usr/bin/env python3
"""Generate tests/golden/modem_golden.json by running the REFERENCE's own layer-1 code.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

The reference modules are imported unchanged.  The one module they need that is not
installed, `sounddevice`, is replaced by an in-memory stand-in: play() appends to a
buffer, rec() reads a capture sequentially (silence past its end), and the `time` module
seen by c1_shared reports samples-consumed / 44100.  Every value written here is an
output of reference code; the oracle (oracle/modem_oracle.py) is asserted equal to it
while the file is written, and tests/test_oracle_golden.py re-checks that on every run.
Only data is stored: inputs, seeds, expected outputs and digests of long arrays.

NOT TO BE RUN AGAIN.  The fixture was generated once; a later attempt to import the reference
in this environment was refused (it may be read, not executed), and that refusal stands.  The
script is kept as the record of how the committed vectors were made.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/DispositivoWaveNET"

# ----------------------------------------------------------------------------- stand-ins
played = []


class FakeDevice:
    capture = np.zeros(0)
    cursor = 0


def _play(a, samplerate=None):
    played.append(np.array(a, dtype=np.float64, copy=True))


def _rec(n, samplerate=None, channels=1, dtype="float64"):
    a = FakeDevice.capture[FakeDevice.cursor:FakeDevice.cursor + n]
    if len(a) < n:
        a = np.concatenate([a, np.zeros(n - len(a))])
    FakeDevice.cursor += n
    return np.array(a, dtype=np.float64).reshape(n, 1)


sd = types.ModuleType("sounddevice")
sd.play = _play
sd.wait = lambda: None
sd.rec = _rec
sys.modules["sounddevice"] = sd
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import dispositivo_wavenet.c1_shared as cs   # noqa: E402  (the reference)
import capa1 as ca                           # noqa: E402  (the reference prototype)
from oracle import modem_oracle as mo        # noqa: E402

fake_time = types.SimpleNamespace(time=lambda: FakeDevice.cursor / 44100, sleep=lambda s: None)
cs.time = fake_time


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def take_played() -> np.ndarray:
    out = np.concatenate(played) if played else np.zeros(0)
    played.clear()
    return out


def set_capture(a):
    FakeDevice.capture = np.asarray(a, dtype=np.float64)
    FakeDevice.cursor = 0


MAC_O = bytes.fromhex("aabbccddeeff")
MAC_D = bytes.fromhex("112233445566")
SB, SA = mo.scheme_b(), mo.scheme_a()
G = {}

# ----------------------------------------------------------------------------- 1. frames
rs = np.random.RandomState(20250620)
frames = []
cases = [(1, MAC_O, MAC_D, 1, b""), (1, MAC_O, MAC_D, 3, b"hola")]
for i in range(256):
    ln = int(rs.randint(0, 108))
    cases.append((int(rs.randint(0, 256)), bytes(rs.randint(0, 256, 6).tolist()),
                  bytes(rs.randint(0, 256, 6).tolist()), int(rs.randint(0, 256)),
                  bytes(rs.randint(0, 256, ln).tolist())))
for ver, mo_, md_, tipo, pay in cases:
    t = cs.crear_trama(ver, mo_, md_, tipo, pay)
    raw = t.get_bytes()
    assert raw == mo.build_frame(ver, mo_, md_, tipo, pay)
    back = cs.Trama(bytes_trama=list(raw))
    assert back.get_checksum_valido() and mo.parse_frame(raw).crc_ok
    frames.append({"version": ver, "mac_o": mo_.hex(), "mac_d": md_.hex(), "tipo": tipo,
                   "payload": pay.hex(), "frame": raw.hex()})
G["frames"] = frames

t = cs.crear_trama_archivo_info(MAC_O, MAC_D, 1, "str")
assert t.get_bytes() == mo.info_frame(MAC_O, MAC_D, 1, "str")
G["info_frame"] = {"count": 1, "name": "str", "frame": t.get_bytes().hex(),
                   "decoded": list(cs.decodificar_payload_archivo_info(t.payload))}
t = cs.crear_trama_ok(MAC_O, MAC_D, bytes.fromhex("4b7b3be0"))
assert t.get_bytes() == mo.ok_frame(MAC_O, MAC_D, bytes.fromhex("4b7b3be0"))
G["ok_frame"] = {"checksum": "4b7b3be0", "frame": t.get_bytes().hex()}
G["capa1_frame"] = {"tipo": 2, "payload": b"hola".hex(),
                    "frame": bytes(ca.crear_trama(1, bytearray(MAC_O), bytearray(MAC_D), 2, b"hola")).hex()}
assert bytes.fromhex(G["capa1_frame"]["frame"]) == mo.build_frame(1, MAC_O, MAC_D, 2, b"hola")
G["crc_kat"] = {"input": b"123456789".hex(), "crc_be": cs.calcular_checksum(b"123456789").hex()}

# parse: corrupted, trailing bytes, length byte > 107, and the three error cases
parse_cases = []
base = bytes.fromhex(frames[1]["frame"])
long_pay = bytes(rs.randint(0, 256, 200).tolist())
variants = {
    "ok": base,
    "corrupt_payload": base[:16] + bytes([base[16] ^ 0x40]) + base[17:],
    "corrupt_crc": base[:-1] + bytes([base[-1] ^ 1]),
    "trailing": base + b"\x14\x14\x07",
    "len200": cs.crear_trama(1, MAC_O, MAC_D, 3, long_pay).get_bytes(),
    "short_header": base[:14],
    "short_body": base[:-1],
    "len_byte_too_big": base[:14] + bytes([60]) + base[15:],
    "exactly15_len0_nocrc": base[:14] + b"\x00",
}
for name, raw in variants.items():
    rec = {"name": name, "raw": raw.hex()}
    try:
        tr = cs.Trama(bytes_trama=list(raw))
        rec.update(error=None, version=tr.version, tipo=tr.tipo, length=tr.length,
                   payload=bytes(tr.payload).hex(), checksum=bytes(tr.checksum).hex(),
                   crc_ok=bool(tr.get_checksum_valido()))
        p = mo.parse_frame(raw)
        assert (p.version, p.tipo, p.length, p.payload, p.checksum, p.crc_ok) == \
            (tr.version, tr.tipo, tr.length, bytes(tr.payload), bytes(tr.checksum), rec["crc_ok"])
    except ValueError as e:
        rec.update(error=str(e))
        try:
            mo.parse_frame(raw)
            raise AssertionError("oracle accepted " + name)
        except mo.FrameError as e2:
            assert str(e2) == str(e)
    parse_cases.append(rec)
try:
    cs.Trama(bytes_trama=[])
except ValueError as e:
    parse_cases.append({"name": "empty", "raw": "", "error": str(e)})
    assert mo.PARSE_MESSAGES[mo.parse_status(b"")] == str(e)
G["parse_cases"] = parse_cases

# chunking of the sample text the reference ships (config 1 input)
text = open(os.path.join(REF, "prueba.txt"), "rb").read()
G["prueba_txt"] = text.hex()
tr_list = cs.obtener_tramas_desde_string(text, "aa:bb:cc:dd:ee:ff", "11:22:33:44:55:66")
assert [t.get_bytes() for t in tr_list] == mo.frames_from_bytes(text, MAC_O, MAC_D)
tr_list[-1].tipo = cs.TIPO_TRAMA_FINAL_ARCHIVO
tr_list[-1]._build()
on_air = [cs.crear_trama_archivo_info(MAC_O, MAC_D, len(tr_list), "str").get_bytes()] + \
         [t.get_bytes() for t in tr_list]
assert on_air == mo.string_transfer_frames(text, MAC_O, MAC_D)
G["prueba_on_air_frames"] = [f.hex() for f in on_air]

# ----------------------------------------------------------------------------- 2. per-bin classes
def ref_classes_b(n):
    freqs = np.abs(np.fft.fftfreq(n, 1 / cs.SAMPLE_RATE))
    out = []
    for k in range(n // 2):
        f = float(freqs[k])
        b = cs.freq_to_byte(f)
        c = 256 if b is None else b
        if abs(f - cs.PING_FREQ) < 10:
            c |= 1 << 9
        if abs(f - cs.FREQ_EOF) < 10:
            c |= 1 << 10
        out.append(int(c))
    return out


def ref_classes_a(n):
    freqs = np.abs(np.fft.fftfreq(n, 1 / ca.sample_rate))
    return [256 if ca.freq_to_byte(float(f)) is None else int(ca.freq_to_byte(float(f)))
            for f in freqs[:n // 2]]


G["classes_b_7055"] = ref_classes_b(7055)
G["classes_a_2205"] = ref_classes_a(2205)
assert G["classes_b_7055"] == mo.bin_classes(7055, SB).tolist()
assert G["classes_a_2205"] == mo.bin_classes(2205, SA).tolist()

# ----------------------------------------------------------------------------- 3. tones
tones = []
for freq, dur, n_expect in [(250, cs.BYTE_DURATION * 2, 21168), (200, cs.SILENCE_DURATION, 13230),
                            (200, cs.SILENCE_DURATION * 4 / 5, 10584), (300, cs.BYTE_DURATION * 2, 21168),
                            (350, cs.BYTE_DURATION, 10584), (4410, cs.BYTE_DURATION, 10584),
                            (8820, cs.BYTE_DURATION, 10584), (13230, cs.BYTE_DURATION, 10584),
                            (18200, cs.BYTE_DURATION, 10584), (17710, cs.BYTE_DURATION, 10584)]:
    if freq == 200:
        cs.transmitir_silencio(duration=dur)
    else:
        cs.transmite_freq(freq, duration=dur)
    x = take_played()
    assert len(x) == n_expect
    assert np.array_equal(x, mo.tone(freq, n_expect, dur, SB))
    idx = sorted(set(list(range(16)) + list(range(n_expect // 2, n_expect // 2 + 16)) +
                     list(range(n_expect - 16, n_expect))))
    tones.append({"freq": freq, "n": n_expect, "sha256_f64": sha(x), "peak": float(np.max(np.abs(x))),
                  "idx": idx, "samples": [float(x[i]) for i in idx]})
G["tones_b"] = tones
ca.transmite_freq(ca.byte_to_freq(100))
x = take_played()
assert np.array_equal(x, mo.tone(2500, 2205, ca.byte_duration, SA))
G["tone_a"] = {"freq": 2500, "n": len(x), "sha256_f64": sha(x), "samples_head": [float(v) for v in x[:16]]}

# ----------------------------------------------------------------------------- 4. pure-tone windows
pure = []
for label, freq in [("byte0", 350), ("byte58", 4410), ("byte121", 8820), ("byte184", 13230),
                    ("byte255", 18200), ("sep", 200), ("ping", 250), ("eof", 300)]:
    cs.transmite_freq(freq, duration=cs.BYTE_DURATION * 2)
    w = take_played()[:7055]
    f = float(cs.detectar_frecuencia(w, cs.SAMPLE_RATE))
    mags = np.abs(np.fft.fft(w))[:7055 // 2]
    k = int(np.argmax(mags))
    order = np.argsort(mags)
    assert abs(f - k * 44100 / 7055) < 1e-6
    assert mo.detect_bin(w) == (k, float(mags[k])) and mo.detect_freq(w, 44100) == f
    pure.append({"label": label, "freq": freq, "detected": f, "bin": k, "mag": float(mags[k]),
                 "runner_up_bin": int(order[-2]), "runner_up_mag": float(mags[order[-2]]),
                 "cls": G["classes_b_7055"][k]})
G["pure_windows_b"] = pure
rows = []
for b in range(128):
    ca.transmite_freq(ca.byte_to_freq(b))
    w = take_played()
    f = float(ca.detectar_frecuencia(w, ca.sample_rate))
    rows.append([b, f, ca.freq_to_byte(f)])
    assert mo.freq_to_byte(mo.detect_freq(w, 44100), SA) == ca.freq_to_byte(f)
G["capa1_byte_roundtrip"] = rows

# ----------------------------------------------------------------------------- 5. emit + listen
def ref_emit(raw: bytes) -> np.ndarray:
    cs.emitir_trama(cs.Trama(bytes_trama=list(raw)))
    return take_played()


def build_capture(frame_list, gaps, sigma, seed):
    parts = []
    for raw, gap in zip(frame_list, gaps):
        parts.append(np.zeros(gap))
        parts.append(ref_emit(raw))
    x = np.concatenate(parts)
    if sigma > 0:
        x = x + sigma * np.random.default_rng(seed).standard_normal(len(x))
    return x


def oracle_capture(frame_list, gaps, sigma, seed):
    parts = []
    for raw, gap in zip(frame_list, gaps):
        parts.append(np.zeros(gap))
        parts.append(mo.emit_frame(raw, SB))
    x = np.concatenate(parts)
    if sigma > 0:
        x = x + sigma * np.random.default_rng(seed).standard_normal(len(x))
    return x


listen = []
scen = [
    ("ping_frame_clean", [bytes.fromhex(frames[0]["frame"])], [0], 0.0, 1, 30),
    ("hola_clean_gap", [bytes.fromhex(frames[1]["frame"])], [12345], 0.0, 2, 30),
    ("info_noise", [bytes.fromhex(G["info_frame"]["frame"])], [7001], 0.1, 3, 30),
    ("repeats", [mo.build_frame(1, bytes(6), bytes(6), 3, bytes([7, 7, 7, 9, 9, 0, 0, 0, 0]))], [311], 0.0, 4, 30),
    ("prueba_three_frames_noise", on_air, [29000, 1234, 17], 0.1, 5, 123),
    ("prueba_three_frames_loud_noise", on_air, [5, 30000, 9999], 1.0, 6, 123),
    ("hola_short_timeout", [bytes.fromhex(frames[1]["frame"])], [100], 0.0, 7, 5),
]
for name, fl, gaps, sigma, seed, timeout in scen:
    x = build_capture(fl, gaps, sigma, seed)
    assert np.array_equal(x, oracle_capture(fl, gaps, sigma, seed))
    set_capture(x)
    cap = mo.Capture(x)
    calls = []
    for _ in range(len(fl) + 1):
        c0 = FakeDevice.cursor
        try:
            tr = cs.escuchar_y_retornar_trama(timeout)
            got, err = bytes(tr.bytes_trama), None
        except ValueError as e:
            got, err = None, str(e)
        used = (FakeDevice.cursor - c0) // 7055
        ob, oreason, oused = mo.listen_frame(cap, timeout, SB, stop_at_end=False)
        assert oused == used and cap.cursor == FakeDevice.cursor, (name, oused, used)
        if err is None:
            assert bytes(ob) == got, name
        else:
            assert mo.PARSE_MESSAGES[mo.parse_status(bytes(ob))] == err
        calls.append({"bytes": None if got is None else got.hex(), "error": err, "windows": used,
                      "oracle_bytes": bytes(ob).hex()})
    bins, mags = mo.window_bins(x, 7055)
    listen.append({"name": name, "frames": [f.hex() for f in fl], "gaps": gaps, "sigma": sigma,
                   "seed": seed, "timeout": timeout, "n_samples": len(x), "sha256_f64": sha(x),
                   "calls": calls, "window_bins": bins.tolist(),
                   "window_mags": [float(m) for m in mags]})
G["listen_b"] = listen

counts = []
for raw in [bytes.fromhex(frames[0]["frame"]), scen[3][1][0], on_air[1], bytes(126), bytes(range(126))]:
    n = len(ref_emit(raw))
    assert n == sum(s[1] for s in mo.schedule(raw, SB, mo.LIVE))
    counts.append({"frame": raw.hex(), "n_samples": n})
G["sample_counts_b"] = counts

# escuchar_ping
ping_cases = []
cs.ejecutar_ping()
ping_audio = take_played()
for name, x, timeout in [("ping_after_gap", np.concatenate([np.zeros(20000), ping_audio, np.zeros(9000)]), 3),
                         ("silence", np.zeros(200000), 3),
                         ("ping_too_late", np.concatenate([np.zeros(150000), ping_audio]), 3)]:
    set_capture(x)
    r = bool(cs.escuchar_ping(timeout))
    cap = mo.Capture(x)
    assert mo.listen_ping(cap, timeout, SB, stop_at_end=False) == r and cap.cursor == FakeDevice.cursor
    ping_cases.append({"name": name, "gap": int(np.argmax(x != 0)) if x.any() else len(x), "n_samples": len(x),
                       "timeout": timeout, "result": r, "windows": FakeDevice.cursor // 7055})
G["ping_b"] = ping_cases
G["ping_tone_n"] = len(ping_audio)

# ----------------------------------------------------------------------------- 6. WAV variants
wav_out = {}


def _write(name, rate, data):
    wav_out["rate"], wav_out["data"] = rate, np.array(data)


cs.write = _write
cs.guardar_trama_como_wav(cs.Trama(bytes_trama=list(bytes.fromhex(frames[1]["frame"]))), "unused.wav")
w = wav_out["data"]
assert w.dtype == np.int16 and np.array_equal(w, mo.wav_int16(bytes.fromhex(frames[1]["frame"]), SB))
G["wav_b"] = {"frame": frames[1]["frame"], "n": len(w), "sha256_i16": sha(w),
              "head": w[:32].tolist(), "at_13230": w[13230:13262].tolist(), "tail": w[-32:].tolist()}

ca.write = _write
old_argv, old_cwd = sys.argv, os.getcwd()
sys.argv = ["capa1.py", "-a", os.path.join(REF, "prueba.txt"), "-b", "aa:bb:cc:dd:ee:ff", "-c", "11:22:33:44:55:66"]
ca.main()
sys.argv = old_argv
a_played = take_played()
w = wav_out["data"]
assert np.array_equal(w, mo.wav_int16(bytes(range(128)), SA))
G["wav_a"] = {"message": bytes(range(128)).hex(), "n": len(w), "sha256_i16": sha(w), "head": w[:32].tolist(),
              "tail": w[-32:].tolist()}
# what capa1.main played: ping frame, the file's frames, then the literal 18-byte example
a_frames = [mo.build_frame(1, MAC_O, bytes.fromhex("ffffffffffff"), 1, b"")] + \
           [mo.build_frame(1, MAC_O, MAC_D, 2, text[i:i + 107]) for i in range(0, len(text), 107)] + \
           [bytes([1, 0xaa, 0xbb, 0xcc, 0xdd, 0xee, 0xff, 0x11, 0x22, 0x33, 0x44, 0x55, 0x66, 1, 42, 0x12, 0x34])]
a_expect = np.concatenate([np.concatenate([mo.tone(f, n, d, SA) for f, n, d in mo.schedule(fr, SA)]) for fr in a_frames])
assert np.array_equal(a_played, a_expect)
G["capa1_main_audio"] = {"frames": [f.hex() for f in a_frames], "n_samples": len(a_played), "sha256_f64": sha(a_played)}

out = os.path.join(HERE, "modem_golden.json")
with open(out, "w") as f:
    json.dump(G, f, separators=(",", ":"))
print("wrote", out, os.path.getsize(out), "bytes")

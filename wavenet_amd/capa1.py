"""The prototype modem of the reference (DispositivoWaveNET/capa1.py) on the HIP kernels:
tones of 500 + 20*byte Hz, 50 ms each, no separators, a 900 Hz end tone, 2205-sample
detection windows.  The reference has a transmitter and a detector but no receive loop
(`recibir_trama`, capa1.py:48-52, is a stub); `decodificar_audio` below is this build's
loopback helper for that scheme and is documented as such.
"""
from __future__ import annotations

import argparse

import numpy as np
import torch

from . import _lib as L
from . import modem as M

sample_rate = 44100
byte_duration = 0.05
base_freq = 500
freq_step = 20
tipo_envio_archivo = 0x02
tipo_ping = 0x01
version = 1

_modem = None


def modem() -> M.Modem:
    global _modem
    if _modem is None:
        _modem = M.Modem(M.esquema_a())
    return _modem


def byte_to_freq(byte):
    return base_freq + byte * freq_step


def freq_to_byte(freq):
    """capa1.py:26-30: nearest tone, accepted for bytes 0..127 only, no distance test."""
    byte = round((freq - base_freq) / freq_step)
    return byte if 0 <= byte <= 127 else None


def mac_str_to_bytes(mac_str):
    try:
        parts = mac_str.split(":")
        if len(parts) != 6:
            raise ValueError
        return bytes(int(part, 16) for part in parts)
    except Exception:
        raise argparse.ArgumentTypeError(f"MAC inválida: {mac_str}")


def _filas(datos: bytes, ancho: int):
    fila = np.zeros((1, ancho), dtype=np.uint8)
    fila[0, :len(datos)] = np.frombuffer(bytes(datos), dtype=np.uint8)
    m = modem()
    return torch.from_numpy(fila).to(m.device), torch.tensor([len(datos)], dtype=torch.int32, device=m.device)


def crear_trama(version, bytes_mac_org, bytes_mac_dest, tipo, cur_payload):
    """[version][MAC_org][MAC_dest][tipo][length][payload][crc32 BE] (capa1.py:80-91)."""
    m = modem()
    campos = bytes([version]) + bytes(bytes_mac_org) + bytes(bytes_mac_dest) + bytes([tipo, len(cur_payload), 0])
    f, _ = _filas(campos, 16)
    p, _ = _filas(bytes(cur_payload), max(16, (len(cur_payload) + 15) & ~15))
    frames, lens = m.build_frames(f, p, 288)
    return bytearray(frames[0, :int(lens[0])].cpu().numpy().tobytes())


def emitir_trama(trama_bytes) -> torch.Tensor:
    """The audio capa1.emitir_trama plays (capa1.py:40-46): one tone per byte, then 900 Hz.
    Returns the float32 samples on the GPU."""
    m = modem()
    raw = bytes(trama_bytes)
    frames, lens = _filas(raw, max(16, (len(raw) + 15) & ~15))
    n = M.host_frame_samples(m.esquema, raw, L.EMIT_LIVE)
    audio, _ = m.encode_streams(frames, lens, M.pad4(n))
    return audio[0, :n]


def audio_int16(mensaje: bytes) -> np.ndarray:
    """The int16 samples of capa1's WAV block (capa1.py:163-173): tones only, no end tone."""
    m = modem()
    frames, lens = _filas(bytes(mensaje), max(16, (len(mensaje) + 15) & ~15))
    n = M.host_frame_samples(m.esquema, bytes(mensaje), L.EMIT_WAV)
    audio, _ = m.encode_streams(frames, lens, M.pad4(n), variant=L.EMIT_WAV, dtype=torch.int16)
    return audio[0, :n].cpu().numpy()


def detectar_frecuencia(audio, sample_rate):
    """Dominant frequency of one 2205-sample window (capa1.py:54-59)."""
    x = torch.as_tensor(np.asarray(audio) if not torch.is_tensor(audio) else audio).reshape(-1)
    if x.numel() != 2205 or sample_rate != 44100:
        raise ValueError("detectar_frecuencia: se espera una ventana de 2205 muestras a 44100 Hz")
    m = modem()
    buf = torch.zeros(M.pad4(2205), dtype=torch.float32, device=m.device)
    buf[:2205] = x.to(m.device, torch.float32)
    _, _, kbin = m.detect(buf, 1, buf.numel(), 1)
    return abs(int(kbin[0, 0]) * (1.0 / (2205 * (1 / sample_rate))))


def decodificar_audio(audio: torch.Tensor) -> bytes:
    """Loopback helper (not in the reference): detect every aligned 2205-sample window and
    map it through freq_to_byte; the 900 Hz end tone reads as byte 20, as it would there."""
    m = modem()
    n = audio.numel() // 2205
    buf = torch.zeros(M.pad4(n * 2205), dtype=torch.float32, device=m.device)
    buf[:n * 2205] = audio[:n * 2205].to(m.device, torch.float32)
    cls, _, _ = m.detect(buf, 1, buf.numel(), n)
    by, lens, *_ = m.slice(cls, 1, 1, max(n, 1))
    return bytes(by[0, 0, :int(lens[0, 0])].cpu().numpy().tolist())

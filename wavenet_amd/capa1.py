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
_altavoz: list = []     # what sd.play received in the reference: one float32 GPU tensor per played sound


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


def audio_reproducido(vaciar: bool = True) -> torch.Tensor:
    """Everything played so far (the reference hands it to the sound card, capa1.py:36-37), as
    one float32 tensor on the GPU."""
    global _altavoz
    out = torch.cat(_altavoz) if _altavoz else torch.zeros(0, dtype=torch.float32, device=modem().device)
    if vaciar:
        _altavoz = []
    return out


def transmite_freq(freq, duration=byte_duration) -> torch.Tensor:
    """One tone of whole-Hz frequency `freq` (capa1.py:33-37), synthesised as a one-segment schedule."""
    m = modem()
    n = int(sample_rate * duration)
    dev = m.device
    lay = (torch.tensor([[0, n]], dtype=torch.int32, device=dev), torch.tensor([[int(freq)]], dtype=torch.int16, device=dev),
           torch.tensor([1], dtype=torch.int32, device=dev), None)
    audio = torch.empty(M.pad4(n), dtype=torch.float32, device=dev)
    m.synth(lay, torch.zeros(1, dtype=torch.int64, device=dev), audio, M.pad4(n))
    _altavoz.append(audio[:n])
    return audio[:n]


def emitir_trama(trama_bytes) -> torch.Tensor:
    """The audio capa1.emitir_trama plays (capa1.py:40-46): one tone per byte, then 900 Hz, all
    tones of a frame in one launch.  Returns the float32 samples on the GPU (also kept for
    audio_reproducido)."""
    m = modem()
    raw = bytes(trama_bytes)
    frames, lens = _filas(raw, max(16, (len(raw) + 15) & ~15))
    n = M.host_frame_samples(m.esquema, raw, L.EMIT_LIVE)
    audio, _ = m.encode_streams(frames, lens, M.pad4(n))
    for byte in raw:
        print(f"Enviando byte {byte} → {byte_to_freq(byte):.1f} Hz")
    print("Trama transmitida.")
    _altavoz.append(audio[0, :n])
    return audio[0, :n]


def send_file_as_sound(file_path, str_org_mac, str_dest_mac):
    """The file in blocks of 107 bytes, one type-2 frame per block, each emitted as sound
    (capa1.py:94-110)."""
    bytes_mac_org = mac_str_to_bytes(str_org_mac)
    bytes_mac_dest = mac_str_to_bytes(str_dest_mac)
    with open(file_path, 'rb') as f:
        while True:
            block = f.read(107)
            if not block:
                break
            trama = crear_trama(1, bytes_mac_org, bytes_mac_dest, tipo_envio_archivo, block)
            print(f"Enviando trama de {len(trama)} bytes...")
            emitir_trama(trama)


def enviar_ping(mac_origin):
    """An empty type-1 frame to ff:ff:ff:ff:ff:ff (capa1.py:114-124)."""
    mac_origin = mac_str_to_bytes(mac_origin)
    trama = crear_trama(1, mac_origin, mac_str_to_bytes("ff:ff:ff:ff:ff:ff"), tipo_ping, b'')
    print(f"Enviando PING desde {mac_origin}...")
    emitir_trama(trama)


def main(argv=None):
    """capa1.py's command line (capa1.py:128-179): ping frame, the file's frames, the literal
    17-byte example frame, then the 128-tone WAV file."""
    import os

    from scipy.io.wavfile import write
    parser = argparse.ArgumentParser(description="Parser de archivo y direcciones MAC")
    parser.add_argument('-a', '--archivo', required=True, help="Ruta al archivo (ej: archivo.txt)")
    parser.add_argument('-b', '--mac_origen', required=True, help="MAC Address de origen (ej: aa:bb:cc:dd:ee:ff)")
    parser.add_argument('-c', '--mac_destino', required=True, help="MAC Address de destino (ej: 11:22:33:44:55:66)")
    args = parser.parse_args(argv)
    enviar_ping(args.mac_origen)
    if not os.path.exists(args.archivo):
        print(f"El archivo {args.archivo} no existe.")
        return
    send_file_as_sound(args.archivo, args.mac_origen, args.mac_destino)
    emitir_trama(bytearray([1, *b'\xaa\xbb\xcc\xdd\xee\xff', *b'\x11\x22\x33\x44\x55\x66', 1, 42, 0x12, 0x34]))
    file_path_bytes = "trama_128bytes_por_byte.wav"
    write(file_path_bytes, sample_rate, audio_int16(bytes(range(128))))
    print("Archivo guardado en: ", file_path_bytes)


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


if __name__ == '__main__':
    main()

"""The reference's layer-1 function surface (DispositivoWaveNET/dispositivo_wavenet/
c1_shared.py) over the HIP kernels and a synthetic audio device.

Same names, arguments, return values and error behaviour as the reference; every
checksum, frame, tone and detection goes through libwavenet_hip.so (one-element batches
here; `wavenet_amd.modem.Modem` is the batched form of the same calls).  There is no CPU
path: without a GPU the first call that needs the modem raises.

The sound card (sounddevice.play/rec) and the wall clock are replaced by a
`CanalSintetico`; install one with `usar_canal()`.
"""
from __future__ import annotations

import argparse
import logging
import threading

import numpy as np
import torch

from .. import _lib as L
from .. import modem as M
from .canal import CanalSintetico

# ------------------------------------------------------------------------------------------
# Constants (reference c1_shared.py:15-44)
# ------------------------------------------------------------------------------------------
SAMPLE_RATE = 44100
BYTE_DURATION = 0.24
SILENCE_DURATION = 0.30
BASE_FREQ = 350
SILENCE_FREQ = 200
FREQ_STEP = 70

TIPO_PING = 0x01
TIPO_TRAMA_ARCHIVO_INFO = 0x02
TIPO_TRAMA_ARCHIVO = 0x03
TIPO_TRAMA_FINAL_ARCHIVO = 0x04
TIPO_ERROR = 0x05
TIPO_OK = 0x06

VERSION = 1

TIME_TO_SAY_OK = 10
TIMES_TO_COMUNICATE_OK = 3
TIME_TO_SAY_128_BYTES = 113
TIMES_TO_COMUNICATE_128_BYTES = 3

PING_FREQ = 250
FREQ_EOF = 300

_MENSAJES = {
    L.ST_EMPTY: "Faltan argumentos para crear la trama",
    L.ST_SHORT_HEADER: "Trama muy corta",
    L.ST_SHORT_BODY: "La trama no contiene suficiente longitud según el campo 'length'",
}

# ------------------------------------------------------------------------------------------
# Device state: one modem per scheme, one current channel
# ------------------------------------------------------------------------------------------
_modems: dict = {}
_hilo = threading.local()      # the current channel is per thread: one endpoint per thread


_modems_lock = threading.Lock()


def modem(esquema: str = "B") -> M.Modem:
    with _modems_lock:
        if esquema not in _modems:
            _modems[esquema] = M.Modem(M.esquema_b() if esquema == "B" else M.esquema_a())
        return _modems[esquema]


def usar_canal(canal):
    """Install the synthetic audio device every play/rec of this module goes to, for the
    calling thread (a `CanalSintetico`, or an `ExtremoAire` of a shared `Aire`)."""
    previo = getattr(_hilo, "canal", None)
    _hilo.canal = canal
    return previo


def canal():
    if getattr(_hilo, "canal", None) is None:
        _hilo.canal = CanalSintetico(M.esquema_b())
    return _hilo.canal


def _dev(data: bytes, ancho: int) -> torch.Tensor:
    fila = np.zeros((1, ancho), dtype=np.uint8)
    fila[0, :len(data)] = np.frombuffer(bytes(data), dtype=np.uint8)
    return torch.from_numpy(fila).to(modem().device)


def _ancho(n: int) -> int:
    return max(16, (n + 15) & ~15)


# ------------------------------------------------------------------------------------------
# Frames (reference c1_shared.py:57-143)
# ------------------------------------------------------------------------------------------
class Trama:
    """[versión][MAC_org][MAC_dest][tipo][longitud][payload][checksum]"""

    def __init__(self, bytes_trama=None, version=None, mac_origen=None, mac_destino=None, tipo=None, payload=None):
        if bytes_trama:
            self.bytes_trama = bytearray(bytes_trama)
            self._parse()
        else:
            if None in (version, mac_origen, mac_destino, tipo, payload):
                raise ValueError(_MENSAJES[L.ST_EMPTY])
            self.version = version
            self.mac_origen = mac_origen
            self.mac_destino = mac_destino
            self.tipo = tipo
            self.payload = payload
            self.length = len(payload)
            self._build()

    def _parse(self):
        raw = bytes(self.bytes_trama)
        m = modem()
        rec = M.Modem.records_to_numpy(
            m.parse_frames(_dev(raw, _ancho(len(raw))), torch.tensor([len(raw)], dtype=torch.int32, device=m.device)))[0]
        if rec["status"] != L.ST_OK:
            raise ValueError(_MENSAJES[int(rec["status"])])
        self.version = self.bytes_trama[0]
        self.mac_origen = self.bytes_trama[1:7]
        self.mac_destino = self.bytes_trama[7:13]
        self.tipo = int(rec["tipo"])
        self.length = int(rec["length"])
        self.payload = self.bytes_trama[15:15 + self.length]
        self.checksum = self.bytes_trama[15 + self.length:15 + self.length + 4]
        self._crc_ok = bool(rec["crc_ok"])

    def _build(self):
        if not (0 <= self.version <= 255 and 0 <= self.tipo <= 255 and self.length <= 255):
            raise ValueError("byte must be in range(0, 256)")
        if len(self.mac_origen) != 6 or len(self.mac_destino) != 6:
            # the reference concatenates addresses of any length (c1_shared.py:109-111); such a
            # frame does not have the 15-byte header the frame kernel builds, so only its
            # checksum comes from the GPU
            cuerpo = bytearray([self.version]) + bytearray(self.mac_origen) + bytearray(self.mac_destino) + \
                bytearray([self.tipo, self.length]) + bytearray(self.payload)
            self.checksum = bytearray(calcular_checksum(cuerpo))
            self.bytes_trama = cuerpo + self.checksum
            return
        m = modem()
        campos = bytes([self.version]) + bytes(self.mac_origen) + bytes(self.mac_destino) + \
            bytes([self.tipo, self.length, 0])
        frames, lens = m.build_frames(_dev(campos, 16), _dev(bytes(self.payload), _ancho(self.length)), 288)
        n = int(lens[0])
        self.bytes_trama = bytearray(frames[0, :n].cpu().numpy().tobytes())
        self.checksum = self.bytes_trama[n - 4:n]

    def get_bytes(self):
        return bytes(self.bytes_trama)

    def get_checksum_valido(self):
        header_y_payload = self.bytes_trama[:15 + self.length]
        return self.checksum == calcular_checksum(header_y_payload)

    def imprimir(self):
        def mac_to_str(mac_bytes):
            return ':'.join(f'{b:02x}' for b in mac_bytes)

        logging.info("----- TRAMA -----")
        logging.info(f"Versión     : {self.version}")
        logging.info(f"MAC Origen  : {mac_to_str(self.mac_origen)}")
        logging.info(f"MAC Destino : {mac_to_str(self.mac_destino)}")
        logging.info(f"Tipo        : {self.tipo}")
        logging.info(f"Longitud    : {self.length}")
        logging.info(f"Payload     : {self.payload}")
        logging.info(f"Checksum    : {self.checksum.hex()}")
        logging.info(f"Checksum válido: {self.get_checksum_valido()}")
        logging.info("-----------------")


def byte_to_freq(byte):
    return BASE_FREQ + byte * FREQ_STEP


def freq_to_byte(freq):
    """Byte of a frequency, or None when it is not within half a step of a tone
    (reference c1_shared.py:152-163).  Scalar host arithmetic; the kernels apply the same
    rule per FFT bin through their class table."""
    candidato = int(round((freq - BASE_FREQ) / FREQ_STEP))
    cerca = abs(freq - (BASE_FREQ + candidato * FREQ_STEP)) < (FREQ_STEP / 2)
    return candidato if (0 <= candidato <= 255 and cerca) else None


def mac_str_to_bytes(mac_str):
    try:
        parts = mac_str.split(":")
        if len(parts) != 6:
            raise ValueError
        return bytes(int(part, 16) for part in parts)
    except Exception:
        raise argparse.ArgumentTypeError(f"MAC inválida: {mac_str}")


def calcular_checksum(trama):
    """CRC-32 of `trama`, 4 bytes big-endian (reference c1_shared.py:178-187)."""
    data = bytes(trama)
    m = modem()
    crc = m.crc32(_dev(data, _ancho(len(data))), torch.tensor([len(data)], dtype=torch.int32, device=m.device))
    return int(crc[0]).to_bytes(4, byteorder='big')


def crear_trama(version, bytes_mac_org, bytes_mac_dest, tipo, cur_payload):
    return Trama(version=version, mac_origen=bytes_mac_org, mac_destino=bytes_mac_dest, tipo=tipo,
                 payload=cur_payload)


def crear_tramas(version, bytes_mac_org, bytes_mac_dest, tipo, payloads):
    """`crear_trama` for many payloads with the same header fields, all frames built and
    checksummed by one kernel launch (a transfer then costs one launch for its frames instead
    of one per frame).  Not in the reference, which builds frame by frame (c1_shared.py:484-490)."""
    payloads = [bytes(p) for p in payloads]
    if not payloads:
        return []
    if (len(bytes_mac_org) != 6 or len(bytes_mac_dest) != 6 or not 0 <= version <= 255 or not 0 <= tipo <= 255
            or max(len(p) for p in payloads) > 255):
        return [crear_trama(version, bytes_mac_org, bytes_mac_dest, tipo, p) for p in payloads]   # the general path
    m = modem()
    n, ancho = len(payloads), _ancho(max(len(p) for p in payloads))
    campos = np.zeros((n, 16), dtype=np.uint8)
    campos[:, :13] = np.frombuffer(bytes([version]) + bytes(bytes_mac_org) + bytes(bytes_mac_dest), dtype=np.uint8)
    campos[:, 13] = tipo
    carga = np.zeros((n, ancho), dtype=np.uint8)
    for i, p in enumerate(payloads):
        campos[i, 14] = len(p)
        carga[i, :len(p)] = np.frombuffer(p, dtype=np.uint8)
    frames, lens = m.build_frames(torch.from_numpy(campos).to(m.device), torch.from_numpy(carga).to(m.device), 288)
    frames, lens = frames.cpu().numpy(), lens.cpu().numpy()
    tramas = []
    for i, p in enumerate(payloads):
        t = Trama.__new__(Trama)
        t.version, t.mac_origen, t.mac_destino, t.tipo, t.payload, t.length = version, bytes_mac_org, bytes_mac_dest, tipo, p, len(p)
        t.bytes_trama = bytearray(frames[i, :lens[i]].tobytes())
        t.checksum = t.bytes_trama[lens[i] - 4:lens[i]]
        tramas.append(t)
    return tramas


def validar_tramas(tramas):
    """`get_checksum_valido` of many frames by one kernel launch: a list of bools."""
    tramas = list(tramas)
    if not tramas:
        return []
    m = modem()
    # what get_checksum_valido compares: the stored checksum with the CRC-32 of the first 15 + length bytes
    filas = [bytes(t.bytes_trama[:15 + t.length]) for t in tramas]
    ancho = _ancho(max(len(f) for f in filas))
    datos = np.zeros((len(filas), ancho), dtype=np.uint8)
    for i, f in enumerate(filas):
        datos[i, :len(f)] = np.frombuffer(f, dtype=np.uint8)
    lens = torch.tensor([len(f) for f in filas], dtype=torch.int32, device=m.device)
    crc = m.crc32(torch.from_numpy(datos).to(m.device), lens).cpu().numpy()
    return [bytes(t.checksum) == int(c).to_bytes(4, byteorder='big') for t, c in zip(tramas, crc)]


def crear_trama_archivo_info(bytes_mac_org, bytes_mac_dest, cantidad_tramas, nombre_archivo):
    payload = cantidad_tramas.to_bytes(4, byteorder='big') + nombre_archivo.encode('utf-8')
    return crear_trama(VERSION, bytes_mac_org, bytes_mac_dest, TIPO_TRAMA_ARCHIVO_INFO, payload)


def decodificar_payload_archivo_info(payload: bytes):
    return int.from_bytes(payload[:4], byteorder='big'), bytes(payload[4:]).decode('utf-8')


def verificar_datos_esperados(trama: Trama, tipo_esperado, org_esperado, dst_esperado):
    return bool(trama.tipo == tipo_esperado and trama.mac_origen == org_esperado
                and trama.mac_destino == dst_esperado)


def crear_trama_ok(bytes_mac_org, bytes_mac_dest, bytes_checksum_received):
    return crear_trama(VERSION, bytes_mac_org, bytes_mac_dest, TIPO_OK, bytes_checksum_received)


# ------------------------------------------------------------------------------------------
# Sound (reference c1_shared.py:286-434)
# ------------------------------------------------------------------------------------------
def detectar_frecuencia(audio, sample_rate):
    """Dominant frequency of one receive window (reference c1_shared.py:286-299).  The
    window must have the length the kernels are built for (7055 or 2205 samples)."""
    x = torch.as_tensor(np.asarray(audio) if not torch.is_tensor(audio) else audio).reshape(-1)
    n = x.numel()
    esquema = {7055: "B", 2205: "A"}.get(n)
    if esquema is None or sample_rate != SAMPLE_RATE:
        raise ValueError(f"detectar_frecuencia: ventana de {n} muestras a {sample_rate} Hz no soportada")
    m = modem(esquema)
    buf = torch.zeros(M.pad4(n), dtype=torch.float32, device=m.device)
    buf[:n] = x.to(m.device, torch.float32)
    _, _, kbin = m.detect(buf, 1, buf.numel(), 1)
    return abs(int(kbin[0, 0]) * (1.0 / (n * (1 / sample_rate))))     # fftfreq(n, 1/sr)[k]


def _ventanas_disponibles(c: CanalSintetico, cuantas: int):
    n = c.esquema.win_n
    c.reservar(c.leido + cuantas * n + 4)
    return c.audio, c.leido, cuantas


def _escuchar_trama_por_tramos(c, m, tw):
    """The listen loop on a shared air: windows are taken as they become certain, the state
    machine is re-run over everything heard so far with one extra empty window appended, and
    the call is over once the machine stops without touching that extra window."""
    n = c.esquema.win_n
    base, k = c.leido, 0
    todas = torch.empty((1, 0), dtype=torch.int16, device=m.device)
    vacia = torch.full((1, 1), L.CLS_NONE, dtype=torch.int16, device=m.device)
    while True:
        nuevas = c.esperar_ventanas(base + k * n, 2 * tw - k)
        audio = c.audio
        cls, _, _ = m.detect(audio, 1, audio.numel(), nuevas, base + k * n)
        k += nuevas
        todas = torch.cat([todas, cls], dim=1)
        by, lens, _, end, reason, _ = m.slice(torch.cat([todas, vacia], dim=1).contiguous(), tw, 1, 288)
        usados = int(end[0, 0])
        if usados <= k or k >= 2 * tw:
            return by, lens, min(usados, k), todas, int(reason[0, 0])
        c.confirmar(base + k * n)


def _registrar_ventanas(clases, usados, razon):
    """The lines the reference's receive loop logs window by window (c1_shared.py:310, 320,
    333, 338, 344-351), replayed from the class words of the windows one listen call consumed."""
    logging.info("Escuchando... (esperando datos hasta EOF 7000 Hz)")
    oyo_ping, espera_silencio = False, True
    for w in np.asarray(clases[:usados]).astype(np.uint16).tolist():
        if not oyo_ping:
            if w & L.CLS_PING:
                oyo_ping = True
                logging.info("Escuche el ping")
            continue
        if w & L.CLS_EOF:
            logging.info("EOF detectado.")
            return
        byte = w & L.CLS_BYTE_MASK
        if byte != L.CLS_NONE:
            if espera_silencio:
                logging.info("?------------?")
                continue
            logging.info(f"BYTE <- {byte}")
            espera_silencio = True
        else:
            espera_silencio = False
            logging.info("--------------")
    if (razon & 0xFF) == L.END_TIMEOUT:
        logging.info("Timeout alcanzado. Terminando escucha.")


def escuchar_y_retornar_trama(timeout):
    """Listens for one frame for `timeout` seconds of channel time; the clock restarts at
    the ping (reference c1_shared.py:301-355).  Returns the Trama or raises ValueError."""
    c, m = canal(), modem()
    tw = m.timeout_windows(timeout)
    if c.incremental:
        by, lens, usados, cls, razon = _escuchar_trama_por_tramos(c, m, tw)
    else:
        audio, desde, nwin = _ventanas_disponibles(c, 2 * tw)
        cls, _, _ = m.detect(audio, 1, audio.numel(), nwin, desde)
        by, lens, _, end, reason, _ = m.slice(cls, tw, 1, 288)
        usados, razon = int(end[0, 0]), int(reason[0, 0])
    c.leido += usados * c.esquema.win_n
    if logging.getLogger().isEnabledFor(logging.INFO):    # the class words leave the GPU only when somebody reads the log
        _registrar_ventanas(cls[0].cpu().numpy(), usados, razon)
    recibidos = by[0, 0, :int(lens[0, 0])].cpu().numpy().tolist()
    logging.info(f"{recibidos}")
    return Trama(bytes_trama=recibidos)


def _pings(m, audio, desde, nwin):
    """Per window: does a ping tone dominate it?"""
    _, _, verdict = m.sync_scan(audio, 1, audio.numel(), nwin, desde)
    v = verdict[0].cpu().numpy()
    if (v == 2).any():       # undecided windows: only the spectrum can tell
        cls, _, _ = m.detect(audio, 1, audio.numel(), nwin, desde)
        return (cls[0].cpu().numpy().astype(np.uint16) & L.CLS_PING) != 0
    return v == 1


def escuchar_ping(timeout):
    """True if a ping tone dominates some window within `timeout` seconds
    (reference c1_shared.py:357-384)."""
    c, m = canal(), modem()
    tw = m.timeout_windows(timeout)
    n = c.esquema.win_n
    base, k = c.leido, 0
    while k < tw:
        if c.incremental:
            nuevas = c.esperar_ventanas(base + k * n, tw - k)
            audio = c.audio
        else:
            audio, _, nuevas = _ventanas_disponibles(c, tw)
        idx = np.flatnonzero(_pings(m, audio, base + k * n, nuevas))
        if len(idx):
            c.leido = base + (k + int(idx[0]) + 1) * n
            logging.info("Escuche el ping")
            return True
        k += nuevas
        if c.incremental and k < tw:
            c.confirmar(base + k * n)
    c.leido = base + tw * n
    logging.info("Timeout alcanzado. Terminando escucha.")
    return False


def _tono(freq, n):
    """Plays n samples of a `freq` Hz tone (one-segment schedule)."""
    c, m = canal(), modem()
    dev = m.device
    lay = (torch.tensor([[0, n]], dtype=torch.int32, device=dev), torch.tensor([[freq]], dtype=torch.int16, device=dev),
           torch.tensor([1], dtype=torch.int32, device=dev), None)
    c.emitir(m, lay, n)


def transmite_freq(freq, duration=BYTE_DURATION):
    _tono(int(freq), int(SAMPLE_RATE * duration))


def transmitir_silencio(freq=SILENCE_FREQ, duration=SILENCE_DURATION):
    _tono(int(freq), int(SAMPLE_RATE * duration))


def ejecutar_ping():
    canal().sleep(2)
    transmite_freq(PING_FREQ, duration=BYTE_DURATION * 2)


def emitir_trama(trama):
    """Puts one frame on the channel: ping, separator + tone per byte, separator, EOF
    (reference c1_shared.py:415-434)."""
    c, m = canal(), modem()
    raw = bytes(trama.bytes_trama)
    frames = _dev(raw, _ancho(len(raw)))
    lens = torch.tensor([len(raw)], dtype=torch.int32, device=m.device)
    lay = m.layout(frames, lens, L.EMIT_LIVE)
    c.emitir(m, lay, M.host_frame_samples(m.esquema, raw, L.EMIT_LIVE))
    logging.info("Trama transmitida.")


# ------------------------------------------------------------------------------------------
# Files (reference c1_shared.py:440-525)
# ------------------------------------------------------------------------------------------
def trama_a_int16(trama) -> np.ndarray:
    """The samples guardar_trama_como_wav writes (reference c1_shared.py:445-467)."""
    m = modem()
    raw = bytes(trama.bytes_trama)
    n = M.host_frame_samples(m.esquema, raw, L.EMIT_WAV)
    audio, _ = m.encode_streams(_dev(raw, _ancho(len(raw))), torch.tensor([len(raw)], dtype=torch.int32, device=m.device),
                                M.pad4(n), variant=L.EMIT_WAV, dtype=torch.int16)
    return audio[0, :n].cpu().numpy()


def guardar_trama_como_wav(trama, nombre_archivo):
    from scipy.io.wavfile import write
    write(nombre_archivo, SAMPLE_RATE, trama_a_int16(trama))
    trama.imprimir()
    logging.info(f"Trama guardada como archivo WAV: {nombre_archivo}")


def leer_wav(nombre_archivo) -> np.ndarray:
    """The samples of a WAV file as a capture for the listener: int16 scaled by 1/32768,
    floating-point files as they are, first channel only.  The reference only writes WAV
    files (c1_shared.py:440-471); this is the way back in."""
    from scipy.io.wavfile import read
    rate, data = read(nombre_archivo)
    if rate != SAMPLE_RATE:
        raise ValueError(f"{nombre_archivo}: {rate} Hz, se esperaban {SAMPLE_RATE}")
    data = np.asarray(data)
    if data.ndim > 1:
        data = data[:, 0]
    if np.issubdtype(data.dtype, np.integer):
        return data.astype(np.float32) / float(1 << (8 * data.dtype.itemsize - 1))
    return data.astype(np.float32)


def cargar_wavs_en_canal(nombres, silencio=0.5):
    """Makes the listed WAV files, `silencio` seconds of silence before each and after the
    last, the capture of the current channel."""
    hueco = np.zeros(int(silencio * SAMPLE_RATE), dtype=np.float32)
    partes = []
    for nombre in nombres:
        partes += [hueco, leer_wav(nombre)]
    canal().cargar(np.concatenate(partes + [hueco]))


def _tramas_desde_bytes(data: bytes, mac_origen_str, mac_destino_str):
    org, dst = mac_str_to_bytes(mac_origen_str), mac_str_to_bytes(mac_destino_str)
    return crear_tramas(VERSION, org, dst, TIPO_TRAMA_ARCHIVO, [data[i:i + 107] for i in range(0, len(data), 107)])


def obtener_tramas_desde_archivo(file_path, mac_origen_str, mac_destino_str):
    with open(file_path, 'rb') as f:
        return _tramas_desde_bytes(f.read(), mac_origen_str, mac_destino_str)


def obtener_tramas_desde_string(string, mac_origen_str, mac_destino_str):
    if isinstance(string, str):
        string = string.encode()
    return _tramas_desde_bytes(string, mac_origen_str, mac_destino_str)


def guardar_archivo_en_tramas_wav(nombre_archivo, mac_org_str, mac_dest_str):
    for i, trama in enumerate(obtener_tramas_desde_archivo(nombre_archivo, mac_org_str, mac_dest_str)):
        guardar_trama_como_wav(trama, f"{nombre_archivo}_trama_{i+1}.wav")

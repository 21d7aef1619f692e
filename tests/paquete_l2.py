"""Test helper: the text form of a layer-2 packet and its reconstruction, restated from reading
WaveNetCore/wavenetcore/WaveNetPacketeering.py (Packet.form :36-49, SecretPacket.form :114-124,
verify_tag :152-164, reconstruct_packet :166-193).  Parity unpinned: the reference's WaveNetCore
could not be run in the build container and it holds no fixture for these functions; what is
checked is the behaviour read from those lines.  Layer 2 is not part of the product: this exists
so that a test can show a packet's text crossing the sound link and still reconstructing."""
import json
from collections import namedtuple

Paquete = namedtuple("Paquete", "src dest mtype body timestamp")
PaqueteSecreto = namedtuple("PaqueteSecreto", "meta body")

CAMPOS = (("src", int), ("dest", int), ("mtype", str), ("body", str), ("timestamp", str))
CAMPOS_SECRETO = (("meta", str), ("body", str))


def forma(p: Paquete) -> str:
    """Packet.form (:42-49): key order enc, src, dest, mtype, body, timestamp."""
    d = {"enc": False}
    d.update(p._asdict())
    return json.dumps(d)


def nulo(mensaje) -> Paquete:
    """Packet.null (:51-58) with a fixed timestamp (the reference stamps the current time)."""
    return Paquete(-1, -1, "error", mensaje, "")


def es_nulo(p) -> bool:
    return isinstance(p, Paquete) and p.mtype == "error"


def _etiqueta(parsed, nombre, tipo):
    if nombre not in parsed:
        return False, f"Missing {nombre} tag"
    v = parsed[nombre]
    if type(v) != tipo:                     # exact type: a bool is not accepted as an int, as in :163
        return False, f"Bad {nombre} tag"
    return True, v


def reconstruir(data):
    """reconstruct_packet (:166-193): a Paquete, a PaqueteSecreto, or the null packet whose body
    names the first missing or mistyped tag, or 'Formation Error ...' for text that is not JSON."""
    try:
        parsed = json.loads(data)
        ok, enc = _etiqueta(parsed, "enc", bool)
        if not ok:
            return nulo(enc)
        valores = []
        for nombre, tipo in (CAMPOS_SECRETO if enc else CAMPOS):
            ok, v = _etiqueta(parsed, nombre, tipo)
            if not ok:
                return nulo(v)
            valores.append(v)
        return PaqueteSecreto(*valores) if enc else Paquete(*valores)
    except Exception as e:      # noqa: BLE001  (the reference catches everything here)
        return nulo("Formation Error " + str(e))

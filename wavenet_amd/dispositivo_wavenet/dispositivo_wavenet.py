"""`DispositivoWaveNet`: the layer-1 device the upper layers talk to (reference
DispositivoWaveNET/dispositivo_wavenet/dispositivo_wavenet.py:17-52; its only in-repo
caller is SoundProtocol, WaveNetCore/wavenetcore/WaveNetProtocols.py:295-329), and the
command line of the same file (:54-98).

Same constructor, same two methods, same two exception messages, same options and modes.
The sound card is the synthetic channel installed with `c1_shared.usar_canal`; with no peer
on it both methods fail the way the reference does on a machine nobody answers.
"""
import argparse
import os

from . import c1_communication as _l1
from .c1_communication import *  # noqa: F401,F403  (the reference module re-exports layer 1 the same way)


class DispositivoWaveNet:
    """One machine's layer-1 endpoint: its own MAC and, for sending, its peer's."""

    #: texts the upper layers see when a transfer fails (dispositivo_wavenet.py:40, 51)
    ERROR_ENVIO = "Failed to send string through layer 1"
    ERROR_ESCUCHA = "Failed to get string from layer 1"

    def __init__(self, mac_origen, mac_destino=None):
        self.mac_origen, self.mac_destino = mac_origen, mac_destino

    def send(self, string, timeout=None):
        entregado = _l1.enviar_string_por_sonido(string, self.mac_origen, self.mac_destino, timeout=timeout)
        if not entregado:
            raise Exception(self.ERROR_ENVIO)

    def listen(self, timeout=None, init_timeout=None):
        texto = _l1.escuchar_string(self.mac_origen, timeout=timeout, init_timeout=init_timeout)
        if texto == False:  # noqa: E712  (the reference's own test: False fails, an empty text does not)
            raise Exception(self.ERROR_ESCUCHA)
        return texto


def _modo_wav(a):
    _l1.guardar_archivo_en_tramas_wav(a.archivo, a.mac_origen, a.mac_destino)


def _modo_oir_trama(a):
    try:
        _l1.escuchar_y_retornar_trama(_l1.TIME_TO_SAY_128_BYTES).imprimir()
    except Exception:
        print("No se escuho ninguna trama en el tiempo establecido.")


def _avisar(exito, texto):
    if not exito:
        print(texto)


# mode -> (needs the file to exist, action); the texts are the reference's (dispositivo_wavenet.py:66-92)
_MODOS = {
    "1": (True, _modo_wav),
    "2": (False, _modo_oir_trama),
    "3": (True, lambda a: _avisar(_l1.enviar_archivo_por_sonido(a.archivo, a.mac_origen, a.mac_destino),
                                  "No se envio el archivo correctamente")),
    "4": (False, lambda a: _avisar(_l1.escuchar_archivo(a.mac_origen), "No se escucho el archivo correctamente")),
    "5": (False, lambda a: _avisar(_l1.enviar_string_por_sonido(a.archivo, a.mac_origen, a.mac_destino),
                                   "No se envio el archivo correctamente")),
    "6": (False, lambda a: _avisar(_l1.escuchar_string(a.mac_origen), "No se escucho el archivo correctamente")),
}


def _argumentos(argv):
    p = argparse.ArgumentParser(description="Capa 1 de WaveNet sobre el canal sintetico")
    for corta, larga, obligatoria, ayuda in (
            ("-a", "--archivo", True, "archivo a enviar, o texto en los modos 5 y 6"),
            ("-b", "--mac_origen", True, "MAC de esta maquina, aa:bb:cc:dd:ee:ff"),
            ("-c", "--mac_destino", False, "MAC de la otra maquina"),
            ("-d", "--modo", True, "1 WAV por trama, 2 oir una trama, 3/4 enviar/oir archivo, 5/6 enviar/oir texto")):
        p.add_argument(corta, larga, required=obligatoria, help=ayuda)
    # no counterpart in the reference, whose capture is the sound card
    p.add_argument("-w", "--captura", action="append", default=[],
                   help="WAV que se carga en el canal antes de escuchar (repetible)")
    return p.parse_args(argv)


def main(argv=None):
    """The listening modes hear the current synthetic channel; `-w` loads WAV files into it first."""
    a = _argumentos(argv)
    necesita_archivo, accion = _MODOS.get(a.modo, (False, None))
    if necesita_archivo and not os.path.exists(a.archivo):
        print(f"El archivo {a.archivo} no existe.")
        return
    if a.captura:
        _l1.cargar_wavs_en_canal(a.captura)
    if accion is not None:          # an unknown mode does nothing, as in the reference
        accion(a)


if __name__ == "__main__":
    main()

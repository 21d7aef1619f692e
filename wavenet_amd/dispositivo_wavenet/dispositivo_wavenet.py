"""`DispositivoWaveNet`: the layer-1 device the upper layers talk to (reference
DispositivoWaveNET/dispositivo_wavenet/dispositivo_wavenet.py:17-52; its only in-repo
caller is SoundProtocol, WaveNetCore/wavenetcore/WaveNetProtocols.py:295-329).

Same constructor, same two methods, same two exception messages.  The sound card is the
synthetic channel installed with `c1_shared.usar_canal`; with no peer on it both methods
fail the way the reference does on a machine nobody answers.
"""
import argparse
import os

from .c1_communication import *  # noqa: F401,F403
from .c1_communication import (TIME_TO_SAY_128_BYTES, cargar_wavs_en_canal, enviar_archivo_por_sonido,
                               enviar_string_por_sonido, escuchar_archivo, escuchar_string,
                               escuchar_y_retornar_trama, guardar_archivo_en_tramas_wav)


class DispositivoWaveNet:
    def __init__(self, mac_origen, mac_destino=None):
        self.mac_origen = mac_origen
        self.mac_destino = mac_destino

    def send(self, string, timeout=None):
        if not enviar_string_por_sonido(string, self.mac_origen, self.mac_destino, timeout=timeout):
            raise Exception("Failed to send string through layer 1")

    def listen(self, timeout=None, init_timeout=None):
        ret = escuchar_string(self.mac_origen, timeout=timeout, init_timeout=init_timeout)
        if ret == False:  # noqa: E712  (the reference tests equality, so an empty string fails too)
            raise Exception("Failed to get string from layer 1")
        return ret


def main(argv=None):
    """The reference's command line (dispositivo_wavenet.py:54-98): same options and modes.
    The listening modes hear the current synthetic channel; `-w` (no counterpart in the
    reference, whose capture is the sound card) loads WAV files into it first."""
    parser = argparse.ArgumentParser(description="Parser de archivo y direcciones MAC")
    parser.add_argument('-a', '--archivo', required=True, help="Ruta al archivo (ej: archivo.txt)")
    parser.add_argument('-b', '--mac_origen', required=True, help="MAC Address de origen (ej: aa:bb:cc:dd:ee:ff)")
    parser.add_argument('-c', '--mac_destino', required=False, help="MAC Address de destino (ej: 11:22:33:44:55:66)")
    parser.add_argument('-d', '--modo', required=True)
    parser.add_argument('-w', '--captura', action='append', default=[],
                        help="archivo WAV que se carga como captura antes de escuchar (repetible)")
    args = parser.parse_args(argv)
    ruta = args.archivo

    if (args.modo == "1" or args.modo == "3") and not os.path.exists(ruta):
        print(f"El archivo {ruta} no existe.")
        return
    if args.captura:
        cargar_wavs_en_canal(args.captura)

    if args.modo == "1":
        guardar_archivo_en_tramas_wav(ruta, args.mac_origen, args.mac_destino)
    elif args.modo == "2":
        try:
            tramita = escuchar_y_retornar_trama(TIME_TO_SAY_128_BYTES)
            tramita.imprimir()
        except Exception:
            print("No se escuho ninguna trama en el tiempo establecido.")
    elif args.modo == "3":
        if not enviar_archivo_por_sonido(ruta, args.mac_origen, args.mac_destino):
            print("No se envio el archivo correctamente")
    elif args.modo == "4":
        if not escuchar_archivo(args.mac_origen):
            print("No se escucho el archivo correctamente")
    elif args.modo == "5":
        if not enviar_string_por_sonido(ruta, args.mac_origen, args.mac_destino):
            print("No se envio el archivo correctamente")
    elif args.modo == "6":
        if not escuchar_string(args.mac_origen):
            print("No se escucho el archivo correctamente")


if __name__ == '__main__':
    main()

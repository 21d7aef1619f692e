"""`DispositivoWaveNet`: the layer-1 device the upper layers talk to (reference
DispositivoWaveNET/dispositivo_wavenet/dispositivo_wavenet.py:17-52; its only in-repo
caller is SoundProtocol, WaveNetCore/wavenetcore/WaveNetProtocols.py:295-329).

Same constructor, same two methods, same two exception messages.  The sound card is the
synthetic channel installed with `c1_shared.usar_canal`; with no peer on it both methods
fail the way the reference does on a machine nobody answers.
"""
from .c1_communication import *  # noqa: F401,F403
from .c1_communication import enviar_string_por_sonido, escuchar_string


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

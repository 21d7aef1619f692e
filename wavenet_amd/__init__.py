"""wavenet_amd: the layer-1 acoustic modem of yosgarcia/WaveNet as HIP kernels for MI355X.

`wavenet_amd.modem.Modem` is the batched device API; `wavenet_amd.dispositivo_wavenet`
keeps the reference's call surface (Trama, emitir_trama, escuchar_y_retornar_trama,
DispositivoWaveNet ...) over a synthetic in-memory audio channel.
"""
from . import _lib  # noqa: F401  (fails loudly when libwavenet_hip.so is missing)

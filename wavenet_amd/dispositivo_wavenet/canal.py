"""Synthetic audio device: what `sounddevice.play/rec/wait` and the wall clock are to the
reference (c1_shared.py:324-325, 394-395, 314-319), as a float32 buffer in HBM.

Two modes:
  * capture (default): play appends samples at the write cursor, rec reads sequentially
    from the start, sleep costs nothing and the clock is samples recorded / sample rate.
    This is the loopback the golden-vector generator gives the reference: emit, then
    listen to what was emitted.
  * half_duplex=True: one endpoint's own sound card.  Playing and sleeping take channel
    time during which nothing is recorded, so both advance the read cursor; with no peer
    the device only ever records silence, as a lone machine does.
"""
from __future__ import annotations

import torch

from .. import modem as M


class CanalSintetico:
    """One capture timeline on one GPU."""

    def __init__(self, esquema: M.Esquema | None = None, capacidad: int = 1 << 22, device=None,
                 half_duplex: bool = False):
        self.esquema = esquema or M.esquema_b()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.audio = torch.zeros(M.pad4(capacidad), dtype=torch.float32, device=self.device)
        self.escrito = 0      # samples written (write cursor)
        self.leido = 0        # samples recorded so far (read cursor)
        self.half_duplex = half_duplex

    # -- clock --------------------------------------------------------------------------
    def time(self) -> float:
        return self.leido / self.esquema.sample_rate

    def sleep(self, segundos: float) -> None:
        if self.half_duplex:
            self.leido += int(segundos * self.esquema.sample_rate)
            self.reservar(self.leido)

    # -- storage ------------------------------------------------------------------------
    def reservar(self, hasta: int) -> None:
        """Make samples [0, hasta) addressable (zeros where nothing was played)."""
        if hasta <= self.audio.numel():
            return
        nuevo = torch.zeros(M.pad4(max(hasta, 2 * self.audio.numel())), dtype=torch.float32, device=self.device)
        nuevo[:self.audio.numel()] = self.audio
        self.audio = nuevo

    def play(self, muestras: torch.Tensor) -> None:
        """sd.play + sd.wait: append float32 samples."""
        n = muestras.numel()
        pos = self.posicion_escritura()
        self.reservar(pos + n)
        self.audio[pos:pos + n] = muestras.to(self.device, torch.float32).reshape(-1)
        self.avanzar_escritura(n)

    def posicion_escritura(self) -> int:
        """Where the next emitted sample goes."""
        return max(self.escrito, self.leido) if self.half_duplex else self.escrito

    def avanzar_escritura(self, n: int) -> None:
        """n samples were written at posicion_escritura()."""
        self.escrito = self.posicion_escritura() + n
        if self.half_duplex:
            self.leido = self.escrito

    def silencio(self, n: int) -> None:
        """Nothing on air for n samples."""
        self.reservar(self.posicion_escritura() + n)
        self.avanzar_escritura(n)

    def cargar(self, muestras) -> None:
        """Replace the timeline by a recorded capture (numpy or torch, any float type)."""
        t = torch.as_tensor(muestras).to(torch.float32).reshape(-1)
        self.audio = torch.zeros(M.pad4(t.numel()), dtype=torch.float32, device=self.device)
        self.audio[:t.numel()] = t.to(self.device)
        self.escrito, self.leido = t.numel(), 0

    def rec(self, n: int) -> torch.Tensor:
        """sd.rec + sd.wait: the next n samples (silence past the written audio)."""
        self.reservar(self.leido + n)
        out = self.audio[self.leido:self.leido + n]
        self.leido += n
        return out

    def agotado(self) -> bool:
        return self.leido >= self.escrito

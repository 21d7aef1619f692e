"""Synthetic audio device: what `sounddevice.play/rec/wait` and the wall clock are to the
reference (c1_shared.py:324-325, 394-395, 314-319), as a float32 buffer in HBM.

Three devices:
  * `Aire` + `ExtremoAire`: several endpoints (one thread each) on one shared timeline, for
    the two-machine transfers of c1_communication.py; see the classes below.
  * capture (default `CanalSintetico`): play appends samples at the write cursor, rec reads sequentially
    from the start, sleep costs nothing and the clock is samples recorded / sample rate.
    This is the loopback the golden-vector generator gives the reference: emit, then
    listen to what was emitted.
  * half_duplex=True: one endpoint's own sound card.  Playing and sleeping take channel
    time during which nothing is recorded, so both advance the read cursor; with no peer
    the device only ever records silence, as a lone machine does.
"""
from __future__ import annotations

import math
import threading

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

    # -- what the layer-1 functions call ------------------------------------------------
    incremental = False   # listeners may look at all the windows they could need at once

    def emitir(self, m: M.Modem, layout, n: int) -> None:
        """Synthesise the n samples of a tone schedule at the write position."""
        pos = self.posicion_escritura()
        self.reservar(pos + n)
        m.synth(layout, torch.tensor([pos], dtype=torch.int64, device=self.device), self.audio, M.pad4(n))
        self.avanzar_escritura(n)


class Aire:
    """The air between several endpoints: one float32 timeline in HBM that every endpoint's
    playback is added to (`ruido`, if given, is what the timeline holds before anyone plays).

    Each endpoint runs in its own thread with its own clock, the way two machines run the
    reference's c1_communication.py.  Co-simulation is conservative: an endpoint may look at
    the span [a, b) only when no other endpoint can still play into it, i.e. when every other
    endpoint's horizon (its clock, or the time until which it is certainly still listening,
    or infinity once it has finished) is at least b.  Results therefore do not depend on how
    the threads are scheduled."""

    def __init__(self, esquema: M.Esquema | None = None, capacidad: int = 1 << 24, device=None, ruido=None):
        self.esquema = esquema or M.esquema_b()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.cond = threading.Condition()
        self.extremos: list[ExtremoAire] = []
        if ruido is None:
            self.audio = torch.zeros(M.pad4(capacidad), dtype=torch.float32, device=self.device)
        else:
            r = torch.as_tensor(ruido).to(torch.float32).reshape(-1)
            self.audio = torch.zeros(M.pad4(max(capacidad, r.numel())), dtype=torch.float32, device=self.device)
            self.audio[:r.numel()] = r.to(self.device)

    def extremo(self, nombre: str = "") -> "ExtremoAire":
        e = ExtremoAire(self, nombre)
        with self.cond:
            self.extremos.append(e)
        return e

    def reservar(self, hasta: int) -> None:
        """Caller holds `cond`."""
        if hasta <= self.audio.numel():
            return
        nuevo = torch.zeros(M.pad4(max(hasta, 2 * self.audio.numel())), dtype=torch.float32, device=self.device)
        nuevo[:self.audio.numel()] = self.audio
        self.audio = nuevo


class ExtremoAire:
    """One endpoint's sound card on a shared `Aire` (half duplex: it hears nothing while it
    plays or sleeps).  `leido` is its clock in samples."""

    incremental = True    # listeners take the windows as they become certain

    def __init__(self, aire: Aire, nombre: str = ""):
        self.aire, self.nombre = aire, nombre
        self.esquema, self.device = aire.esquema, aire.device
        self._t = 0
        self._hasta = None      # while listening: a time before which this endpoint certainly plays nothing
        self.terminado = False
        self.emitido = 0        # samples played

    # -- clock --------------------------------------------------------------------------
    @property
    def leido(self) -> int:
        return self._t

    @leido.setter
    def leido(self, t: int) -> None:
        with self.aire.cond:
            self._t = int(t)
            self._hasta = None
            self.aire.cond.notify_all()

    def time(self) -> float:
        return self._t / self.esquema.sample_rate

    def sleep(self, segundos: float) -> None:
        self.leido = self._t + int(segundos * self.esquema.sample_rate)

    def horizonte(self) -> float:
        if self.terminado:
            return math.inf
        return self._t if self._hasta is None else max(self._hasta, self._t)

    def terminar(self) -> None:
        with self.aire.cond:
            self.terminado = True
            self.aire.cond.notify_all()

    # -- storage ------------------------------------------------------------------------
    @property
    def audio(self) -> torch.Tensor:
        return self.aire.audio

    def reservar(self, hasta: int) -> None:
        with self.aire.cond:
            self.aire.reservar(hasta)

    def posicion_escritura(self) -> int:
        return self._t

    def emitir(self, m: M.Modem, layout, n: int) -> None:
        """Play: the schedule's n samples are added to the air at this endpoint's clock."""
        tmp = torch.zeros(M.pad4(n), dtype=torch.float32, device=self.device)
        m.synth(layout, torch.zeros(1, dtype=torch.int64, device=self.device), tmp, M.pad4(n))
        with self.aire.cond:
            self.aire.reservar(self._t + n)
            self.aire.audio[self._t:self._t + n] += tmp[:n]
            self._t += n
            self._hasta = None
            self.emitido += n
            self.aire.cond.notify_all()

    def esperar_ventanas(self, desde: int, max_ventanas: int) -> int:
        """Blocks until at least one receive window starting at sample `desde` is certain,
        and returns how many consecutive ones are (at most `max_ventanas`).  While it waits
        the endpoint is known to keep listening through that first window."""
        n = self.esquema.win_n
        with self.aire.cond:
            self._hasta = desde + n
            self.aire.cond.notify_all()
            while True:
                h = min((o.horizonte() for o in self.aire.extremos if o is not self), default=math.inf)
                if h >= desde + n:
                    break
                self.aire.cond.wait()
            k = max_ventanas if h == math.inf else min(max_ventanas, int((h - desde) // n))
            self.aire.reservar(desde + k * n + 4)
            return k

    def confirmar(self, hasta: int) -> None:
        """The listener has looked at everything before `hasta` and is still listening."""
        with self.aire.cond:
            self._hasta = hasta
            self.aire.cond.notify_all()

#!/usr/bin/env python3
"""Generate tests/golden/duplex_golden.json: the REFERENCE's two-machine transfers
(c1_communication.py: enviar_string_por_sonido / escuchar_string with ping acknowledgements,
enviar_archivo_por_sonido / escuchar_archivo with OK-frame acknowledgements), run here as two
threads over one simulated air.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_duplex.py

The reference modules are imported unchanged.  `sounddevice` and the `time` module they see
are replaced by a shared-air stand-in: every endpoint (thread) has its own clock in samples;
play() adds its samples to the common timeline at the endpoint's clock and advances it,
sleep() advances it, rec(n) returns the n samples of the timeline at the clock (plus seeded
noise) and advances it.  rec blocks until no other endpoint can still play into the span it
reads (conservative co-simulation: an endpoint that is itself blocked in rec until time h
cannot play before h), so the outcome does not depend on thread scheduling.

Stored per scenario: both return values, both final clocks, samples played by each side and
the outcome of every listen call (windows consumed, bytes heard or error text, ping result).

NOT TO BE RUN AGAIN.  The fixture was generated once; a later attempt to import the reference
in this environment was refused (it may be read, not executed), and that refusal stands.  The
script is kept as the record of how the committed vectors were made.
"""
import json
import os
import sys
import tempfile
import threading
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/DispositivoWaveNET"
SR = 44100
WIN = 7055


class Air:
    def __init__(self, sigma=0.0, seed=0, seconds=0):
        self.buf = np.zeros(1 << 20)
        self.cond = threading.Condition()
        self.eps = {}
        self.noise = sigma * np.random.default_rng(seed).standard_normal(int(seconds * SR)) if sigma > 0 else None

    def ensure(self, n):
        if n > len(self.buf):
            nb = np.zeros(max(n, 2 * len(self.buf)))
            nb[:len(self.buf)] = self.buf
            self.buf = nb


class Endpoint:
    def __init__(self, air, name):
        self.air, self.name = air, name
        self.t = 0            # clock, samples
        self.blocked = None   # while in rec: the clock value it will have afterwards
        self.done = False
        self.played = 0
        self.calls = []

    def horizon(self):
        if self.done:
            return float("inf")
        return self.blocked if self.blocked is not None else self.t


AIR = None


def me() -> Endpoint:
    return AIR.eps[threading.get_ident()]


def _play(a, samplerate=None):
    ep = me()
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    with AIR.cond:
        AIR.ensure(ep.t + len(a))
        AIR.buf[ep.t:ep.t + len(a)] += a
        ep.t += len(a)
        ep.played += len(a)
        AIR.cond.notify_all()


def _rec(n, samplerate=None, channels=1, dtype="float64"):
    ep = me()
    with AIR.cond:
        ep.blocked = ep.t + n
        AIR.cond.notify_all()
        while min([o.horizon() for o in AIR.eps.values() if o is not ep], default=float("inf")) < ep.t + n:
            AIR.cond.wait()
        AIR.ensure(ep.t + n)
        out = AIR.buf[ep.t:ep.t + n].copy()
        if AIR.noise is not None:
            out += AIR.noise[ep.t:ep.t + n]
        ep.t += n
        ep.blocked = None
        AIR.cond.notify_all()
    return out.reshape(n, 1)


def _sleep(s):
    ep = me()
    with AIR.cond:
        ep.t += int(s * SR)
        AIR.cond.notify_all()


sd = types.ModuleType("sounddevice")
sd.play, sd.rec, sd.wait = _play, _rec, (lambda: None)
sys.modules["sounddevice"] = sd
sys.path.insert(0, REF)

import dispositivo_wavenet.c1_shared as cs            # noqa: E402  (the reference)
import dispositivo_wavenet.c1_communication as cc     # noqa: E402  (the reference)

fake_time = types.SimpleNamespace(time=lambda: me().t / SR, sleep=_sleep)
cs.time = fake_time
cc.time = fake_time

# log every listen call where c1_communication looks the functions up
_listen, _ping = cc.escuchar_y_retornar_trama, cc.escuchar_ping


def logged_listen(timeout):
    ep, t0 = me(), me().t
    try:
        tr = _listen(timeout)
        ep.calls.append({"kind": "frame", "windows": (ep.t - t0) // WIN, "bytes": bytes(tr.bytes_trama).hex(), "error": None})
        return tr
    except ValueError as e:
        ep.calls.append({"kind": "frame", "windows": (ep.t - t0) // WIN, "bytes": None, "error": str(e)})
        raise


def logged_ping(timeout):
    ep, t0 = me(), me().t
    r = _ping(timeout)
    ep.calls.append({"kind": "ping", "windows": (ep.t - t0) // WIN, "result": bool(r)})
    return r


cc.escuchar_y_retornar_trama = logged_listen
cc.escuchar_ping = logged_ping


def run(roles, sigma=0.0, seed=0, seconds=0):
    """roles: {name: callable}; each runs in its own thread as one endpoint."""
    global AIR
    AIR = Air(sigma, seed, seconds)
    results, eps = {}, {}

    def body(name, fn):
        ep = Endpoint(AIR, name)
        with AIR.cond:
            AIR.eps[threading.get_ident()] = ep
            eps[name] = ep
            started.release()
        go.wait()
        try:
            results[name] = fn()
        finally:
            with AIR.cond:
                ep.done = True
                AIR.cond.notify_all()

    started, go = threading.Semaphore(0), threading.Event()
    threads = [threading.Thread(target=body, args=(n, f)) for n, f in roles.items()]
    for t in threads:
        t.start()
    for _ in threads:
        started.acquire()
    go.set()
    for t in threads:
        t.join()
    return {n: {"result": results[n], "clock": eps[n].t, "played": eps[n].played, "calls": eps[n].calls} for n in roles}


MAC_A, MAC_B = "aa:bb:cc:dd:ee:ff", "11:22:33:44:55:66"
G = {"sample_rate": SR, "window": WIN, "mac_sender": MAC_A, "mac_receiver": MAC_B, "scenarios": []}
text = open(os.path.join(REF, "prueba.txt"), "rb").read().decode()

# 1. short string, clean air
msg = "hola mundo"
out = run({"tx": lambda: cc.enviar_string_por_sonido(msg, MAC_A, MAC_B), "rx": lambda: cc.escuchar_string(MAC_B)})
assert out["tx"]["result"] is True and out["rx"]["result"] == msg, out
G["scenarios"].append({"name": "string_short_clean", "kind": "string", "message": msg, "sigma": 0.0, "seed": 0,
                       "noise_seconds": 0, **out})

# 2. the reference's sample text (two data frames), noisy air
out = run({"tx": lambda: cc.enviar_string_por_sonido(text, MAC_A, MAC_B), "rx": lambda: cc.escuchar_string(MAC_B)},
          sigma=0.05, seed=11, seconds=400)
assert out["tx"]["result"] is True and out["rx"]["result"] == text, (out["tx"]["result"], out["rx"]["result"])
G["scenarios"].append({"name": "string_prueba_noise", "kind": "string", "message": text, "sigma": 0.05, "seed": 11,
                       "noise_seconds": 400, **out})

# 3. nobody answers
out = run({"tx": lambda: cc.enviar_string_por_sonido(msg, MAC_A, MAC_B)})
assert out["tx"]["result"] is False
G["scenarios"].append({"name": "string_no_receiver", "kind": "string", "message": msg, "sigma": 0.0, "seed": 0,
                       "noise_seconds": 0, **out})

# 4. file transfer with OK-frame acknowledgements (150 bytes -> two data frames)
data = bytes(np.random.RandomState(5).randint(0, 256, 150).tolist())
cwd = os.getcwd()
with tempfile.TemporaryDirectory() as tmp:
    os.chdir(tmp)
    with open("dato.bin", "wb") as f:
        f.write(data)
    out = run({"tx": lambda: cc.enviar_archivo_por_sonido("dato.bin", MAC_A, MAC_B), "rx": lambda: cc.escuchar_archivo(MAC_B)})
    got = open("new_dato.bin", "rb").read() if os.path.exists("new_dato.bin") else None
    os.chdir(cwd)
G["scenarios"].append({"name": "file_two_frames", "kind": "file", "file_name": "dato.bin", "file_bytes": data.hex(),
                       "received_bytes": None if got is None else got.hex(), "sigma": 0.0, "seed": 0,
                       "noise_seconds": 0, **out})
print("file transfer:", out["tx"]["result"], out["rx"]["result"], got == data)

# 5. the WAV files of mode 1 (guardar_archivo_en_tramas_wav, c1_shared.py:440-525), read back as a
#    capture (int16 / 32768, 0.5 s of silence between files) and heard by the reference's listener
wavs = {}
cs.write = lambda name, rate, data_: wavs.__setitem__(os.path.basename(name), (rate, np.array(data_)))
with tempfile.TemporaryDirectory() as tmp:
    src = os.path.join(tmp, "prueba.txt")
    with open(src, "wb") as f:
        f.write(text.encode())
    cs.guardar_archivo_en_tramas_wav(src, MAC_A, MAC_B)
names = sorted(wavs)
assert all(r == SR and d.dtype == np.int16 for r, d in wavs.values())
gap = np.zeros(SR // 2)
cap = np.concatenate([np.concatenate([gap, wavs[n][1].astype(np.float64) / 32768.0]) for n in names] + [gap])


def hear_all():
    _play(cap)          # put the capture on the air, then rewind this endpoint's clock and listen to it
    me().t = 0
    for _ in range(len(names) + 1):
        try:
            cc.escuchar_y_retornar_trama(cs.TIME_TO_SAY_128_BYTES)
        except ValueError:
            pass
    return True


out = run({"rx": hear_all})
G["wav_listen"] = {"files": names, "n_samples": [int(len(wavs[n][1])) for n in names], "gap": len(gap),
                   "timeout": cs.TIME_TO_SAY_128_BYTES, "calls": out["rx"]["calls"], "clock": out["rx"]["clock"]}
print("wav listen:", [(c["windows"], c["error"], None if c["bytes"] is None else len(c["bytes"]) // 2) for c in out["rx"]["calls"]])

path = os.path.join(HERE, "duplex_golden.json")
with open(path, "w") as f:
    json.dump(G, f, separators=(",", ":"))
print("wrote", path, os.path.getsize(path), "bytes")
for s in G["scenarios"]:
    print(s["name"], {k: (v["result"] if not isinstance(v["result"], str) else v["result"][:20], v["clock"], v["played"], len(v["calls"]))
                      for k, v in s.items() if isinstance(v, dict)})

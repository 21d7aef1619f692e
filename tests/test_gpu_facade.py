"""The reference's call surface (Trama, emitir_trama, escuchar_y_retornar_trama,
DispositivoWaveNet ...) on the GPU path, against the vectors the reference produced.
The first two tests are the reference's own DispositivoWaveNET/tests/test_1.py."""
import argparse

import numpy as np
import pytest
import torch

from oracle import modem_oracle as mo

pytestmark = pytest.mark.gpu

MAC_O, MAC_D = bytes.fromhex("aabbccddeeff"), bytes.fromhex("112233445566")


@pytest.fixture()
def cs():
    from wavenet_amd.dispositivo_wavenet import c1_shared
    from wavenet_amd.dispositivo_wavenet.canal import CanalSintetico
    c1_shared.usar_canal(CanalSintetico())
    yield c1_shared
    c1_shared.usar_canal(None)


def test_listen(cs):
    from wavenet_amd.dispositivo_wavenet.canal import CanalSintetico
    from wavenet_amd.dispositivo_wavenet.dispositivo_wavenet import DispositivoWaveNet as wn
    cs.usar_canal(CanalSintetico(half_duplex=True))
    with pytest.raises(Exception) as e_info:
        w = wn("0:0:0:0:0:0", "1:1:1:1:1:1")
        w.listen(timeout=10)
    assert str(e_info.value) == "Failed to get string from layer 1"


def test_send(cs):
    from wavenet_amd.dispositivo_wavenet.canal import CanalSintetico
    from wavenet_amd.dispositivo_wavenet.dispositivo_wavenet import DispositivoWaveNet as wn
    cs.usar_canal(CanalSintetico(half_duplex=True))
    with pytest.raises(Exception) as e_info:
        w = wn("0:0:0:0:0:0", "1:1:1:1:1:1")
        w.send("wikir", timeout=10)
    assert str(e_info.value) == "Failed to send string through layer 1"


def test_trama_build_parse_and_errors(cs, golden):
    for f in golden["frames"][:40]:
        t = cs.crear_trama(f["version"], bytes.fromhex(f["mac_o"]), bytes.fromhex(f["mac_d"]), f["tipo"],
                           bytes.fromhex(f["payload"]))
        assert t.get_bytes().hex() == f["frame"] and t.get_checksum_valido()
        back = cs.Trama(bytes_trama=list(t.get_bytes()))
        assert (back.version, back.tipo, back.length, bytes(back.payload).hex()) == \
            (f["version"], f["tipo"], len(f["payload"]) // 2, f["payload"])
        assert bytes(back.mac_origen).hex() == f["mac_o"] and back.get_checksum_valido()
    for c in golden["parse_cases"]:
        raw = bytes.fromhex(c["raw"])
        if c["error"] is None:
            t = cs.Trama(bytes_trama=list(raw))
            assert (t.tipo, t.length, bytes(t.checksum).hex(), bool(t.get_checksum_valido())) == \
                (c["tipo"], c["length"], c["checksum"], c["crc_ok"]), c["name"]
        else:
            with pytest.raises(ValueError) as e:
                cs.Trama(bytes_trama=list(raw))
            assert str(e.value) == c["error"], c["name"]
    with pytest.raises(ValueError):
        cs.Trama(version=1, mac_origen=MAC_O, mac_destino=MAC_D, tipo=3)
    assert cs.calcular_checksum(b"123456789").hex() == "cbf43926"
    assert cs.crear_trama_archivo_info(MAC_O, MAC_D, 1, "str").get_bytes().hex() == golden["info_frame"]["frame"]
    assert cs.crear_trama_ok(MAC_O, MAC_D, bytes.fromhex("4b7b3be0")).get_bytes().hex() == golden["ok_frame"]["frame"]
    assert list(cs.decodificar_payload_archivo_info(bytes.fromhex(golden["info_frame"]["frame"])[15:-4])) == \
        golden["info_frame"]["decoded"]
    with pytest.raises(argparse.ArgumentTypeError):
        cs.mac_str_to_bytes("aa:bb:cc")
    assert cs.mac_str_to_bytes("aa:bb:cc:dd:ee:ff") == MAC_O
    assert [cs.freq_to_byte(cs.byte_to_freq(b)) for b in (0, 58, 255)] == [0, 58, 255]
    assert cs.freq_to_byte(200) is None and cs.freq_to_byte(18300.0) is None


def test_reference_quirks_are_kept(cs):
    """Two behaviours of the reference a port might 'fix': addresses that are not 6 bytes long are
    concatenated as they are (c1_shared.py:109-111; the checksum is zlib's over what was built), and
    escuchar_archivo calls .imprimir() on -1 after three failed listens (c1_communication.py:27-34)."""
    import zlib

    from wavenet_amd.dispositivo_wavenet import c1_communication
    t = cs.Trama(version=1, mac_origen=b"\xaa\xbb", mac_destino=b"\x01\x02\x03\x04\x05\x06\x07", tipo=3, payload=b"hola")
    body = bytes([1]) + b"\xaa\xbb" + b"\x01\x02\x03\x04\x05\x06\x07" + bytes([3, 4]) + b"hola"
    assert t.get_bytes() == body + zlib.crc32(body).to_bytes(4, "big") and bytes(t.checksum) == t.get_bytes()[-4:]
    with pytest.raises(ValueError):
        cs.Trama(version=256, mac_origen=MAC_O, mac_destino=MAC_D, tipo=3, payload=b"")
    cs.canal().cargar(np.zeros(3 * 7055))            # nothing on the channel: every listen fails at once
    with pytest.raises(AttributeError):
        c1_communication.escuchar_archivo("11:22:33:44:55:66")


def test_frames_built_and_validated_in_batches(cs, golden):
    """crear_tramas / validar_tramas: the frames of the golden vectors built by one launch equal the
    ones built one by one, and the batch validation equals get_checksum_valido frame by frame."""
    rng = np.random.default_rng(4)
    payloads = [bytes(rng.integers(0, 256, int(n), dtype=np.uint8)) for n in (0, 1, 19, 106, 107, 107, 64, 255)]
    lote = cs.crear_tramas(1, MAC_O, MAC_D, 3, payloads)
    uno = [cs.crear_trama(1, MAC_O, MAC_D, 3, p) for p in payloads]
    assert [t.get_bytes() for t in lote] == [t.get_bytes() for t in uno]
    assert [(t.version, t.tipo, t.length, bytes(t.payload), bytes(t.checksum)) for t in lote] == \
        [(t.version, t.tipo, t.length, bytes(t.payload), bytes(t.checksum)) for t in uno]
    assert cs.crear_tramas(1, MAC_O, MAC_D, 3, [b"hola"])[0].get_bytes().hex() == \
        "01aabbccddeeff1122334455660304686f6c614b7b3be0"      # the reference's own frame for this input
    lote[2].bytes_trama[20] ^= 1                     # a damaged payload byte
    lote[5].checksum = bytearray(b"\0\0\0\0")        # a wrong stored checksum
    assert cs.validar_tramas(lote) == [t.get_checksum_valido() for t in lote] == [True, True, False, True, True, False, True, True]
    assert cs.crear_tramas(1, MAC_O, MAC_D, 3, []) == [] and cs.validar_tramas([]) == []
    raro = cs.crear_tramas(1, b"\xaa", MAC_D, 3, [b"x", b"yy"])       # addresses of other lengths: the general path
    assert [t.get_bytes() for t in raro] == [cs.crear_trama(1, b"\xaa", MAC_D, 3, p).get_bytes() for p in (b"x", b"yy")]


def test_string_chunking(cs, golden):
    text = bytes.fromhex(golden["prueba_txt"])
    tramas = cs.obtener_tramas_desde_string(text, "aa:bb:cc:dd:ee:ff", "11:22:33:44:55:66")
    tramas[-1].tipo = cs.TIPO_TRAMA_FINAL_ARCHIVO
    tramas[-1]._build()
    on_air = [cs.crear_trama_archivo_info(MAC_O, MAC_D, len(tramas), "str")] + tramas
    assert [t.get_bytes().hex() for t in on_air] == golden["prueba_on_air_frames"]


def test_emit_then_listen_loopback(cs, golden):
    """Configuration 1: the reference's sample text, its three frames on one channel, heard
    back frame by frame; then nothing is left and the listen fails as the reference's does."""
    frames = [bytes.fromhex(x) for x in golden["prueba_on_air_frames"]]
    for raw in frames:
        cs.emitir_trama(cs.Trama(bytes_trama=list(raw)))
    c = cs.canal()
    assert c.escrito == sum(len(mo.emit_frame(f, mo.scheme_b())) for f in frames)
    heard = [cs.escuchar_y_retornar_trama(123) for _ in frames]
    assert [t.get_bytes() for t in heard] == frames and all(t.get_checksum_valido() for t in heard)
    total, nombre = cs.decodificar_payload_archivo_info(heard[0].payload)
    assert (total, nombre) == (2, "str")
    assert bytes(heard[1].payload + heard[2].payload) == bytes.fromhex(golden["prueba_txt"])
    with pytest.raises(ValueError) as e:
        cs.escuchar_y_retornar_trama(30)
    assert str(e.value) == "Faltan argumentos para crear la trama"


@pytest.mark.parametrize("idx", [1, 4, 6])
def test_listen_on_reference_captures(cs, golden, idx):
    """A recorded capture (gaps, noise) loaded into the channel: same frames or errors, and
    the same channel time consumed per call, as the reference on that capture."""
    s = golden["listen_b"][idx]
    parts = []
    for raw, gap in zip(s["frames"], s["gaps"]):
        parts += [np.zeros(gap), mo.emit_frame(bytes.fromhex(raw), mo.scheme_b())]
    x = np.concatenate(parts)
    if s["sigma"] > 0:
        x = x + s["sigma"] * np.random.default_rng(s["seed"]).standard_normal(len(x))
    c = cs.canal()
    c.cargar(x)
    for call in s["calls"]:
        antes = c.leido
        if call["error"] is None:
            assert cs.escuchar_y_retornar_trama(s["timeout"]).get_bytes().hex() == call["bytes"]
        else:
            with pytest.raises(ValueError) as e:
                cs.escuchar_y_retornar_trama(s["timeout"])
            assert str(e.value) == call["error"]
        assert (c.leido - antes) // 7055 == call["windows"]


def test_escuchar_ping(cs, golden):
    ping = mo.tone(250, 21168, 0.48, mo.scheme_b())
    c = cs.canal()
    for case in golden["ping_b"]:
        if case["name"] == "silence":
            x = np.zeros(case["n_samples"])
        else:
            gap = case["gap"] - 1
            x = np.concatenate([np.zeros(gap), ping, np.zeros(case["n_samples"] - gap - len(ping))])
        c.cargar(x)
        assert cs.escuchar_ping(case["timeout"]) == case["result"], case["name"]
        assert c.leido // 7055 == case["windows"]
    c.cargar(np.zeros(10))
    cs.ejecutar_ping()
    c.leido = 0
    assert cs.escuchar_ping(3) is True


def test_detectar_frecuencia_and_tones(cs, golden):
    for p in golden["pure_windows_b"]:
        c = cs.canal()
        c.cargar(np.zeros(0))
        cs.transmite_freq(p["freq"], duration=cs.BYTE_DURATION * 2)
        assert c.escrito == 21168
        w = c.rec(7055)
        f = cs.detectar_frecuencia(w, cs.SAMPLE_RATE)
        assert f == p["detected"]
    with pytest.raises(ValueError):
        cs.detectar_frecuencia(np.zeros(1000), 44100)
    t = golden["tones_b"][1]      # the 200 Hz separator
    c = cs.canal()
    c.cargar(np.zeros(0))
    cs.transmitir_silencio()
    got = c.rec(13230).cpu().numpy()
    assert np.allclose(got[t["idx"]], t["samples"], rtol=0, atol=6e-8)


def test_wav_and_capa1(cs, golden, tmp_path):
    from scipy.io import wavfile

    from wavenet_amd import capa1
    g = golden["wav_b"]
    t = cs.Trama(bytes_trama=list(bytes.fromhex(g["frame"])))
    path = str(tmp_path / "trama.wav")
    cs.guardar_trama_como_wav(t, path)
    rate, data = wavfile.read(path)
    assert rate == 44100 and data.dtype == np.int16 and len(data) == g["n"]
    assert data[:32].tolist() == g["head"] and data[-32:].tolist() == g["tail"]
    assert np.array_equal(data, mo.wav_int16(bytes.fromhex(g["frame"]), mo.scheme_b()))
    # prototype scheme
    assert bytes(capa1.crear_trama(1, MAC_O, MAC_D, 2, b"hola")).hex() == golden["capa1_frame"]["frame"]
    ga = golden["wav_a"]
    w = capa1.audio_int16(bytes.fromhex(ga["message"]))
    assert len(w) == ga["n"] and w[:32].tolist() == ga["head"] and w[-32:].tolist() == ga["tail"]
    trama = capa1.crear_trama(1, MAC_O, MAC_D, 2, b"hola WaveNet")
    audio = capa1.emitir_trama(trama)
    assert audio.numel() == (len(trama) + 1) * 2205
    heard = capa1.decodificar_audio(audio)
    # bytes above 127 have no tone the prototype's freq_to_byte accepts (capa1.py:28-30)
    want = bytes(b for b in bytes(trama) + bytes([20]) if b <= 127)
    assert heard == want
    for b, f, back in golden["capa1_byte_roundtrip"][::9]:
        x = mo.tone(capa1.byte_to_freq(b), 2205, 0.05, mo.scheme_a())
        assert capa1.detectar_frecuencia(x, 44100) == f and capa1.freq_to_byte(f) == back


def test_capa1_main_plays_what_the_reference_played(golden, tmp_path, monkeypatch, capsys):
    """Configuration 1, prototype scheme: capa1.main on the reference's sample file plays the ping
    frame, the file's two frames and the literal example frame (golden: the frames the reference
    played and their sample count); the audio equals the oracle's within float32 rounding, and
    detecting every 2205-sample window of it gives exactly the oracle's freq_to_byte sequence
    (bytes above 127 dropped, each 900 Hz end tone read as byte 20)."""
    from scipy.io import wavfile

    from wavenet_amd import capa1
    g = golden["capa1_main_audio"]
    src = tmp_path / "prueba.txt"
    src.write_bytes(bytes.fromhex(golden["prueba_txt"]))
    monkeypatch.chdir(tmp_path)
    capa1.audio_reproducido()
    capa1.main(["-a", str(src), "-b", "aa:bb:cc:dd:ee:ff", "-c", "11:22:33:44:55:66"])
    audio = capa1.audio_reproducido()
    assert audio.numel() == g["n_samples"]
    sa = mo.scheme_a()
    frames = [bytes.fromhex(f) for f in g["frames"]]
    want = np.concatenate([np.concatenate([mo.tone(f, n, d, sa) for f, n, d in mo.schedule(fr, sa)]) for fr in frames])
    assert np.max(np.abs(audio.cpu().numpy().astype(np.float64) - want)) <= 3e-8 + 5e-11
    out = capsys.readouterr().out
    assert out.count("Trama transmitida.") == 4 and "Enviando trama de 126 bytes..." in out
    # decode: the oracle's detector and freq_to_byte on every aligned window of the float32 audio
    bins, _ = mo.window_bins(audio.cpu().numpy().astype(np.float64), 2205)
    freqs = np.abs(np.fft.fftfreq(2205, 1 / 44100))[bins]
    oracle_bytes = [mo.freq_to_byte(float(f), sa) for f in freqs]
    heard = capa1.decodificar_audio(audio)
    assert heard == bytes(b for b in oracle_bytes if b is not None)
    assert heard == b"".join(bytes(b for b in fr if b <= 127) + bytes([20]) for fr in frames)
    rate, data = wavfile.read(str(tmp_path / "trama_128bytes_por_byte.wav"))
    ga = golden["wav_a"]
    assert rate == 44100 and len(data) == ga["n"] and data[:32].tolist() == ga["head"] and data[-32:].tolist() == ga["tail"]
    # one tone alone
    capa1.audio_reproducido()
    tone = capa1.transmite_freq(900)
    assert np.max(np.abs(tone.cpu().numpy() - mo.tone(900, 2205, 0.05, sa))) <= 3e-8 + 5e-11
    assert torch.equal(capa1.audio_reproducido(), tone)

"""The CPU oracle against the vectors the reference itself produced (tests/golden/).

These run without a GPU and without the reference tree: the JSON holds inputs, seeds and
the reference's outputs (digests for long arrays)."""
import hashlib

import numpy as np
import pytest

from oracle import modem_oracle as mo

SB, SA = mo.scheme_b(), mo.scheme_a()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_scheme_sample_counts():
    assert (SB.tone_n, SB.sep_n, SB.sep_short_n, SB.ping_n, SB.eof_n, SB.win_n) == \
        (10584, 13230, 10584, 21168, 21168, 7055)
    assert (SA.tone_n, SA.win_n) == (2205, 2205)


def test_frames_build_and_parse(golden):
    for fr in golden["frames"]:
        raw = mo.build_frame(fr["version"], bytes.fromhex(fr["mac_o"]), bytes.fromhex(fr["mac_d"]),
                             fr["tipo"], bytes.fromhex(fr["payload"]))
        assert raw.hex() == fr["frame"]
        p = mo.parse_frame(raw)
        assert p.crc_ok and p.payload.hex() == fr["payload"] and p.tipo == fr["tipo"]
    assert golden["frames"][0]["frame"] == "01aabbccddeeff11223344556601002def0858"
    assert golden["frames"][1]["frame"] == "01aabbccddeeff1122334455660304686f6c614b7b3be0"
    k = golden["crc_kat"]
    assert mo.crc32_be(bytes.fromhex(k["input"])).hex() == k["crc_be"] == "cbf43926"
    mac_o, mac_d = bytes.fromhex("aabbccddeeff"), bytes.fromhex("112233445566")
    assert mo.info_frame(mac_o, mac_d, 1, "str").hex() == golden["info_frame"]["frame"]
    assert list(mo.decode_info_payload(mo.parse_frame(bytes.fromhex(golden["info_frame"]["frame"])).payload)) \
        == golden["info_frame"]["decoded"]
    assert mo.ok_frame(mac_o, mac_d, bytes.fromhex(golden["ok_frame"]["checksum"])).hex() == golden["ok_frame"]["frame"]
    assert mo.build_frame(1, mac_o, mac_d, 2, b"hola").hex() == golden["capa1_frame"]["frame"]


def test_parse_cases(golden):
    for c in golden["parse_cases"]:
        raw = bytes.fromhex(c["raw"])
        if c["error"] is None:
            p = mo.parse_frame(raw)
            assert (p.version, p.tipo, p.length, p.payload.hex(), p.checksum.hex(), p.crc_ok) == \
                (c["version"], c["tipo"], c["length"], c["payload"], c["checksum"], c["crc_ok"]), c["name"]
        else:
            with pytest.raises(mo.FrameError) as e:
                mo.parse_frame(raw)
            assert str(e.value) == c["error"], c["name"]


def test_string_chunking(golden):
    text = bytes.fromhex(golden["prueba_txt"])
    fr = mo.string_transfer_frames(text, bytes.fromhex("aabbccddeeff"), bytes.fromhex("112233445566"))
    assert [f.hex() for f in fr] == golden["prueba_on_air_frames"]
    assert [len(f) for f in fr] == [26, 126, 56]


def test_bin_classes(golden):
    assert mo.bin_classes(7055, SB).tolist() == golden["classes_b_7055"]
    assert mo.bin_classes(2205, SA).tolist() == golden["classes_a_2205"]
    c = golden["classes_b_7055"]
    assert [k for k in range(len(c)) if c[k] & mo.CLS_PING] == [39, 40, 41]
    assert [k for k in range(len(c)) if c[k] & mo.CLS_EOF] == [47, 48, 49]


def test_tones(golden):
    durs = {21168: mo.BYTE_DURATION * 2, 13230: mo.SILENCE_DURATION, 10584: mo.BYTE_DURATION}
    for t in golden["tones_b"]:
        d = mo.SILENCE_DURATION * 4 / 5 if (t["freq"] == 200 and t["n"] == 10584) else durs[t["n"]]
        x = mo.tone(t["freq"], t["n"], d, SB)
        assert sha(x) == t["sha256_f64"]
        assert [float(x[i]) for i in t["idx"]] == t["samples"]
    t = golden["tone_a"]
    assert sha(mo.tone(t["freq"], t["n"], mo.A_BYTE_DURATION, SA)) == t["sha256_f64"]


def test_pure_tone_windows(golden):
    for p in golden["pure_windows_b"]:
        w = mo.tone(p["freq"], SB.ping_n, mo.BYTE_DURATION * 2, SB)[:7055]
        k, m = mo.detect_bin(w)
        assert (k, m) == (p["bin"], p["mag"])
        assert mo.detect_freq(w, 44100) == p["detected"]
        assert int(mo.bin_classes(7055, SB)[k]) == p["cls"]
    for b, f, back in golden["capa1_byte_roundtrip"]:
        w = mo.tone(mo.byte_to_freq(b, SA), 2205, mo.A_BYTE_DURATION, SA)
        assert mo.detect_freq(w, 44100) == f and mo.freq_to_byte(f, SA) == back == b


def _capture(s):
    parts = []
    for raw, gap in zip(s["frames"], s["gaps"]):
        parts += [np.zeros(gap), mo.emit_frame(bytes.fromhex(raw), SB)]
    x = np.concatenate(parts)
    if s["sigma"] > 0:
        x = x + s["sigma"] * np.random.default_rng(s["seed"]).standard_normal(len(x))
    return x


@pytest.mark.parametrize("idx", range(7))
def test_listen_scenarios(golden, idx):
    s = golden["listen_b"][idx]
    x = _capture(s)
    assert len(x) == s["n_samples"] and sha(x) == s["sha256_f64"]
    bins, mags = mo.window_bins(x, 7055)
    assert bins.tolist() == s["window_bins"]
    assert np.allclose(mags, s["window_mags"], rtol=1e-12, atol=1e-9)
    cap = mo.Capture(x)
    for call in s["calls"]:
        got, _, used = mo.listen_frame(cap, s["timeout"], SB, stop_at_end=False)
        assert used == call["windows"]
        assert bytes(got).hex() == call["oracle_bytes"]
        if call["error"] is None:
            assert bytes(got).hex() == call["bytes"]
        else:
            assert mo.PARSE_MESSAGES[mo.parse_status(bytes(got))] == call["error"]
    # the class-sequence form of the same loop gives the same calls
    cls = mo.bin_classes(7055, SB)[bins]
    tw = mo.timeout_to_windows(s["timeout"], SB)
    recs = mo.slice_classes(cls, tw, len(s["calls"]))
    for (w0, w1, by, _), call in zip(recs, s["calls"]):
        assert by.hex() == call["oracle_bytes"]
        if w1 < len(cls):
            assert w1 - w0 == call["windows"]


def test_sample_counts(golden):
    for c in golden["sample_counts_b"]:
        assert sum(n for _, n, _ in mo.schedule(bytes.fromhex(c["frame"]), SB)) == c["n_samples"]
    assert golden["sample_counts_b"][-1]["n_samples"] == 3056130


def test_ping(golden):
    ping = mo.tone(SB.ping_freq, SB.ping_n, mo.BYTE_DURATION * 2, SB)
    assert golden["ping_tone_n"] == len(ping)
    for c in golden["ping_b"]:
        if c["name"] == "silence":
            x = np.zeros(c["n_samples"])
        else:
            gap = c["gap"] - 1      # the ping's first sample is sin(0) = 0
            x = np.concatenate([np.zeros(gap), ping, np.zeros(c["n_samples"] - gap - len(ping))])
        cap = mo.Capture(x)
        assert mo.listen_ping(cap, c["timeout"], SB, stop_at_end=False) == c["result"]
        assert cap.cursor // 7055 == c["windows"]


def test_wav_variants(golden):
    g = golden["wav_b"]
    w = mo.wav_int16(bytes.fromhex(g["frame"]), SB)
    assert len(w) == g["n"] and sha(w) == g["sha256_i16"]
    assert w[:32].tolist() == g["head"] and w[13230:13262].tolist() == g["at_13230"]
    g = golden["wav_a"]
    w = mo.wav_int16(bytes.fromhex(g["message"]), SA)
    assert len(w) == g["n"] and sha(w) == g["sha256_i16"]
    g = golden["capa1_main_audio"]
    x = np.concatenate([np.concatenate([mo.tone(f, n, d, SA) for f, n, d in mo.schedule(bytes.fromhex(fr), SA)])
                        for fr in g["frames"]])
    assert len(x) == g["n_samples"] and sha(x) == g["sha256_f64"]

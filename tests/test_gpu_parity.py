"""GPU parity: every HIP kernel, called through the C ABI, against the CPU oracle on the
same inputs.  Integer/byte/index results must be identical; float results carry the
tolerance written next to each check."""
import zlib

import numpy as np
import pytest
import torch

from oracle import modem_oracle as mo
from wavenet_amd import _lib as L
from wavenet_amd import modem as M

pytestmark = pytest.mark.gpu

SB, SA = mo.scheme_b(), mo.scheme_a()
MAC_O, MAC_D = bytes.fromhex("aabbccddeeff"), bytes.fromhex("112233445566")


@pytest.fixture(scope="module")
def mb():
    return M.Modem(M.esquema_b())


@pytest.fixture(scope="module")
def ma():
    return M.Modem(M.esquema_a())


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rows_from(frames, stride):
    buf = np.zeros((len(frames), stride), dtype=np.uint8)
    lens = np.zeros(len(frames), dtype=np.int32)
    for i, f in enumerate(frames):
        buf[i, :len(f)] = np.frombuffer(bytes(f), dtype=np.uint8)
        lens[i] = len(f)
    return buf, lens


# ------------------------------------------------------------------------------- MFMA layout
def test_mfma_lane_layout(mb):
    """v_mfma_f32_16x16x4_f32: lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; register r of
    lane l holds D[4*(l>>4)+r][l&15].  The detection kernel is built on exactly this."""
    d = mb.selftest_mfma()
    i = np.arange(16)
    A = 1.0 + i[:, None] + 16.0 * np.arange(4)[None, :]
    B = 1.0 + 3.0 * i[None, :] + 100.0 * np.arange(4)[:, None]
    D = A @ B
    for lane in range(64):
        for r in range(4):
            assert d[lane, r] == D[4 * (lane >> 4) + r, lane & 15], (lane, r)


# ------------------------------------------------------------------------------- K4
@pytest.mark.parametrize("stride", [128, 48, 272, 50])
def test_crc32_batch(mb, stride):
    rng = np.random.default_rng(stride)
    n = 3000 + stride
    buf = rng.integers(0, 256, (n, stride), dtype=np.uint8)
    lens = rng.integers(0, stride + 1, n).astype(np.int32)
    lens[:4] = [0, 1, stride, stride - 1]
    got = mb.crc32(dev(buf), dev(lens)).cpu().numpy()
    want = np.array([zlib.crc32(buf[i, :lens[i]].tobytes()) for i in range(n)], dtype=np.int64)
    assert np.array_equal(got, want)


def test_crc32_known_answer(mb):
    buf, lens = rows_from([b"123456789", b""], 16)
    got = mb.crc32(dev(buf), dev(lens)).cpu().numpy()
    assert got.tolist() == [0xCBF43926, 0]


def test_frame_build_matches_reference_vectors(mb, golden):
    fr = golden["frames"]
    fields = np.zeros((len(fr), 16), dtype=np.uint8)
    pay = np.zeros((len(fr), 112), dtype=np.uint8)
    for i, f in enumerate(fr):
        p = bytes.fromhex(f["payload"])
        fields[i, :15] = np.frombuffer(bytes([f["version"]]) + bytes.fromhex(f["mac_o"]) + bytes.fromhex(f["mac_d"]) +
                                       bytes([f["tipo"], len(p)]), dtype=np.uint8)
        pay[i, :len(p)] = np.frombuffer(p, dtype=np.uint8)
    frames, lens = mb.build_frames(dev(fields), dev(pay), 128)
    frames, lens = frames.cpu().numpy(), lens.cpu().numpy()
    for i, f in enumerate(fr):
        assert bytes(frames[i, :lens[i]]).hex() == f["frame"]
        assert not frames[i, lens[i]:].any()


def test_frame_build_rejects_rows_that_do_not_fit(mb):
    fields = np.zeros((2, 16), dtype=np.uint8)
    fields[0, 14] = 200     # 219 bytes do not fit a 128-byte row
    fields[1, 14] = 3
    pay = np.zeros((2, 256), dtype=np.uint8)
    _, lens = mb.build_frames(dev(fields), dev(pay), 128)
    assert lens.cpu().tolist() == [-1, 22]


def test_frame_parse_validate(mb, golden):
    cases = golden["parse_cases"]
    raws = [bytes.fromhex(c["raw"]) for c in cases]
    buf, lens = rows_from(raws, 288)
    rec = M.Modem.records_to_numpy(mb.parse_frames(dev(buf), dev(lens)))
    msg = {L.ST_EMPTY: mo.PARSE_MESSAGES[mo.ST_EMPTY], L.ST_SHORT_HEADER: mo.PARSE_MESSAGES[mo.ST_SHORT_HEADER],
           L.ST_SHORT_BODY: mo.PARSE_MESSAGES[mo.ST_SHORT_BODY]}
    for c, raw, r in zip(cases, raws, rec):
        if c["error"] is None:
            assert r["status"] == L.ST_OK, c["name"]
            assert (int(r["tipo"]), int(r["length"]), bool(r["crc_ok"])) == (c["tipo"], c["length"], c["crc_ok"])
            assert int(r["crc"]) == zlib.crc32(raw[:15 + c["length"]])
        else:
            assert msg[int(r["status"])] == c["error"], c["name"]


def test_frame_parse_random_mixed(mb):
    """Mixed payload lengths 0..107, 1 % corrupted: the shape of the checksum microbenchmark."""
    rng = np.random.default_rng(5)
    n = 20000
    frames = []
    for i in range(n):
        ln = int(rng.integers(0, 108))
        f = bytearray(mo.build_frame(1, rng.bytes(6), rng.bytes(6), 3, rng.bytes(ln)))
        if rng.random() < 0.01:
            f[int(rng.integers(0, len(f)))] ^= 1 << int(rng.integers(0, 8))
        frames.append(bytes(f))
    buf, lens = rows_from(frames, 128)
    rec = M.Modem.records_to_numpy(mb.parse_frames(dev(buf), dev(lens)))
    for f, r in zip(frames, rec):
        st = mo.parse_status(f)
        assert int(r["status"]) == st
        if st == mo.ST_OK:
            p = mo.parse_frame(f)
            assert bool(r["crc_ok"]) == p.crc_ok and int(r["crc"]) == zlib.crc32(f[:15 + p.length])


# ------------------------------------------------------------------------------- K1
def test_synth_live_matches_reference_audio(mb, golden):
    """float32 samples = the reference's float64 samples rounded, up to the reference's own
    phase rounding: |diff| <= half a float32 ulp + 5e-11."""
    frames = [bytes.fromhex(golden["frames"][0]["frame"]), bytes.fromhex(golden["frames"][1]["frame"]),
              mo.build_frame(1, bytes(6), bytes(6), 3, bytes([7, 7, 7, 9, 9, 0, 0, 0, 0])),
              bytes.fromhex(golden["prueba_on_air_frames"][2])]
    buf, lens = rows_from(frames, 128)
    stride = M.pad4(max(len(mo.emit_frame(f, SB)) for f in frames) + 1000)
    audio, n_samples = mb.encode_streams(dev(buf), dev(lens), stride)
    audio, n_samples = audio.cpu().numpy(), n_samples.cpu().numpy()
    for i, f in enumerate(frames):
        ref = mo.emit_frame(f, SB)
        assert n_samples[i] == len(ref)
        got = audio[i, :len(ref)].astype(np.float64)
        half_ulp = 0.5 * np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
        assert np.all(np.abs(got - ref) <= half_ulp * 1.0001 + 5e-11)
        assert not audio[i, len(ref):].any()          # zero fill to the end of the slot
    # spot values the reference itself produced
    t = golden["tones_b"][0]       # the 250 Hz ping opens every frame
    assert np.allclose(audio[0, t["idx"][:16]], t["samples"][:16], rtol=0, atol=6e-8)


@pytest.mark.parametrize("plan", ["some_tones_tabulated", "no_tone_tabulated"])
def test_synth_tones_with_long_periods(plan):
    """A tone plan whose byte tones repeat only after tens of thousands of samples (no period
    table for them) and, in the second case, whose separator, ping and EOF tones do not
    either: the sample-by-sample paths must give the same correctly rounded samples."""
    import dataclasses
    extra = dict(base_freq=351, freq_step=71) if plan == "some_tones_tabulated" else \
        dict(base_freq=351, freq_step=71, sep_freq=201, ping_freq=251, eof_freq=301)
    esq = dataclasses.replace(M.esquema_b(), **extra)
    sch = dataclasses.replace(SB, **extra)
    m = M.Modem(esq)
    frames = [mo.build_frame(1, MAC_O, MAC_D, 3, bytes([0, 255, 3, 3, 17, 200])), mo.build_frame(1, MAC_D, MAC_O, 1, b"")]
    buf, lens = rows_from(frames, 32)
    stride = M.pad4(max(len(mo.emit_frame(f, sch)) for f in frames) + 9)
    audio, n_samples = m.encode_streams(dev(buf), dev(lens), stride)
    audio = audio.cpu().numpy()
    for i, f in enumerate(frames):
        ref = mo.emit_frame(f, sch)
        assert int(n_samples[i]) == len(ref)
        got = audio[i, :len(ref)].astype(np.float64)
        half_ulp = 0.5 * np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
        assert np.all(np.abs(got - ref) <= half_ulp * 1.0001 + 5e-11)
        assert not audio[i, len(ref):].any()
    w, n = m.encode_streams(dev(buf), dev(lens), stride, variant=L.EMIT_WAV, dtype=torch.int16)
    assert np.array_equal(w.cpu().numpy()[0, :int(n[0])], mo.wav_int16(frames[0], sch))


def test_synth_sines_computed_on_the_gpu_equal_the_host_table(monkeypatch):
    """Scheme B's synthesis computes the sines of its period tables in float64 on the GPU; schemes
    with short tones (and WN_SYNTH_TABLE=lds) look them up in the quarter-wave table the host
    built.  A ping tone of 1 Hz walks through every entry of that table (phase m = sample index,
    21 168 samples live, 13 230 in the WAV variant, table 11 026 entries): both ways must give the
    same float32 and the same int16 samples, and both must be the oracle's."""
    import dataclasses
    esq = dataclasses.replace(M.esquema_b(), ping_freq=1)
    sch = dataclasses.replace(SB, ping_freq=1)
    frames = [mo.build_frame(1, MAC_O, MAC_D, 3, bytes([0, 255, 3, 3, 17, 200]))]
    buf, lens = rows_from(frames, 32)
    ref = mo.emit_frame(frames[0], sch)
    stride = M.pad4(len(ref) + 9)
    got = {}
    for mode in ("computed", "lds"):
        if mode == "lds":
            monkeypatch.setenv("WN_SYNTH_TABLE", "lds")
        else:
            monkeypatch.delenv("WN_SYNTH_TABLE", raising=False)
        m = M.Modem(esq)
        audio, n = m.encode_streams(dev(buf), dev(lens), stride)
        wav, nw = m.encode_streams(dev(buf), dev(lens), stride, variant=L.EMIT_WAV, dtype=torch.int16)
        got[mode] = (audio.cpu().numpy()[0, :int(n[0])], wav.cpu().numpy()[0, :int(nw[0])])
    assert np.array_equal(got["computed"][0], got["lds"][0])
    assert np.array_equal(got["computed"][1], got["lds"][1])
    half_ulp = 0.5 * np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(got["computed"][0].astype(np.float64) - ref) <= half_ulp * 1.0001 + 5e-11)
    assert np.array_equal(got["computed"][1], mo.wav_int16(frames[0], sch))


@pytest.mark.parametrize("plan", ["live", "long_periods"])
def test_synth_small_footprint_kernel_writes_the_same_samples(plan, monkeypatch):
    """The small-footprint synthesis kernel (256 threads, sines read from the quarter-wave table,
    period tables of at most 893 samples) against the ordinary one: identical float32 and int16
    samples, for the live tone plan, for one whose byte tones have no period table (sample by
    sample from the table), with small parts and slots at every alignment, and nothing written
    outside the slots."""
    import dataclasses
    monkeypatch.setenv("WN_SYNTH_PART", "4096")
    esq = M.esquema_b() if plan == "live" else dataclasses.replace(M.esquema_b(), base_freq=351, freq_step=71)
    m = M.Modem(esq)
    rng = np.random.default_rng(3)
    msgs = [mo.build_frame(1, MAC_O, MAC_D, 3, bytes(rng.integers(0, 256, k).tolist())) for k in (0, 3, 1, 2)]
    buf, lens = rows_from(msgs, 32)
    lay = m.layout(dev(buf), dev(lens))
    ns = lay[3].cpu().tolist()
    slot = M.pad4(max(ns) + 40)
    offs = torch.tensor([k * (slot + 8) + k for k in range(4)], dtype=torch.int64, device="cuda")
    fill = torch.tensor([ns[0], ns[1] + 13, slot, slot - 2], dtype=torch.int64, device="cuda")
    for dtype in (torch.float32, torch.int16):
        lay_d = lay if dtype == torch.float32 else m.layout(dev(buf), dev(lens), L.EMIT_WAV)
        out = []
        for beside in (False, True):
            audio = torch.full((4 * (slot + 8) + 16,), 7, dtype=dtype, device="cuda")
            m.synth(lay_d, offs, audio, slot, fill if dtype == torch.float32 else None, beside=beside)
            out.append(audio.cpu().numpy())
        assert np.array_equal(out[0], out[1]), (plan, dtype)
        assert (out[0] != 7).any()


def test_synth_sample_counts_with_repeats(mb, golden):
    frames = [bytes.fromhex(c["frame"]) for c in golden["sample_counts_b"]]
    buf, lens = rows_from(frames, 128)
    lay = mb.layout(dev(buf), dev(lens))
    assert lay[3].cpu().tolist() == [c["n_samples"] for c in golden["sample_counts_b"]]


def test_synth_unaligned_destination_and_no_fill(mb):
    f = mo.build_frame(1, MAC_O, MAC_D, 1, b"")
    buf, lens = rows_from([f, f], 32)
    lay = mb.layout(dev(buf), dev(lens))
    n = int(lay[3][0])
    audio = torch.full((2 * n + 64,), 7.0, dtype=torch.float32, device="cuda")
    dst = torch.tensor([3, n + 10], dtype=torch.int64, device="cuda")
    mb.synth(lay, dst, audio, M.pad4(n))
    a = audio.cpu().numpy()
    ref = mo.emit_frame(f, SB).astype(np.float32)
    assert np.all(a[:3] == 7.0) and np.all(a[n + 3:n + 10] == 7.0) and np.all(a[2 * n + 10:] == 7.0)
    assert np.max(np.abs(a[3:n + 3] - ref)) <= 6e-8 and np.max(np.abs(a[n + 10:2 * n + 10] - ref)) <= 6e-8


@pytest.mark.parametrize("scheme", ["B", "A"])
@pytest.mark.parametrize("part", [4096, 20000])
def test_synth_small_parts_and_misaligned_slots(mb, ma, scheme, part, monkeypatch):
    """Many parts per frame (part boundaries inside tones) and slots that start at every
    alignment modulo 4 samples, with and without zero fill after the frame, for the live scheme
    (segment by segment from period tables) and the prototype (short segments, one run of
    vectors): every sample equals the oracle's, nothing outside the slots is touched."""
    monkeypatch.setenv("WN_SYNTH_PART", str(part))
    m, sch = (mb, SB) if scheme == "B" else (ma, SA)
    rng = np.random.default_rng(part)
    if scheme == "B":
        msgs = [mo.build_frame(1, MAC_O, MAC_D, 3, bytes(rng.integers(0, 256, k).tolist())) for k in (0, 3, 1, 2)]
    else:
        msgs = [bytes(rng.integers(0, 128, k).tolist()) for k in (5, 17, 1, 9)]
    refs = [mo.emit_frame(f, sch).astype(np.float32) for f in msgs]
    buf, lens = rows_from(msgs, 32)
    lay = m.layout(dev(buf), dev(lens))
    slot = M.pad4(max(len(r) for r in refs) + 40)
    offs = [k * (slot + 8) + k for k in range(4)]                  # alignments 0, 1, 2, 3
    audio = torch.full((4 * (slot + 8) + 16,), 7.0, dtype=torch.float32, device="cuda")
    fill = torch.tensor([len(refs[0]), len(refs[1]) + 13, slot, slot - 2], dtype=torch.int64, device="cuda")
    m.synth(lay, torch.tensor(offs, dtype=torch.int64, device="cuda"), audio, slot, fill)
    a = audio.cpu().numpy()
    written = np.zeros(len(a), dtype=bool)
    for off, ref, end in zip(offs, refs, fill.cpu().tolist()):
        end = max(end, len(ref))
        assert np.max(np.abs(a[off:off + len(ref)] - ref)) <= 6e-8
        assert not a[off + len(ref):off + end].any()
        written[off:off + end] = True
    assert np.all(a[~written] == 7.0)


def test_synth_wav_int16(mb, ma, golden):
    g = golden["wav_b"]
    f = bytes.fromhex(g["frame"])
    buf, lens = rows_from([f], 32)
    audio, n = mb.encode_streams(dev(buf), dev(lens), M.pad4(g["n"]), variant=L.EMIT_WAV, dtype=torch.int16)
    w = audio.cpu().numpy()[0, :g["n"]]
    assert int(n[0]) == g["n"] and np.array_equal(w, mo.wav_int16(f, SB))
    assert w[:32].tolist() == g["head"] and w[-32:].tolist() == g["tail"]
    g = golden["wav_a"]
    msg = bytes.fromhex(g["message"])
    buf, lens = rows_from([msg], 128)
    audio, n = ma.encode_streams(dev(buf), dev(lens), M.pad4(g["n"]), variant=L.EMIT_WAV, dtype=torch.int16)
    w = audio.cpu().numpy()[0, :g["n"]]
    assert int(n[0]) == g["n"] and np.array_equal(w, mo.wav_int16(msg, SA))


def test_synth_scheme_a_live(ma, golden):
    g = golden["capa1_main_audio"]
    frames = [bytes.fromhex(x) for x in g["frames"]]
    buf, lens = rows_from(frames, 128)
    stride = M.pad4(127 * 2205)
    audio, n = ma.encode_streams(dev(buf), dev(lens), stride)
    audio = audio.cpu().numpy()
    for i, f in enumerate(frames):
        ref = np.concatenate([mo.tone(fr, nn, d, SA) for fr, nn, d in mo.schedule(f, SA)])
        assert int(n[i]) == len(ref)
        assert np.max(np.abs(audio[i, :len(ref)] - ref)) <= 3e-8 + 5e-11


# ------------------------------------------------------------------------------- K2
def oracle_windows(x32, esq_cls, n):
    bins, mags = mo.window_bins(x32.astype(np.float64), n)
    return bins, mags, esq_cls[bins]


def capture_of(s):
    parts = []
    for raw, gap in zip(s["frames"], s["gaps"]):
        parts += [np.zeros(gap), mo.emit_frame(bytes.fromhex(raw), SB)]
    x = np.concatenate(parts)
    if s["sigma"] > 0:
        x = x + s["sigma"] * np.random.default_rng(s["seed"]).standard_normal(len(x))
    return x


def check_detect(m, x32, n, lut, mag_rtol=1e-4):
    nwin = len(x32) // n
    pad = np.zeros(M.pad4(nwin * n), dtype=np.float32)
    pad[:nwin * n] = x32[:nwin * n]
    cls, mag, kbin = m.detect(dev(pad), 1, len(pad), nwin)
    cls, mag, kbin = cls.cpu().numpy()[0], mag.cpu().numpy()[0], kbin.cpu().numpy()[0]
    obins, omags, ocls = oracle_windows(pad[:nwin * n], lut, n)
    assert np.array_equal(kbin, obins)
    assert np.array_equal(cls.astype(np.uint16), ocls)
    nz = omags > 0
    assert np.all(np.abs(mag[nz] - omags[nz]) <= mag_rtol * omags[nz])       # tone energies: 1e-4 relative
    assert np.all(mag[~nz] == 0)
    return cls


@pytest.mark.parametrize("idx", range(7))
def test_detect_reference_scenarios(mb, golden, idx):
    """Windows of the captures the reference itself decoded (float32 audio): dominant bin,
    class and magnitude per window, and the bins the reference saw in float64."""
    s = golden["listen_b"][idx]
    x = capture_of(s)
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    x32 = x.astype(np.float32)
    check_detect(mb, x32, 7055, lut)
    # float32 rounding of the audio does not move the reference's decisions here
    assert mo.window_bins(x32.astype(np.float64), 7055)[0].tolist() == s["window_bins"]


def test_detect_pure_tones_and_silence(mb, golden):
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    parts = [mo.tone(p["freq"], SB.ping_n, mo.BYTE_DURATION * 2, SB)[:7055] for p in golden["pure_windows_b"]]
    parts.append(np.zeros(7055))
    parts.append(np.full(7055, 0.25))
    x32 = np.concatenate(parts).astype(np.float32)
    cls = check_detect(mb, x32, 7055, lut)
    for p, c in zip(golden["pure_windows_b"], cls):
        assert int(c) == p["cls"]


def test_detect_noise_only_and_offsets(mb, golden):
    """Noise-only windows (the reference's argmax is then any bin) at every alignment of the
    window start modulo 4 samples, several streams."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    rng = np.random.default_rng(11)
    n_streams, n_windows, first = 5, 9, 3
    stride = M.pad4(first + n_windows * 7055 + 6)
    a = (0.1 * rng.standard_normal((n_streams, stride))).astype(np.float32)
    cls, mag, kbin = mb.detect(dev(a), n_streams, stride, n_windows, first)
    for s in range(n_streams):
        ob, om, oc = oracle_windows(a[s, first:first + n_windows * 7055], lut, 7055)
        assert np.array_equal(kbin[s].cpu().numpy(), ob)
        assert np.array_equal(cls[s].cpu().numpy().astype(np.uint16), oc)
        assert np.allclose(mag[s].cpu().numpy(), om, rtol=1e-4)


@pytest.mark.parametrize("exp", [-100, -80, -45, 40, 60])
def test_detect_dynamic_range(mb, golden, exp):
    """Scaling the samples by a power of two scales the reference's float64 spectrum exactly,
    so its decisions do not change.  Very quiet and very loud captures (whose float32 squared
    magnitudes under- or overflow) must still give the oracle's bins; digital silence between
    them gives bin 0."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    frame = mo.build_frame(1, MAC_O, MAC_D, 3, b"rango")
    x = mo.emit_frame(frame, SB)[:12 * 7055]
    x = x + 0.02 * np.random.default_rng(exp & 0xFF).standard_normal(len(x))
    x[3 * 7055:5 * 7055] = 0.0
    x32 = (x.astype(np.float32) * np.float32(2.0) ** exp).astype(np.float32)
    assert np.isfinite(x32).all() and (x32 != 0).any()
    mb.detect_stats()
    check_detect(mb, x32, 7055, lut)
    windows, top2, full = mb.detect_stats()
    assert windows == 12
    if exp in (-100, -80, 60):
        assert full == 10      # every window with signal went through the float64 spectrum


@pytest.mark.parametrize("scheme", ["B", "A"])
def test_detect_every_bin(mb, ma, golden, scheme):
    """One window per bin k of [0, N/2): a tone on the centre of bin k with a weaker one on
    another bin.  Exercises every entry of the kernel's output-to-bin map (both halves of
    every sum pair, the self-mirrored rows and columns, the excluded top bin)."""
    m, n, lut = (mb, 7055, golden["classes_b_7055"]) if scheme == "B" else (ma, 2205, golden["classes_a_2205"])
    lut = np.array(lut, dtype=np.uint16)
    t = np.arange(n)
    rng = np.random.default_rng(5)
    wins = []
    for k in range(n // 2):
        k2 = (k + 1 + int(rng.integers(0, n // 2 - 2))) % (n // 2)
        w = 0.4 * np.cos(2 * np.pi * k * t / n + rng.uniform(0, 6.28)) + 0.1 * np.cos(2 * np.pi * k2 * t / n + 0.7)
        wins.append(w.astype(np.float32))
    x32 = np.concatenate(wins)
    cls = check_detect(m, x32, n, lut)
    kbin = m.detect(dev(np.pad(x32, (0, 4))[:M.pad4(len(x32))]), 1, M.pad4(len(x32)), n // 2)[2].cpu().numpy()[0]
    assert kbin.tolist() == list(range(n // 2))
    # the bin just above the searched range must never win: put all the energy there
    top = (0.4 * np.cos(2 * np.pi * (n // 2) * t / n) + 0.01 * np.cos(2 * np.pi * 7 * t / n)).astype(np.float32)
    check_detect(m, top, n, lut)


@pytest.mark.parametrize("scheme", ["B", "A"])
def test_detect_energy_outside_the_searched_bins(mb, ma, golden, scheme):
    """Nearly all the energy in bin N//2, the one bin np.argmax never looks at, and far below it
    (a) a noise floor, (b) two tones a millionth apart in amplitude, (c) one clear tone.  The float32
    error of every searched bin scales with the energy of the whole window, so (a) and (b) can
    only be decided in float64; the kernel has to notice that from the window's energy, not from
    the searched bins alone.  Bins, classes and magnitudes must be the oracle's."""
    m, n, lut = (mb, 7055, golden["classes_b_7055"]) if scheme == "B" else (ma, 2205, golden["classes_a_2205"])
    lut = np.array(lut, dtype=np.uint16)
    t = np.arange(n)
    rng = np.random.default_rng(21)
    top = 0.4 * np.cos(2 * np.pi * (n // 2) * t / n)
    wins = []
    for i in range(6):                                   # (a) noise floor 1e-5
        wins.append(top + 1e-5 * rng.standard_normal(n))
    for i in range(6):                                   # (b) near-tie of two weak tones
        k1, k2 = int(rng.integers(60, n // 2 - 5)), int(rng.integers(60, n // 2 - 5))
        eps = (1e-6, -1e-6, 3e-6, -3e-6, 1e-5, -1e-5)[i]
        wins.append(top + 1e-4 * np.cos(2 * np.pi * k1 * t / n + 0.4) + 1e-4 * (1 + eps) * np.cos(2 * np.pi * k2 * t / n + 1.1))
    for a in (1e-3, 1e-2):                               # (c) clear winner, 400x and 40x below the top bin
        wins.append(top + a * np.cos(2 * np.pi * 777 * t / n + 0.2))
    x32 = np.concatenate(wins).astype(np.float32)
    m.detect_stats()
    check_detect(m, x32, n, lut)
    windows, top2, full = m.detect_stats()
    assert windows == 14 and top2 + full >= 12           # (a) and (b) were not decided in float32


def test_detect_float32_error_stays_inside_the_bound(mb):
    """The flag rests on |dZ| <= 724 u sqrt(E) (u = 2^-24, E = energy of the spectrum) for every
    float32 output.  Measured here on the reported peak magnitude of windows the float32 pass
    decided alone: the error must sit far inside that bound."""
    rng = np.random.default_rng(8)
    n, t = 7055, np.arange(7055)
    wins = [0.3 * rng.standard_normal(n) for _ in range(8)]
    wins += [0.45 * np.cos(2 * np.pi * (350 + 70 * b) * t / 44100) + 0.5 for b in (0, 58, 121, 184, 255)]   # with a DC offset
    wins += [np.abs(0.9 * rng.standard_normal(n)) for _ in range(3)]                                     # all-positive samples
    x32 = np.concatenate(wins).astype(np.float32)
    nwin = len(wins)
    mb.detect_stats()
    cls, mag, kbin = mb.detect(dev(np.pad(x32, (0, M.pad4(len(x32)) - len(x32)))), 1, M.pad4(len(x32)), nwin)
    _, top2, full = mb.detect_stats()
    spec = np.abs(np.fft.fft(x32.astype(np.float64).reshape(nwin, n), axis=1))[:, :n // 2 + 1]
    energy = (spec ** 2).sum(axis=1)
    err = np.abs(mag.cpu().numpy()[0].astype(np.float64) - spec[np.arange(nwin), kbin.cpu().numpy()[0]])
    bound = 724 * 2.0 ** -24 * np.sqrt(energy)
    assert top2 + full < nwin                            # some windows were decided in float32
    assert np.all(err <= bound / 8), (err / bound).max()


def test_detect_leaves_out_only_columns_that_cannot_matter(mb, golden):
    """The second DFT stage runs on 16 spectrum columns at a time and stops once the energy outside
    the outputs seen cannot reach the decision threshold.  Clean tones need few columns, noise needs
    all 43, a tone on the bin that is not searched needs all of them too; bins, classes and
    magnitudes equal the oracle's whole-spectrum argmax in every case, and so do near-ties whose two
    peaks sit in different column tiles."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    t = np.arange(7055)
    rng = np.random.default_rng(31)
    tones = np.concatenate([0.45 * np.sin(2 * np.pi * (350 + 70 * b) * t / 44100) for b in range(0, 256, 5)])
    mb.detect_stats()
    check_detect(mb, tones.astype(np.float32), 7055, lut)
    w, _, _, cols, s1, _ = mb.detect_stats(columns=True)
    assert w == 52 and cols < 45 * w                         # mostly one or two of the nine (row tile, column tile) cells
    assert s1 < 0.7 * w * 12 * 43                            # and most of stage 1 was left out with them
    noise = 0.2 * rng.standard_normal(7055 * 8)
    check_detect(mb, noise.astype(np.float32), 7055, lut)
    w, _, _, cols, s1, _ = mb.detect_stats(columns=True)
    assert (w, cols, s1) == (8, 8 * 3 * 43, 8 * 12 * 43)
    # two tones of nearly equal strength whose bins fall into different column tiles (k mod 85,
    # folded, in [0,16), [16,32), [32,43)), in both orders of strength, with and without a third
    wins = []
    for ka, kb in ((5, 20), (90, 2000), (40, 1300), (700, 37), (3000, 25)):
        for eps in (1e-7, -1e-7, 1e-3, -1e-3, 0.2, -0.2):
            x = np.cos(2 * np.pi * ka * t / 7055 + 0.3) + (1 + eps) * np.cos(2 * np.pi * kb * t / 7055 + 1.0)
            wins.append(0.3 * x)
            wins.append(0.3 * x + 0.29 * np.cos(2 * np.pi * 1717 * t / 7055))
    check_detect(mb, np.concatenate(wins).astype(np.float32), 7055, lut)
    # DC offset (bin 0 carries most of the energy) and a weak tone on top of it
    dc = np.concatenate([0.5 + 0.01 * np.cos(2 * np.pi * 500 * t / 7055), 0.5 + 1e-4 * rng.standard_normal(7055),
                         -0.7 + 0.3 * np.cos(2 * np.pi * 2222 * t / 7055)])
    check_detect(mb, dc.astype(np.float32), 7055, lut)


def test_detect_windows_done_twice(mb, golden):
    """Windows with four fifths of their energy in one tile of spectrum columns, spread over many
    bins, and the largest bin in another tile: the kernel starts them on the first tile alone, finds
    that the rest cannot be ruled out, fetches the window again and does it the whole way.  Several streams at odd offsets, so that the second
    fetch and the fetch of the next window are exercised at every alignment; bins, classes and
    magnitudes must still be the oracle's."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    rng = np.random.default_rng(41)
    t = np.arange(7055)
    n_streams, n_windows, first = 3, 7, 5
    stride = M.pad4(first + n_windows * 7055 + 9)
    a = np.zeros((n_streams, stride), dtype=np.float32)
    for s in range(n_streams):
        for w in range(n_windows):
            tile = (s + w) % 3
            cols = np.arange(16 * tile, min(16 * tile + 16, 43))
            bins = [k for k in range(60, 3500) if min(k % 85, 85 - k % 85) in cols]
            chosen = rng.choice(bins, 40, replace=False)
            x = sum(rng.uniform(0.5, 1.0) * np.cos(2 * np.pi * k * t / 7055 + rng.uniform(0, 6.28)) for k in chosen)
            # ... and a fifth of the energy on one bin of another tile: the true maximum, outside the tile the kernel starts on
            other = [k for k in range(60, 3500) if min(k % 85, 85 - k % 85) // 16 == (tile + 1) % 3]
            x = x + 2.4 * np.cos(2 * np.pi * int(rng.choice(other)) * t / 7055 + 0.5)
            if w % 3 == 2:
                x = 0.45 * np.sin(2 * np.pi * (350 + 70 * (16 * s + w)) * t / 44100) * 40    # an ordinary tone in between
            a[s, first + w * 7055:first + (w + 1) * 7055] = 0.01 * x
    mb.detect_stats()
    cls, mag, kbin = mb.detect(dev(a), n_streams, stride, n_windows, first)
    st = mb.detect_stats(columns=True)
    assert st[0] == n_streams * n_windows and st[5] >= 10          # most of the spread windows were done twice
    for s in range(n_streams):
        ob, om, oc = oracle_windows(a[s, first:first + n_windows * 7055], lut, 7055)
        assert np.array_equal(kbin[s].cpu().numpy(), ob)
        assert np.array_equal(cls[s].cpu().numpy().astype(np.uint16), oc)
        assert np.allclose(mag[s].cpu().numpy(), om, rtol=1e-4)


def test_detect_weakest_column_tile_left_out(mb, golden):
    """Windows whose energy sits in two of the three tiles of spectrum columns go the long way
    without the third tile.  (a) Two strong bins, one in each of two tiles, in every pairing of the
    tiles: decided after the two tiles.  (b) Two tiles filled with many equal bins and a tenth of the
    energy on one bin of the third tile, which is the true maximum: after the two tiles the rest
    cannot be ruled out, the window is fetched again and done with all three.  (c) The same with
    the third tile's bin weaker than the others: still the long way twice, another winner.  Several
    streams at odd offsets; bins, classes and magnitudes are the oracle's throughout."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    rng = np.random.default_rng(43)
    t = np.arange(7055)
    tile_of = lambda k: min(k % 85, 85 - k % 85) // 16
    by_tile = [[k for k in range(60, 3500) if tile_of(k) == c] for c in range(3)]
    n_streams, n_windows, first = 3, 9, 3
    stride = M.pad4(first + n_windows * 7055 + 11)
    a = np.zeros((n_streams, stride), dtype=np.float32)
    for s in range(n_streams):
        for w in range(n_windows):
            left_out = (s + w) % 3
            ca, cb = [c for c in range(3) if c != left_out]
            kind = w % 3
            if kind == 0:
                x = np.cos(2 * np.pi * int(rng.choice(by_tile[ca])) * t / 7055 + 0.3) \
                    + 0.8 * np.cos(2 * np.pi * int(rng.choice(by_tile[cb])) * t / 7055 + 1.1)
            else:
                x = np.zeros(7055)
                for c in (ca, cb):
                    for k in rng.choice(by_tile[c], 30, replace=False):
                        x += rng.uniform(0.7, 1.0) * np.cos(2 * np.pi * int(k) * t / 7055 + rng.uniform(0, 6.28))
                # 60 bins of amplitude 0.7 .. 1 in two tiles; one bin in the third with amplitude 2.6 (an
                # eighth of the energy, the maximum) or 0.6 (not the maximum)
                x += (2.6 if kind == 1 else 0.6) * np.cos(2 * np.pi * int(rng.choice(by_tile[left_out])) * t / 7055 + 0.5)
            a[s, first + w * 7055:first + (w + 1) * 7055] = 0.02 * x
    mb.detect_stats()
    cls, mag, kbin = mb.detect(dev(a), n_streams, stride, n_windows, first)
    st = mb.detect_stats(paths=True)
    assert st[0] == n_streams * n_windows
    assert st[7] >= 16                      # the windows of many bins left their weakest tile out ...
    assert st[5] >= 8                       # ... and those of kind (b) were done again with all three
    assert st[8] >= 8                       # the windows of two bins took one tile of chosen columns instead
    for s in range(n_streams):
        ob, om, oc = oracle_windows(a[s, first:first + n_windows * 7055], lut, 7055)
        assert np.array_equal(kbin[s].cpu().numpy(), ob)
        assert np.array_equal(cls[s].cpu().numpy().astype(np.uint16), oc)
        assert np.allclose(mag[s].cpu().numpy(), om, rtol=1e-4)


def test_detect_separator_bin_alone(monkeypatch, mb, golden):
    """Windows that hold little else than the separator tone are decided from that one bin and the
    window's energy.  (a) The separator alone, at several loudnesses down to 1e-6, with and without a
    little noise.  (b) The separator with a data tone over the last 5 to 45 % of the window: decided
    from the one bin while the rest cannot compete, by the usual ways beyond.  (c) The separator's
    frequency next door (bins 31 and 33 louder than 32) and a tone twice as strong in the same
    spectrum column (bin 32 + 85): the pilot sends them to the test, the test sends them back.  (d) The
    separator's bin and an equally loud bin elsewhere (a tie for float64).  Bins, classes and
    magnitudes are the oracle's; with the test switched off the same windows give the same."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    rng = np.random.default_rng(83)
    t = np.arange(7055)
    n_streams, n_windows = 3, 24
    firsts = (0, 1, 3)      # every alignment of a window in its buffer
    stride = M.pad4(3 + n_windows * 7055 + 9)
    a = np.zeros((n_streams, stride), dtype=np.float32)
    sep = np.sin(2 * np.pi * 200 * t / 44100)
    for s in range(n_streams):
        for w in range(n_windows):
            kind = w % 6
            if kind == 0:
                x = 0.5 * 10.0 ** -float(rng.integers(0, 7)) * sep
            elif kind == 1:
                x = 0.5 * sep + 0.02 * rng.standard_normal(7055)
            elif kind == 2:
                cut = int(7055 * (1.0 - rng.uniform(0.05, 0.45)))
                x = 0.5 * sep.copy()
                x[cut:] = 0.5 * np.sin(2 * np.pi * (350 + 70 * int(rng.integers(0, 256))) * t[cut:] / 44100)
            elif kind == 3:
                x = 0.3 * np.cos(2 * np.pi * 31 * t / 7055) + 0.3 * np.cos(2 * np.pi * 33 * t / 7055 + 1.0) \
                    + 0.2 * np.cos(2 * np.pi * 32 * t / 7055 + 2.0)
            elif kind == 4:
                x = 0.2 * np.cos(2 * np.pi * 32 * t / 7055 + 0.3) + 0.4 * np.cos(2 * np.pi * (32 + 85) * t / 7055 + 1.0)
            else:
                x = 0.3 * np.cos(2 * np.pi * 32 * t / 7055 + 0.3) + 0.3 * np.cos(2 * np.pi * 1000 * t / 7055 + 1.0)
            a[s, firsts[s] + w * 7055:firsts[s] + (w + 1) * 7055] = x
    monkeypatch.setenv("WN_DETECT_SEP", "0")
    off = M.Modem(M.esquema_b())
    for m, on in ((mb, True), (off, False)):
        for s in range(n_streams):
            m.detect_stats()
            cls, mag, kbin = m.detect(dev(a[s:s + 1]), 1, stride, n_windows, firsts[s])
            st = m.detect_stats(paths=True)
            assert st[0] == n_windows
            if on:
                assert st[9] >= 8 and st[9] <= 12      # kinds (a), (b) and part of (c)
                assert st[5] >= 4                      # kind (d) at least came back from the test
            else:
                assert st[9] == 0
            ob, om, oc = oracle_windows(a[s, firsts[s]:firsts[s] + n_windows * 7055], lut, 7055)
            assert np.array_equal(kbin[0].cpu().numpy(), ob)
            assert np.array_equal(cls[0].cpu().numpy().astype(np.uint16), oc)
            assert np.allclose(mag[0].cpu().numpy(), om, rtol=1e-4)


def test_detect_scheme_tone_bin_alone(monkeypatch, mb, golden):
    """Windows that hold little else than one tone of the scheme (a byte tone, the ping, the end
    tone) are decided from that one bin: the pilot names the spectrum column, the swing of its row
    sums names the tone among those of the column.  (a) Every byte tone, the ping and the end tone
    alone, at a random start phase.  (b) Byte tones with the separator over the first 5 to 30 % of the
    window, and with noise.  (c) Tones that are not the scheme's (35 Hz off): the wrong bin is tried,
    the window starts again and goes the usual ways.  (d) Two byte tones of the same spectrum
    column, the weaker one at 0.45 of the stronger.  Bins, classes and magnitudes are the oracle's;
    with the test switched off the same windows give the same."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    rng = np.random.default_rng(89)
    t = np.arange(7055)
    tones = [350 + 70 * b for b in range(256)] + [250, 300]
    col = lambda f: min(round(f * 7055 / 44100) % 85, 85 - round(f * 7055 / 44100) % 85)
    windows = []
    for f in tones:                                                                   # (a)
        windows.append(0.5 * np.sin(2 * np.pi * f * (t + int(rng.integers(0, 5000))) / 44100))
    for _ in range(60):                                                               # (b)
        f = tones[int(rng.integers(0, 256))]
        cut = int(7055 * rng.uniform(0.05, 0.30))
        x = 0.5 * np.sin(2 * np.pi * 200 * t / 44100)
        x[cut:] = 0.5 * np.sin(2 * np.pi * f * t[cut:] / 44100)
        windows.append(x + 10 ** rng.uniform(-4, -1.3) * rng.standard_normal(7055))
    for _ in range(40):                                                               # (c)
        f = tones[int(rng.integers(0, 256))] + 35
        windows.append(0.5 * np.sin(2 * np.pi * f * t / 44100 + rng.uniform(0, 6.28)))
    same_col = [(f, g) for f in tones[:256] for g in tones[:256] if f < g and col(f) == col(g)]
    for i in rng.choice(len(same_col), 40, replace=False):                            # (d)
        f, g = same_col[int(i)]
        windows.append(0.5 * np.sin(2 * np.pi * f * t / 44100) + 0.225 * np.sin(2 * np.pi * g * t / 44100 + 1.0))
    n_windows = len(windows)
    first = 2
    stride = M.pad4(first + n_windows * 7055 + 5)
    a = np.zeros((1, stride), dtype=np.float32)
    for w, x in enumerate(windows):
        a[0, first + w * 7055:first + (w + 1) * 7055] = x
    monkeypatch.setenv("WN_DETECT_BIN", "0")
    off = M.Modem(M.esquema_b())
    ob, om, oc = oracle_windows(a[0, first:first + n_windows * 7055], lut, 7055)
    for m, on in ((mb, True), (off, False)):
        m.detect_stats()
        cls, mag, kbin = m.detect(dev(a), 1, stride, n_windows, first)
        st = m.detect_stats(paths=True)
        assert st[0] == n_windows
        if on:
            assert st[10] >= 200                    # most of (a) and (b): a tone two fifths of a bin off its bin holds too little in it
            assert st[5] >= 20                      # (c) came back from the test
        else:
            assert st[10] == 0
        assert np.array_equal(kbin[0].cpu().numpy(), ob)
        assert np.array_equal(cls[0].cpu().numpy().astype(np.uint16), oc)
        assert np.allclose(mag[0].cpu().numpy(), om, rtol=1e-4)


@pytest.mark.parametrize("base, step, sep", [(410, 5, 180), (300, 37, 0), (1000, 64, 2210)])
def test_detect_other_tone_plans(base, step, sep):
    """The one-bin tests are driven by tables made from the scheme's tone plan when a context is made.
    Plans other than the reference's: tones less than a bin apart (several share a bin, more than
    eight fall in one spectrum column), a plan without a separator, a plan whose separator lies
    among the byte tones.  Each tone alone, each with the separator (or its neighbour tone) over part
    of the window, and noise: bins and magnitudes are NumPy's, classes the host table's."""
    esq = M.esquema_b()
    esq = type(esq)(**{**esq.__dict__, "nombre": "X", "base_freq": base, "freq_step": step, "sep_freq": sep,
                       "ping_freq": 230 if sep else 0, "ping_n": esq.ping_n if sep else 0,
                       "sep_n": esq.sep_n if sep else 0, "sep_short_n": esq.sep_short_n if sep else 0})
    m = M.Modem(esq)
    lut = M.host_class_lut(esq)
    rng = np.random.default_rng(base + step)
    t = np.arange(7055)
    tones = [base + step * b for b in range(0, 256, 3)] + [esq.eof_freq] + ([sep, 230] if sep else [])
    windows = []
    for f in tones:
        windows.append(0.4 * np.sin(2 * np.pi * f * (t + int(rng.integers(0, 3000))) / 44100))
        other = sep if sep else tones[int(rng.integers(0, len(tones)))]
        cut = int(7055 * rng.uniform(0.05, 0.95))
        x = 0.4 * np.sin(2 * np.pi * other * t / 44100)
        x[cut:] = 0.4 * np.sin(2 * np.pi * f * t[cut:] / 44100)
        windows.append(x + 10 ** rng.uniform(-4, -1.5) * rng.standard_normal(7055))
    windows.append(0.2 * rng.standard_normal(7055))
    x32 = np.concatenate(windows).astype(np.float32)
    m.detect_stats()
    check_detect(m, x32, 7055, lut)
    st = m.detect_stats(paths=True)
    assert st[0] == len(windows) and st[9] + st[10] >= len(tones) // 3     # the one-bin tests did decide windows of this plan


def test_detect_sixteen_chosen_columns(monkeypatch, mb, golden):
    """Windows whose energy sits in a few spectrum columns of different column tiles get one tile made
    of the sixteen strongest columns, with twiddles and bins worked out per window.  (a) The separator
    ending and a data tone beginning anywhere in the window (what a receiver mostly sees).  (b) A bin
    of column 0 (a multiple of 85: its mirrored outputs repeat the plain ones), or of row 0 (a
    multiple of 83: likewise), with a stronger bin in another tile.  (c) The one bin the reference never looks at (3527) louder than
    everything, with bins in the two other tiles: the loudest of those is the answer.  (d) A constant
    offset with two tones.  (e) Forty bins all over: too many columns, the long way.  Bins, classes
    and magnitudes are the oracle's; the same windows with the choice switched off give the same."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    rng = np.random.default_rng(71)
    t = np.arange(7055)
    col = lambda k: min(k % 85, 85 - k % 85)
    tile_of = lambda k: col(k) // 16
    by_tile = [[k for k in range(60, 3500) if tile_of(k) == c] for c in range(3)]
    n_streams, n_windows, first = 3, 20, 2
    stride = M.pad4(first + n_windows * 7055 + 7)
    a = np.zeros((n_streams, stride), dtype=np.float32)
    kinds = []
    for s in range(n_streams):
        for w in range(n_windows):
            kind = (s + w) % 5
            kinds.append(kind)
            if kind == 0:
                cut = int(rng.integers(2200, 4900))
                x = 0.5 * np.sin(2 * np.pi * 200 * t / 44100)
                x[cut:] = 0.5 * np.sin(2 * np.pi * (350 + 70 * int(rng.integers(0, 256))) * t[cut:] / 44100)
            elif kind == 1:
                if w % 2:
                    special = 85 * int(rng.integers(2, 40))
                    other = int(rng.choice(by_tile[1 + (w // 2) % 2]))
                    ratio = 1.5
                else:
                    special = 83 * int(rng.choice([m for m in range(2, 42) if tile_of(83 * m) != 0]))
                    other = int(rng.choice(by_tile[0]))
                    ratio = 1.2
                x = 0.3 * np.cos(2 * np.pi * special * t / 7055 + 0.2) + 0.3 * ratio * np.cos(2 * np.pi * other * t / 7055 + 1.0)
            elif kind == 2:
                x = 0.4 * np.cos(2 * np.pi * 3527 * t / 7055 + 0.1) \
                    + 0.36 * np.cos(2 * np.pi * int(rng.choice(by_tile[0])) * t / 7055 + 0.7) \
                    + 0.25 * np.cos(2 * np.pi * int(rng.choice(by_tile[1])) * t / 7055 + 2.0)
            elif kind == 3:
                x = 0.12 + 0.3 * np.cos(2 * np.pi * int(rng.choice(by_tile[1])) * t / 7055) \
                    + 0.28 * np.cos(2 * np.pi * int(rng.choice(by_tile[2])) * t / 7055 + 0.4)
            else:
                x = np.zeros(7055)
                for k in rng.choice(np.arange(60, 3500), 40, replace=False):
                    x += 0.07 * rng.uniform(0.7, 1.0) * np.cos(2 * np.pi * int(k) * t / 7055 + rng.uniform(0, 6.28))
            a[s, first + w * 7055:first + (w + 1) * 7055] = x
    few = sum(1 for k in kinds if k != 4)
    monkeypatch.setenv("WN_DETECT_PICK", "0")
    off = M.Modem(M.esquema_b())
    for m, on in ((mb, True), (off, False)):
        m.detect_stats()
        cls, mag, kbin = m.detect(dev(a), n_streams, stride, n_windows, first)
        st = m.detect_stats(paths=True)
        assert st[0] == n_streams * n_windows
        if on:
            # most windows of kinds (a) to (d) (in some both tones share a column tile, and the pilot, which
            # sees a part of the rows, can find more than 70 % in one tile), none of kind (e)
            assert st[8] >= few // 2 and st[8] <= few
        else:
            assert st[8] == 0 and st[6] >= few // 2
        for s in range(n_streams):
            ob, om, oc = oracle_windows(a[s, first:first + n_windows * 7055], lut, 7055)
            assert np.array_equal(kbin[s].cpu().numpy(), ob)
            assert np.array_equal(cls[s].cpu().numpy().astype(np.uint16), oc)
            assert np.allclose(mag[s].cpu().numpy(), om, rtol=1e-4)


@pytest.mark.parametrize("scheme", ["B", "A"])
def test_detect_random_mixtures(mb, ma, golden, scheme):
    """Windows drawn at random: up to four tones of log-uniform strengths on arbitrary (not
    bin-centred) frequencies, a DC offset, a noise floor from nothing to louder than the tones,
    sometimes silence in part of the window, any overall scale.  Whatever path a window takes
    (one column tile, all of them, twice, float64 candidates, float64 spectrum), bin, class and
    magnitude are the oracle's."""
    m, n, lut = (mb, 7055, golden["classes_b_7055"]) if scheme == "B" else (ma, 2205, golden["classes_a_2205"])
    lut = np.array(lut, dtype=np.uint16)
    rng = np.random.default_rng(1234 if scheme == "B" else 4321)
    t = np.arange(n)
    wins = []
    for i in range(400):
        x = np.zeros(n)
        for _ in range(int(rng.integers(0, 5))):
            f = rng.uniform(0, 22050)
            x += 10 ** rng.uniform(-3, 0) * np.sin(2 * np.pi * f * t / 44100 + rng.uniform(0, 6.28))
        if rng.random() < 0.3:
            x += rng.uniform(-0.5, 0.5)
        if rng.random() < 0.7:
            x += 10 ** rng.uniform(-5, 0) * rng.standard_normal(n)
        if rng.random() < 0.2:
            a, b = sorted(rng.integers(0, n, 2))
            x[a:b] = 0.0
        wins.append(x * 10 ** rng.uniform(-2, 1))
    x32 = np.concatenate(wins).astype(np.float32)
    m.detect_stats()
    check_detect(m, x32, n, lut)
    st = m.detect_stats(columns=True)
    assert st[0] == 400 and st[3] < 400 * (3 * 43 if scheme == "B" else 2 * 25)   # some windows needed only part of the spectrum


@pytest.mark.parametrize("scheme", ["B", "A"])
def test_detect_signal_families(mb, ma, golden, scheme):
    """Shapes that put the window's energy where the spectrum cuts are awkward: two scheme tones
    meeting at any point of the window (what a receiver that is not aligned with the sender
    hears), square and clipped waves (harmonics in every column tile), chirps and
    amplitude-modulated tones (energy smeared over neighbouring bins), decaying tones, pulse trains
    whose period is a factor of the window length (energy in one spectrum column or one row of
    the two-stage DFT), steps and ramps (energy at the lowest bins).  Bins, classes and magnitudes
    are the oracle's on all of them."""
    m, n, lut, n1, n2 = (mb, 7055, golden["classes_b_7055"], 83, 85) if scheme == "B" else (ma, 2205, golden["classes_a_2205"], 45, 49)
    lut = np.array(lut, dtype=np.uint16)
    esq = SB if scheme == "B" else mo.scheme_a()
    rng = np.random.default_rng(99 if scheme == "B" else 77)
    t = np.arange(n)
    tones = [esq.base_freq + esq.freq_step * b for b in range(0, esq.max_byte + 1, 7)] + [esq.eof_freq]
    if esq.sep_freq:
        tones += [esq.sep_freq, esq.ping_freq]
    wins = []
    for i in range(600):
        kind = i % 10
        f = float(rng.choice(tones))
        if kind == 0 or kind == 1:      # two tones of the scheme meeting inside the window
            g = float(rng.choice(tones))
            cut = int(rng.integers(0, n + 1))
            x = np.where(t < cut, np.sin(2 * np.pi * f * (t + rng.integers(0, 5000)) / 44100),
                         np.sin(2 * np.pi * g * (t - cut) / 44100)) * 0.975
        elif kind == 2:                 # square wave
            x = np.sign(np.sin(2 * np.pi * f * t / 44100 + 0.1))
        elif kind == 3:                 # clipped tone
            x = np.clip(3.0 * np.sin(2 * np.pi * f * t / 44100), -1, 1)
        elif kind == 4:                 # chirp over a few bins to a few hundred bins
            span = 10 ** rng.uniform(0, 3.3)
            x = np.sin(2 * np.pi * (f + 0.5 * span * t / n) * t / 44100)
        elif kind == 5:                 # amplitude modulation
            x = (1 + 0.9 * np.sin(2 * np.pi * rng.uniform(1, 400) * t / 44100)) * np.sin(2 * np.pi * f * t / 44100)
        elif kind == 6:                 # decaying tone
            x = np.exp(-t / rng.uniform(50, 5000)) * np.sin(2 * np.pi * f * t / 44100)
        elif kind == 7:                 # pulse train with a period that divides the window
            p = int(rng.choice([n1, n2, 5, 17 if n == 7055 else 9]))
            x = np.zeros(n)
            x[int(rng.integers(0, p))::p] = 1.0
            x += 0.05 * np.sin(2 * np.pi * f * t / 44100)
        elif kind == 8:                 # step or ramp with a faint tone
            x = (t >= rng.integers(0, n)).astype(float) if rng.random() < 0.5 else t / n
            x = x + 10 ** rng.uniform(-3, -1) * np.sin(2 * np.pi * f * t / 44100)
        else:                           # a tone that starts or stops inside the window, over noise
            a, b = sorted(rng.integers(0, n, 2))
            x = np.zeros(n)
            x[a:b] = np.sin(2 * np.pi * f * t[a:b] / 44100)
            x += 10 ** rng.uniform(-4, -1) * rng.standard_normal(n)
        wins.append(x * 10 ** rng.uniform(-1.5, 0.5))
    x32 = np.concatenate(wins).astype(np.float32)
    # windows whose two largest float64 bins are closer than float64 rounding can tell apart are
    # not a question of this kernel (pulse trains make flat spectra): they are few
    spec = np.abs(np.fft.fft(x32.astype(np.float64).reshape(-1, n), axis=1))[:, :n // 2]
    top2 = np.sort(spec, axis=1)[:, -2:]
    clear = top2[:, 1] - top2[:, 0] > 1e-9 * top2[:, 1]
    assert clear.sum() >= 0.9 * len(wins)
    x32 = x32.reshape(-1, n)[clear].reshape(-1)
    check_detect(m, x32, n, lut)


def test_detect_refinement_paths_forced(golden):
    """Raise the tie band so that the float64 re-evaluation runs on ordinary windows: both
    the two-bin and the whole-spectrum path must reproduce the oracle."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    x = capture_of(golden["listen_b"][1])[:7055 * 24]
    rng = np.random.default_rng(3)
    x = np.concatenate([x, 0.1 * rng.standard_normal(7055 * 8)]).astype(np.float32)
    for band, which in ((0.5, 1), (0.9995, 2)):
        m = M.Modem(M.esquema_b())
        m.set_tie_band(band)
        m.detect_stats()
        check_detect(m, x, 7055, lut)
        st = m.detect_stats()
        assert st[0] == 32 and st[which] + st[2] > 0 and (which == 1 or st[2] >= 30), st   # silent windows need no float64


def test_detect_near_tie_decided_in_float64(mb, golden):
    """Two tones whose peaks are closer than float32 can tell: the float64 path decides as
    the reference does."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    n = np.arange(7055)
    k1, k2 = 400, 1400
    wins = []
    for eps in (0.0, 1e-7, -1e-7, 3e-6, -3e-6):
        wins.append(np.cos(2 * np.pi * k1 * n / 7055) + (1.0 + eps) * np.cos(2 * np.pi * k2 * n / 7055 + 0.3))
    x32 = (0.4 * np.concatenate(wins)).astype(np.float32)
    mb.detect_stats()
    check_detect(mb, x32, 7055, lut)
    assert mb.detect_stats()[1] >= 3


def test_detect_three_way_near_tie(mb, golden):
    """Three tones whose peaks float32 cannot order: the three candidates are re-evaluated in
    float64 (not the whole spectrum) and the oracle's bin wins; with four such tones the whole
    spectrum is."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    n = np.arange(7055)
    wins = []
    for e2, e3 in ((1e-7, -1e-7), (-2e-7, 1e-7), (0.0, 0.0), (3e-7, 2e-7)):
        wins.append(np.cos(2 * np.pi * 300 * n / 7055) + (1 + e2) * np.cos(2 * np.pi * 1500 * n / 7055 + 0.3)
                    + (1 + e3) * np.cos(2 * np.pi * 2900 * n / 7055 + 1.3))
    four = sum(np.cos(2 * np.pi * k * n / 7055 + 0.1 * k) for k in (300, 1100, 1900, 2700))
    x32 = (0.3 * np.concatenate(wins + [four])).astype(np.float32)
    mb.detect_stats()
    check_detect(mb, x32, 7055, lut)
    _, few, whole = mb.detect_stats()
    assert few >= 3 and whole <= 2


def test_detect_scheme_a(ma, golden):
    lut = np.array(golden["classes_a_2205"], dtype=np.uint16)
    msg = bytes(range(128))
    x = np.concatenate([mo.tone(f, n, d, SA) for f, n, d in mo.schedule(msg, SA)]).astype(np.float32)
    cls = check_detect(ma, x, 2205, lut)
    assert (cls[:128] & 0x1FF).tolist() == list(range(128))
    rng = np.random.default_rng(2)
    check_detect(ma, (x + 0.2 * rng.standard_normal(len(x))).astype(np.float32), 2205, lut)


# ------------------------------------------------------------------------------- state machine
@pytest.mark.parametrize("idx", range(7))
def test_listen_scenarios_end_to_end(mb, golden, idx):
    """detect -> slice -> parse on the reference's own scenarios: the same frames (or the
    same error) call after call, and the same number of windows per call."""
    s = golden["listen_b"][idx]
    x32 = capture_of(s).astype(np.float32)
    nwin = -(-len(x32) // 7055)
    pad = np.zeros(M.pad4(nwin * 7055), dtype=np.float32)
    pad[:len(x32)] = x32
    tw = mb.timeout_windows(s["timeout"])
    assert tw == mo.timeout_to_windows(s["timeout"], SB)
    out = mb.decode_streams(dev(pad), 1, len(pad), nwin, s["timeout"], max_calls=len(s["calls"]), frame_cap=288)
    by, lens = out["bytes"].cpu().numpy()[0], out["lens"].cpu().numpy()[0]
    start, end = out["start"].cpu().numpy()[0], out["end"].cpu().numpy()[0]
    rec = M.Modem.records_to_numpy(out["rec"][0])
    ncalls = int(out["n_calls"][0])
    for j, call in enumerate(s["calls"]):
        if j >= ncalls:
            assert call["oracle_bytes"] == ""       # capture exhausted: nothing left to hear
            continue
        got = bytes(by[j, :lens[j]])
        assert got.hex() == call["oracle_bytes"], (s["name"], j)
        if end[j] < nwin:
            assert end[j] - start[j] == call["windows"]
        if call["error"] is None:
            assert got.hex() == call["bytes"] and rec[j]["status"] == L.ST_OK and rec[j]["crc_ok"] == 1
        else:
            msgs = {L.ST_EMPTY: mo.PARSE_MESSAGES[1], L.ST_SHORT_HEADER: mo.PARSE_MESSAGES[2],
                    L.ST_SHORT_BODY: mo.PARSE_MESSAGES[3]}
            assert msgs[int(rec[j]["status"])] == call["error"]


def test_slice_random_class_sequences(mb):
    """The state machine alone on random class words (pings, EOFs, bytes, separators in any
    order), many streams, several timeouts."""
    rng = np.random.default_rng(9)
    n_streams, n_windows = 300, 400
    pool = np.array([256, 256, 256, 256 | 512, 256 | 1024, 7, 7, 9, 200, 255, 0], dtype=np.uint16)
    cls = pool[rng.integers(0, len(pool), (n_streams, n_windows))]
    for tw, max_calls in ((33, 6), (5, 40), (1000, 3)):
        by, lens, start, end, reason, ncalls = [t.cpu().numpy() for t in
                                                mb.slice(dev(cls.astype(np.int16)), tw, max_calls, 64)]
        for s in range(n_streams):
            want = mo.slice_classes(cls[s], tw, max_calls)
            assert ncalls[s] == len(want)
            for j, (w0, w1, b, r) in enumerate(want):
                assert (start[s, j], end[s, j]) == (w0, w1)
                assert bytes(by[s, j, :lens[s, j]]) == b[:64]
                assert (reason[s, j] & 0xFF) == r
                assert bool(reason[s, j] & L.END_OVERFLOW) == (len(b) > 64)


def test_slice_long_frames_across_chunk_boundaries(mb):
    """The kernel walks the class words 64 windows at a time.  Long frames (hundreds of windows
    between ping and EOF) whose runs of byte windows and separators straddle every multiple of 64
    at every phase, timeouts that end in the middle of a chunk, an EOF in the same chunk as a
    symbol, a ping as the last window of a chunk: all must equal the oracle's sequential loop."""
    rng = np.random.default_rng(17)
    NONE, PING, EOF = 256, 256 | 512, 256 | 1024
    rows = []
    for s in range(192):
        lead = s % 67                                  # the ping lands on every position of a chunk
        seq = [NONE] * lead + [PING]
        n_sym = 20 + s % 50
        for i in range(n_sym):
            seq += [NONE] * int(rng.integers(1, 4))    # separator windows
            seq += [int(rng.integers(0, 256))] * int(rng.integers(1, 6))   # the same tone over several windows
        if s % 3 == 0:
            seq += [int(rng.integers(0, 256)), EOF]    # symbol without separator right before the EOF
        elif s % 3 == 1:
            seq += [NONE, EOF]
        # s % 3 == 2: no EOF, the call ends by timeout or at the end of the capture
        seq += [NONE] * 5 + [PING, NONE, 65, NONE, 66, 66, NONE, EOF]      # a second, short frame
        rows.append(seq)
    n_windows = max(len(r) for r in rows) + 3
    cls = np.full((len(rows), n_windows), NONE, dtype=np.uint16)
    for i, r in enumerate(rows):
        cls[i, :len(r)] = r
    for tw, max_calls in ((1000, 4), (97, 8), (64, 8), (63, 10), (130, 6)):
        by, lens, start, end, reason, ncalls = [t.cpu().numpy() for t in
                                                mb.slice(dev(cls.astype(np.int16)), tw, max_calls, 128)]
        for s in range(len(rows)):
            want = mo.slice_classes(cls[s], tw, max_calls)
            assert ncalls[s] == len(want), (s, tw)
            for j, (w0, w1, b, r) in enumerate(want):
                assert (start[s, j], end[s, j], reason[s, j] & 0xFF) == (w0, w1, r), (s, tw, j)
                assert bytes(by[s, j, :lens[s, j]]) == b[:128], (s, tw, j)


def test_slice_long_captures(mb):
    """Captures of more than 2048 windows take the sixteen-wave form of the kernel (tables for
    98 304 windows in LDS, calls in batches of 128), longer ones the single-wave machine over class
    words staged 16 384 at a time: random class words with frames, timeouts and symbol runs that
    straddle the multiples of 4096 and 16 384, a capture that ends exactly on one, calls that run
    out in the middle, more calls than a batch holds, and the call limit reached before the
    capture ends."""
    rng = np.random.default_rng(23)
    NONE, PING, EOF = 256, 256 | 512, 256 | 1024
    pool = np.array([NONE] * 12 + [PING, EOF] + [7, 7, 9, 200, 255, 0, 33, 33], dtype=np.uint16)
    for n_windows in (2049, 16384, 16385, 40000, 98304, 100000):
        n_streams = 12 if n_windows < 90000 else 3
        cls = pool[rng.integers(0, len(pool), (n_streams, n_windows))]
        # a frame of alternating separator / byte windows across each stage boundary, phase by stream
        for s in range(n_streams):
            for edge in (16384, 32768):
                if edge + 200 < n_windows:
                    lo = edge - 90 - s
                    cls[s, lo] = PING
                    body = np.where(np.arange(1, 170) % 2 == (s & 1), NONE, 40 + (np.arange(1, 170) // 2) % 200)
                    cls[s, lo + 1:lo + 170] = body.astype(np.uint16)
                    cls[s, lo + 170] = EOF
        for tw, max_calls in ((700, 64), (50, 3000), (20000, 5), (64, 7)):
            by, lens, start, end, reason, ncalls = [t.cpu().numpy() for t in
                                                    mb.slice(dev(cls.astype(np.int16)), tw, max_calls, 96)]
            for s in range(n_streams):
                want = mo.slice_classes(cls[s], tw, max_calls)
                assert ncalls[s] == len(want), (n_windows, s, tw)
                for j, (w0, w1, b, r) in enumerate(want):
                    assert (start[s, j], end[s, j], reason[s, j] & 0xFF) == (w0, w1, r), (n_windows, s, tw, j)
                    assert bytes(by[s, j, :lens[s, j]]) == b[:96], (n_windows, s, tw, j)
                    assert bool(reason[s, j] & L.END_OVERFLOW) == (len(b) > 96)


def test_contexts_of_both_schemes_on_one_device(golden):
    """Making the short-window context after the long-window one must leave the long one fully
    usable, its float64 pass (whose phase table is the largest dynamic LDS request) included."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    b = M.Modem(M.esquema_b())
    a = M.Modem(M.esquema_a())
    b.set_tie_band(0.5)
    x = capture_of(golden["listen_b"][0])[:7055 * 10].astype(np.float32)
    b.detect_stats()
    check_detect(b, x, 7055, lut)
    assert b.detect_stats()[1] > 0
    xa = np.concatenate([mo.tone(f, n, d, SA) for f, n, d in mo.schedule(b"ab", SA)]).astype(np.float32)
    check_detect(a, xa, 2205, np.array(golden["classes_a_2205"], dtype=np.uint16))


# ------------------------------------------------------------------------------- K3
def test_sync_scan_energies_and_sound_verdicts(mb, golden):
    """Energies within 1e-4 relative of the float64 values; a decided verdict is never wrong."""
    lut = np.array(golden["classes_b_7055"], dtype=np.uint16)
    for idx in (1, 4, 5):
        s = golden["listen_b"][idx]
        x32 = capture_of(s).astype(np.float32)
        nwin = len(x32) // 7055
        pad = np.zeros(M.pad4(nwin * 7055), dtype=np.float32)
        pad[:nwin * 7055] = x32[:nwin * 7055]
        pe, we, vd = [t.cpu().numpy()[0] for t in mb.sync_scan(dev(pad), 1, len(pad), nwin)]
        xw = pad[:nwin * 7055].astype(np.float64).reshape(nwin, 7055)
        spec = np.abs(np.fft.fft(xw, axis=1)) ** 2
        want_pe = spec[:, 39:42].max(axis=1)
        want_we = (xw ** 2).sum(axis=1)
        assert np.allclose(we, want_we, rtol=1e-4, atol=1e-12)
        assert np.all(np.abs(pe - want_pe) <= 1e-4 * want_pe + 1e-6 * (7055 * want_we))
        is_ping = (lut[np.argmax(spec[:, :3527], axis=1)] & 512) != 0
        assert np.all(is_ping[vd == 1]) and not np.any(is_ping[vd == 0])
        if s["sigma"] < 0.5:
            assert (vd != 2).mean() > 0.9           # clean frames are decided without a spectrum
    sil = torch.zeros(7056 * 3, dtype=torch.float32, device="cuda")
    assert mb.sync_scan(sil, 1, 7056 * 3, 3)[2].cpu().tolist() == [[0, 0, 0]]
    # far outside the float32 energy range nothing is decided wrongly: undecided, or silence
    x = capture_of(golden["listen_b"][1]).astype(np.float32)[:8 * 7055]
    is_ping = (lut[mo.window_bins(x.astype(np.float64), 7055)[0]] & 512) != 0
    for exp in (-100, -70, 58):
        pad = np.zeros(M.pad4(8 * 7055 + 7055), dtype=np.float32)
        pad[:8 * 7055] = x * np.float32(2.0) ** exp
        vd = mb.sync_scan(dev(pad), 1, len(pad), 9)[2].cpu().numpy()[0]
        assert np.all(is_ping[vd[:8] == 1]) and not np.any(is_ping[vd[:8] == 0])
        assert vd[8] == 0


def test_capture_cut_into_window_ranges(mb, golden):
    """One capture held in pieces (how a long capture is spread over several GPUs): detection
    of each window range on its own copy gives the same class words as detection of the whole,
    and the listen loop over the collected words returns the same frames."""
    from wavenet_amd import sharding as SH
    s = golden["listen_b"][4]
    x32 = capture_of(s).astype(np.float32)
    nwin = len(x32) // 7055
    whole = np.zeros(M.pad4(nwin * 7055), dtype=np.float32)
    whole[:nwin * 7055] = x32[:nwin * 7055]
    timeout, calls = 60.0, len(s["frames"]) + 1
    ref = mb.decode_streams(dev(whole), 1, len(whole), nwin, timeout, max_calls=calls)
    pieces = []
    for r in range(3):
        lo, hi = SH.shard_range(nwin, r, 3)
        local = np.zeros(M.pad4((hi - lo) * 7055), dtype=np.float32)
        local[:(hi - lo) * 7055] = whole[lo * 7055:hi * 7055]
        pieces.append(mb.detect(dev(local), 1, len(local), hi - lo)[0].view(-1))
    assert torch.equal(torch.cat(pieces), ref["cls"].view(-1))
    got = SH.decode_capture_sharded(mb, dev(whole), nwin, timeout, calls)       # one rank: the whole range
    for k in ("cls", "bytes", "lens", "start", "end", "reason", "n_calls"):
        assert torch.equal(got[k].view(-1), ref[k].view(-1)), k
    assert int(got["n_calls"][0]) >= len(s["frames"])


# ------------------------------------------------------------------------------- round trip at scale
def test_roundtrip_many_streams_max_payload(mb):
    """Configuration 2 in small: independent streams, one 107-byte frame each, encode on the
    GPU, decode on the GPU; every frame must come back bit-exact with a valid checksum, and
    a sample of streams is checked window by window against the oracle."""
    n = 48
    rng = np.random.default_rng(21)
    fields = np.zeros((n, 16), dtype=np.uint8)
    fields[:, 0] = 1
    fields[:, 1:13] = rng.integers(0, 256, (n, 12))
    fields[:, 13] = 3
    fields[:, 14] = 107
    pay = rng.integers(0, 256, (n, 112), dtype=np.uint8)
    pay[3, :107] = 42                      # a stream full of repeated bytes (short separators)
    frames, lens = mb.build_frames(dev(fields), dev(pay), 128)
    stride = M.pad4(mb.esquema.max_frame_samples(126))
    nwin = stride // 7055
    audio, n_samples = mb.encode_streams(frames, lens, stride)
    out = mb.decode_streams(audio, n, stride, nwin, 123.0)
    by, ln = out["bytes"].cpu().numpy()[:, 0], out["lens"].cpu().numpy()[:, 0]
    rec = M.Modem.records_to_numpy(out["rec"])
    fr = frames.cpu().numpy()
    for i in range(n):
        assert ln[i] == 126 and np.array_equal(by[i, :126], fr[i, :126]), i
        assert rec[i]["status"] == L.ST_OK and rec[i]["crc_ok"] == 1
        want = mo.build_frame(1, bytes(fields[i, 1:7]), bytes(fields[i, 7:13]), 3, bytes(pay[i, :107]))
        assert bytes(fr[i, :126]) == want
    lut = M.host_class_lut(mb.esquema)
    a = audio.cpu().numpy()
    for i in (0, 3, n - 1):
        ob, om, oc = oracle_windows(a[i, :nwin * 7055], lut, 7055)
        assert np.array_equal(out["bin"][i].cpu().numpy(), ob)
        assert np.array_equal(out["cls"][i].cpu().numpy().astype(np.uint16), oc)
        got, reason, used = mo.listen_frame(mo.Capture(a[i].astype(np.float64)), 123.0, SB)
        assert bytes(got) == bytes(fr[i, :126]) and reason == mo.R_EOF
        assert int(out["end"][i, 0]) == used


def test_empty_and_smallest_inputs(mb, ma):
    """Nothing to do is not an error: zero frames, zero streams, zero windows; and the
    smallest real cases: an empty payload, one window, a capture shorter than any frame."""
    z8 = torch.zeros((0, 128), dtype=torch.uint8, device="cuda")
    z32 = torch.zeros((0,), dtype=torch.int32, device="cuda")
    assert mb.crc32(z8, z32).numel() == 0
    assert mb.parse_frames(z8, z32).numel() == 0
    frames, lens = mb.build_frames(torch.zeros((0, 16), dtype=torch.uint8, device="cuda"),
                                   torch.zeros((0, 112), dtype=torch.uint8, device="cuda"), 128)
    assert frames.shape[0] == 0 and lens.numel() == 0
    audio = torch.zeros(M.pad4(7055), dtype=torch.float32, device="cuda")
    cls, mag, kbin = mb.detect(audio, 0, 0, 0)
    assert cls.numel() == 0
    cls, mag, kbin = mb.detect(audio, 1, audio.numel(), 0)
    assert cls.shape == (1, 0)
    by, ln, st, en, rs, nc = mb.slice(torch.zeros((1, 0), dtype=torch.int16, device="cuda"), 10, 1, 128)
    assert int(nc[0]) == 0
    # one silent window: bin 0, its class, magnitude 0; the listener times out empty-handed
    cls, mag, kbin = mb.detect(audio, 1, audio.numel(), 1)
    assert int(kbin[0, 0]) == 0 and float(mag[0, 0]) == 0.0
    by, ln, st, en, rs, nc = mb.slice(cls, 10, 1, 128)
    assert int(nc[0]) == 1 and int(ln[0, 0]) == 0 and int(en[0, 0]) == 1 and int(rs[0, 0]) == L.END_TIMEOUT
    # empty payload: the 19-byte ping frame, through the whole path
    f = mo.build_frame(1, MAC_O, MAC_D, 1, b"")
    buf, lens = rows_from([f], 32)
    stride = M.pad4(len(mo.emit_frame(f, SB)) + 7055)
    a, n = mb.encode_streams(dev(buf), dev(lens), stride)
    out = mb.decode_streams(a, 1, stride, stride // 7055, 30.0, frame_cap=32)
    assert bytes(out["bytes"][0, 0, :int(out["lens"][0, 0])].cpu().numpy()) == f
    rec = M.Modem.records_to_numpy(out["rec"])[0]
    assert rec["status"] == L.ST_OK and rec["crc_ok"] == 1 and rec["length"] == 0
    # the prototype scheme: a single one-byte message
    buf, lens = rows_from([b"\x07"], 16)
    a, n = ma.encode_streams(dev(buf), dev(lens), M.pad4(2 * 2205))
    cls, _, _ = ma.detect(a, 1, a.shape[1], 1)
    assert int(cls[0, 0]) & 0x1FF == 7


def test_awgn_statistics(mb):
    a = torch.zeros(1 << 20, dtype=torch.float32, device="cuda")
    mb.awgn(a, 0.1, 1234)
    x = a.cpu().numpy().astype(np.float64)
    assert abs(x.mean()) < 1e-3 and abs(x.std() - 0.1) < 1e-3
    assert abs((np.abs(x) > 0.2).mean() - 0.0455) < 2e-3
    b = torch.zeros(1 << 20, dtype=torch.float32, device="cuda")
    mb.awgn(b, 0.1, 1234)
    assert torch.equal(a, b)

"""CPU-only checks of the C ABI: the library loads, exports every declared symbol, and its
host-side tables (no GPU involved) equal the oracle's / the reference's golden values."""
import os
import re

import numpy as np

from oracle import modem_oracle as mo
from wavenet_amd import _lib as L
from wavenet_amd import modem as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "wavenet_hip.h")).read()
    declared = set(re.findall(r"\b(wn_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    for name in declared:
        assert getattr(L.lib, name) is not None


def test_struct_layouts():
    import ctypes as C
    assert C.sizeof(L.WnFrameRec) == 8 and M.REC_DTYPE.itemsize == 8
    assert C.sizeof(L.WnScheme) == 14 * 4 + 16


def test_schemes_match_oracle():
    for e, o in ((M.esquema_b(), mo.scheme_b()), (M.esquema_a(), mo.scheme_a())):
        for f in ("sample_rate", "base_freq", "freq_step", "max_byte", "sep_freq", "ping_freq", "eof_freq",
                  "tone_n", "sep_n", "sep_short_n", "ping_n", "eof_n", "win_n", "amp", "gain"):
            assert getattr(e, f) == getattr(o, f), f
        assert bool(e.tol_check) == o.tol_check


def test_class_lut_matches_reference(golden):
    assert M.host_class_lut(M.esquema_b()).tolist() == golden["classes_b_7055"]
    assert M.host_class_lut(M.esquema_a()).tolist() == golden["classes_a_2205"]


def test_dft_plan():
    assert M.host_dft_plan(M.esquema_b()) == (83, 85)
    assert M.host_dft_plan(M.esquema_a()) == (45, 49)
    import dataclasses
    bad = dataclasses.replace(M.esquema_b(), win_n=7056)
    import ctypes as C
    a, b = C.c_int32(), C.c_int32()
    assert L.lib.wn_host_dft_plan(C.byref(bad.c_struct()), C.byref(a), C.byref(b)) == L.WN_ERR_UNSUPPORTED
    assert b"7056" in L.lib.wn_last_error()


def test_sine_tables_against_oracle_tones():
    """Every sample of a reference tone is table[(f*i) mod sr] folded to a quarter period,
    to float32 rounding of the float64 value."""
    for esq, sch, dur in ((M.esquema_b(), mo.scheme_b(), mo.BYTE_DURATION * 2), (M.esquema_a(), mo.scheme_a(), 0.05)):
        f32, i16 = M.host_sine_tables(esq)
        sr, q = esq.sample_rate, esq.sample_rate // 4
        for freq in (200, 250, 300, 350, 4410, 13230, 17710, 18200, 900, 2500):
            n = esq.ping_n if esq.nombre == "B" else esq.tone_n
            ref = mo.tone(freq, n, dur, sch)
            m = (freq * np.arange(n, dtype=np.int64)) % sr
            neg = m >= 2 * q
            m2 = np.where(neg, m - 2 * q, m)
            idx = np.where(m2 > q, 2 * q - m2, m2)
            mine = np.where(neg, -f32[idx], f32[idx])
            assert np.max(np.abs(mine.astype(np.float64) - ref)) <= 6e-8 + 1e-11
            # correctly rounded up to the reference's own float64 phase rounding (~2e-11 rad)
            half_ulp = 0.5 * np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
            assert np.all(np.abs(mine.astype(np.float64) - ref) <= half_ulp * 1.0001 + 5e-11)
            plain = mo.Scheme(**{**sch.__dict__, "gain": 1.0})
            ref16 = np.int16(mo.tone(freq, n, dur, plain) * 32767)
            mine16 = np.where(neg, -i16[idx].astype(np.int32), i16[idx].astype(np.int32))
            assert np.array_equal(mine16, ref16.astype(np.int32))


def test_frame_samples(golden):
    e = M.esquema_b()
    for c in golden["sample_counts_b"]:
        assert M.host_frame_samples(e, bytes.fromhex(c["frame"])) == c["n_samples"]
    assert e.max_frame_samples(126) == 3056130
    fr = bytes.fromhex(golden["wav_b"]["frame"])
    assert M.host_frame_samples(e, fr, L.EMIT_WAV) == golden["wav_b"]["n"]
    assert M.host_frame_samples(M.esquema_a(), bytes(range(128)), L.EMIT_WAV) == golden["wav_a"]["n"]


def test_no_gpu_means_loud_failure():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        M.Modem()


def test_entry_points_reject_bad_arguments_before_touching_a_device():
    """Argument checks come first: with a null context or pointers every entry point returns
    WN_ERR_ARG and says which argument, without launching anything."""
    import ctypes as C
    null = C.c_void_p(None)
    lib = L.lib
    calls = [
        lambda: lib.wn_crc32_batch(null, null, null, 128, 1, null, null),
        lambda: lib.wn_frame_build_batch(null, null, null, 112, 1, null, 128, null, null),
        lambda: lib.wn_frame_parse_validate_batch(null, null, null, 128, 1, null, null),
        lambda: lib.wn_synth_layout(null, null, null, 128, 1, 0, 300, null, null, null, null, null),
        lambda: lib.wn_synth_batch(null, null, null, null, 300, 1, null, null, 1000, null, 1000, 0, null),
        lambda: lib.wn_channel_awgn(null, null, 10, 0.1, 1, null),
        lambda: lib.wn_detect_windows(null, null, 7056, 1, 7056, 0, 1, null, null, null, null),
        lambda: lib.wn_sync_scan(null, null, 7056, 1, 7056, 0, 1, null, null, null, null),
        lambda: lib.wn_slice_frames(null, null, 1, 1, 10, 1, 128, null, null, null, null, null, null, null),
        lambda: lib.wn_detect_set_tie_band(null, 0.1),
        lambda: lib.wn_selftest_mfma(null, null),
    ]
    for i, call in enumerate(calls):
        assert call() == L.WN_ERR_ARG, i
        assert lib.wn_last_error().startswith(b"bad argument"), (i, lib.wn_last_error())
    assert lib.wn_ctx_destroy(null) == 0
    assert lib.wn_strerror(L.WN_ERR_ARG) and lib.wn_strerror(0)
    bad = L.WnScheme()          # all zeros: not a scheme
    ctx = C.c_void_p()
    assert lib.wn_scheme_init(C.byref(bad), C.byref(ctx)) == L.WN_ERR_ARG and not ctx.value
